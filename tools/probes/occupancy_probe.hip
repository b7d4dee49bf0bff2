// How many 256-thread workgroups with N bytes of dynamic LDS does one CU hold?  (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(float* p) { extern __shared__ float s[]; s[threadIdx.x] = 1.f; __syncthreads(); if (p) p[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int lds : {49152, 65536, 73728, 77824, 80896, 81920, 98304}) {
        int n = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds);
        printf("dynamic LDS %6d B -> %d workgroups per CU (%s)\n", lds, n, hipGetErrorString(e));
    }
    return 0;
}
