// hbm_phase_probe.hip -- what does HBM sustain for a 1 : 1 read / write mix when every workgroup of the chip alternates between a READ
// burst and a WRITE burst of U x 16 B per thread (persistent grid, one workgroup per CU, identical work: the workgroups run in step, so
// the whole chip alternates)?  U = 1 is the finely mixed copy of a trivial kernel; csrc/mpx_convw.h moves 128 KB per CU and burst.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/hbm_phase_probe tools/probes/hbm_phase_probe.hip      run: ./hbm_phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int U, int R>      // U 16-B units per thread and burst; R = reads per write burst (1: copy; 2: a second input stream is read and added)
__global__ __launch_bounds__(256) void phased_copy(const u4* __restrict__ a, const u4* __restrict__ b, u4* __restrict__ c, long long n_units) {
    const long long per_wg = (long long)U * 256;
    const long long n_chunks = n_units / per_wg;
    for (long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const long long base = ch * per_wg + threadIdx.x;
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + base + u * 256);
        if (R == 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] += __builtin_nontemporal_load(b + base + u * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], c + base + u * 256);
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int U, int R>
int run(const u4* a, const u4* b, u4* c, long long n_units, int grid, const char* what) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((phased_copy<U, R>), dim3(grid), dim3(256), 0, 0, a, b, c, n_units);
    CHECK(hipDeviceSynchronize());
    const int reps = 5;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((phased_copy<U, R>), dim3(grid), dim3(256), 0, 0, a, b, c, n_units);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)n_units * 16 * (R + 1);
    printf("%-22s grid %4d  burst %3d x 16 B per thread = %4d KB per workgroup: %7.3f ms  %5.2f TB/s\n", what, grid, U, U * 4, ms, bytes / ms / 1e9);
    return 0;
}

int main() {
    const long long n_units = (3LL << 30) / 16 * 2;       // 6 GiB per array
    u4 *a, *b, *c;
    CHECK(hipMalloc(&a, n_units * 16)); CHECK(hipMalloc(&b, n_units * 16)); CHECK(hipMalloc(&c, n_units * 16));
    CHECK(hipMemset(a, 1, n_units * 16)); CHECK(hipMemset(b, 2, n_units * 16));
    for (int grid : {256, 512, 1024}) {
        if (run<1, 1>(a, b, c, n_units, grid, "copy (1 : 1)")) return 1;
        if (run<4, 1>(a, b, c, n_units, grid, "copy (1 : 1)")) return 1;
        if (run<16, 1>(a, b, c, n_units, grid, "copy (1 : 1)")) return 1;
        if (run<32, 1>(a, b, c, n_units, grid, "copy (1 : 1)")) return 1;
        if (run<1, 2>(a, b, c, n_units, grid, "add  (2 : 1)")) return 1;
        if (run<16, 2>(a, b, c, n_units, grid, "add  (2 : 1)")) return 1;
    }
    if (run<1, 1>(a, b, c, n_units, 65536, "copy (1 : 1), big grid")) return 1;
    if (run<4, 1>(a, b, c, n_units, 65536, "copy (1 : 1), big grid")) return 1;
    return 0;
}
