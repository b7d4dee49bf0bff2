// f16x3 MFMA loops WITH their operand traffic from LDS, under the package power limit (round 6): what does the wave tile cost in watts?
// profiles/r06_conv3pp_ablation_and_power.txt says the persistent 3x3 kernel is throttled (1.98 of 2.4 GHz at 1380 W) by its LDS -> register
// operand reads of FRESH data far more than by its MFMAs (constant operands: 1010 W at 2.4 GHz).  A wave tile of M x N reads (M + N) rows
// per K step for M * N / 256 * 3 MFMAs, so a 128 x 128 tile moves half the LDS bytes per MFMA of the kernels' 64 x 64.
//   mode 0   8 waves (2 per SIMD), wave tile 64 x 64:  16 fragment reads (ds_read_b128) per 48 MFMAs     -- the kernels' shape
//   mode 1   4 waves (1 per SIMD), wave tile 128 x 128: 32 fragment reads per 192 MFMAs                  -- accumulators = 256 registers
//   mode 2   8 waves, wave tile 128 x 64:  24 fragment reads per 96 MFMAs
// Every CU runs one workgroup; the LDS holds 64 KB of random ReLU-like planes and every K step reads another slice of it (fresh data in the
// registers each step, as in a kernel); no DMA, no global traffic in the loop.  Prints ms, algorithmic TFLOP/s of the chip.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_tile_probe mfma_lds_tile_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int LDS_BYTES = 65536;

template <int CF, int PF, int NT>
__global__ __launch_bounds__(NT, 1) void probe(const h8* src, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += NT) ((h8*)smem)[i] = src[(blockIdx.x * (LDS_BYTES / 16) + i) & 0xffff];
    __syncthreads();
    f4 acc[CF][PF];
#pragma unroll
    for (int a = 0; a < CF; ++a)
#pragma unroll
        for (int b = 0; b < PF; ++b) acc[a][b] = f4{0, 0, 0, 0};
    h8 ah[CF], al[CF], bh[PF], bl[PF];
    // per-lane base: a fragment is 16 rows x 64 B; lanes read 16 B each: row = lane & 15, chunk = lane >> 4
    const int lbase = (lane & 15) * 64 + (lane >> 4) * 16;
    int off = (wave * 4096) & (LDS_BYTES - 1);
    for (int s = 0; s < steps; ++s) {
        // fresh fragments of this step (hi planes in the lower half of the LDS, lo planes in the upper half)
#pragma unroll
        for (int a = 0; a < CF; ++a) {
            ah[a] = *(const h8*)(smem + ((off + a * 1024 + lbase) & (LDS_BYTES / 2 - 1)));
            al[a] = *(const h8*)(smem + LDS_BYTES / 2 + ((off + a * 1024 + lbase) & (LDS_BYTES / 2 - 1)));
        }
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            bh[b] = *(const h8*)(smem + ((off + 8192 + b * 1024 + lbase) & (LDS_BYTES / 2 - 1)));
            bl[b] = *(const h8*)(smem + LDS_BYTES / 2 + ((off + 8192 + b * 1024 + lbase) & (LDS_BYTES / 2 - 1)));
        }
        off = (off + 2048 + 64) & (LDS_BYTES / 2 - 1);
#pragma unroll
        for (int a = 0; a < CF; ++a)
#pragma unroll
            for (int b = 0; b < PF; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            }
    }
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < CF; ++a)
#pragma unroll
        for (int b = 0; b < PF; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * NT + tid] = sum;
}

int main() {
    const int N = 1 << 16;
    std::vector<_Float16> h(N * 8);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
    for (int i = 0; i < N * 8; ++i) {
        const float x = rnd() > 0 ? rnd() * 2.f : 0.f;             // ReLU-like: half the values zero
        h[i] = (_Float16)(x < 0 ? -x : x);
    }
    h8* d; float* dout;
    hipMalloc(&d, N * 16);
    hipMemcpy(d, h.data(), N * 16, hipMemcpyHostToDevice);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipMalloc(&dout, cus * 512 * 4);
    hipFuncSetAttribute((const void*)probe<4, 4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)probe<8, 8, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)probe<8, 4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"8 waves, wave tile  64 x  64 (16 reads / 48 MFMAs) ", "4 waves, wave tile 128 x 128 (32 reads / 192 MFMAs)", "8 waves, wave tile 128 x  64 (24 reads / 96 MFMAs) "};
    // the same number of MFMAs per CU in every mode: steps x waves x MFMAs per step
    const int steps[3] = {40000, 20000, 20000};
    const double mfmas[3] = {40000.0 * 8 * 48, 20000.0 * 4 * 192, 20000.0 * 8 * 96};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(e0);
                for (int k = 0; k < 8; ++k) {
                    if (mode == 0) hipLaunchKernelGGL((probe<4, 4, 512>), dim3(cus), dim3(512), LDS_BYTES, 0, d, dout, steps[0]);
                    else if (mode == 1) hipLaunchKernelGGL((probe<8, 8, 256>), dim3(cus), dim3(256), LDS_BYTES, 0, d, dout, steps[1]);
                    else hipLaunchKernelGGL((probe<8, 4, 512>), dim3(cus), dim3(512), LDS_BYTES, 0, d, dout, steps[2]);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = mfmas[mode] * 8 * 16384.0 * cus;          // 8 launches; 2 * 16 * 16 * 32 flop per MFMA
            printf("%s  %.1f ms for 8 launches = %.0f TFLOP/s issued (%.0f algorithmic)\n", names[mode], ms, flops / ms / 1e9, flops / ms / 1e9 / 3);
        }
    return 0;
}
