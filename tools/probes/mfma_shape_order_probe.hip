// Bare f16x3 MFMA loops under the package power limit (round 6): does the MFMA SHAPE or the ORDER of the three products change what a
// power-limited kernel can issue?  Same harness as mfma_i8corr_probe.hip (one wave per SIMD, a 64 x 64 wave tile, operands in registers,
// ReLU-like data, every CU busy).
//   0  16x16x32, per fragment pair: hi.lo, lo.hi, hi.hi                      (the engine's order inside a pair)
//   1  16x16x32, per weight fragment a: hi.lo over all b, lo.hi over all b, hi.hi over all b      (mpx_conv3pp.h's order: A shared by 4 MFMAs)
//   2  16x16x32, per pair: hi.lo, hi.hi, lo.hi                              (consecutive MFMAs always share one operand)
//   3  32x32x16, per pair: hi.lo, lo.hi, hi.hi     (2 x 2 fragments of 32: twice the MACs per operand register read)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_order_probe mfma_shape_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const h8* ah, const h8* al, const h8* bh, const h8* bl, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    h8 Ah[4], Al[4], Bh[4], Bl[4];
    for (int i = 0; i < 4; ++i) {
        Ah[i] = ah[(t * 4 + i) & 0xffff]; Al[i] = al[(t * 4 + i) & 0xffff];
        Bh[i] = bh[(t * 4 + i) & 0xffff]; Bl[i] = bl[(t * 4 + i) & 0xffff];
    }
    float s = 0.f;
    if constexpr (MODE == 3) {
        // 64 x 64 x 32 per step = 2 x 2 fragments of 32 x 32, two K halves of 16: Ah[2 * a + k], Bh[2 * b + k]
        f16v acc[2][2];
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[2 * a + k], Bl[2 * b + k], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[2 * a + k], Bh[2 * b + k], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[2 * a + k], Bh[2 * b + k], acc[a][b], 0, 0, 0);
                    }
            asm volatile("" : "+v"(Ah[0]), "+v"(Bh[0]));
        }
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int j = 0; j < 16; ++j) s += acc[a][b][j];
    } else {
        f4 acc[4][4];
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) acc[a][b] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[a], Bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bh[b], acc[a][b], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bl[b], acc[a][b], 0, 0, 0);
                        if (MODE == 0) {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[a], Bh[b], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bh[b], acc[a][b], 0, 0, 0);
                        } else {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bh[b], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[a], Bh[b], acc[a][b], 0, 0, 0);
                        }
                    }
            }
            asm volatile("" : "+v"(Ah[0]), "+v"(Bh[0]));
        }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
                for (int j = 0; j < 4; ++j) s += acc[a][b][j];
    }
    out[t] = s;
}

int main() {
    const int N = 1 << 16;
    std::vector<_Float16> hh(N * 8), hl(N * 8), xh(N * 8), xl(N * 8);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
    for (int i = 0; i < N * 8; ++i) {
        const float w = rnd() * 700.f, x = rnd() > 0 ? rnd() * 2.f : 0.f;          // weights scaled like the engine's planes; half the pixels zero
        hh[i] = (_Float16)w; hl[i] = (_Float16)(w - (float)hh[i]);
        xh[i] = (_Float16)(x < 0 ? -x : x); xl[i] = (_Float16)((x < 0 ? -x : x) - (float)xh[i]);
    }
    h8 *dah, *dal, *dbh, *dbl; float* dout;
    hipMalloc(&dah, N * 16); hipMalloc(&dal, N * 16); hipMalloc(&dbh, N * 16); hipMalloc(&dbl, N * 16);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipMalloc(&dout, cus * 256 * 4);
    hipMemcpy(dah, hh.data(), N * 16, hipMemcpyHostToDevice); hipMemcpy(dal, hl.data(), N * 16, hipMemcpyHostToDevice);
    hipMemcpy(dbh, xh.data(), N * 16, hipMemcpyHostToDevice); hipMemcpy(dbl, xl.data(), N * 16, hipMemcpyHostToDevice);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"16x16x32  pair: hi.lo lo.hi hi.hi            ", "16x16x32  per a: hi.lo x4, lo.hi x4, hi.hi x4", "16x16x32  pair: hi.lo hi.hi lo.hi            ",
                            "32x32x16  pair: hi.lo lo.hi hi.hi            "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            for (int w = 0; w < 2; ++w) {       // the second launch is the timed one (the first warms clocks and power state up)
                hipEventRecord(e0);
                for (int k = 0; k < 24; ++k) {
                    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(256), 0, 0, dah, dal, dbh, dbl, dout, iters);
                    else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(cus), dim3(256), 0, 0, dah, dal, dbh, dbl, dout, iters);
                    else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(cus), dim3(256), 0, 0, dah, dal, dbh, dbl, dout, iters);
                    else hipLaunchKernelGGL(probe<3>, dim3(cus), dim3(256), 0, 0, dah, dal, dbh, dbl, dout, iters);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double ns_step = ms * 1e6 / (24.0 * iters);
            const double alg = 2.0 * 64 * 64 * 32 * cus * 4 / ns_step / 1e3;      // algorithmic TFLOP/s: one 64 x 64 x 32 product per step and wave
            printf("%s  %.1f ns per 32-k step of a 64 x 64 wave tile = %.0f TFLOP/s algorithmic (x3 issued = %.0f) on %d CUs\n", names[mode], ns_step, alg, 3 * alg, cus);
        }
    return 0;
}
