// Bare MFMA loops under the package power limit: what would integer-slice corrections buy an MFMA-bound kernel?
//   f16x3   per (cout fragment, pixel fragment) and 32-k step: W_hi.X_lo + W_lo.X_hi + W_hi.X_hi on v_mfma_f32_16x16x32_f16 (the engine's arithmetic)
//   f16+i8  W_hi.X_hi on v_mfma_f32_16x16x32_f16 and BOTH corrections K-concatenated on ONE v_mfma_i32_16x16x64_i8 ([W_hi8 | W_lo8] . [X_lo8 ; X_hi8])
//   f16x1   the main product alone (for scale)
// One wave per SIMD (256-thread workgroups, launch_bounds(256, 1)), a 64 x 64 wave tile (4 x 4 fragments), operands in registers (random
// data, loaded once: a ReLU-like half of the pixel values zero), every CU busy.  Prints ns per 32-k step of the wave tile and the equivalent
// algorithmic TFLOP/s of the whole chip.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_i8corr_probe mfma_i8corr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const h8* ah, const h8* al, const h8* bh, const h8* bl, const i4* a8, const i4* b8, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    h8 Ah[4], Al[4], Bh[4], Bl[4];
    i4 A8[4], B8[4];
    for (int i = 0; i < 4; ++i) {
        Ah[i] = ah[(t * 4 + i) & 0xffff]; Al[i] = al[(t * 4 + i) & 0xffff];
        Bh[i] = bh[(t * 4 + i) & 0xffff]; Bl[i] = bl[(t * 4 + i) & 0xffff];
        A8[i] = a8[(t * 4 + i) & 0xffff]; B8[i] = b8[(t * 4 + i) & 0xffff];
    }
    f4 acc[4][4];
    i4 iac[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) { acc[a][b] = f4{0, 0, 0, 0}; iac[a][b] = i4{0, 0, 0, 0}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (MODE == 0) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[a], Bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bh[b], acc[a][b], 0, 0, 0);
                } else if (MODE == 1) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bh[b], acc[a][b], 0, 0, 0);
                    iac[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A8[a], B8[b], iac[a][b], 0, 0, 0);
                } else {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[a], Bh[b], acc[a][b], 0, 0, 0);
                }
            }
        // keep the operands opaque so that nothing is hoisted or folded
        asm volatile("" : "+v"(Ah[0]), "+v"(Bh[0]), "+v"(A8[0]), "+v"(B8[0]));
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            for (int j = 0; j < 4; ++j) s += acc[a][b][j] + (float)iac[a][b][j];
    out[t] = s;
}

int main() {
    const int N = 1 << 16;
    std::vector<_Float16> hh(N * 8), hl(N * 8), xh(N * 8), xl(N * 8);
    std::vector<int> w8(N * 4), x8(N * 4);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
    for (int i = 0; i < N * 8; ++i) {
        const float w = rnd() * 700.f, x = rnd() > 0 ? rnd() * 2.f : 0.f;          // weights scaled like the engine's planes; half the pixels zero
        hh[i] = (_Float16)w; hl[i] = (_Float16)(w - (float)hh[i]);
        xh[i] = (_Float16)(x < 0 ? -x : x); xl[i] = (_Float16)((x < 0 ? -x : x) - (float)xh[i]);
    }
    for (int i = 0; i < N * 4; ++i) {
        w8[i] = rand() ^ (rand() << 16);
        const int z = rand();
        x8[i] = (z & 1) ? (rand() ^ (rand() << 16)) : ((rand() & 0xffff) << 16);   // the X_lo8 half random, half of the X_hi8 bytes zero-ish
    }
    h8 *dah, *dal, *dbh, *dbl; i4 *da8, *db8; float* dout;
    hipMalloc(&dah, N * 16); hipMalloc(&dal, N * 16); hipMalloc(&dbh, N * 16); hipMalloc(&dbl, N * 16);
    hipMalloc(&da8, N * 16); hipMalloc(&db8, N * 16);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipMalloc(&dout, cus * 256 * 4);
    hipMemcpy(dah, hh.data(), N * 16, hipMemcpyHostToDevice); hipMemcpy(dal, hl.data(), N * 16, hipMemcpyHostToDevice);
    hipMemcpy(dbh, xh.data(), N * 16, hipMemcpyHostToDevice); hipMemcpy(dbl, xl.data(), N * 16, hipMemcpyHostToDevice);
    hipMemcpy(da8, w8.data(), N * 16, hipMemcpyHostToDevice); hipMemcpy(db8, x8.data(), N * 16, hipMemcpyHostToDevice);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"f16x3  (3 fp16 MFMAs per fragment pair and 32 k)", "f16+i8 (1 fp16 MFMA + 1 int8 MFMA of K = 64)   ", "f16x1  (the main product alone)                 "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 2; ++w) {       // the second launch is the timed one (the first warms clocks and power state up)
                hipEventRecord(e0);
                for (int k = 0; k < 24; ++k) {
                    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(256), 0, 0, dah, dal, dbh, dbl, da8, db8, dout, iters);
                    else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(cus), dim3(256), 0, 0, dah, dal, dbh, dbl, da8, db8, dout, iters);
                    else hipLaunchKernelGGL(probe<2>, dim3(cus), dim3(256), 0, 0, dah, dal, dbh, dbl, da8, db8, dout, iters);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double ns_step = ms * 1e6 / (24.0 * iters);
            const double alg = 2.0 * 64 * 64 * 32 * cus * 4 / ns_step / 1e3;      // algorithmic TFLOP/s: one 64 x 64 x 32 product per step and wave
            printf("%s  %.1f ns per 32-k step of a 64 x 64 wave tile = %.0f TFLOP/s algorithmic on %d CUs\n", names[mode], ns_step, alg, cus);
        }
    return 0;
}
