// Probe (GPU box): v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands.
//  1. operand lane map with exact small integers: lane l holds A[l&15][32*(l>>4) + t], t = 0..31 (assumed)
//  2. where a lane's E8M0 scale byte applies (row l&15, K block l>>4 assumed)
//  3. throughput: per "step" of 64 K on a 64x64 wave tile,  A: 96 x f16 16x16x32  vs  C: 32 x f16 16x16x32 + 16 x fp8 16x16x128
// hipcc --offload-arch=gfx950 -O2 mfma_scale_probe.hip -o mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void mm(const i8v* a, const i8v* b, const int* sa, const int* sb, float* d) {
    int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int j = 0; j < 4; ++j) d[((l >> 4) * 4 + j) * 16 + (l & 15)] = c[j];
}

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void loop(const h8* fa, const i8v* f8, float* out, int iters) {
    const int l = threadIdx.x;
    h8 a[4], b[4];
    i8v a8[4], b8[4];
    for (int i = 0; i < 4; ++i) { a[i] = fa[(l + i * 64) & 1023]; b[i] = fa[(l + 256 + i * 64) & 1023]; a8[i] = f8[(l + i * 64) & 1023]; b8[i] = f8[(l + 300 + i * 64) & 1023]; }
    f4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0, 0, 0, 0};
    const int sc = 0x7f7f7f7f;      // E8M0 127 = 2^0
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, sc, 0, sc);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + l] = s;
}

static uint8_t enc(int v) {   // e4m3fn of small integers
    if (v == 0) return 0;
    int s = v < 0; float a = fabsf((float)v); int e = (int)floorf(log2f(a)); int m = (int)roundf((a / exp2f((float)e) - 1.f) * 8.f);
    return (uint8_t)((s << 7) | ((e + 7) << 3) | m);
}

int main() {
    static int A[16][128], B[128][16];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) A[i][k] = ((i * 7 + k * 3) % 9) - 4;
    for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) B[k][j] = ((k * 5 + j * 11) % 7) - 3;
    std::vector<uint8_t> ha(64 * 32), hb(64 * 32);
    for (int l = 0; l < 64; ++l) for (int t = 0; t < 32; ++t) {
        ha[l * 32 + t] = enc(A[l & 15][32 * (l >> 4) + t]);
        hb[l * 32 + t] = enc(B[32 * (l >> 4) + t][l & 15]);
    }
    i8v *da, *db; int *dsa, *dsb; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    auto run = [&](const int* sa, const int* sb, float* hd) {
        hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    };
    int sa[64], sb[64]; float hd[256];
    for (int l = 0; l < 64; ++l) sa[l] = sb[l] = 127;
    run(sa, sb, hd);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = 0; for (int k = 0; k < 128; ++k) r += A[i][k] * B[k][j]; if (hd[i * 16 + j] != (float)r) ++bad; }
    printf("lane map A[l&15][32(l>>4)+t] B[32(l>>4)+t][l&15], scales 2^0: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    // scale semantics: A scale of lane (row r, block g) = 2^(g) -> expect sum_g 2^g * partial(i, j, block g) if lane's scale applies to (row l&15, block l>>4)
    for (int l = 0; l < 64; ++l) { sa[l] = 127 + (l >> 4); sb[l] = 127; }
    run(sa, sb, hd);
    bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float r = 0; for (int g = 0; g < 4; ++g) { int p = 0; for (int k = 32 * g; k < 32 * g + 32; ++k) p += A[i][k] * B[k][j]; r += p * exp2f((float)g); } if (hd[i * 16 + j] != r) ++bad; }
    printf("A scale byte0 of lane l = scale of (row l&15, K block l>>4): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127 + (l & 15 ? 0 : 3); }      // B scale: column 0 only, all blocks
    run(sa, sb, hd);
    bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = 0; for (int k = 0; k < 128; ++k) r += A[i][k] * B[k][j]; float w = (j == 0) ? 8.f * r : (float)r; if (hd[i * 16 + j] != w) ++bad; }
    printf("B scale byte0 of lane l = scale of (col l&15, K block l>>4): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);

    // throughput
    std::vector<_Float16> hf(1024 * 8); for (auto& x : hf) x = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    std::vector<uint8_t> h8v(1024 * 32); for (auto& x : h8v) x = (uint8_t)(rand() & 0x77);
    h8* dfa; i8v* df8; float* dout;
    hipMalloc(&dfa, hf.size() * 2); hipMalloc(&df8, h8v.size()); hipMalloc(&dout, 512 * 256 * 4);
    hipMemcpy(dfa, hf.data(), hf.size() * 2, hipMemcpyHostToDevice); hipMemcpy(df8, h8v.data(), h8v.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(loop<0>, dim3(512), dim3(256), 0, 0, dfa, df8, dout, iters);
            else hipLaunchKernelGGL(loop<1>, dim3(512), dim3(256), 0, 0, dfa, df8, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // algorithmic flops per iteration per wave: 64x64x64 MACs (one 64-K step of the f16x3 product) = 2*64^3
            double alg = 2.0 * 64 * 64 * 64 * (double)iters * 512 * 4;
            printf("variant %s rep %d: %.2f ms  -> %.1f TFLOP/s algorithmic (f16x3-equivalent), %.0f ns per 64-K step\n",
                   v == 0 ? "A (96 f16)" : "C (32 f16 + 16 fp8x128)", rep, ms, alg / ms / 1e9, ms * 1e6 / iters);
        }
    }
    return 0;
}
