// Probe (GPU box): v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3): which of a lane's 32 operand bytes belong to which
// 32-wide K block, which lane's scale byte scales which (row, block); throughput of the f16 32x32x16 + fp8 32x32x64 mix.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void mm(const i8v* a, const i8v* b, const int* sa, const int* sb, float* d) {
    int l = threadIdx.x;
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void loop(const h8* fa, const i8v* f8, float* out, int iters) {
    const int l = threadIdx.x;
    h8 a[4], b[4];          // 2 row blocks x 2 k halves
    i8v a8[2], b8[2];
    for (int i = 0; i < 4; ++i) { a[i] = fa[(l + i * 64) & 1023]; b[i] = fa[(l + 256 + i * 64) & 1023]; }
    for (int i = 0; i < 2; ++i) { a8[i] = f8[(l + i * 64) & 1023]; b8[i] = f8[(l + 300 + i * 64) & 1023]; }
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    const int sc = 0x7f7f7f7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * i], b[2 * j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * i + 1], b[2 * j + 1], acc[i][j], 0, 0, 0);
                if (VARIANT == 1) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, sc, 0, sc);
                else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * i], b[2 * j + 1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * i + 1], b[2 * j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * i], b[2 * j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * i + 1], b[2 * j + 1], acc[i][j], 0, 0, 0);
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 256 + l] = s;
}
int main() {
    const uint8_t one = 0x38, enc[4] = {0x38, 0x40, 0x44, 0x48};     // 1, 2, 3, 4
    std::vector<uint8_t> ha(64 * 32, one), hb(64 * 32);
    // B byte t of lane l: value index = 2*(l>>5) + (t>>4)   -> 1: (h0, first 16 B), 2: (h0, second), 3: (h1, first), 4: (h1, second)
    for (int l = 0; l < 64; ++l) for (int t = 0; t < 32; ++t) hb[l * 32 + t] = enc[2 * (l >> 5) + (t >> 4)];
    i8v *da, *db; int *dsa, *dsb; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 4096);
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    float base[1024], hd[1024];
    int s1[64], s0[64];
    for (int l = 0; l < 64; ++l) s0[l] = 0x7f7f7f7f;
    for (int which = 0; which < 2; ++which) {
        hipMemcpy(dsa, s0, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, s0, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        hipMemcpy(base, dd, 4096, hipMemcpyDeviceToHost);
        printf("%s scale: base D[0][0]=%g (expect 16*(1+2+3+4)=160)\n", which ? "B" : "A", base[0]);
        for (int L = 0; L < 64; L += (L < 4 ? 1 : 7)) {
            for (int l = 0; l < 64; ++l) s1[l] = 0x7f7f7f7f;
            s1[L] = 0x7f7f7f82;
            hipMemcpy(which ? dsb : dsa, s1, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost);
            int cnt = 0, row = -1, col = -1; float delta = 0;
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (hd[i * 32 + j] != base[i * 32 + j]) { ++cnt; row = i; col = j; delta = hd[i * 32 + j] - base[i * 32 + j]; }
            printf("  lane %2d (r %2d, h %d): %3d outputs changed (last row %2d col %2d), delta/7/16 = %g   [1+3: first halves, 2+4: second halves, 1+2: h0, 3+4: h1]\n", L, L & 31, L >> 5, cnt, row, col, delta / 7 / 16);
        }
    }
    std::vector<_Float16> hf(1024 * 8); for (auto& x : hf) x = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    std::vector<uint8_t> h8v(1024 * 32); for (auto& x : h8v) x = (uint8_t)(rand() & 0x77);
    h8* dfa; i8v* df8; float* dout;
    hipMalloc(&dfa, hf.size() * 2); hipMalloc(&df8, h8v.size()); hipMalloc(&dout, 512 * 256 * 4);
    hipMemcpy(dfa, hf.data(), hf.size() * 2, hipMemcpyHostToDevice); hipMemcpy(df8, h8v.data(), h8v.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    for (int v = 0; v < 2; ++v) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(loop<0>, dim3(512), dim3(256), 0, 0, dfa, df8, dout, iters);
        else hipLaunchKernelGGL(loop<1>, dim3(512), dim3(256), 0, 0, dfa, df8, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double alg = 2.0 * 64 * 64 * 32 * (double)iters * 512 * 4;
        printf("variant %s rep %d: %.2f ms -> %.1f TFLOP/s algorithmic (f16x3-equivalent), %.0f ns per 32-K step\n",
               v == 0 ? "A32 (24 f16 32x32x16)" : "B (8 f16 32x32x16 + 4 fp8 32x32x64)", rep, ms, alg / ms / 1e9, ms * 1e6 / iters);
    }
    return 0;
}
