// Probe (GPU box): which (row/col, K block) a lane's E8M0 scale byte applies to in v_mfma_scale_f32_16x16x128_f8f6f4.
// One lane at a time gets scale 2^3 (others 2^0); A = B = all ones, so D[i][j] = sum over the 4 K blocks of 32 * (block scales).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
__global__ void mm(const i8v* a, const i8v* b, const int* sa, const int* sb, float* d, int opsel) {
    int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 1, sa[l], 1, sb[l]);
    for (int j = 0; j < 4; ++j) d[((l >> 4) * 4 + j) * 16 + (l & 15)] = c[j];
}
int main() {
    // A[i][k] = 1 for k in block g -> value (g+1) to tell blocks apart: A[i][k] = 1, B[k][j] = g+1 (exact in e4m3: 1,2,3,4)
    const uint8_t one = 0x38, enc[4] = {0x38, 0x40, 0x44, 0x48};
    std::vector<uint8_t> ha(64 * 32, one), hb(64 * 32);
    for (int l = 0; l < 64; ++l) for (int t = 0; t < 32; ++t) hb[l * 32 + t] = enc[l >> 4];
    i8v *da, *db; int *dsa, *dsb; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    float base[256], hd[256];
    int s1[64], s0[64];
    for (int l = 0; l < 64; ++l) s0[l] = 0x7f7f7f7f;
    for (int which = 0; which < 2; ++which) for (int opsel = 0; opsel < 2; ++opsel) {
        hipMemcpy(dsa, s0, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, s0, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, opsel);
        hipMemcpy(base, dd, 1024, hipMemcpyDeviceToHost);
        printf("%s scale, opsel %d, base D[0][0]=%g (expect 32*(1+2+3+4)=320)\n", which ? "B" : "A", opsel, base[0]);
        for (int L = 0; L < 64; ++L) {
            for (int l = 0; l < 64; ++l) s1[l] = 0x7f7f7f7f;
            s1[L] = opsel == 0 ? 0x7f7f7f82 : 0x7f7f827f;        // byte opsel = 127+3
            hipMemcpy(which ? dsb : dsa, s1, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, opsel);
            hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
            // which rows/cols changed, and by how much: delta = 7 * 32 * (g+1) for block g
            int rows = 0, cols = 0; float delta = 0; int cnt = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (hd[i * 16 + j] != base[i * 16 + j]) { rows |= 1 << i; cols |= 1 << j; delta = hd[i * 16 + j] - base[i * 16 + j]; ++cnt; }
            if (L < 20 || L % 16 == 0 || L == 63) printf("  lane %2d: %3d outputs changed, rows mask %04x cols mask %04x, delta %g (= 224 * (block+1): block %g)\n", L, cnt, rows, cols, delta, delta / 224 - 1);
        }
    }
    return 0;
}
