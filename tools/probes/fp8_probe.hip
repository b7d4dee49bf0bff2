// Probe (run on the GPU box): operand lane map of v_mfma_f32_16x16x32_fp8_fp8 with exact small integers, and
// rounding / saturation of v_cvt_pk_fp8_f32.   hipcc --offload-arch=gfx950 -O2 fp8_probe.hip -o fp8_probe && ./fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void mm(const long* a, const long* b, float* d) {
    int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a[l], b[l], c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) d[((l >> 4) * 4 + j) * 16 + (l & 15)] = c[j];   // row=(l>>4)*4+j, col=l&15
}
__global__ void cvt(const float* in, uint8_t* o8, float* back, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false);
    o8[i] = (uint8_t)(p & 0xff);
    f2 d = __builtin_amdgcn_cvt_pk_f32_fp8(p, false);
    back[i] = d[0];
}
static uint8_t enc_small_int(int v) {   // e4m3fn encoding of integers -8..8 (exact)
    float f = (float)v; if (v == 0) return 0;
    int s = v < 0; float a = fabsf(f); int e = (int)floorf(log2f(a)); int m = (int)roundf((a / exp2f((float)e) - 1.f) * 8.f);
    return (uint8_t)((s << 7) | ((e + 7) << 3) | m);
}
int main() {
    // A[i][k], B[k][j] small integers; lane l holds A[l&15][8*(l>>4)+t], B[8*(l>>4)+t][l&15] for t=0..7 (assumed map)
    int A[16][32], B[32][16];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i][k] = ((i * 7 + k * 3) % 9) - 4;
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k][j] = ((k * 5 + j * 11) % 7) - 3;
    long ha[64], hb[64];
    for (int l = 0; l < 64; ++l) {
        uint64_t pa = 0, pb = 0;
        for (int t = 0; t < 8; ++t) {
            pa |= (uint64_t)enc_small_int(A[l & 15][8 * (l >> 4) + t]) << (8 * t);
            pb |= (uint64_t)enc_small_int(B[8 * (l >> 4) + t][l & 15]) << (8 * t);
        }
        ha[l] = (long)pa; hb[l] = (long)pb;
    }
    long *da, *db; float* dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dd);
    float hd[256]; hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        int ref = 0; for (int k = 0; k < 32; ++k) ref += A[i][k] * B[k][j];
        if (hd[i * 16 + j] != (float)ref) ++bad;
    }
    printf("fp8 mfma lane map (A[l&15][8(l>>4)+t], B[8(l>>4)+t][l&15], D row=(l>>4)*4+j col=l&15): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    float vin[16] = {0.f, 1.f, 1.0625f, 1.1875f, 0.0019f, 0.00098f, 0.0009f, 447.f, 448.f, 460.f, 1000.f, -1000.f, 17.f, 0.3f, -0.07f, 3e-4f};
    float *di, *dbk; uint8_t* d8;
    hipMalloc(&di, 64); hipMalloc(&dbk, 64); hipMalloc(&d8, 16);
    hipMemcpy(di, vin, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, di, d8, dbk, 16);
    float hb2[16]; uint8_t h8[16]; hipMemcpy(hb2, dbk, 64, hipMemcpyDeviceToHost); hipMemcpy(h8, d8, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("cvt %12.6g -> 0x%02x -> %12.6g\n", vin[i], h8[i], hb2[i]);
    return 0;
}
