// HBM rate of the conv epilogue's traffic by activation layout (standalone probe, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o epilogue_shape_probe epilogue_shape_probe.hip && ./epilogue_shape_probe
// One workgroup = one 128-pixel x 128-channel tile of a [M][C] tensor pair (hi / lo fp16 planes): read the residual tile,
// add, write the output tile -- nothing else (no K loop).  LAYOUT 0 = NHWC as the engine stores it (a wave instruction covers
// 4 pixels x 256 B); LAYOUT 1 = blocked [M/16][C/32][16][32] (a wave instruction covers one contiguous 1-KiB block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int LAYOUT>
__global__ __launch_bounds__(512) void epi_probe(const _Float16* r_hi, const _Float16* r_lo, _Float16* y_hi, _Float16* y_lo, int M, int C, int n_tiles_c) {
    const int tid = threadIdx.x;
    const int nb = gridDim.x, b = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int mt = L / n_tiles_c, nt = L - mt * n_tiles_c;
    const int m0 = mt * 128, n0 = nt * 128;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        size_t o;
        if (LAYOUT == 0) {
            const int g = tid & 15, prow = tid >> 4;              // 16 threads per pixel, 32 pixels per iteration
            const int pix = m0 + it * 32 + prow;
            if (pix >= M) continue;
            o = (size_t)pix * C + n0 + g * 8;
        } else {
            const int q = tid & 3, p = (tid >> 2) & 15, blk = tid >> 6;      // a wave = one 16-pixel x 32-channel block
            const int pb = it * 2 + (blk >> 2), cb = blk & 3;                  // 8 pixel blocks x 4 channel blocks per tile
            const int pix = m0 + pb * 16 + p;
            if (pix >= M) continue;
            o = (((size_t)(pix >> 4) * (C >> 5) + ((n0 >> 5) + cb)) * 16 + (pix & 15)) * 32 + q * 8;
        }
        h8 a = __builtin_nontemporal_load((const h8*)(r_hi + o));
        h8 c = __builtin_nontemporal_load((const h8*)(r_lo + o));
        h8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = fmaxf((float)a[j] + (float)c[j] + 0.5f, 0.f);
            oh[j] = (_Float16)v;
            ol[j] = (_Float16)(v - (float)oh[j]);
        }
        __builtin_nontemporal_store(oh, (h8*)(y_hi + o));
        __builtin_nontemporal_store(ol, (h8*)(y_lo + o));
    }
}

int main() {
    struct Case { int M, C; const char* name; } cases[] = {{2048 * 196, 1024, "14x14 x 1024 ch (256->1024)"}, {2048 * 3136, 256, "56x56 x 256 ch (64->256)"},
                                                           {2048 * 784, 512, "28x28 x 512 ch (128->512)"}};
    for (auto& cs : cases) {
        const size_t n = (size_t)cs.M * cs.C;
        _Float16 *rh, *rl, *yh, *yl;
        hipMalloc(&rh, n * 2); hipMalloc(&rl, n * 2); hipMalloc(&yh, n * 2); hipMalloc(&yl, n * 2);
        hipMemset(rh, 0, n * 2); hipMemset(rl, 0, n * 2);
        const int n_tiles_c = cs.C / 128, n_tiles_p = (cs.M + 127) / 128;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int layout = 0; layout < 2; ++layout) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (layout == 0) hipLaunchKernelGGL(epi_probe<0>, dim3(n_tiles_p * n_tiles_c), dim3(512), 0, 0, rh, rl, yh, yl, cs.M, cs.C, n_tiles_c);
                else hipLaunchKernelGGL(epi_probe<1>, dim3(n_tiles_p * n_tiles_c), dim3(512), 0, 0, rh, rl, yh, yl, cs.M, cs.C, n_tiles_c);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("%-30s %-8s %8.3f ms  %6.2f TB/s (%.2f GB read + %.2f GB written)\n", cs.name, layout ? "blocked" : "NHWC", best, n * 8 / best / 1e9, n * 4 / 1e9, n * 4 / 1e9);
        }
        hipFree(rh); hipFree(rl); hipFree(yh); hipFree(yl);
    }
    return 0;
}
