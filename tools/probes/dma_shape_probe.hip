// L2 -> LDS DMA rate by piece shape (standalone probe, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o dma_shape_probe dma_shape_probe.hip && ./dma_shape_probe
// Every wave streams 1-KiB buffer_load...lds pieces out of a table that stays in its XCD's L2 (2 MB per XCD class), with
// DEPTH pieces in flight, in three shapes:
//   0: 16 rows x 64 B   (what the conv kernels' rings use: a 32-deep K slice of 16 rows, row pitch = K*2 bytes)
//   1:  8 rows x 128 B  (a 64-deep K slice, or hi|lo of a 32-deep slice interleaved in one line)
//   2:  1 row  x 1 KiB  (contiguous)
// Reports GB/s per CU and chip-wide for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int SHAPE, int DEPTH>
__global__ __launch_bounds__(256) void dma_probe(const char* table, int table_bytes, int pitch, int iters, unsigned* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // each XCD class (blockIdx % 8) reads its own 2-MB window, so the data stays in that XCD's L2
    const char* base = table + (size_t)(blockIdx.x & 7) * (table_bytes / 8);
    const int window = table_bytes / 8;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, window, 0x00020000);
    int lane_off;
    if (SHAPE == 0) lane_off = (lane >> 2) * pitch + (lane & 3) * 16;          // 16 rows x 64 B
    else if (SHAPE == 1) lane_off = (lane >> 3) * pitch + (lane & 7) * 16;     //  8 rows x 128 B
    else lane_off = lane * 16;                                                 // 1 KiB contiguous
    char* dst = smem + wave * (DEPTH * 1024);
    // walk the window: piece p of this wave starts at a different row block / column each time
    unsigned pos = (blockIdx.x >> 3) * 7919u + wave * 104729u;
    const int rows_per_piece = SHAPE == 0 ? 16 : (SHAPE == 1 ? 8 : 1);
    const int row_bytes = SHAPE == 0 ? 64 : (SHAPE == 1 ? 128 : 1024);
    const int n_rowblocks = window / (pitch * rows_per_piece) - 1;
    const int n_cols = pitch / row_bytes;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            pos = pos * 1664525u + 1013904223u;
            const int rb = (pos >> 8) % n_rowblocks, col = (pos >> 20) % n_cols;
            const int soff = SHAPE == 2 ? (int)(((pos >> 8) % (window / 1024 - 1)) * 1024) : rb * rows_per_piece * pitch + col * row_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + d * 1024), 16, lane_off, soff, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && threadIdx.x == 0 && blockIdx.x == 0xffffffffu) sink[0] = *(unsigned*)smem;
#endif
}

template <int SHAPE, int DEPTH>
double run(const char* table, int table_bytes, int pitch, int wg_per_cu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    const size_t lds = 4 * DEPTH * 1024;
    hipFuncSetAttribute((const void*)dma_probe<SHAPE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dma_probe<SHAPE, DEPTH>), dim3(grid), dim3(256), lds, 0, table, table_bytes, pitch, iters, nullptr);
    hipEventRecord(e0);
    hipLaunchKernelGGL((dma_probe<SHAPE, DEPTH>), dim3(grid), dim3(256), lds, 0, table, table_bytes, pitch, iters, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 4 * iters * DEPTH * 1024.0;
    return bytes / (ms * 1e-3) / 1e9;      // GB/s chip-wide
}

int main() {
    const int table_bytes = 16 << 20;
    char* table;
    hipMalloc(&table, table_bytes);
    hipMemset(table, 1, table_bytes);
    const int pitch = 2048;        // a 1024-channel fp16 plane row
    printf("L2->LDS DMA (1-KiB pieces, table 2 MB per XCD class, row pitch %d B)\n", pitch);
    printf("%-22s %8s %14s %14s\n", "shape", "WG/CU", "GB/s per CU", "TB/s chip");
    for (int wg = 1; wg <= 4; wg *= 2) {
        const int iters = 4000 / wg;
        double a = run<0, 8>(table, table_bytes, pitch, wg, iters);
        double b = run<1, 8>(table, table_bytes, pitch, wg, iters);
        double c = run<2, 8>(table, table_bytes, pitch, wg, iters);
        printf("%-22s %8d %14.1f %14.2f\n", "16 rows x 64 B", wg, a / 256, a / 1e3);
        printf("%-22s %8d %14.1f %14.2f\n", "8 rows x 128 B", wg, b / 256, b / 1e3);
        printf("%-22s %8d %14.1f %14.2f\n", "1 KiB contiguous", wg, c / 256, c / 1e3);
    }
    for (int wg = 2; wg <= 2; ++wg) {
        double a = run<0, 16>(table, table_bytes, pitch, wg, 1000);
        double b = run<1, 16>(table, table_bytes, pitch, wg, 1000);
        printf("%-22s %8d %14.1f %14.2f   (16 pieces in flight per wave)\n", "16 rows x 64 B", wg, a / 256, a / 1e3);
        printf("%-22s %8d %14.1f %14.2f   (16 pieces in flight per wave)\n", "8 rows x 128 B", wg, b / 256, b / 1e3);
    }
    hipFree(table);
    return 0;
}
