// mpx_conv256.h -- 1x1 stride-1 conv + BN + residual + ReLU on a 256(cout) x 256(pixel) tile (f16x3 arithmetic of mpx_conv.h).
//
// Why: the per-workgroup timelines (tools/probes/conv_timeline.py) show the K loop of the 128x128 tiles running at
// 1.1-1.7 us per 32-deep step where the MFMAs need 0.4-0.8 us: every step waits for an LDS-DMA round trip, and what a CU
// can keep in flight is bounded by its LDS (throughput ~ bytes in flight x MACs per byte).  A 256x256 tile does twice the
// MACs per operand byte of a 128x128 one: 64 KB per K step for 768 MFMAs per SIMD (3072 cycles ~ 1.6 us, about one DMA
// round trip), so one stage in flight is enough.
//
// Workgroup = 8 waves as 2 (cout) x 4 (pixel); wave tile 128 cout x 64 pixels = acc[8][4] (128 VGPRs).  LDS: two stages of
// [W_hi 256 x 64 B | W_lo | X_hi 256 x 64 B | X_lo] = 2 x 64 KB.  A K step is four QUADRANTS of 24 MFMAs (4 cout fragments x 2
// pixel fragments x 3 products); fragment registers are A0, A1 (4 fragments x hi/lo each) and B0, B1 (2 x hi/lo): 96 VGPRs.
// The quadrant order snakes so that every quadrant changes ONE operand half and the half it needs next is read from LDS a
// whole quadrant (384 cycles) earlier:
//     even step:  (A0,B0) [read B1]  (A0,B1) [read A1]  | barrier |  (A1,B1) [read A0']  (A1,B0) [read B1']
//     odd  step:  (A0,B1) [read B0]  (A0,B0) [read A1]  | barrier |  (A1,B0) [read A0']  (A1,B1) [read B0']
// (primes = next stage).  The ONE barrier per step sits mid-step: behind it every wave has finished reading stage s (so its
// slot is refilled with stage s+2 right away, 8 one-KiB pieces per wave in the next quadrant) and stage s+1
// has landed (each wave waited for its own pieces with vmcnt(0) just before): a stage has one full step to arrive.
// Epilogue: the ring is free; the two cout halves go through it one after the other as an fp32 [256 px][128 cout] tile
// (XOR-swizzled 16-B chunks), read back as 8 channels per thread: whole 256-B runs per pixel row, as mpx_conv.h.
#pragma once
#include "mpx_conv.h"

namespace mpx {

struct Conv256 {
    static constexpr int TC = 256, TP = 256, NW = 8, NT = 512;
    static constexpr int STAGE = 65536;                 // [W_hi | W_lo | X_hi | X_lo], 16 KB each
    static constexpr int OFF_WHI = 0, OFF_WLO = 16384, OFF_XHI = 32768, OFF_XLO = 49152;
    static constexpr int LDS = 2 * STAGE;
};

__global__ __launch_bounds__(512, 2) void conv256_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Conv256 C;
    MPX_STAMP(t_start);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int L;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    }
    const int mt = L / p.n_tiles_c;
    const int nt = L - mt * p.n_tiles_c;
    const int m0 = mt * C::TP, n0 = nt * C::TC;
    const int K = p.ktot, nk = K >> 5;

    // ---- DMA: 1x1 stride 1, so pixel m's K vector is the row m of [M][K]; descriptors start at the tile's first row --------
    const int prow = lane >> 2;
    const int src_q = ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 16;        // swizzled 16-B chunk of the 64-B K slice (bytes)
    constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t x_hi, x_lo, w_hi, w_lo;
    {
        const long long xrem = ((long long)p.M - m0) * K * 2;              // rows >= M are out of range: zeros
        const int xrec = xrem > 0x7fffffffLL ? 0x7fffffff : (int)xrem;
        x_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_hi + (size_t)m0 * K), 0, xrec, 0x00020000);
        x_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_lo + (size_t)m0 * K), 0, xrec, 0x00020000);
        const int wrec = C::TC * K * 2;
        w_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * K), 0, wrec, 0x00020000);
        w_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * K), 0, wrec, 0x00020000);
    }
    // this wave moves rows [wave*32, wave*32+32) of each of the four planes: two 16-row pieces per plane
    const int roff0 = ((wave * 32 + prow) * K) * 2 + src_q;
    const int roff1 = roff0 + 16 * K * 2;
    const int w_lane = lane * 16;            // W planes are piece-major (w_packed_index): byte lane*16 of the piece, its position in the soffset
    auto dma_piece = [&](int slot, int ks, int which) {     // which = 0..7: W_hi p0, W_lo p0, W_hi p1, W_lo p1, X_hi p0, X_lo p0, X_hi p1, X_lo p1
        char* sb = smem + slot * C::STAGE;
        const int soff = ks * 64;
        const int dead = ks < nk ? 0 : (int)OOB;
        const int pc = (which >> 1) & 1;
        const int voff = (pc ? roff1 : roff0) | dead;
        const int d = (wave * 2 + pc) * 1024;
        const int wvoff = w_lane | dead;
        const int wsoff = ks * 1024 + (wave * 2 + pc) * 16 * K * 2;
        switch (which & 5) {
            case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_hi, MPX_LDS_PTR(sb + C::OFF_WHI + d), 16, wvoff, wsoff, 0, 0); break;
            case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_lo, MPX_LDS_PTR(sb + C::OFF_WLO + d), 16, wvoff, wsoff, 0, 0); break;
            case 4: __builtin_amdgcn_raw_ptr_buffer_load_lds(x_hi, MPX_LDS_PTR(sb + C::OFF_XHI + d), 16, voff, soff, 0, 0); break;
            default: __builtin_amdgcn_raw_ptr_buffer_load_lds(x_lo, MPX_LDS_PTR(sb + C::OFF_XLO + d), 16, voff, soff, 0, 0); break;
        }
    };

    // ---- fragments ------------------------------------------------------------------------------------------------------
    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = (wr * 128 + lrow) * 64 + qsw;          // + half*4096 + f*1024
    const int b_off = (wc * 64 + lrow) * 64 + qsw;           // + half*2048 + f*1024
    struct AH { h8 hi[4], lo[4]; };
    struct BH { h8 hi[2], lo[2]; };
    AH A0, A1;
    BH B0, B1;
    f4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
    auto read_a = [&](AH& r, int slot, int half, int j) {        // j = 0..7: hi f0..3, lo f0..3
        const char* s = smem + slot * C::STAGE + (j < 4 ? C::OFF_WHI : C::OFF_WLO) + a_off + half * 4096 + (j & 3) * 1024;
        if (j < 4) r.hi[j] = *(const h8*)s;
        else r.lo[j - 4] = *(const h8*)s;
    };
    auto read_b = [&](BH& r, int slot, int half, int j) {        // j = 0..3: hi f0,f1, lo f0,f1
        const char* s = smem + slot * C::STAGE + (j < 2 ? C::OFF_XHI : C::OFF_XLO) + b_off + half * 2048 + (j & 1) * 1024;
        if (j < 2) r.hi[j] = *(const h8*)s;
        else r.lo[j - 2] = *(const h8*)s;
    };
    // one MFMA of quadrant (ah, bh): i = 0..23 in (a, term, b) order
    auto mfma_q = [&](const AH& a, int ah, const BH& b, int bh, int i) {
        const int fa = i / 6, r = i % 6, term = r >> 1, fb = r & 1;
        f4& d = acc[ah * 4 + fa][bh * 2 + fb];
        if (term == 0) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.lo[fb], d, 0, 0, 0);
        else if (term == 1) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo[fa], b.hi[fb], d, 0, 0, 0);
        else d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.hi[fb], d, 0, 0, 0);
    };
    // A quadrant: 24 MFMAs with `nread` fragment reads (after MFMAs 1, 4, 7, ...) and `ndma` DMA pieces (after MFMAs 2, 5, 8,
    // ...) hand-placed between them; everything fenced, as in mpx_conv.h.
    auto quadrant = [&](const AH& a, int ah, const BH& b, int bh, auto&& reader, int nread, auto&& dma, int ndma) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            mfma_q(a, ah, b, bh, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i % 3 == 1 && i / 3 < nread) {
                reader(i / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i % 3 == 2 && i / 3 < ndma) {
                dma(i / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto none = [](int) {};

    // ---- prologue: stages 0 and 1 ----------------------------------------------------------------------------------------
#pragma unroll
    for (int w = 0; w < 8; ++w) dma_piece(0, 0, w);
#pragma unroll
    for (int w = 0; w < 8; ++w) dma_piece(1, 1, w);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // stage 0 (this wave's pieces); stage 1 stays in flight
    __builtin_amdgcn_s_barrier();
    MPX_STAMP(t_pro);
#pragma unroll
    for (int j = 0; j < 8; ++j) read_a(A0, 0, 0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_b(B0, 0, 0, j);

    // mid-step rendezvous: this wave's reads of stage s have returned, its pieces of stage s+1 have landed; after the barrier
    // that holds for every wave
    auto mid = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int ks = 0; ks < nk; ks += 2) {
        // ---- even step ks: stage in slot 0, next stage in slot 1 -------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quadrant(A0, 0, B0, 0, [&](int j) { read_b(B1, 0, 1, j); }, 4, none, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quadrant(A0, 0, B1, 1, [&](int j) { read_a(A1, 0, 1, j); }, 8, none, 0);
        mid();
        quadrant(A1, 1, B1, 1, [&](int j) { read_a(A0, 1, 0, j); }, 8, [&](int w) { dma_piece(0, ks + 2, w); }, 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quadrant(A1, 1, B0, 0, [&](int j) { read_b(B1, 1, 1, j); }, 4, none, 0);
        // ---- odd step ks+1: stage in slot 1, next stage in slot 0 ------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quadrant(A0, 0, B1, 1, [&](int j) { read_b(B0, 1, 0, j); }, 4, none, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quadrant(A0, 0, B0, 0, [&](int j) { read_a(A1, 1, 1, j); }, 8, none, 0);
        mid();
        quadrant(A1, 1, B0, 0, [&](int j) { read_a(A0, 0, 0, j); }, 8, [&](int w) { dma_piece(1, ks + 3, w); }, 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quadrant(A1, 1, B1, 1, [&](int j) { read_b(B0, 0, 0, j); }, 4, none, 0);
    }
    // the trailing dead DMAs and the last (unused) fragment reads must be over before the ring is reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MPX_STAMP(t_kend);
    MPX_STAMP(t_epi);

    // ---- epilogue: the two cout halves through the ring as an fp32 [256 px][128 cout] tile --------------------------------
    constexpr int RP = 128 * 4;                 // row pitch of the fp32 tile
    const int g = tid & 15;                     // 16 threads per pixel row, 8 channels each
    const int prow2 = tid >> 4;                 // 32 pixel rows per iteration, 8 iterations
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
        const int co8 = n0 + c * 128 + g * 8;
        const bool co_ok = co8 < p.cout;
        h8 rh[8], rl[8];
        if (p.r_hi) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int pix = m0 + it * 32 + prow2;
                if (co_ok && pix < p.M) {
                    const size_t o = (size_t)pix * p.cout + co8;
                    rh[it] = __builtin_nontemporal_load((const h8*)(p.r_hi + o));
                    rl[it] = __builtin_nontemporal_load((const h8*)(p.r_lo + o));
                }
            }
        }
        __syncthreads();                        // the previous half has been read out (c = 1); K loop reads are over (c = 0)
        if (wr == c) {
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int col = a * 16 + (lane >> 4) * 4;                      // cout within the 128-wide half
                const f4 sc = *(const f4*)(p.scale + n0 + c * 128 + col);
                const f4 sh = *(const f4*)(p.shift + n0 + c * 128 + col);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int pl = wc * 64 + b * 16 + lrow;
                    const f4 v = acc[a][b] * sc + sh;
                    *(f4*)(smem + pl * RP + (((col >> 2) ^ (pl & 7)) << 4)) = v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pl = it * 32 + prow2;
            const int pix = m0 + pl;
            if (!(co_ok && pix < p.M)) continue;
            const f4 v0 = *(const f4*)(smem + pl * RP + (((2 * g) ^ (pl & 7)) << 4));
            const f4 v1 = *(const f4*)(smem + pl * RP + (((2 * g + 1) ^ (pl & 7)) << 4));
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.r_hi) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)rh[it][j] + (float)rl[it][j];
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            h8 oh, ol;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t hi, lo;
                split_f32(v[j], hi, lo);
                oh[j] = hi;
                ol[j] = lo;
            }
            const size_t o = (size_t)pix * p.cout + co8;
            __builtin_nontemporal_store(oh, (h8*)(p.y_hi + o));
            __builtin_nontemporal_store(ol, (h8*)(p.y_lo + o));
        }
    }
    MPX_STAMP_WRITE(p, t_start, t_pro, t_kend, t_epi);
#endif
}

}  // namespace mpx
