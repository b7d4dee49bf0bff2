// mpx_convx.h -- EXPANDING 1x1 stride-1 conv + BN + residual + ReLU (the last conv of a bottleneck: 256->1024, 128->512,
// 512->2048) as ONE persistent workgroup per CU with a continuous K pipeline (f16x3 arithmetic of mpx_conv.h).
//
// What bounds these layers in mpx_conv.h's kernel (tools/probes/conv_timeline.py): K is short (8 steps for
// 256->1024), every step waits for an LDS-DMA round trip (two-stage ring = one step of lookahead), and per tile a prologue
// round trip, a residual round trip and ~2.4 us of dispatch gap are exposed; the 256x256 kernel (mpx_conv256.h) halves the
// operand bytes but has nothing to cover its long epilogue with.  Here:
//   * tile 256 (cout) x 128 (pixels), 8 waves as 4 x 2, wave tile 64 x 64 (acc 64 VGPRs);
//   * THREE ring stages of 48 KB (144 KB): a stage has two steps to land, so a 32-deep K step is paced by the MFMAs;
//   * the ring never drains: stage numbers run on across the tiles a workgroup walks (fixed cout tile, pixel tiles G apart),
//     so the first stages of the next tile are already in LDS when the current tile ends -- no prologue, no dispatch gap;
//   * K step = the quadrant snake of mpx_conv256.h (fragment halves A0/A1, B0/B1 = 64 VGPRs, one barrier per step);
//   * the epilogue works from the accumulator registers (v_permlane16_swap + DPP row_ror:8 regrouping into full 128-B lines,
//     tools/probes/experimental/mpx_convp.h) and needs no LDS and no barrier; its residual lines are requested two K steps before the tile ends, into
//     registers that only the epilogue uses, and its stores retire under the next tile's K loop.
// vmcnt bookkeeping (loads, LDS-DMAs and stores retire in issue order; every wave issues the same instructions -- masked lanes
// carry an out-of-range offset): at the mid-step rendezvous of step s the stage s+1 must have landed; the instructions issued
// after its pieces are the pieces of stage s+2 (6) plus, near a tile boundary, the 16 epilogue loads or the 16 epilogue stores:
// the wait count depends on the step's position in the tile only (WAIT_* below).
#pragma once
#include "mpx_conv.h"

namespace mpx {

// timing-only ablations (probe builds, wrong results; tools/ablate_convx.sh): CX_ABL bit 0 = pixel pieces, 1 = weight pieces, 2 = residual
// loads, 3 = output stores carry an out-of-range offset -- still issued and counted, no memory access
#ifndef CX_ABL
#define CX_ABL 0
#endif

struct ConvX {
    static constexpr int TC = 256, TP = 128, NW = 8, NT = 512;
    static constexpr int STAGE = 49152;                 // [W_hi 16 KB | W_lo 16 KB | X_hi 8 KB | X_lo 8 KB]
    static constexpr int OFF_WHI = 0, OFF_WLO = 16384, OFF_XHI = 32768, OFF_XLO = 40960;
    static constexpr int NS = 3;
    static constexpr int OFF_SCALE = NS * STAGE;        // f32[256] scale, f32[256] shift of the workgroup's cout tile
    static constexpr int LDS = NS * STAGE + 2048;
    static constexpr int PIECES = 6;                    // DMA instructions per wave and stage
    static constexpr int EPI_LOADS = 16;                // residual: 4 pixel fragments x 2 lines x hi/lo
    static constexpr int EPI_STORES = 16;
};

__global__ __launch_bounds__(512, 2) void convx_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef ConvX C;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int K = p.ktot, nk = K >> 5;
    constexpr unsigned OOB = 0x80000000u;

    // ---- the tiles of this workgroup: logical ids v0, v0 + G, v0 + 2G, ... (cout tile fastest, so nt is the same for all) ----
    const int G = gridDim.x;                                           // a multiple of 8 and of n_tiles_c (host)
    const int v0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);    // blocks of one XCD walk neighbouring tiles
    const int n0 = (v0 % p.n_tiles_c) * C::TC;
    const int mt0 = v0 / p.n_tiles_c, mt_step = G / p.n_tiles_c;
    const int n_mt = (p.M + C::TP - 1) / C::TP;
    const int my_tiles = mt0 < n_mt ? (n_mt - 1 - mt0) / mt_step + 1 : 0;
    if (my_tiles == 0) return;

    // ---- DMA ---------------------------------------------------------------------------------------------------------------
    const int prow = lane >> 2;
    const int src_q = ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 16;
    __amdgpu_buffer_rsrc_t w_hi, w_lo, x_hi, x_lo;
    {
        const int wrec = C::TC * K * 2;
        w_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * K), 0, wrec, 0x00020000);
        w_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * K), 0, wrec, 0x00020000);
    }
    const int w_lane = lane * 16;                                       // W planes are piece-major (w_packed_index); rows [wave*32, +32): two pieces per plane
    const int xrow = ((wave * 16 + prow) * K) * 2 + src_q;             // X rows [wave*16, +16): one piece per plane
    int f_tile = 0, f_ks = 0, f_slot = 0;                               // the next stage to issue: tile index, K step, ring slot
    int f_dead = 0;
    auto set_x_desc = [&](int ti) {
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * K * 2;
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (rem < 0 ? 0 : (int)rem);
        x_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_hi + (size_t)m0 * K), 0, rec, 0x00020000);
        x_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_lo + (size_t)m0 * K), 0, rec, 0x00020000);
    };
    set_x_desc(0);
    auto dma_piece = [&](int which) {           // 0..5: W_hi p0, W_lo p0, W_hi p1, W_lo p1, X_hi, X_lo of the stage (f_tile, f_ks)
        char* sb = smem + f_slot * C::STAGE;
        const int soff = f_ks * 64;
        if (which < 4) {
            const int pc = which >> 1;
            const int voff = w_lane | f_dead | ((CX_ABL & 2) ? (int)OOB : 0);
            const int d = (wave * 2 + pc) * 1024;
            const int wsoff = f_ks * 1024 + (wave * 2 + pc) * 16 * K * 2;
            if ((which & 1) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_hi, MPX_LDS_PTR(sb + C::OFF_WHI + d), 16, voff, wsoff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_lo, MPX_LDS_PTR(sb + C::OFF_WLO + d), 16, voff, wsoff, 0, 0);
        } else {
            const int voff = xrow | f_dead | ((CX_ABL & 1) ? (int)OOB : 0);
            const int d = wave * 1024;
            if (which == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_hi, MPX_LDS_PTR(sb + C::OFF_XHI + d), 16, voff, soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(x_lo, MPX_LDS_PTR(sb + C::OFF_XLO + d), 16, voff, soff, 0, 0);
        }
    };
    auto next_fill = [&]() {                    // advance (f_tile, f_ks, f_slot) after a stage has been issued
        f_slot = f_slot == C::NS - 1 ? 0 : f_slot + 1;
        f_ks += 1;
        if (f_ks == nk) {
            f_ks = 0;
            f_tile += 1;
            if (f_tile < my_tiles) set_x_desc(f_tile);
            else f_dead = (int)OOB;             // past the last tile: the pieces still count, but touch no memory
        }
    };

    // ---- fragments ---------------------------------------------------------------------------------------------------------
    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = (wr * 64 + lrow) * 64 + qsw;            // + half*2048 + f*1024
    const int b_off = (wc * 64 + lrow) * 64 + qsw;
    struct FH { h8 hi[2], lo[2]; };
    FH A0, A1, B0, B1;
    f4 acc[4][4];
    auto read_a = [&](FH& r, int slot, int half, int j) {      // j = 0..3: hi f0, hi f1, lo f0, lo f1
        const char* s = smem + slot * C::STAGE + (j < 2 ? C::OFF_WHI : C::OFF_WLO) + a_off + half * 2048 + (j & 1) * 1024;
        if (j < 2) r.hi[j] = *(const h8*)s;
        else r.lo[j - 2] = *(const h8*)s;
    };
    auto read_b = [&](FH& r, int slot, int half, int j) {
        const char* s = smem + slot * C::STAGE + (j < 2 ? C::OFF_XHI : C::OFF_XLO) + b_off + half * 2048 + (j & 1) * 1024;
        if (j < 2) r.hi[j] = *(const h8*)s;
        else r.lo[j - 2] = *(const h8*)s;
    };
    auto mfma_q = [&](const FH& a, int ah, const FH& b, int bh, int i) {      // i = 0..11 in (a, term, b) order
        const int fa = i / 6, r = i % 6, term = r >> 1, fb = r & 1;
        f4& d = acc[ah * 2 + fa][bh * 2 + fb];
        if (term == 0) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.lo[fb], d, 0, 0, 0);
        else if (term == 1) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo[fa], b.hi[fb], d, 0, 0, 0);
        else d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.hi[fb], d, 0, 0, 0);
    };
    // a quadrant: 12 MFMAs, four fragment reads (after MFMAs 1, 4, 7, 10) and up to six DMA pieces (after MFMAs 0, 2, 5, 6, 8, 11)
    auto quadrant = [&](const FH& a, int ah, const FH& b, int bh, auto&& reader, bool with_dma) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            mfma_q(a, ah, b, bh, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i % 3 == 1) {
                reader(i / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (with_dma && i % 3 != 1 && i < 9 + 0) {      // i = 0, 2, 3, 5, 6, 8 -> pieces 0..5
                dma_piece(i - (i + 1) / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- epilogue state: lane geometry of the regrouped 16-B chunks (as tools/probes/experimental/mpx_convp.h) -------------------------------------------
    const int erow = lane >> 4;
    const bool lo8 = (lane & 8) == 0;
    int offA[4];
    {
        const int co = n0 + wr * 64 + (2 * (lo8 ? 0 : 1) + (erow & 1)) * 16 + (erow >> 1) * 8;
        const int dead = (p.cout - 1 - co) & (int)OOB;
#pragma unroll
        for (int b = 0; b < 4; ++b) offA[b] = ((wc * 64 + b * 16 + (lane & 7)) * p.cout + co) * 2 | dead;
    }
    const int row8 = 8 * p.cout * 2;
    u4 rh[4][2], rl[4][2];
    auto ror8 = [](float old, float src, auto mask_tag) {
        constexpr int MASK = decltype(mask_tag)::value;
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, MASK, false));
    };
    // C::EPI_LOADS loads of tile `ti`: its residual lines
    auto issue_epilogue_loads = [&](int ti) {
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = p.r_hi ? (rem > 0x7fffffffLL ? 0x7fffffff : (int)rem) : 0;          // no residual: every lane out of range
        const half_t* rbh = p.r_hi ? p.r_hi : p.y_hi;
        const half_t* rbl = p.r_hi ? p.r_lo : p.y_lo;
        const __amdgpu_buffer_rsrc_t r_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(rbh + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(rbl + (size_t)m0 * p.cout), 0, rec, 0x00020000);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rh[b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_hi_rs, (offA[b] + k * row8) | ((CX_ABL & 4) ? (int)OOB : 0), 0, 2);
                rl[b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_lo_rs, (offA[b] + k * row8) | ((CX_ABL & 4) ? (int)OOB : 0), 0, 2);
            }
    };
    auto epilogue = [&](int ti) {               // C::EPI_STORES stores
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (int)rem;
        const __amdgpu_buffer_rsrc_t y_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_hi + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t y_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_lo + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        f4 sc[2][2], sh[2][2];                  // scale / shift of this lane's channels in the accumulator layout, from LDS
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = wr * 64 + (2 * q + (erow & 1)) * 16 + (erow >> 1) * 8;
            sc[q][0] = *(const f4*)(smem + C::OFF_SCALE + col * 4);
            sc[q][1] = *(const f4*)(smem + C::OFF_SCALE + col * 4 + 16);
            sh[q][0] = *(const f4*)(smem + C::OFF_SCALE + 1024 + col * 4);
            sh[q][1] = *(const f4*)(smem + C::OFF_SCALE + 1024 + col * 4 + 16);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float ve[8], vo[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e0 = acc[0][b][j], e1 = acc[1][b][j], o0 = acc[2][b][j], o1 = acc[3][b][j];
                const auto se = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
                const auto so = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0), __float_as_uint(o1), false, false);
                ve[j] = __uint_as_float((unsigned)se[0]) * sc[0][0][j] + sh[0][0][j];
                ve[4 + j] = __uint_as_float((unsigned)se[1]) * sc[0][1][j] + sh[0][1][j];
                vo[j] = __uint_as_float((unsigned)so[0]) * sc[1][0][j] + sh[1][0][j];
                vo[4 + j] = __uint_as_float((unsigned)so[1]) * sc[1][1][j] + sh[1][1][j];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = k == 0 ? ror8(ve[j], vo[j], std::integral_constant<int, 0xC>{}) : ror8(vo[j], ve[j], std::integral_constant<int, 0x3>{});
                {
                    const u4 ra = rh[b][k], rc = rl[b][k];      // zeros when the layer has no residual
                    const h8 a = __builtin_bit_cast(h8, ra);
                    const h8 c = __builtin_bit_cast(h8, rc);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)a[j] + (float)c[j];
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, oh), y_hi_rs, (offA[b] + k * row8) | ((CX_ABL & 8) ? (int)OOB : 0), 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, ol), y_lo_rs, (offA[b] + k * row8) | ((CX_ABL & 8) ? (int)OOB : 0), 0, 2);
            }
        }
    };

    // scale / shift of the cout tile into LDS (the first barrier below publishes them)
    if (tid < 128) {
        const float* src = tid < 64 ? p.scale + n0 + tid * 4 : p.shift + n0 + (tid - 64) * 4;
        *(f4*)(smem + C::OFF_SCALE + tid * 16) = *(const f4*)src;
    }
    // ---- prologue: stages 0, 1, 2 -------------------------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
#pragma unroll
        for (int w = 0; w < C::PIECES; ++w) dma_piece(w);
        next_fill();
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");         // stage 0
    __builtin_amdgcn_s_barrier();
    int rslot = 0;                                              // ring slot of the current K step
#pragma unroll
    for (int j = 0; j < 4; ++j) read_a(A0, 0, 0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_b(B0, 0, 0, j);

    // Mid-step rendezvous.  NWAIT = instructions this wave issued after the pieces of the stage the next quadrants read.
    auto mid = [&](auto nwait_tag) {
        constexpr int NWAIT = decltype(nwait_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, C::PIECES> WaitSteady;                                   // pieces of stage s+2
    typedef std::integral_constant<int, C::PIECES + C::EPI_STORES> WaitAfterEpilogue;           // + the previous tile's stores
    typedef std::integral_constant<int, C::PIECES + C::EPI_LOADS> WaitLast;                     // + this tile's epilogue loads
    auto lg0 = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto nxt = [&](int s) { return s == C::NS - 1 ? 0 : s + 1; };
    // two K steps: even (fragments A0/B0 of its stage are in registers on entry) and odd (A0/B1 are); mid_e / mid_o = the wait
    // counts of their rendezvous; `loads_tile` >= 0: the epilogue loads of that tile are issued behind the even step's DMA
    auto two_steps = [&](auto mid_e, auto mid_o, int loads_tile) {
        const int s0 = rslot, s1 = nxt(s0), s2 = nxt(s1);
        lg0();
        quadrant(A0, 0, B0, 0, [&](int j) { read_b(B1, s0, 1, j); }, false);
        lg0();
        quadrant(A0, 0, B1, 1, [&](int j) { read_a(A1, s0, 1, j); }, false);
        mid(mid_e);
        quadrant(A1, 1, B1, 1, [&](int j) { read_a(A0, s1, 0, j); }, true);      // refills slot s0
        next_fill();
        if (loads_tile >= 0) issue_epilogue_loads(loads_tile);
        lg0();
        quadrant(A1, 1, B0, 0, [&](int j) { read_b(B1, s1, 1, j); }, false);
        lg0();
        quadrant(A0, 0, B1, 1, [&](int j) { read_b(B0, s1, 0, j); }, false);
        lg0();
        quadrant(A0, 0, B0, 0, [&](int j) { read_a(A1, s1, 1, j); }, false);
        mid(mid_o);
        quadrant(A1, 1, B0, 0, [&](int j) { read_a(A0, s2, 0, j); }, true);      // refills slot s1
        next_fill();
        lg0();
        quadrant(A1, 1, B1, 1, [&](int j) { read_b(B0, s2, 0, j); }, false);
        rslot = s2;
    };

#ifdef MPX_DIAG
    const unsigned long long d_start = __builtin_amdgcn_s_memtime();
    unsigned long long d_k = 0, d_wait = 0, d_epi = 0;
#endif
    for (int ti = 0; ti < my_tiles; ++ti) {
#ifdef MPX_DIAG
        const unsigned long long d_t0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        // K steps 0, 1: behind a tile boundary the previous tile's stores are still among the younger instructions
        // (nk >= 4: the host sends K = 64 layers to mpx_conv.h's kernel)
        if (nk == 4) {
            if (ti == 0) two_steps(WaitSteady{}, WaitSteady{}, -1);
            else two_steps(WaitAfterEpilogue{}, WaitAfterEpilogue{}, -1);
            two_steps(WaitSteady{}, WaitLast{}, ti);
        } else {
            if (ti == 0) two_steps(WaitSteady{}, WaitSteady{}, -1);
            else two_steps(WaitAfterEpilogue{}, WaitAfterEpilogue{}, -1);
            for (int ks = 2; ks + 2 < nk; ks += 2) two_steps(WaitSteady{}, WaitSteady{}, -1);
            two_steps(WaitSteady{}, WaitLast{}, ti);
        }
        // the epilogue loads were issued behind the DMA of step nk-2; since then only the pieces of step nk-1
        __builtin_amdgcn_sched_barrier(0);
#ifdef MPX_DIAG
        const unsigned long long d_t1 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PIECES) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef MPX_DIAG
        const unsigned long long d_t2 = __builtin_amdgcn_s_memtime();
#endif
        epilogue(ti);
#ifdef MPX_DIAG
        const unsigned long long d_t3 = __builtin_amdgcn_s_memtime();
        d_k += d_t1 - d_t0; d_wait += d_t2 - d_t1; d_epi += d_t3 - d_t2;
#endif
    }
#ifdef MPX_DIAG
    // per workgroup: SUMS over its tiles, laid out so that tools/probes/conv_timeline.py prints them as its phases:
    // "prologue" = K loops, "k-loop" = waits for the residual lines, "epilogue-1" = epilogue arithmetic + store issue
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long* o_ = p.stamps + (size_t)blockIdx.x * 8;
        o_[0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        o_[1] = d_start; o_[2] = d_start + d_k; o_[3] = d_start + d_k + d_wait; o_[4] = d_start + d_k + d_wait + d_epi;
        o_[5] = __builtin_amdgcn_s_memtime();
        o_[6] = __builtin_amdgcn_s_memrealtime();
        o_[7] = (unsigned long long)my_tiles;
    }
#endif
    // dead pieces still target this workgroup's LDS: they are older than the last stores
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::EPI_STORES) : "memory");
#endif
}

}  // namespace mpx
