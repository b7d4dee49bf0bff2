// mpx_conv256p.h -- the 256x256-tile kernel of mpx_conv256.h as ONE persistent workgroup per CU (tile id 13).
//
// Timeline of mpx_conv256.h on 1024 -> 256 at batch 2340 (tools/probes/conv_timeline.py): K loop 80.7 % of a workgroup's life at 2.7 us
// per step -- paced by the LDS-DMA round trip of its 64-KB stage, not by its 96 MFMAs per wave -- prologue 5.1 %, epilogue through the
// ring (two cout halves, four block barriers, 256 KB of stores that nothing overlaps) 14.1 %, empty slot 1.4 %.  Here:
//   * a fixed grid walks the tiles v0, v0 + G, ... (one cout tile per workgroup); the two-stage ring runs on ACROSS tiles: the stages
//     issued in the last two steps of a tile are the first two of the next one (pixel descriptor switched when the fill wraps) -- no
//     prologue, no dispatch gap;
//   * the epilogue works from the 128 accumulator registers (v_permlane16_swap + DPP row_ror:8 regrouping into whole 128-B lines, as
//     mpx_convx.h; scale / shift from LDS, applied after the regrouping: 16 registers instead of 32): no LDS tile, no block barrier, its 32
//     stores per wave retire under the next tile's first step, whose rendezvous allows them among the younger instructions (vmcnt(32));
//   * the K step is mpx_conv256.h's quadrant snake, unchanged.
// Layers without a residual operand only (every layer the 256x256 tile is a default for: the reducing 1x1 convs).  Same summation order
// and epilogue arithmetic as mpx_conv256.h: results are bit-identical to tile 9.
#pragma once
#include "mpx_conv256.h"

namespace mpx {

struct Conv256P {
    static constexpr int TC = 256, TP = 256, NW = 8, NT = 512;
    static constexpr int STAGE = Conv256::STAGE;
    static constexpr int OFF_SCALE = 2 * STAGE;         // f32[256] scale, f32[256] shift of the workgroup's cout tile
    static constexpr int LDS = 2 * STAGE + 2048;
    static constexpr int EPI_STORES = 32;               // 2 cout groups x 4 pixel fragments x 2 half-fragments x hi/lo
};

__global__ __launch_bounds__(512, 2) void conv256p_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Conv256 C;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = p.ktot, nk = K >> 5;                  // nk even (K % 64 == 0, host)
    constexpr unsigned OOB = 0x80000000u;

    // ---- the tiles of this workgroup: logical ids v0, v0 + G, ... (cout tile fastest, so it is the same for all of them) ----
    const int G = gridDim.x;                                           // a multiple of 8 and of n_tiles_c (host)
    const int v0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);    // blocks of one XCD walk neighbouring tiles
    const int n0 = (v0 % p.n_tiles_c) * C::TC;
    const int mt0 = v0 / p.n_tiles_c, mt_step = G / p.n_tiles_c;
    const int n_mt = (p.M + C::TP - 1) / C::TP;
    const int my_tiles = mt0 < n_mt ? (n_mt - 1 - mt0) / mt_step + 1 : 0;
    if (my_tiles == 0) return;

    // ---- DMA ------------------------------------------------------------------------------------------------------------
    const int prow = lane >> 2;
    const int src_q = ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 16;
    __amdgpu_buffer_rsrc_t x_hi, x_lo, w_hi, w_lo;
    {
        const int wrec = C::TC * K * 2;
        w_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * K), 0, wrec, 0x00020000);
        w_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * K), 0, wrec, 0x00020000);
    }
    auto set_x_desc = [&](int ti) {
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * K * 2;               // rows >= M are out of range: zeros
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (rem < 0 ? 0 : (int)rem);
        x_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_hi + (size_t)m0 * K), 0, rec, 0x00020000);
        x_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_lo + (size_t)m0 * K), 0, rec, 0x00020000);
    };
    set_x_desc(0);
    const int roff0 = ((wave * 32 + prow) * K) * 2 + src_q;
    const int roff1 = roff0 + 16 * K * 2;
    const int w_lane = lane * 16;
    int f_tile = 0, f_ks = 0, f_dead = 0;        // the next stage to issue: tile index, K step; dead behind the last tile
    auto dma_piece = [&](int slot, int which) {  // which = 0..7: W_hi p0, W_lo p0, W_hi p1, W_lo p1, X_hi p0, X_lo p0, X_hi p1, X_lo p1
        char* sb = smem + slot * C::STAGE;
        const int soff = f_ks * 64;
        const int pc = (which >> 1) & 1;
        const int voff = (pc ? roff1 : roff0) | f_dead;
        const int d = (wave * 2 + pc) * 1024;
        const int wvoff = w_lane | f_dead;
        const int wsoff = f_ks * 1024 + (wave * 2 + pc) * 16 * K * 2;
        switch (which & 5) {
            case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_hi, MPX_LDS_PTR(sb + C::OFF_WHI + d), 16, wvoff, wsoff, 0, 0); break;
            case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_lo, MPX_LDS_PTR(sb + C::OFF_WLO + d), 16, wvoff, wsoff, 0, 0); break;
            case 4: __builtin_amdgcn_raw_ptr_buffer_load_lds(x_hi, MPX_LDS_PTR(sb + C::OFF_XHI + d), 16, voff, soff, 0, 0); break;
            default: __builtin_amdgcn_raw_ptr_buffer_load_lds(x_lo, MPX_LDS_PTR(sb + C::OFF_XLO + d), 16, voff, soff, 0, 0); break;
        }
    };
    auto next_fill = [&]() {                     // advance (f_tile, f_ks) after a stage has been issued
        f_ks += 1;
        if (f_ks == nk) {
            f_ks = 0;
            f_tile += 1;
            if (f_tile < my_tiles) set_x_desc(f_tile);
            else f_dead = (int)OOB;              // past the last tile: the pieces still count, but touch no memory
        }
    };

    // ---- fragments (mpx_conv256.h) ----------------------------------------------------------------------------------------
    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = (wr * 128 + lrow) * 64 + qsw;
    const int b_off = (wc * 64 + lrow) * 64 + qsw;
    struct AH { h8 hi[4], lo[4]; };
    struct BH { h8 hi[2], lo[2]; };
    AH A0, A1;
    BH B0, B1;
    f4 acc[8][4];
    auto read_a = [&](AH& r, int slot, int half, int j) {
        const char* s = smem + slot * C::STAGE + (j < 4 ? C::OFF_WHI : C::OFF_WLO) + a_off + half * 4096 + (j & 3) * 1024;
        if (j < 4) r.hi[j] = *(const h8*)s;
        else r.lo[j - 4] = *(const h8*)s;
    };
    auto read_b = [&](BH& r, int slot, int half, int j) {
        const char* s = smem + slot * C::STAGE + (j < 2 ? C::OFF_XHI : C::OFF_XLO) + b_off + half * 2048 + (j & 1) * 1024;
        if (j < 2) r.hi[j] = *(const h8*)s;
        else r.lo[j - 2] = *(const h8*)s;
    };
    auto mfma_q = [&](const AH& a, int ah, const BH& b, int bh, int i) {
        const int fa = i / 6, r = i % 6, term = r >> 1, fb = r & 1;
        f4& d = acc[ah * 4 + fa][bh * 2 + fb];
        if (term == 0) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.lo[fb], d, 0, 0, 0);
        else if (term == 1) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo[fa], b.hi[fb], d, 0, 0, 0);
        else d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.hi[fb], d, 0, 0, 0);
    };
    auto quadrant = [&](const AH& a, int ah, const BH& b, int bh, auto&& reader, int nread, int dma_slot) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            mfma_q(a, ah, b, bh, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i % 3 == 1 && i / 3 < nread) {
                reader(i / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i % 3 == 2 && dma_slot >= 0) {      // i = 2, 5, ..., 23: the stage's eight pieces
                dma_piece(dma_slot, i / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- epilogue from the accumulator registers ------------------------------------------------------------------------------
    const int row8 = 8 * p.cout * 2;
    auto ror8 = [](float old, float src, auto mask_tag) {
        constexpr int MASK = decltype(mask_tag)::value;
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, MASK, false));
    };
    auto epilogue = [&](int ti) {
        int l_ = lane;                          // opaque copy: keeps hipcc from hoisting the lane geometry out of the tile loop (and spilling it)
        asm volatile("" : "+v"(l_));
        const int erow = l_ >> 4;
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (int)rem;
        const __amdgpu_buffer_rsrc_t y_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_hi + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t y_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_lo + (size_t)m0 * p.cout), 0, rec, 0x00020000);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {        // the wave's two groups of four cout fragments (64 channels each)
            // this lane's 8 consecutive channels after the regrouping (the same for every pixel fragment and half-fragment)
            const int col = wr * 128 + gq * 64 + (2 * ((l_ & 8) ? 1 : 0) + (erow & 1)) * 16 + (erow >> 1) * 8;
            const int dead = (p.cout - 1 - (n0 + col)) & (int)OOB;
            const int obase = ((wc * 64 + (l_ & 7)) * p.cout + n0 + col) * 2 | dead;
            const f4 sc0 = *(const f4*)(smem + Conv256P::OFF_SCALE + col * 4), sc1 = *(const f4*)(smem + Conv256P::OFF_SCALE + col * 4 + 16);
            const f4 sh0 = *(const f4*)(smem + Conv256P::OFF_SCALE + 1024 + col * 4), sh1 = *(const f4*)(smem + Conv256P::OFF_SCALE + 1024 + col * 4 + 16);
            const float sc[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
            const float sh[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float ve[8], vo[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e0 = acc[gq * 4 + 0][b][j], e1 = acc[gq * 4 + 1][b][j], o0 = acc[gq * 4 + 2][b][j], o1 = acc[gq * 4 + 3][b][j];
                    const auto se = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
                    const auto so = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0), __float_as_uint(o1), false, false);
                    ve[j] = __uint_as_float((unsigned)se[0]);
                    ve[4 + j] = __uint_as_float((unsigned)se[1]);
                    vo[j] = __uint_as_float((unsigned)so[0]);
                    vo[4 + j] = __uint_as_float((unsigned)so[1]);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float raw = k == 0 ? ror8(ve[j], vo[j], std::integral_constant<int, 0xC>{}) : ror8(vo[j], ve[j], std::integral_constant<int, 0x3>{});
                        v[j] = raw * sc[j] + sh[j];
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
                    const int o = obase + b * 16 * p.cout * 2 + k * row8;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, oh), y_hi_rs, o, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, ol), y_lo_rs, o, 0, 2);
                }
            }
        }
    };

    // scale / shift of the cout tile into LDS (the prologue's barrier publishes them)
    if (tid < 128) {
        const float* src = tid < 64 ? p.scale + n0 + tid * 4 : p.shift + n0 + (tid - 64) * 4;
        *(f4*)(smem + Conv256P::OFF_SCALE + tid * 16) = *(const f4*)src;
    }
    // ---- prologue (once per workgroup): stages 0 and 1 of the first tile ---------------------------------------------------------
#pragma unroll
    for (int w = 0; w < 8; ++w) dma_piece(0, w);
    next_fill();
#pragma unroll
    for (int w = 0; w < 8; ++w) dma_piece(1, w);
    next_fill();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // stage 0 (this wave's pieces); stage 1 stays in flight
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) read_a(A0, 0, 0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_b(B0, 0, 0, j);

    // mid-step rendezvous: this wave's reads of stage s have returned, its pieces of stage s+1 have landed; after the barrier that
    // holds for every wave.  `after_epi`: the first rendezvous of a tile that follows another one -- the previous tile's stores were
    // issued after the stage this rendezvous waits for (loads, LDS-DMAs and stores retire in issue order)
    auto mid = [&](bool after_epi) {
        __builtin_amdgcn_sched_barrier(0);
        if (after_epi) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Conv256P::EPI_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto lg0 = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nk; ks += 2) {
            // ---- even step: stage in slot 0, next stage in slot 1 -------------------------------------------------------------
            lg0();
            quadrant(A0, 0, B0, 0, [&](int j) { read_b(B1, 0, 1, j); }, 4, -1);
            lg0();
            quadrant(A0, 0, B1, 1, [&](int j) { read_a(A1, 0, 1, j); }, 8, -1);
            mid(ti > 0 && ks == 0);
            quadrant(A1, 1, B1, 1, [&](int j) { read_a(A0, 1, 0, j); }, 8, 0);        // refills slot 0
            next_fill();
            lg0();
            quadrant(A1, 1, B0, 0, [&](int j) { read_b(B1, 1, 1, j); }, 4, -1);
            // ---- odd step: stage in slot 1, next stage in slot 0 --------------------------------------------------------------
            lg0();
            quadrant(A0, 0, B1, 1, [&](int j) { read_b(B0, 1, 0, j); }, 4, -1);
            lg0();
            quadrant(A0, 0, B0, 0, [&](int j) { read_a(A1, 1, 1, j); }, 8, -1);
            mid(false);
            quadrant(A1, 1, B0, 0, [&](int j) { read_a(A0, 0, 0, j); }, 8, 1);        // refills slot 1
            next_fill();
            lg0();
            quadrant(A1, 1, B1, 1, [&](int j) { read_b(B0, 0, 0, j); }, 4, -1);
        }
        __builtin_amdgcn_sched_barrier(0);
        epilogue(ti);
        __builtin_amdgcn_sched_barrier(0);
    }
    // dead pieces still target this workgroup's LDS, and the last stores are in flight
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}

}  // namespace mpx
