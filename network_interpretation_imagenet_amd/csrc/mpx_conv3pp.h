// mpx_conv3pp.h -- the 3x3 patch kernel (mpx_conv3p.h) as ONE persistent workgroup per CU (tile id 12).
//
// What the per-workgroup timeline of the patch kernel shows on the 14x14 layers (tools/probes/conv_timeline.py, batch 2340,
// 256 -> 256): the K loop is MFMA-paced (1.1 us per 48-MFMA step at two waves per SIMD) and takes 88 % of a workgroup's life;
// the rest is the prologue (first patch + three weight stages: one DMA round trip, 6.3 %), the epilogue through the LDS (two
// block barriers, 5.3 %) and the dispatch gap (1.5 %) -- 13 % of the CU time without an MFMA, fourteen times per launch.  Here:
//   * a fixed grid (one workgroup per CU) walks the tiles v0, v0 + G, ... (same cout tile for all of them, pixel tiles G /
//     n_tiles_c apart); the weight ring and the two patch buffers run on ACROSS tiles: during the last chunk of a tile the
//     patch pieces that the K loop issues anyway fetch chunk 0 of the NEXT tile, and W(step + 3) wraps to its first taps --
//     no prologue after the first tile, no dispatch gap;
//   * the epilogue works from the accumulator registers (v_permlane16_swap + DPP row_ror:8 regrouping into whole 128-B lines,
//     as mpx_convx.h): no LDS, no block barrier; its stores retire under the next tile's K loop;
//   * the next tile's patch offsets and descriptors are computed at the start of the current tile's last chunk, when every patch
//     piece of the current tile has been issued, and its fragment rows in the tile's last step (float-reciprocal divisions).
// vmcnt bookkeeping as mpx_conv3p.h (every wait an immediate); the two steps behind a tile boundary allow the EPI_STORES
// stores of the previous tile among the younger instructions (loads, LDS-DMAs and stores retire in issue order).
// Same accumulation order and the same epilogue arithmetic as mpx_conv3p.h: results are bit-identical to tile 6.  Layers WITHOUT a
// residual operand only (the 3x3 conv of a bottleneck block, the first 3x3 conv of a BasicBlock): the residual lines of a tile would
// need 64 more registers next to the fragments read ahead for the next tile, and hipcc then spills inside the K loop.
#pragma once
#include "mpx_conv3p.h"

// timing-only ablations (probe builds, wrong results): bit 0 = no fragment reads from LDS, 1 = weight DMAs carry an out-of-range offset
// (issued and counted, no access), 2 = patch DMAs likewise, 3 = stores likewise, 4 = no step barrier, 5 = no MFMAs,
// 6 = the fragments ARE read from LDS (fresh data, into the set `fb`) but every MFMA takes the set `fa` of the prologue (constant operands)
#ifndef P3_ABL
#define P3_ABL 0
#endif

namespace mpx {

template <class C>
struct PatchPersistent {
    static constexpr int EPI_STORES = 4 * C::PF;            // PF pixel fragments x 2 half-fragments x hi/lo
    static int lds_bytes(int patch_rows) {                  // ring + patches + dump pieces + scale / shift of the cout tile
        return 3 * C::WSTAGE + 2 * patch_rows * 128 + C::NW * 1024 + C::TC * 8;
    }
};

template <class C>
__global__ __launch_bounds__(C::NT, 1) void conv3x3pp_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    constexpr int TC = C::TC, TP = C::TP, NW = C::NW, XJP = C::XJP;
    constexpr int CF = C::CF, PF = C::PF, WJ = C::WJ, WSTAGE = C::WSTAGE;
    static_assert(CF == 4, "the register epilogue regroups four cout fragments (64 channels) per wave");
    constexpr int OFF_WHI = 0, OFF_WLO = TC * 64;
    constexpr int XBASE = 3 * WSTAGE;
    constexpr int EPI_STORES = PatchPersistent<C>::EPI_STORES;
    constexpr unsigned OOB = 0x80000000u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::NWC, wc = wave % C::NWC;

    // ---- the tiles of this workgroup: logical ids v0, v0 + G, ... (cout tile fastest, so it is the same for all of them) ----
    const int G = gridDim.x;                                           // a multiple of 8 and of n_tiles_c (host)
    const int v0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);    // blocks of one XCD walk neighbouring tiles
    const int n0 = (v0 % p.n_tiles_c) * TC;
    const int mt0 = v0 / p.n_tiles_c, mt_step = G / p.n_tiles_c;
    const int n_mt = (p.M + TP - 1) / TP;
    const int my_tiles = mt0 < n_mt ? (n_mt - 1 - mt0) / mt_step + 1 : 0;
    if (my_tiles == 0) return;

    const int H = p.hin, W = p.win, cin = p.k_per_tap;
    const int PW = W + 2, PIMG = (H + 2) * PW, howo = H * W;
    const int R = p.patch_rows;
    const int PSTG = R * 128;
    const int nchunks = cin >> 5;                           // even (cin % 64 == 0, host)
    const int n_img = p.M / howo;
    const int OFF_SCALE = XBASE + 2 * PSTG + NW * 1024;
    // x / d for 0 <= x < 2^23 (the host checks the ranges): float reciprocal, one correction step -- ~8 VALU instead of the ~35 of
    // an integer division; the tile geometry below is recomputed per tile INSIDE the K loop
    const float r_howo = 1.0f / (float)howo, r_w = 1.0f / (float)W, r_pimg = 1.0f / (float)PIMG, r_pw = 1.0f / (float)PW;
    auto fdiv = [](int x, int d, float rd) {
        int q = (int)((float)x * rd);
        const int r = x - q * d;
        q += (r >= d) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return q;
    };
    auto pb = [&](int m) {
        const int n = fdiv(m, howo, r_howo);
        const int rem = m - n * howo;
        const int oy = fdiv(rem, W, r_w);
        return n * PIMG + oy * PW + (rem - oy * W);
    };

    // ---- DMA addressing ---------------------------------------------------------------------------------------------------
    const int lrow = lane & 15, lq = lane >> 4;
    // lane-derived values of the per-tile code come from an opaque copy of `lane`: hipcc would otherwise hoist them out of the tile
    // loop and, with 192 registers of accumulators and fragments live in it, spill them (a scratch reload drains vmcnt)
    auto opaque_lane = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    __amdgpu_buffer_rsrc_t x_rs_hi, x_rs_lo, w_rs_hi, w_rs_lo;
    {
        const int wrec = TC * p.ktot * 2;
        w_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
        w_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
    }
    int xo[XJP];        // this lane's byte offsets of the patch pieces of the tile being PREFETCHED (or out of range)
    int rb[PF];         // patch row of this lane's pixels (tap 0) in the tile being COMPUTED
    // patch offsets and pixel descriptors of tile ti; past the last tile every piece is dead (it still counts for vmcnt)
    auto set_patch = [&](int ti) {
        const int l_ = opaque_lane();
        const int prow = l_ >> 2;
        const int xsrc_q = ((l_ & 3) ^ (((prow >> 2) & 1) << 1)) * 16;
        const int live = ti < my_tiles ? 0 : (int)OOB;
        const int m0 = (mt0 + ti * mt_step) * TP;
        const int q0 = pb(m0);
        const int m_last = (m0 + TP < p.M ? m0 + TP : p.M) - 1;
        const int r_tile = pb(m_last) - q0 + 2 * PW + 3;
        const int n_first = fdiv(q0, PIMG, r_pimg);
        {
            const size_t img_bytes = (size_t)howo * cin * 2;
            const size_t rem = (size_t)(n_img - n_first) * img_bytes;
            const int nrec = rem > 0x7fffffffu ? 0x7fffffff : (int)rem;
            x_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x_hi + (size_t)n_first * img_bytes), 0, nrec, 0x00020000);
            x_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x_lo + (size_t)n_first * img_bytes), 0, nrec, 0x00020000);
        }
#pragma unroll
        for (int i = 0; i < XJP; ++i) {
            const int pr = (i * NW + wave) * 16 + prow;
            const int q = q0 + pr;
            const int n = fdiv(q, PIMG, r_pimg);
            const int rem = q - n * PIMG;
            const int py = fdiv(rem, PW, r_pw);
            const int px = rem - py * PW;
            const bool ok = pr < r_tile && pr < R && n < n_img && py >= 1 && py <= H && px >= 1 && px <= W;
            const int off = ((((n - n_first) * H + py - 1) * W + px - 1) * cin) * 2 + xsrc_q;
            xo[i] = (ok ? off : (int)OOB) | live;
        }
    };
    auto set_rows = [&](int ti) {          // fragment rows of tile ti (any values behind the last tile: nothing reads them)
        const int lrow_ = opaque_lane() & 15;
        const int m0 = (mt0 + ti * mt_step) * TP;
        const int q0 = pb(m0);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int m = m0 + wc * (TP / C::NWC) + j * 16 + lrow_;
            rb[j] = m < p.M ? pb(m) - q0 : 0;
        }
    };
    const int w_lane = lane * 16;
    // W(chunk, tap); chunk == nchunks (+1) is chunk 0 (1) of the NEXT tile: the same weights again, dead behind the last tile
    auto stage_w = [&](int slot, int chunk, int tap, bool last_tile) {
        char* sb = smem + slot * WSTAGE;
        const bool wrap = chunk >= nchunks;
        const int ch = wrap ? chunk - nchunks : chunk;
        const int soff = (tap * cin + ch * 32) * 32;
        const int dead = (wrap && last_tile) ? (int)OOB : 0;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int d = (j * NW + wave) * 1024;
            const int ps = soff + (j * NW + wave) * 16 * p.ktot * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + OFF_WHI + d), 16, w_lane | dead | ((P3_ABL & 2) ? (int)OOB : 0), ps, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_lo, MPX_LDS_PTR(sb + OFF_WLO + d), 16, w_lane | dead | ((P3_ABL & 2) ? (int)OOB : 0), ps, 0, 0);
        }
    };
    // piece i of this wave for `chunk` (nchunks = chunk 0 of the next tile, whose offsets xo already holds) into patch buffer chunk & 1
    auto stage_patch = [&](int i, int chunk) {
        const bool inside = (i * NW + wave) * 16 < R;
        char* sb = inside ? smem + XBASE + (chunk & 1) * PSTG + (i * NW + wave) * 1024 : smem + XBASE + 2 * PSTG + wave * 1024;
        char* sl = inside ? sb + R * 64 : sb;
        const int dead = inside ? 0 : (int)OOB;
        const int soff = (chunk >= nchunks ? chunk - nchunks : chunk) * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_hi, MPX_LDS_PTR(sb), 16, xo[i] | dead | ((P3_ABL & 4) ? (int)OOB : 0), soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_lo, MPX_LDS_PTR(sl), 16, xo[i] | dead | ((P3_ABL & 4) ? (int)OOB : 0), soff, 0, 0);
    };

    f4 acc[CF][PF];

    // ---- fragment addressing --------------------------------------------------------------------------------------------------
    const int a_off = (wr * (TC / C::NWR) + lrow) * 64 + ((lq ^ (((lane >> 3) & 1) << 1)) * 16);
    struct Frags {
        h8 a_hi[CF], a_lo[CF], b_hi[PF], b_lo[PF];
    };
    constexpr int NF = 2 * (CF + PF);
    constexpr int NM = 3 * CF * PF;
    int baddr[PF];
    auto set_baddr = [&](int tap) {
        const int td = (tap / 3) * PW + (tap % 3);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int row = rb[j] + td;
            baddr[j] = row * 64 + ((lq ^ (((row >> 2) & 1) << 1)) << 4);
        }
    };
    auto load_frag = [&](int wslot, int pbuf, Frags& f, int j) {
        if (P3_ABL & 1) {
            asm volatile("" : "+v"(f.a_hi[0]), "+v"(f.b_hi[0]));
            return;
        }
        const char* sw = smem + wslot * WSTAGE;
        const char* sx = smem + XBASE + pbuf * PSTG;
        if (j < CF) f.a_hi[j] = *(const h8*)(sw + OFF_WHI + a_off + j * 1024);
        else if (j < 2 * CF) f.a_lo[j - CF] = *(const h8*)(sw + OFF_WLO + a_off + (j - CF) * 1024);
        else if (j < 2 * CF + PF) f.b_hi[j - 2 * CF] = *(const h8*)(sx + baddr[j - 2 * CF]);
        else f.b_lo[j - 2 * CF - PF] = *(const h8*)(sx + R * 64 + baddr[j - 2 * CF - PF]);
    };
    auto mfma_one = [&](const Frags& f, int i) {
        if (P3_ABL & 32) {
            asm volatile("" : "+v"(acc[0][0]));
            return;
        }
        const int a = i / (3 * PF), r = i % (3 * PF), term = r / PF, b = r % PF;
        if (term == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_lo[b], acc[a][b], 0, 0, 0);
        else if (term == 1) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_lo[a], f.b_hi[b], acc[a][b], 0, 0, 0);
        else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_hi[b], acc[a][b], 0, 0, 0);
    };

    // ---- epilogue from the accumulator registers (lane geometry of the regrouped 16-B chunks as mpx_convx.h) -------------------
    const int row8 = 8 * p.cout * 2;
    auto ror8 = [](float old, float src, auto mask_tag) {
        constexpr int MASK = decltype(mask_tag)::value;
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, MASK, false));
    };
    auto epilogue = [&](int ti) {
        const int l_ = opaque_lane();
        const int erow = l_ >> 4;
        int offA[PF];
        {
            const int co = n0 + wr * 64 + (2 * ((l_ & 8) ? 1 : 0) + (erow & 1)) * 16 + (erow >> 1) * 8;
            const int dead = ((p.cout - 1 - co) & (int)OOB) | ((P3_ABL & 8) ? (int)OOB : 0);
#pragma unroll
            for (int b = 0; b < PF; ++b) offA[b] = ((wc * (TP / C::NWC) + b * 16 + (l_ & 7)) * p.cout + co) * 2 | dead;
        }
        const int m0 = (mt0 + ti * mt_step) * TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (int)rem;
        const __amdgpu_buffer_rsrc_t y_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_hi + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t y_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_lo + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        f4 sc[2][2], sh[2][2];                  // scale / shift of this lane's channels in the accumulator layout, from LDS
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = wr * 64 + (2 * q + (erow & 1)) * 16 + (erow >> 1) * 8;
            sc[q][0] = *(const f4*)(smem + OFF_SCALE + col * 4);
            sc[q][1] = *(const f4*)(smem + OFF_SCALE + col * 4 + 16);
            sh[q][0] = *(const f4*)(smem + OFF_SCALE + TC * 4 + col * 4);
            sh[q][1] = *(const f4*)(smem + OFF_SCALE + TC * 4 + col * 4 + 16);
        }
#pragma unroll
        for (int b = 0; b < PF; ++b) {
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, oh), y_hi_rs, offA[b] + k * row8, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, ol), y_lo_rs, offA[b] + k * row8, 0, 2);
            }
        }
    };

    // scale / shift of the cout tile into LDS (the prologue's barrier publishes them)
    if (tid < TC / 2) {
        const float* src = tid < TC / 4 ? p.scale + n0 + tid * 4 : p.shift + n0 + (tid - TC / 4) * 4;
        *(f4*)(smem + OFF_SCALE + tid * 16) = *(const f4*)src;
    }
    // ---- prologue (once per workgroup): patch(0), W(0..2) of the first tile ---------------------------------------------------
    set_patch(0);
    set_rows(0);
#pragma unroll
    for (int i = 0; i < XJP; ++i) stage_patch(i, 0);
    stage_w(0, 0, 0, false);
    stage_w(1, 0, 1, false);
    stage_w(2, 0, 2, false);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    set_baddr(0);
#pragma unroll
    for (int j = 0; j < NF; ++j) load_frag(0, 0, fa, j);

    // Step (chunk c, tap T) as mpx_conv3p.h's; `after_epi`: one of the first two steps of a tile that follows another one -- the
    // previous tile's stores were issued after the weight stage this step waits for; `next_ti` >= 0: the last step of a tile --
    // the fragments it reads ahead are those of tile next_ti, at its rows.
    auto full_step = [&](auto tap_tag, int c, int wslot, const Frags& cur, Frags& nxt, bool after_epi, int next_ti, bool last_tile) {
        constexpr int T = decltype(tap_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        if (T < 2 && after_epi) wait_vmcnt<C::wait_at(T) + EPI_STORES>();
        else wait_vmcnt<C::wait_at(T)>();
        if (!(P3_ABL & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nw = (wslot + 1 == 3) ? 0 : wslot + 1;
        constexpr int TN = (T + 1) % 9;
        const int cn = (T == 8) ? c + 1 : c;
        if (T == 8 && next_ti >= 0) set_rows(next_ti);
        set_baddr(TN);
        __builtin_amdgcn_sched_barrier(0);
        const Frags& use = (P3_ABL & 64) ? fa : cur;
        Frags& fill = (P3_ABL & 64) ? fb : nxt;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma_one(use, i);
            __builtin_amdgcn_sched_barrier(0);
            if ((i & 1) == 0 && i / 2 < NF) {
                load_frag(nw, cn & 1, fill, i / 2);
                if (P3_ABL & 64) asm volatile("" ::"v"(fill.a_hi[0]), "v"(fill.a_lo[0]), "v"(fill.b_hi[0]), "v"(fill.b_lo[0]));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i == 1) {
                constexpr int T3 = (T + 3) % 9;
                stage_w(wslot, c + (T + 3) / 9, T3, last_tile);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (T < XJP / C::PPS) {
#pragma unroll
                for (int q = 0; q < C::PPS; ++q) {
                    if (i == 5 + 4 * q) {
                        stage_patch(T * C::PPS + q, c + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        return nw;
    };

    int ws = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const bool last_tile = ti + 1 == my_tiles;
#pragma unroll
        for (int a = 0; a < CF; ++a)
#pragma unroll
            for (int b = 0; b < PF; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunks; c += 2) {        // two chunks per iteration: 18 steps, so that fa / fb end where they began
            const bool ae = ti > 0 && c == 0;
            const bool bd = c + 2 == nchunks;
            ws = full_step(std::integral_constant<int, 0>{}, c, ws, fa, fb, ae, -1, last_tile);
            ws = full_step(std::integral_constant<int, 1>{}, c, ws, fb, fa, ae, -1, last_tile);
            ws = full_step(std::integral_constant<int, 2>{}, c, ws, fa, fb, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 3>{}, c, ws, fb, fa, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 4>{}, c, ws, fa, fb, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 5>{}, c, ws, fb, fa, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 6>{}, c, ws, fa, fb, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 7>{}, c, ws, fb, fa, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 8>{}, c, ws, fa, fb, false, -1, last_tile);
            // every patch piece of this tile has been issued: the pieces of the last chunk's steps fetch the next tile's chunk 0
            if (bd) set_patch(ti + 1);
            ws = full_step(std::integral_constant<int, 0>{}, c + 1, ws, fb, fa, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 1>{}, c + 1, ws, fa, fb, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 2>{}, c + 1, ws, fb, fa, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 3>{}, c + 1, ws, fa, fb, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 4>{}, c + 1, ws, fb, fa, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 5>{}, c + 1, ws, fa, fb, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 6>{}, c + 1, ws, fb, fa, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 7>{}, c + 1, ws, fa, fb, false, -1, last_tile);
            ws = full_step(std::integral_constant<int, 8>{}, c + 1, ws, fb, fa, false, bd ? ti + 1 : -1, last_tile);
        }
        __builtin_amdgcn_sched_barrier(0);
        epilogue(ti);
        __builtin_amdgcn_sched_barrier(0);
    }
    // dead pieces still target this workgroup's LDS, and the last stores are in flight
    wait_vmcnt<0>();
#endif
}

}  // namespace mpx
