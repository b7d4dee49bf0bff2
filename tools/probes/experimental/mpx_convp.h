// mpx_convp.h -- PERSISTENT variant of mpx_conv.h's kernel (same f16x3 arithmetic, same ring / K-step schedule).
//
// What the per-workgroup timelines of mpx_conv.h's kernel showed on MI355X (tools/probes/conv_timeline.py, expanding 1x1
// layer 256->1024 at batch 2048): a workgroup lives ~19 us of which the K loop is 47 %, the prologue (first DMA round
// trip) 18 %, the epilogue 34 % (residual round trip, fp32 transposition through LDS, two block barriers), and each
// workgroup slot of a CU then stays EMPTY ~2.4 us until the dispatcher has started the next workgroup (22 % of the CU
// time with one workgroup instead of two).  This kernel removes the serial pieces:
//   * a fixed grid (two workgroups per CU) loops over the tiles: no dispatch gap, descriptors / offsets set up once per tile;
//   * the epilogue works from the accumulator registers: four v_permlane16_swap per fragment pair give every lane 8
//     consecutive channels of one pixel (16 B per plane), so residual planes are read and output planes written with
//     dwordx4 buffer instructions in 64-B runs -- no LDS, no block barrier, the ring is free as soon as the K loop ends;
//   * so the NEXT tile's prologue DMAs are issued before the current tile's epilogue and land underneath it.
// vmcnt bookkeeping: loads, LDS-DMAs and stores retire in issue order.  Epilogue loads/stores are buffer instructions
// with out-of-range offsets for masked lanes (never skipped), so every wave issues the same number of VMEM
// instructions per tile and all waits stay immediates.
#pragma once
#include "../../../network_interpretation_imagenet_amd/csrc/mpx_conv.h"

namespace mpx {

template <class C, bool DUAL = false>
__global__ __launch_bounds__(C::NT, C::MINB) void convp_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TC = C::TC, TP = C::TP, NSW = C::NSW, NSX = C::NSX, NW = C::NW;
    constexpr int CF = C::CF, PF = C::PF, WJ = C::WJ, XJ = C::XJ, WSTAGE = C::WSTAGE, XSTAGE = C::XSTAGE, XBASE = C::XBASE;
    constexpr int OFF_WHI = 0, OFF_WLO = TC * 64, OFF_XHI = 0, OFF_XLO = TP * 64;
    static_assert(CF % 4 == 0, "the register epilogue needs 64 output channels per wave (four cout fragments = one 128-B line)");
    constexpr int NQ = CF / 2;                                  // fragment pairs per wave
    constexpr int N_STORES = NQ * PF * 2;                       // epilogue store instructions per wave and tile
    typedef unsigned u4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::NWC, wc = wave % C::NWC;
    const int nk = p.ktot >> 5;
    const int prow = lane >> 2;
    const int src_q = ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 8;
    const int howo = p.ho * p.wo;
    const int img_elems = p.hin * p.win * p.pix_stride;
    const int n_img = p.M / howo;
    constexpr unsigned OOB = 0x80000000u;

    // ---- per-tile DMA state (rewritten by setup()) --------------------------------------------------------------
    int m0 = 0, n0 = 0;
    __amdgpu_buffer_rsrc_t x_rs_hi, x_rs_lo, w_rs_hi, w_rs_lo, x2_rs_hi, x2_rs_lo;
    int x_off0[XJ], x_iy0[XJ], x_ix0[XJ];
    int x2_off0[DUAL ? XJ : 1];
    int ky = 0, kx = 0, c0 = 0, cb = 0;
    auto setup = [&](int vb) {
        // XCD-aware bijective remap over ALL tiles (the grid is a multiple of 8, so vb % 8 == blockIdx.x % 8)
        const int nb = p.n_tiles;
        const int q8 = nb >> 3, r8 = nb & 7, xcd = vb & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
        const int mt = L / p.n_tiles_c;
        const int nt = L - mt * p.n_tiles_c;
        m0 = mt * TP;
        n0 = nt * TC;
        const int n_first = m0 / howo;
        {
            const size_t rem = (size_t)(n_img - n_first) * img_elems * 2;
            const int nrec = rem > 0x7fffffffu ? 0x7fffffff : (int)rem;
            x_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_hi + (size_t)n_first * img_elems), 0, nrec, 0x00020000);
            x_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_lo + (size_t)n_first * img_elems), 0, nrec, 0x00020000);
            const int wrec = TC * p.ktot * 2;
            w_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
            w_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
        }
#pragma unroll
        for (int i = 0; i < XJ; ++i) {
            const int m = m0 + (i * NW + wave) * 16 + prow;
            const int n = m / howo;
            const int rem = m - n * howo;
            const int oy = rem / p.wo;
            const int ox = rem - oy * p.wo;
            x_iy0[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
            x_ix0[i] = ox * p.stride - p.pad;
            x_off0[i] = (((n - n_first) * p.hin + x_iy0[i]) * p.win + x_ix0[i]) * p.pix_stride * 2 + src_q * 2;
            if (DUAL) {
                const int off = (((n - n_first) * p.hin2 + oy * p.stride2) * p.win2 + ox * p.stride2) * p.pix_stride2 * 2 + src_q * 2;
                x2_off0[i] = off | ((p.M - 1 - m) & (int)OOB);
            }
        }
        if (DUAL) {
            const int img2 = p.hin2 * p.win2 * p.pix_stride2;
            const size_t rem = (size_t)(n_img - n_first) * img2 * 2;
            const int nrec = rem > 0x7fffffffu ? 0x7fffffff : (int)rem;
            x2_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2_hi + (size_t)n_first * img2), 0, nrec, 0x00020000);
            x2_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2_lo + (size_t)n_first * img2), 0, nrec, 0x00020000);
        } else {
            x2_rs_hi = x_rs_hi;
            x2_rs_lo = x_rs_lo;
        }
        ky = 0; kx = 0; c0 = 0; cb = 0;
    };
    const int w_lane = lane * 16;    // piece-major planes (mpx_conv.h w_packed_index): byte lane*16 of the piece, piece position in the soffset
    int w_piece[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int piece = C::HALF_W ? (wave % (NW / 2)) : (j * NW + wave);
        w_piece[j] = piece * 16 * p.ktot * 2;
    }

    auto stage_w = [&](int buf, int ks) {
        char* sb = smem + buf * WSTAGE;
        const int soff = ks * 1024;
        const int dead = ks < nk ? 0 : (int)OOB;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (C::HALF_W) {
                if (wave < NW / 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + wave * 1024), 16, w_lane | dead, soff + w_piece[j], 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_lo, MPX_LDS_PTR(sb + wave * 1024), 16, w_lane | dead, soff + w_piece[j], 0, 0);
            } else {
                const int d = (j * NW + wave) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + OFF_WHI + d), 16, w_lane | dead, soff + w_piece[j], 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_lo, MPX_LDS_PTR(sb + OFF_WLO + d), 16, w_lane | dead, soff + w_piece[j], 0, 0);
            }
        }
    };
    auto stage_x = [&](int i, int buf, int ky_, int kx_, int c0_, bool live) {
        char* sb = smem + XBASE + buf * XSTAGE;
        const int iy = x_iy0[i] + ky_, ix = x_ix0[i] + kx_;
        const int delta = ((ky_ * p.win + kx_) * p.pix_stride + c0_) * 2;
        const int voff = (x_off0[i] + delta) | ((iy | (p.hin - 1 - iy) | ix | (p.win - 1 - ix)) & (int)OOB) | (live ? 0 : (int)OOB);
        const int d = (i * NW + wave) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_hi, MPX_LDS_PTR(sb + OFF_XHI + d), 16, voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_lo, MPX_LDS_PTR(sb + OFF_XLO + d), 16, voff, 0, 0, 0);
    };
    auto stage_x2 = [&](int i, int buf, int cb_, bool live) {
        char* sb = smem + XBASE + buf * XSTAGE;
        const int voff = (x2_off0[DUAL ? i : 0] + cb_ * 2) | (live ? 0 : (int)OOB);
        const int d = (i * NW + wave) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x2_rs_hi, MPX_LDS_PTR(sb + OFF_XHI + d), 16, voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x2_rs_lo, MPX_LDS_PTR(sb + OFF_XLO + d), 16, voff, 0, 0, 0);
    };
    auto advance = [&]() {
        c0 += 32;
        const bool wc0 = (c0 == p.k_per_tap);
        c0 = wc0 ? 0 : c0;
        kx += wc0 ? 1 : 0;
        const bool wkx = (kx == p.kw);
        kx = wkx ? 0 : kx;
        ky += wkx ? 1 : 0;
    };
    auto prologue = [&]() {      // stages 0 .. NSX-1 (always from the first operand: k1/32 >= NSX for DUAL)
#pragma unroll
        for (int s = 0; s < NSX; ++s) {
            if (s < NSW) stage_w(s, s);
#pragma unroll
            for (int i = 0; i < XJ; ++i) stage_x(i, s, ky, kx, c0, s < nk);
            advance();
        }
    };

    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = (wr * (TC / C::NWR) + lrow) * 64 + qsw;
    const int b_off = (wc * (TP / C::NWC) + lrow) * 64 + qsw;
    struct Frags {
        h8 a_hi[CF], a_lo[CF], b_hi[PF], b_lo[PF];
    };
    constexpr int NF = 2 * (CF + PF);
    constexpr int NM = 3 * CF * PF;
    f4 acc[CF][PF];
    auto load_frag = [&](int wslot, int xslot, Frags& f, int j) {
        const char* sw = smem + wslot * WSTAGE;
        const char* sx = smem + XBASE + xslot * XSTAGE;
        if (j < CF) f.a_hi[j] = *(const h8*)(sw + OFF_WHI + a_off + j * 1024);
        else if (j < 2 * CF) f.a_lo[j - CF] = *(const h8*)(sw + OFF_WLO + a_off + (j - CF) * 1024);
        else if (j < 2 * CF + PF) f.b_hi[j - 2 * CF] = *(const h8*)(sx + OFF_XHI + b_off + (j - 2 * CF) * 1024);
        else f.b_lo[j - 2 * CF - PF] = *(const h8*)(sx + OFF_XLO + b_off + (j - 2 * CF - PF) * 1024);
    };
    auto mfma_one = [&](const Frags& f, int i) {
        const int a = i / (3 * PF), r = i % (3 * PF), term = r / PF, b = r % PF;
        if (term == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_lo[b], acc[a][b], 0, 0, 0);
        else if (term == 1) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_lo[a], f.b_hi[b], acc[a][b], 0, 0, 0);
        else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_hi[b], acc[a][b], 0, 0, 0);
    };
    auto mfma_all = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < NM; ++i) mfma_one(f, i);
    };

    int wslot = 0, xslot = 0;
    auto full_step = [&](auto seg_tag, int ks, const Frags& cur, Frags& nxt) {
        constexpr bool SEGB = decltype(seg_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<C::WAIT_STEP>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nw = (wslot + 1 == NSW) ? 0 : wslot + 1;
        const int nx = (xslot + 1 == NSX) ? 0 : xslot + 1;
        const bool live = ks + NSX < nk;
        constexpr int G = 1 + XJ;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma_one(cur, i);
            __builtin_amdgcn_sched_barrier(0);
            if ((i & 1) == 0 && i / 2 < NF) {
                load_frag(nw, nx, nxt, i / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (i == C::DMA_FIRST + 4 * g) {
                    if (g == 0) stage_w(wslot, ks + NSW);
                    else if (SEGB) stage_x2(g - 1, xslot, cb, live);
                    else stage_x(g - 1, xslot, ky, kx, c0, live);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (SEGB) cb += 32;
        else advance();
        wslot = nw;
        xslot = nx;
    };
    typedef std::integral_constant<bool, false> SegA;
    typedef std::integral_constant<bool, DUAL> SegLast;

    // ---- epilogue from the accumulator registers ---------------------------------------------------------------------
    // 1. v_permlane16_swap(acc[2q][b][j], acc[2q+1][b][j]): a lane of row r = lane>>4 then holds, for pixel column
    //    lane&15 of pixel fragment b, the 8 consecutive channels  (2q + (r&1)) * 16 + (r>>1) * 8 + [0,8)  of its wave's range
    //    ([0,4) in acc[2q][b], [4,8) in acc[2q+1][b]): 16 B per plane.  One pair q covers 32 channels = 64 B of a pixel row.
    // 2. Half-line (64-B) accesses cost ~1.4x the HBM time of full lines (measured: 64->256 at 60 instead of 84 TFLOP/s), so
    //    two pairs (qe, qo) = 64 channels = one 128-B line are regrouped with two DPP moves per register (row_ror:8, bank
    //    masks): instruction A carries pixels [0,8) of the fragment -- lanes 0-7 of a row keep pair qe, lanes 8-15 receive
    //    pair qo of the pixel 8 lanes below -- and instruction B pixels [8,16) the other way round.  Every load / store
    //    instruction then moves 8 pixel rows x 128 B.
    // All loads / stores are buffer instructions; masked lanes carry an out-of-range offset, so every wave issues exactly
    // N_STORES stores per tile.
    static_assert(NQ % 2 == 0, "the register epilogue regroups pairs of fragment pairs into full 128-B lines");
    const int erow = lane >> 4;
    const bool lo8 = (lane & 8) == 0;
    auto ror8 = [](float old, float src, auto mask_tag) {         // lanes of the banks in `mask` take src from the lane 8 away
        constexpr int MASK = decltype(mask_tag)::value;
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, MASK, false));
    };
    auto epilogue = [&](int m0e, int n0e) {
        const size_t base = (size_t)m0e * p.cout;
        const long long remain = ((long long)p.M - m0e) * p.cout * 2;
        const int nrec = remain > 0x7fffffffLL ? 0x7fffffff : (int)remain;      // rows >= M are out of range
        const __amdgpu_buffer_rsrc_t y_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_hi + base), 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t y_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_lo + base), 0, nrec, 0x00020000);
        f4 sc[NQ][2], sh[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {              // scale / shift in the accumulator layout
            const int co = n0e + wr * (TC / C::NWR) + (2 * q + (erow & 1)) * 16 + (erow >> 1) * 8;
            sc[q][0] = *(const f4*)(p.scale + co);          // scale/shift are padded to cout_pad >= n_tiles_c * TC
            sc[q][1] = *(const f4*)(p.scale + co + 4);
            sh[q][0] = *(const f4*)(p.shift + co);
            sh[q][1] = *(const f4*)(p.shift + co + 4);
        }
        int offA[NQ / 2][PF];                       // byte offset of this lane's 16 B in instruction A; B is 8 pixel rows further
        const int row8 = 8 * p.cout * 2;
#pragma unroll
        for (int t = 0; t < NQ / 2; ++t) {
            const int qsel = 2 * t + (lo8 ? 0 : 1);
            const int co = n0e + wr * (TC / C::NWR) + (2 * qsel + (erow & 1)) * 16 + (erow >> 1) * 8;
            const int dead = (p.cout - 1 - co) & (int)OOB;                     // co >= cout: nothing to store
#pragma unroll
            for (int b = 0; b < PF; ++b) {
                const int pl = wc * (TP / C::NWC) + b * 16 + (lane & 7);
                offA[t][b] = (pl * p.cout + co) * 2 | dead;
            }
        }
        u4 rh[NQ / 2][PF][2], rl[NQ / 2][PF][2];
        if (p.r_hi) {
            const __amdgpu_buffer_rsrc_t r_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.r_hi + base), 0, nrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.r_lo + base), 0, nrec, 0x00020000);
#pragma unroll
            for (int t = 0; t < NQ / 2; ++t)
#pragma unroll
                for (int b = 0; b < PF; ++b)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        rh[t][b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_hi_rs, offA[t][b] + k * row8, 0, 2);      // nt: streamed once
                        rl[t][b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_lo_rs, offA[t][b] + k * row8, 0, 2);
                    }
        }
#pragma unroll
        for (int t = 0; t < NQ / 2; ++t) {
#pragma unroll
            for (int b = 0; b < PF; ++b) {
                float ve[8], vo[8];                 // pairs qe = 2t, qo = 2t+1 in the accumulator layout, BatchNorm applied
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // (copy the vector elements into scalars first: __builtin_bit_cast applied to an ext_vector element
                    // lvalue reads element 0 whatever the subscript -- clang 22 / ROCm 7.2)
                    const float e0 = acc[4 * t][b][j], e1 = acc[4 * t + 1][b][j], o0 = acc[4 * t + 2][b][j], o1 = acc[4 * t + 3][b][j];
                    const auto se = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
                    const auto so = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0), __float_as_uint(o1), false, false);
                    ve[j] = __uint_as_float((unsigned)se[0]) * sc[2 * t][0][j] + sh[2 * t][0][j];
                    ve[4 + j] = __uint_as_float((unsigned)se[1]) * sc[2 * t][1][j] + sh[2 * t][1][j];
                    vo[j] = __uint_as_float((unsigned)so[0]) * sc[2 * t + 1][0][j] + sh[2 * t + 1][0][j];
                    vo[4 + j] = __uint_as_float((unsigned)so[1]) * sc[2 * t + 1][1][j] + sh[2 * t + 1][1][j];
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {       // k = 0: instruction A (pixels 0-7 of the fragment), 1: B (pixels 8-15)
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        v[j] = k == 0 ? ror8(ve[j], vo[j], std::integral_constant<int, 0xC>{})      // lanes 8-15 <- qo of the pixel 8 below
                                      : ror8(vo[j], ve[j], std::integral_constant<int, 0x3>{});     // lanes 0-7 <- qe of the pixel 8 above
                    if (p.r_hi) {
                        const u4 ra = rh[t][b][k], rc = rl[t][b][k];
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
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, oh), y_hi_rs, offA[t][b] + k * row8, 0, 2);    // nt
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, ol), y_lo_rs, offA[t][b] + k * row8, 0, 2);
                }
            }
        }
    };

    // ---- the tile loop -------------------------------------------------------------------------------------------------
    int vb = blockIdx.x;
    setup(vb);
    prologue();
    wait_vmcnt<C::WAIT_PROLOGUE>();
    __builtin_amdgcn_s_barrier();
    for (;;) {
#pragma unroll
        for (int a = 0; a < CF; ++a)
#pragma unroll
            for (int b = 0; b < PF; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        wslot = 0;
        xslot = 0;
        Frags fa, fb;
#pragma unroll
        for (int j = 0; j < NF; ++j) load_frag(0, 0, fa, j);
        int ks = 0;
        if (DUAL) {
            const int n_a = (p.k1 >> 5) - NSX;
            for (; ks < n_a; ks += 2) {
                full_step(SegA{}, ks, fa, fb);
                full_step(SegA{}, ks + 1, fb, fa);
            }
        }
        for (; ks + 2 < nk; ks += 2) {
            full_step(SegLast{}, ks, fa, fb);
            full_step(SegLast{}, ks + 1, fb, fa);
        }
        if (ks + 2 == nk) {
            full_step(SegLast{}, ks, fa, fb);
            mfma_all(fb);
        } else {
            mfma_all(fa);
        }
        // Every wave has read its last fragments (lgkmcnt(0)); after the barrier the ring belongs to the next tile: its
        // prologue is issued BEFORE this tile's epilogue, so the first DMA round trip and the residual round trip overlap.
        // (The trailing dead DMAs of this tile retire in issue order ahead of the prologue's pieces.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int m0e = m0, n0e = n0;
        const int vbn = vb + (int)gridDim.x;
        const bool more = vbn < p.n_tiles;
        if (more) {
            setup(vbn);
            prologue();
        }
        epilogue(m0e, n0e);
        if (!more) {
            // the trailing dead DMAs of the K loop still target this workgroup's LDS: they are older than the stores, so they
            // have retired when at most the stores are outstanding (the LDS may be handed to another workgroup after s_endpgm)
            wait_vmcnt<N_STORES>();
            break;
        }
        vb = vbn;
        // the N_STORES stores are this wave's youngest vector-memory instructions: everything older -- the residual loads and
        // the whole prologue of the new tile -- has retired when at most they are outstanding
        wait_vmcnt<N_STORES>();
        __builtin_amdgcn_s_barrier();
    }
#endif
}

}  // namespace mpx
