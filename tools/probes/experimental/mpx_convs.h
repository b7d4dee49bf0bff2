// mpx_convs.h -- EXPANDING 1x1 stride-1 conv + BN + residual + ReLU with the PIXEL operand resident in LDS ("X-stationary";
// f16x3 arithmetic of mpx_conv.h): the last conv of a bottleneck with K = 128 or 256 (128->512, 256->1024).
//
// Why: in mpx_conv.h's 128x128 tile the texture addresser (which carries every LDS-DMA piece and the epilogue's loads and
// stores) is as busy as the MFMA pipe (profiles/r02_pmc_lds_ta.txt: 71 %): per K step a workgroup moves 16 KB of weights AND 16 KB
// of pixels for 192 MFMAs, and the pixel tile is fetched again by every one of the layer's cout tiles.  K is short in these layers,
// so a pixel tile's WHOLE K extent fits the LDS: 96 pixels x 256 channels x (hi + lo) = 96 KB.  One persistent workgroup per CU
// walks pixel tiles; per pixel tile it loads the pixels once and then sweeps all cout tiles (256 wide) over them, streaming only
// weights (32 KB per step for 288 MFMAs, two-stage ring in the remaining 64 KB; weights are L2-resident and piece-major, so a
// step's 32 pieces are 32 contiguous KiB).  Bytes through the texture path per MAC fall to 0.35x of the 128x128 tile's.
//   * 8 waves as 4 (cout) x 2 (pixels), wave tile 64 x 48, accumulators 48 VGPRs;
//   * K step = two halves of 18 MFMAs: A0 (first two cout fragments of the stage) x the step's three pixel fragments, then A1 x
//     the same, pixel fragment by pixel fragment.  One fragment register set (56 VGPRs), yet every fragment is read from LDS at
//     least 12 MFMAs before its use: A1 during the first half, A0 of the next stage early in the second, and each pixel fragment of
//     the next K step (the resident pixels are always there) into its predecessor's registers as soon as that one is done;
//   * ONE rendezvous per step, between the halves: the weights of the next stage have landed (own pieces: counted vmcnt; everyone's:
//     the barrier), and every wave has finished reading the current stage, whose slot is refilled at once with stage +2;
//   * the ring never drains: stages run on across cout tiles and pixel tiles;
//   * register epilogue of mpx_convx.h (v_permlane16_swap + DPP row_ror:8 into whole 128-B lines, no LDS, no barrier); its residual
//     lines and its scale / shift vectors are requested two and a half steps before the cout tile ends, its stores retire under the next
//     K loop;
//   * at a pixel-tile boundary the next pixel tile is requested BEFORE the last cout tile's epilogue runs (one barrier: every wave
//     has finished reading the old pixels) and awaited after it.
// vmcnt bookkeeping (loads, LDS-DMAs and stores retire in issue order; every wave issues the same instruction counts -- masked
// lanes carry an out-of-range offset): see the WAIT_* constants at the waits.
#pragma once
#include "../../../network_interpretation_imagenet_amd/csrc/mpx_conv.h"

namespace mpx {

struct ConvS {
    static constexpr int TC = 256, TP = 96, NW = 8, NT = 512;
    static constexpr int XSTEP = 12288;                 // one K step of the pixel tile: [hi 96 rows x 64 B | lo 96 rows x 64 B]
    static constexpr int MAXNK = 8;                     // K <= 256
    static constexpr int OFF_W = MAXNK * XSTEP;         // 98304: weight ring behind the pixel tile
    static constexpr int WSTAGE = 32768;                // [W_hi 256 rows x 64 B | W_lo]
    static constexpr int LDS = OFF_W + 2 * WSTAGE;      // 163840 = all of it
    static constexpr int WPIECES = 4;                   // weight DMA instructions per wave and stage
    static constexpr int XPIECES = 12;                  // pixel-tile DMA instructions per wave (96 pieces / 8 waves; dead ones for K = 128)
    static constexpr int EPI_LOADS = 12;                // residual: 3 pixel fragments x 2 lines x hi/lo
    static constexpr int EPI_STORES = 12;
};

__global__ __launch_bounds__(512, 2) void convs_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef ConvS C;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int K = p.ktot, nk = K >> 5;                  // 4 or 8 (host)
    const int n_ct = p.n_tiles_c;                       // cout tiles per pixel tile
    constexpr unsigned OOB = 0x80000000u;

    // ---- the pixel tiles of this workgroup: v0, v0 + G, ... (blocks of one XCD take neighbouring tiles) -------------------------
    const int G = gridDim.x;
    const int n_pt = (p.M + C::TP - 1) / C::TP;
    const int v0 = blockIdx.x;
    const int my_pt = v0 < n_pt ? (n_pt - 1 - v0) / G + 1 : 0;
    if (my_pt == 0) return;
    const int n_stage = my_pt * n_ct * nk;              // weight stages this workgroup consumes

    // ---- DMA ------------------------------------------------------------------------------------------------------------------
    const int prow = lane >> 2;
    const int src_q = ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 16;
    const int w_lane = lane * 16;
    const __amdgpu_buffer_rsrc_t w_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, n_ct * C::TC * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_lo = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_lo, 0, n_ct * C::TC * K * 2, 0x00020000);
    int f_q = 0, f_ct = 0, f_ks = 0;                    // the next weight stage to issue: running number, cout tile, K step
    auto dma_w = [&](int which) {                       // 0..3: W_hi p0, W_lo p0, W_hi p1, W_lo p1 of stage f_q
        char* sb = smem + C::OFF_W + (f_q & 1) * C::WSTAGE;
        const int pc = which >> 1;
        const int voff = w_lane | (f_q < n_stage ? 0 : (int)OOB);      // past the end: the piece still counts, but touches no memory
        const int soff = ((f_ct * 16 + wave * 2 + pc) * nk + f_ks) * 1024;
        const int d = (wave * 2 + pc) * 1024;
        if ((which & 1) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_hi, MPX_LDS_PTR(sb + d), 16, voff, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_lo, MPX_LDS_PTR(sb + 16384 + d), 16, voff, soff, 0, 0);
    };
    auto next_w = [&]() {
        f_q += 1;
        f_ks += 1;
        if (f_ks == nk) {
            f_ks = 0;
            f_ct = f_ct + 1 == n_ct ? 0 : f_ct + 1;
        }
    };
    // pixel tile `pi` of this workgroup into the resident image: piece idx = j*8 + wave -> (K step, plane, 16-row block)
    const int x_lane = prow * K * 2 + src_q;
    auto load_x = [&](int pi) {
        const int m0 = (v0 + pi * G) * C::TP;
        const long long rem = ((long long)p.M - m0) * K * 2;
        const int rec = pi < my_pt ? (rem > 0x7fffffffLL ? 0x7fffffff : (int)rem) : 0;     // no such tile: every lane out of range
        const half_t* bh = pi < my_pt ? p.x_hi + (size_t)m0 * K : p.x_hi;
        const half_t* bl = pi < my_pt ? p.x_lo + (size_t)m0 * K : p.x_lo;
        const __amdgpu_buffer_rsrc_t xh = __builtin_amdgcn_make_buffer_rsrc((void*)bh, 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t xl = __builtin_amdgcn_make_buffer_rsrc((void*)bl, 0, rec, 0x00020000);
#pragma unroll
        for (int j = 0; j < C::XPIECES; ++j) {
            const int idx = j * 8 + wave;               // wave-uniform
            const int ks = idx / 12, r = idx - ks * 12, plane = r / 6, rb = r - plane * 6;
            const int voff = (x_lane + rb * 16 * K * 2) | (ks < nk ? 0 : (int)OOB);      // the row offset stays in the VGPR: the buffer range check (rows >= M read zeros) does not see the SGPR offset
            char* d = smem + ks * C::XSTEP + plane * 6144 + rb * 1024;
            if (plane == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xh, MPX_LDS_PTR(d), 16, voff, ks * 64, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(xl, MPX_LDS_PTR(d), 16, voff, ks * 64, 0, 0);
        }
    };

    // ---- fragments ------------------------------------------------------------------------------------------------------------
    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = C::OFF_W + (wr * 64 + lrow) * 64 + qsw;          // + slot*WSTAGE + plane*16384 + half*2048 + f*1024
    const int b_off = (wc * 48 + lrow) * 64 + qsw;                     // + ks*XSTEP + plane*6144 + b*1024
    struct AH { h8 hi[2], lo[2]; };
    struct BP { h8 hi[3], lo[3]; };                     // the three pixel fragments of one K step
    AH A0, A1;
    BP B0;
    f4 acc[4][3];
    auto read_a = [&](AH& r, int q, int half, int j) {                  // j = 0..3: hi f0, hi f1, lo f0, lo f1 of stage q
        const char* s = smem + a_off + (q & 1) * C::WSTAGE + (j < 2 ? 0 : 16384) + half * 2048 + (j & 1) * 1024;
        if (j < 2) r.hi[j] = *(const h8*)s;
        else r.lo[j - 2] = *(const h8*)s;
    };
    auto read_b = [&](BP& r, int ks, int j) {                           // j = 0..5: hi b0..b2, lo b0..b2 of K step ks
        const char* s = smem + b_off + ks * C::XSTEP + (j < 3 ? 0 : 6144) + (j % 3) * 1024;
        if (j < 3) r.hi[j] = *(const h8*)s;
        else r.lo[j - 3] = *(const h8*)s;
    };
    // half a step: 18 MFMAs, pixel fragment by pixel fragment (two cout fragments x three products each), so that a pixel fragment's
    // registers are free for the next K step's as soon as its six MFMAs have been issued; `extra(i)` runs after MFMA i
    auto half_step = [&](const AH& a, int ah, const BP& b, auto&& extra) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int bi = i / 6, fa = (i % 6) / 3, term = i % 3;
            f4& d = acc[ah * 2 + fa][bi];
            if (term == 0) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.lo[bi], d, 0, 0, 0);
            else if (term == 1) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo[fa], b.hi[bi], d, 0, 0, 0);
            else d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[fa], b.hi[bi], d, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            extra(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#ifdef MPX_DIAG
    unsigned long long d_rv = 0, d_ew = 0, d_epi = 0, d_xw = 0, d_bar = 0 , d_h1 = 0, d_h2 = 0, d_dma = 0;
    const unsigned long long d_begin = __builtin_amdgcn_s_memtime();
#define MPX_T0 const unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define MPX_T1(acc_) acc_ += __builtin_amdgcn_s_memtime() - t0_
#else
#define MPX_T0
#define MPX_T1(acc_)
#endif
    auto rendezvous = [&](auto nwait_tag) {             // NWAIT = instructions this wave issued after the pieces of the next stage
        constexpr int NWAIT = decltype(nwait_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
#ifdef MPX_DIAG
        const unsigned long long ta_ = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");
#ifdef MPX_DIAG
        const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef MPX_DIAG
        const unsigned long long tc_ = __builtin_amdgcn_s_memtime();
        d_rv += tc_ - ta_;
        d_bar += tc_ - tb_;
#endif
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- epilogue state: lane geometry of the regrouped 16-B chunks (mpx_convx.h) -------------------------------------------------
    const int erow = lane >> 4;
    const bool lo8 = (lane & 8) == 0;
    const int co_lane = wr * 64 + (2 * (lo8 ? 0 : 1) + (erow & 1)) * 16 + (erow >> 1) * 8;    // this lane's 8 channels within a cout tile
    const int pix0 = (wc * 48 + (lane & 7)) * p.cout * 2;          // + (2b + k) * row8 for pixel fragment b, line k -- in the VGPR offset: the
                                                                    // buffer range check (rows >= M are neither read nor written) does not see the SGPR offset
    const int row8 = 8 * p.cout * 2;
    u4 rh[3][2], rl[3][2];
    auto ror8 = [](float old, float src, auto mask_tag) {
        constexpr int MASK = decltype(mask_tag)::value;
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, MASK, false));
    };
    // C::EPI_LOADS loads for (pixel tile pi, cout tile ct): its residual lines
    auto issue_epilogue_loads = [&](int pi, int ct) {
        const int m0 = (v0 + pi * G) * C::TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = p.r_hi ? (rem > 0x7fffffffLL ? 0x7fffffff : (int)rem) : 0;          // no residual: every lane out of range (zeros)
        const half_t* rbh = p.r_hi ? p.r_hi : p.y_hi;
        const half_t* rbl = p.r_hi ? p.r_lo : p.y_lo;
        const __amdgpu_buffer_rsrc_t r_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(rbh + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(rbl + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const int co = (ct * C::TC + co_lane) * 2;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rh[b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_hi_rs, pix0 + co + (2 * b + k) * row8, 0, 2);
                rl[b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_lo_rs, pix0 + co + (2 * b + k) * row8, 0, 2);
            }
    };
    // C::EPI_STORES stores; clears the accumulators.  The scale / shift of the 8 channels this lane stores (applied after the
    // regrouping) are loaded here -- L2 hits; hipcc waits for them with vmcnt(0), which also retires the two weight stages in
    // flight -- and `mid` (the next pixel tile's request at a pixel-tile boundary) runs behind that wait, so that nothing the
    // compiler waits for is queued behind it.
    auto epilogue = [&](int pi, int ct, auto&& mid) {
        f4 sc8[2], sh8[2];
        {
            const __amdgpu_buffer_rsrc_t sc_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.scale, 0, n_ct * C::TC * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t sh_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.shift, 0, n_ct * C::TC * 4, 0x00020000);
            const int col8 = (ct * C::TC + co_lane) * 4;
            sc8[0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(sc_rs, col8, 0, 0));
            sc8[1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(sc_rs, col8 + 16, 0, 0));
            sh8[0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(sh_rs, col8, 0, 0));
            sh8[1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(sh_rs, col8 + 16, 0, 0));
            asm volatile("" ::"v"(sc8[0]), "v"(sc8[1]), "v"(sh8[0]), "v"(sh8[1]));      // a use: the compiler's wait lands here
            __builtin_amdgcn_sched_barrier(0);
        }
        mid();
        __builtin_amdgcn_sched_barrier(0);
        const int m0 = (v0 + pi * G) * C::TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (int)rem;
        const __amdgpu_buffer_rsrc_t y_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_hi + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t y_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_lo + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const int co = (ct * C::TC + co_lane) * 2;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float ve[8], vo[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e0 = acc[0][b][j], e1 = acc[1][b][j], o0 = acc[2][b][j], o1 = acc[3][b][j];
                const auto se = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
                const auto so = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0), __float_as_uint(o1), false, false);
                ve[j] = __uint_as_float((unsigned)se[0]);
                ve[4 + j] = __uint_as_float((unsigned)se[1]);
                vo[j] = __uint_as_float((unsigned)so[0]);
                vo[4 + j] = __uint_as_float((unsigned)so[1]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = (k == 0 ? ror8(ve[j], vo[j], std::integral_constant<int, 0xC>{}) : ror8(vo[j], ve[j], std::integral_constant<int, 0x3>{})) *
                               sc8[j >> 2][j & 3] + sh8[j >> 2][j & 3];
                {
                    const h8 a = __builtin_bit_cast(h8, rh[b][k]);      // zeros when the layer has no residual
                    const h8 c = __builtin_bit_cast(h8, rl[b][k]);
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, oh), y_hi_rs, pix0 + co + (2 * b + k) * row8, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, ol), y_lo_rs, pix0 + co + (2 * b + k) * row8, 0, 2);
            }
        }
    };

    // ---- one K step.  q = running stage number (slot q & 1), ks = its K step in the pixel tile.  On entry A0 (first two cout
    // fragments of stage q) and B0 (the pixel fragments of K step ks) are in registers.  First half: A0 x B0 while A1 (stage q) is
    // read.  Rendezvous.  Second half: A1 x B0 -- the four pieces of stage q+2 go out at once (a whole step to land), then what
    // `after` issues (the epilogue loads, in step nk-3); A0 of stage q+1 is read early, and each pixel fragment of K step ks_next (the
    // resident pixels are always there) replaces its predecessor as soon as that one's last MFMA has been issued: every fragment
    // is read at least 12 MFMAs before its first use, with one register set. --------------------------------------------------------
    auto step = [&](int q, int ks, int ks_next, auto nwait, auto&& after) {
#ifdef MPX_DIAG
        const unsigned long long s0_ = __builtin_amdgcn_s_memtime();
#endif
        half_step(A0, 0, B0, [&](int i) { if (i < 4) read_a(A1, q, 1, i); });
#ifdef MPX_DIAG
        const unsigned long long s1_ = __builtin_amdgcn_s_memtime();
        d_h1 += s1_ - s0_;
#endif
        rendezvous(nwait);
#ifdef MPX_DIAG
        const unsigned long long s2_ = __builtin_amdgcn_s_memtime();
#endif
        half_step(A1, 1, B0, [&](int i) {
            // all 32 pieces at once would queue in the CU's one texture addresser and hold the later waves at issue (measured: 600
            // cycles for the second wave of a SIMD): the first wave of each SIMD issues now, the second 8 MFMAs later
            if (i < 4 && wave < 4) dma_w(i);
            if (i >= 8 && i < 12 && wave >= 4) dma_w(i - 8);
            if (i == 12) {
                next_w();
#ifdef MPX_DIAG
                d_dma += __builtin_amdgcn_s_memtime() - s2_;
#endif
                after();
            }
            if (i % 6 == 5) {
                read_b(B0, ks_next, i / 6);
                read_b(B0, ks_next, 3 + i / 6);
            }
            if (i >= 13 && i < 17) read_a(A0, q + 1, 0, i - 13);
        });
#ifdef MPX_DIAG
        d_h2 += __builtin_amdgcn_s_memtime() - s2_;
#endif
    };
    typedef std::integral_constant<int, 0> Wait0;
    typedef std::integral_constant<int, C::EPI_STORES> WaitStores;        // the previous cout tile's stores are younger than the stage
    typedef std::integral_constant<int, C::EPI_LOADS> WaitLoads;          // this cout tile's epilogue loads are
    auto nothing = [] {};

    // ---- prologue: pixel tile 0, weight stages 0 and 1 ----------------------------------------------------------------------------
    load_x(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int w = 0; w < C::WPIECES; ++w) dma_w(w);
        next_w();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) read_a(A0, 0, 0, j);
#pragma unroll
    for (int j = 0; j < 6; ++j) read_b(B0, 0, j);

    int q = 0;
    for (int pi = 0; pi < my_pt; ++pi) {
        for (int ct = 0; ct < n_ct; ++ct) {
            // Per step s the order of a wave's vector-memory instructions is: [rendezvous of s] stage s+2 (4 pieces, at once) [what
            // `after` issues].  The rendezvous of step s needs stage s+1, issued in step s-1: everything issued behind it may stay in
            // flight -- nothing in the steady state; the previous cout tile's 12 stores at K step 0 (they were issued behind the last
            // step's stage); the 12 epilogue loads at step nk-2 (issued behind step nk-3's stage).  The rendezvous of step nk-1 waits
            // for a stage issued AFTER those loads, so it is where they have to be in at the latest (two steps after their issue).
            for (int ks = 0; ks < nk; ks += 2) {
                if (ks == 0) {
                    if (pi == 0 && ct == 0) step(q, 0, 1, Wait0{}, nothing);
                    else step(q, 0, 1, WaitStores{}, nothing);
                } else if (ks == nk - 2) {
                    step(q + ks, ks, ks + 1, WaitLoads{}, nothing);
                } else {
                    step(q + ks, ks, ks + 1, Wait0{}, nothing);
                }
                const int kn = ks + 2 == nk ? 0 : ks + 2;
                if (ks + 1 == nk - 3) step(q + ks + 1, ks + 1, kn, Wait0{}, [&] { issue_epilogue_loads(pi, ct); });
                else step(q + ks + 1, ks + 1, kn, Wait0{}, nothing);
            }
            q += nk;
            if (ct + 1 == n_ct) {
                // every wave has finished reading the resident pixels (its last fragment reads returned): the next pixel tile is
                // requested inside the epilogue, so that it travels under it.  (The pixel fragments read ahead for "K step 0" came
                // from the OLD pixels: re-read below.)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                {
                    MPX_T0;
                    epilogue(pi, ct, [&] { load_x(pi + 1); });
                    MPX_T1(d_epi);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    MPX_T0;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::EPI_STORES) : "memory");                  // own pixel pieces landed
                    __builtin_amdgcn_s_barrier();                                                          // everyone's
                    MPX_T1(d_xw);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) read_b(B0, 0, j);
            } else {
                MPX_T0;
                epilogue(pi, ct, nothing);
                MPX_T1(d_epi);
            }
        }
    }
#ifdef MPX_DIAG
    // per workgroup SUMS, laid out so that tools/probes/conv_timeline.py prints them as its phases: "prologue" = time in the
    // rendezvous (vmcnt wait + barrier), "k-loop" = waits for the epilogue loads, "epilogue-1" = epilogue arithmetic and store
    // issue, "epilogue-2" = waits for the next pixel tile
    const unsigned long long d_start = __builtin_amdgcn_s_memtime() - d_begin;      // the workgroup's whole life, in cycles
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long* o_ = p.stamps + (size_t)blockIdx.x * 8;
        o_[0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        o_[1] = d_start; o_[2] = d_rv; o_[3] = d_rv + d_ew; o_[4] = d_rv + d_ew + d_epi;
        o_[5] = d_rv + d_ew + d_epi + d_xw;
        o_[6] = d_bar;
        o_[7] = (unsigned long long)my_pt;
    }
    if (p.stamps && lane == 0) p.stamps[(size_t)(8192 + blockIdx.x) * 8 + wave] = d_bar;      // third record: barrier wait of every wave
    if (p.stamps && lane == 0 && (wave == 0 || wave == 4)) {                                   // fourth: segments of a step, waves 0 and 4
        unsigned long long* o3_ = p.stamps + (size_t)(12288 + blockIdx.x) * 8 + (wave ? 4 : 0);
        o3_[0] = d_h1; o3_[1] = d_dma; o3_[2] = d_h2;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // dead pieces still target this workgroup's LDS
#undef MPX_T0
#undef MPX_T1
#endif
}

}  // namespace mpx
