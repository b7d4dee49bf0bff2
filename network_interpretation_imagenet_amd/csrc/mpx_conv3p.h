// mpx_conv3p.h -- 3x3 stride-1 conv + BN + residual + ReLU with the input staged ONCE per 32-channel chunk
// ("patch" kernel, f16x3 arithmetic as mpx_conv.h).
//
// mpx_conv.h treats the 9 taps as 9 independent K steps and DMAs the pixel operand for each of them: the same input
// pixel crosses L2 -> LDS nine times, and 2/3 (cout >= 128) to 4/5 (cout = 64) of a step's LDS-DMA pieces are pixel
// rows.  Here the K loop runs chunk-major (32 input channels, then the 9 taps): per chunk the tile's input PATCH --
// every input pixel any of its output pixels touches, in zero-PADDED coordinates -- lands in LDS once, and a tap is
// nothing but a row offset ky*(W+2) + kx on the fragment reads.  DMA pieces per 9 steps: 9 W stages + 1 patch
// instead of 9 W + 9 X stages (-57 % bytes for the 128x256 tile).
//
// Geometry (stride 1, pad 1, square maps): PW = W+2, PIMG = (H+2)*PW; output pixel m = (n, oy, ox) has the padded
// index pb(m) = n*PIMG + oy*PW + ox of its top-left tap; the patch of a tile starts at q0 = pb(m0) and patch row r
// holds padded index q0 + r (zeros where that is padding: the buffer unit returns 0 for the out-of-range offset).
// Tap (ky, kx) of pixel m reads patch row pb(m) - q0 + ky*PW + kx.  The host sizes the patch (p.patch_rows, a
// multiple of 16) for the worst tile of the layer.
//
// LDS: [W ring, 3 stages][patch 0: hi rows | lo rows][patch 1].  Fragment rows of a wave are no longer 16-aligned,
// so the pixel-side chunk swizzle is ((row>>2)&1)<<1, which keeps any 16 consecutive rows conflict-free for
// ds_read_b128 (the weight side keeps mpx_conv.h's).  The 9 taps are unrolled (two chunks = 18 steps per loop
// iteration, for the fragment double buffer), so every s_waitcnt vmcnt is still an immediate.
#pragma once
#include "mpx_conv.h"
#include <type_traits>

namespace mpx {

template <int TC_, int TP_, int NWR_, int NWC_, int XJP_, int PPS_>
struct PatchCfg {
    static constexpr int TC = TC_, TP = TP_, NWR = NWR_, NWC = NWC_;
    static constexpr int XJP = XJP_;      // patch pieces (16 rows x 2 planes) a wave moves per chunk
    static constexpr int PPS = PPS_;      // pieces it issues per step, in taps 0 .. XJP/PPS-1 (must end by tap 5, see full_step)
    static_assert(XJP_ % PPS_ == 0 && XJP_ / PPS_ <= 6, "patch pieces must be issued in taps 0..5");
    static constexpr int NW = NWR * NWC, NT = 64 * NW;
    static constexpr int CF = TC / NWR / 16, PF = TP / NWC / 16;         // 16x16 fragments per wave
    static constexpr int WSTAGE = TC * 128;                                // weight stage: [hi TC x 64 B][lo TC x 64 B], 3-deep ring
    static constexpr int WJ = TC / 16 / NW;                                // weight pieces per wave per plane
    static constexpr int LW = 2 * WJ;                                      // DMA instructions per wave per weight stage
    static_assert((TC / 16) % NW == 0 && (TC / NWR) % 16 == 0 && (TP / NWC) % 16 == 0, "tile shape");
    static constexpr int MAX_PATCH_ROWS = 16 * NW * XJP;
    static constexpr int wait_at(int tap) {                   // vmcnt at the top of the step of `tap` (see full_step)
        return LW + pp(tap + 8) + pp(tap + 7);
    }
    static constexpr int pp(int t) { return (t % 9) < XJP / PPS ? 2 * PPS : 0; }      // patch DMA instructions of the step of tap t
    static int lds_bytes(int patch_rows) {
        const int ring = 3 * WSTAGE + 2 * patch_rows * 128 + NW * 1024;      // + one 1-KiB dump piece per wave
        const int epi = TP * TC * 4;
        return ring > epi ? ring : epi;
    }
};

template <class C>
__global__ __launch_bounds__(C::NT, 1) void conv3x3p_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TC = C::TC, TP = C::TP, NW = C::NW, NT = C::NT, XJP = C::XJP;
    constexpr int CF = C::CF, PF = C::PF, WJ = C::WJ, WSTAGE = C::WSTAGE;
    constexpr int OFF_WHI = 0, OFF_WLO = TC * 64;
    constexpr int XBASE = 3 * WSTAGE;

    MPX_STAMP(t_start);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::NWC, wc = wave % C::NWC;

    int L;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    }
    const int mt = L / p.n_tiles_c;
    const int nt = L - mt * p.n_tiles_c;
    const int m0 = mt * TP, n0 = nt * TC;

    const int H = p.hin, W = p.win, cin = p.k_per_tap;
    const int PW = W + 2, PIMG = (H + 2) * PW, howo = H * W;
    const int R = p.patch_rows;                           // allocated patch rows (multiple of 16)
    const int PSTG = R * 128;                             // bytes of one patch stage: [hi R x 64 B][lo R x 64 B]
    const int nchunks = cin >> 5;
    auto pb = [&](int m) {
        const int n = m / howo;
        const int rem = m - n * howo;
        const int oy = rem / W;
        return n * PIMG + oy * PW + (rem - oy * W);
    };
    const int q0 = pb(m0);
    const int m_last = (m0 + TP < p.M ? m0 + TP : p.M) - 1;
    const int r_tile = pb(m_last) - q0 + 2 * PW + 3;      // patch rows this tile needs (<= R by the host's sizing)
    const int n_first = q0 / PIMG;
    const int n_img = p.M / howo;

    // ---- DMA addressing ----------------------------------------------------------------------------------------
    constexpr unsigned OOB = 0x80000000u;
    const int prow = lane >> 2;
    const int xsrc_q = ((lane & 3) ^ (((prow >> 2) & 1) << 1)) * 16;     // patch rows: alignment-independent swizzle
    __amdgpu_buffer_rsrc_t x_rs_hi, x_rs_lo, w_rs_hi, w_rs_lo;
    {
        const size_t img_bytes = (size_t)howo * cin * 2;
        const size_t rem = (size_t)(n_img - n_first) * img_bytes;
        const int nrec = rem > 0x7fffffffu ? 0x7fffffff : (int)rem;
        x_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x_hi + (size_t)n_first * img_bytes), 0, nrec, 0x00020000);
        x_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x_lo + (size_t)n_first * img_bytes), 0, nrec, 0x00020000);
        const int wrec = TC * p.ktot * 2;
        w_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
        w_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
    }
    int x_poff[XJP];        // byte offset of this lane's chunk of patch piece i (channel chunk 0), or out of range
#pragma unroll
    for (int i = 0; i < XJP; ++i) {
        const int pr = (i * NW + wave) * 16 + prow;       // patch row
        const int q = q0 + pr;
        const int n = q / PIMG;
        const int rem = q - n * PIMG;
        const int py = rem / PW;
        const int px = rem - py * PW;
        const bool ok = pr < r_tile && pr < R && n < n_img && py >= 1 && py <= H && px >= 1 && px <= W;
        const int off = ((((n - n_first) * H + py - 1) * W + px - 1) * cin) * 2 + xsrc_q;
        x_poff[i] = ok ? off : (int)OOB;
    }
    const int w_lane = lane * 16;    // piece-major planes (mpx_conv.h w_packed_index): byte lane*16 of the piece, piece position in the soffset

    // W(step): step = chunk*9 + tap, K offset (tap*cin + chunk*32) elements; steps past the end read nothing
    auto stage_w = [&](int slot, int chunk, int tap) {
        char* sb = smem + slot * WSTAGE;
        const int soff = (tap * cin + chunk * 32) * 32;          // K step (tap*cin + chunk*32)/32, 1 KiB per step
        const int dead = chunk < nchunks ? 0 : (int)OOB;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int d = (j * NW + wave) * 1024;
            const int ps = soff + (j * NW + wave) * 16 * p.ktot * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + OFF_WHI + d), 16, w_lane | dead, ps, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_lo, MPX_LDS_PTR(sb + OFF_WLO + d), 16, w_lane | dead, ps, 0, 0);
        }
    };
    // Piece i of this wave for `chunk` into patch buffer chunk&1.  A piece beyond the allocated rows still has to be an
    // instruction (the vmcnt bookkeeping counts per wave) and an out-of-range load still WRITES its zeros: it goes to the
    // wave's dump piece behind the patches.
    auto stage_patch = [&](int i, int chunk) {
        const bool inside = (i * NW + wave) * 16 < R;
        char* sb = inside ? smem + XBASE + (chunk & 1) * PSTG + (i * NW + wave) * 1024 : smem + XBASE + 2 * PSTG + wave * 1024;
        char* sl = inside ? sb + R * 64 : sb;
        const int dead = (chunk < nchunks && inside) ? 0 : (int)OOB;
        const int soff = chunk * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_hi, MPX_LDS_PTR(sb), 16, x_poff[i] | dead, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_lo, MPX_LDS_PTR(sl), 16, x_poff[i] | dead, soff, 0, 0);
    };

    f4 acc[CF][PF];
#pragma unroll
    for (int a = 0; a < CF; ++a)
#pragma unroll
        for (int b = 0; b < PF; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing ------------------------------------------------------------------------------------
    const int lrow = lane & 15, lq = lane >> 4;
    const int a_off = (wr * (TC / C::NWR) + lrow) * 64 + ((lq ^ (((lane >> 3) & 1) << 1)) * 16);
    int rb[PF];             // patch row of this lane's pixel (tap 0) per pixel fragment
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int m = m0 + wc * (TP / C::NWC) + j * 16 + lrow;
        rb[j] = m < p.M ? pb(m) - q0 : 0;
    }
    struct Frags {
        h8 a_hi[CF], a_lo[CF], b_hi[PF], b_lo[PF];
    };
    constexpr int NF = 2 * (CF + PF);
    constexpr int NM = 3 * CF * PF;
    int baddr[PF];          // byte address (within a patch stage's hi plane) of the next step's pixel fragments
    auto set_baddr = [&](int tap) {
        const int td = (tap / 3) * PW + (tap % 3);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int row = rb[j] + td;
            baddr[j] = row * 64 + ((lq ^ (((row >> 2) & 1) << 1)) << 4);
        }
    };
    auto load_frag = [&](int wslot, int pbuf, Frags& f, int j) {
        const char* sw = smem + wslot * WSTAGE;
        const char* sx = smem + XBASE + pbuf * PSTG;
        if (j < CF) f.a_hi[j] = *(const h8*)(sw + OFF_WHI + a_off + j * 1024);
        else if (j < 2 * CF) f.a_lo[j - CF] = *(const h8*)(sw + OFF_WLO + a_off + (j - CF) * 1024);
        else if (j < 2 * CF + PF) f.b_hi[j - 2 * CF] = *(const h8*)(sx + baddr[j - 2 * CF]);
        else f.b_lo[j - 2 * CF - PF] = *(const h8*)(sx + R * 64 + baddr[j - 2 * CF - PF]);
    };
    auto mfma_one = [&](const Frags& f, int i) {
        const int a = i / (3 * PF), r = i % (3 * PF), term = r / PF, b = r % PF;
        if (term == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_lo[b], acc[a][b], 0, 0, 0);
        else if (term == 1) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_lo[a], f.b_hi[b], acc[a][b], 0, 0, 0);
        else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_hi[b], acc[a][b], 0, 0, 0);
    };

    // ---- prologue: patch(0), W(0..2) -------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < XJP; ++i) stage_patch(i, 0);
    stage_w(0, 0, 0);
    stage_w(1, 0, 1);
    stage_w(2, 0, 2);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    MPX_STAMP(t_pro);
    Frags fa, fb;
    set_baddr(0);
#pragma unroll
    for (int j = 0; j < NF; ++j) load_frag(0, 0, fa, j);

    // Step (chunk c, tap T): MFMAs on `cur`; reads the fragments of the next step into `nxt`; issues W(step+3) into the
    // slot of W(step) and, for T < XJP, piece T of patch(c+1).  At its top, W(step+1) must have landed: it was issued in
    // step-2, and everything issued after it -- the patch pieces of step-2, and W + patch pieces of step-1 -- may stay in
    // flight: vmcnt = LW + pp(T-1) + pp(T-2) = C::wait_at(T).  patch(c+1) is issued in taps 0..XJP-1 of chunk c and first
    // read during tap 8, whose top waits for W issued in tap 6: loads return in order, so the patch is there.
    auto full_step = [&](auto tap_tag, int c, int wslot, const Frags& cur, Frags& nxt) {
        constexpr int T = decltype(tap_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<C::wait_at(T)>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nw = (wslot + 1 == 3) ? 0 : wslot + 1;
        constexpr int TN = (T + 1) % 9;
        const int cn = (T == 8) ? c + 1 : c;
        set_baddr(TN);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma_one(cur, i);
            __builtin_amdgcn_sched_barrier(0);
            if ((i & 1) == 0 && i / 2 < NF) {
                load_frag(nw, cn & 1, nxt, i / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i == 1) {
                constexpr int T3 = (T + 3) % 9;
                stage_w(wslot, c + (T + 3) / 9, T3);
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
    for (int c = 0; c < nchunks; c += 2) {        // two chunks per iteration: 18 steps, so that fa / fb end where they began
        ws = full_step(std::integral_constant<int, 0>{}, c, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 1>{}, c, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 2>{}, c, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 3>{}, c, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 4>{}, c, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 5>{}, c, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 6>{}, c, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 7>{}, c, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 8>{}, c, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 0>{}, c + 1, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 1>{}, c + 1, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 2>{}, c + 1, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 3>{}, c + 1, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 4>{}, c + 1, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 5>{}, c + 1, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 6>{}, c + 1, ws, fb, fa);
        ws = full_step(std::integral_constant<int, 7>{}, c + 1, ws, fa, fb);
        ws = full_step(std::integral_constant<int, 8>{}, c + 1, ws, fb, fa);
    }
    wait_vmcnt<0>();
    MPX_STAMP(t_kend);

    // ---- epilogue (as mpx_conv.h) ------------------------------------------------------------------------------------
    constexpr int GPP = TC / 8;
    constexpr int PPI = NT / GPP;
    constexpr int ITERS = TP / PPI;
    constexpr int RP = TC * 4;
    const int g = tid % GPP;
    const int prow2 = tid / GPP;
    const int co8 = n0 + g * 8;
    const bool co_ok = co8 < p.cout;
    h8 rh[ITERS], rl[ITERS];
    if (p.r_hi) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int pix = m0 + it * PPI + prow2;
            if (co_ok && pix < p.M) {
                const size_t o = (size_t)pix * p.cout + co8;
                rh[it] = __builtin_nontemporal_load((const h8*)(p.r_hi + o));
                rl[it] = __builtin_nontemporal_load((const h8*)(p.r_lo + o));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < CF; ++a) {
        const int col = wr * (TC / C::NWR) + a * 16 + (lane >> 4) * 4;
        const f4 sc = *(const f4*)(p.scale + n0 + col);
        const f4 sh = *(const f4*)(p.shift + n0 + col);
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            const int pl = wc * (TP / C::NWC) + b * 16 + lrow;
            const f4 v = acc[a][b] * sc + sh;
            *(f4*)(smem + pl * RP + (((col >> 2) ^ (pl & 7)) << 4)) = v;
        }
    }
    __syncthreads();
    MPX_STAMP(t_epi);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int pl = it * PPI + prow2;
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
        const size_t o = (size_t)pix * p.cout + co8;
        h8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t hi, lo;
            split_f32(v[j], hi, lo);
            oh[j] = hi;
            ol[j] = lo;
        }
        __builtin_nontemporal_store(oh, (h8*)(p.y_hi + o));
        __builtin_nontemporal_store(ol, (h8*)(p.y_lo + o));
    }
    MPX_STAMP_WRITE(p, t_start, t_pro, t_kend, t_epi);
#endif
}

typedef PatchCfg<128, 256, 2, 4, 4, 1> PatchTile0;    // cout >= 128: 8 waves, patch <= 512 rows
typedef PatchCfg<64, 256, 1, 4, 8, 2> PatchTile1;     // cout = 64: 4 waves side by side, patch <= 512 rows
typedef PatchCfg<128, 192, 2, 4, 3, 1> PatchTile2;    // 7x7 maps: 192 pixels = 3.9 images, patch <= 384 rows

}  // namespace mpx
