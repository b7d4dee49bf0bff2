// mpx_conv.h -- K1/K2: conv (implicit GEMM) + BatchNorm scale/shift + residual + ReLU, gfx950.
//
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],   k = (ky, kx, ci), ci fastest
// on v_mfma_f32_16x16x32_f16 as three products  W_hi*X_lo + W_lo*X_hi + W_hi*X_hi  (split-fp16
// operands, fp32 accumulation).  A = weights (MFMA rows = cout), B = pixels (MFMA cols), so an
// accumulator register quad is 4 consecutive channels of one pixel.
//
// Workgroup = 8 waves (2 cout x 4 pixel), tile TC(cout) x TP(pixel), K advances 32 per step.
// LDS: a ring of NS = 3 stages, each [W_hi TCx64B][W_lo TCx64B][X_hi TPx64B][X_lo TPx64B], filled by
// global_load_lds_dwordx4 in 1-KiB pieces (16 rows x 64 B, lane-linear destination) and consumed behind
// a COUNTED s_waitcnt vmcnt + one raw s_barrier per K step; fragments are double-buffered in registers
// (read step k+1 while the MFMAs of step k run), so two DMA stages stay in flight under the MFMAs (cdna_hip_programming.md 5, "Pipelining across barriers").
// Rows are 64 B, which makes a ds_read_b128 fragment read 2-way bank conflicted; the 16-B chunk index
// is XORed with ((row>>3)&1)<<1 on the DMA's SOURCE address and on the fragment read (rule 21).
// Epilogue: accumulators -> fp32 tile in LDS (XOR-swizzled 16-B chunks) -> each thread owns 8
// consecutive channels of one pixel: residual planes are read and output planes written as whole
// 128-B lines (16 B per lane, 16 or 8 lanes per pixel row).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpx {

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MPX_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MPX_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void split_f32(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)(v - (float)hi);
}

struct ConvParams {
    const half_t* x_hi;      // input planes, NHWC (pix_stride elements per pixel)
    const half_t* x_lo;
    const half_t* w_hi;      // packed weights [cout_pad][ktot]
    const half_t* w_lo;
    const float* scale;      // [cout_pad]
    const float* shift;
    const half_t* r_hi;      // residual planes [M][cout] or null
    const half_t* r_lo;
    half_t* y_hi;            // output planes [M][cout]
    half_t* y_lo;
    float* y_f32;            // fp32 output [M][cout] (fc) or null
    const half_t* zero_page; // >= 64 zero bytes, 16-B aligned
    int hin, win;            // input spatial extent the bounds check uses
    int pix_stride;          // fp16 elements between adjacent input pixels
    int ho, wo;
    int kh, kw, stride, pad;
    int k_per_tap;           // K contributed by one (ky,kx) tap (= cin; 32 for the stem)
    int ktot;                // kh*kw*k_per_tap
    int cout;                // real output channels (store bound and row pitch of y/r); multiple of 8
    int M;                   // B*ho*wo output pixels
    int n_tiles_c;           // cout_pad / TC
    int relu;
};

constexpr int CONV_THREADS = 512;
constexpr int CONV_STAGES = 3;

template <int TC, int TP>
constexpr int conv_lds_bytes() {
    constexpr int ring = CONV_STAGES * (TC + TP) * 128;
    constexpr int epi = TP * TC * 4;
    return ring > epi ? ring : epi;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 5 || N == 6 || N == 10 || N == 12, "add the literal below");
    // lgkmcnt(0): this wave's fragment reads of the previous step have returned before it enters the
    // barrier that frees their ring slot for the next DMA
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
}

template <int TC, int TP>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_f16x3_kernel(const ConvParams p) {
    static_assert((TC == 128 || TC == 64) && TP == 256, "tile shapes wired below");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = CONV_STAGES;
    constexpr int CF = TC / 32;                 // 16-row cout fragments per wave (wave tile TC/2 x TP/4)
    constexpr int PF = TP / 64;                 // 16-col pixel fragments per wave
    constexpr int OFF_WHI = 0, OFF_WLO = TC * 64, OFF_XHI = TC * 128, OFF_XLO = TC * 128 + TP * 64;
    constexpr int STAGE = (TC + TP) * 128;
    constexpr int WPIECES = TC / 16;            // 1-KiB pieces per W plane
    constexpr int LPT = (2 * WPIECES + 2 * (TP / 16)) / 8;   // DMA instructions per wave per stage (6 or 5)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // XCD-aware bijective remap: blocks that share an XCD (b % 8) walk a contiguous range of
    // logical tiles, cout tiles fastest, so the X tile of one pixel range stays in that XCD's L2.
    int L;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    }
    const int mt = L / p.n_tiles_c;
    const int nt = L - mt * p.n_tiles_c;
    const int m0 = mt * TP, n0 = nt * TC;

    // ---- per-thread DMA bookkeeping: lane i of a piece moves row (i>>2), LDS chunk (i&3) ----
    const int prow = lane >> 2;                                   // row within a 16-row piece
    const int src_q = ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 8;   // swizzled source chunk, in elements
    int x_pixbase[2], x_iy0[2], x_ix0[2];
    const int howo = p.ho * p.wo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (i * 8 + wave) * 16 + prow;
        const int n = m / howo;
        const int rem = m - n * howo;
        const int oy = rem / p.wo;
        const int ox = rem - oy * p.wo;
        x_pixbase[i] = n * p.hin * p.win;
        x_iy0[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
        x_ix0[i] = ox * p.stride - p.pad;
    }
    // W: TC=128 -> this wave moves piece `wave` of both planes; TC=64 -> waves 0-3 W_hi, 4-7 W_lo
    const half_t* w_src;
    const half_t* w_src2 = nullptr;
    {
        const int wrow = (TC == 128 ? wave : (wave & 3)) * 16 + prow;
        const size_t o = (size_t)(n0 + wrow) * p.ktot + src_q;
        if (TC == 128) {
            w_src = p.w_hi + o;
            w_src2 = p.w_lo + o;
        } else {
            w_src = (wave < 4 ? p.w_hi : p.w_lo) + o;
        }
    }

    // `live` = false issues the same DMAs from the zero page (a step past the end of K): the ring and the
    // vmcnt bookkeeping then never change shape, so the K loop has no conditional code in it.
    auto stage_w = [&](int buf, int ks, bool live) {
        char* sb = smem + buf * STAGE;
        if (TC == 128) {
            __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(live ? w_src + ks * 32 : p.zero_page), MPX_LDS_PTR(sb + OFF_WHI + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(live ? w_src2 + ks * 32 : p.zero_page), MPX_LDS_PTR(sb + OFF_WLO + wave * 1024), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(live ? w_src + ks * 32 : p.zero_page), MPX_LDS_PTR(sb + wave * 1024), 16, 0, 0);
        }
    };
    auto stage_x = [&](int i, int buf, int ky, int kx, int c0, bool live) {
        char* sb = smem + buf * STAGE;
        const int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
        const bool ok = live & ((unsigned)iy < (unsigned)p.hin) & ((unsigned)ix < (unsigned)p.win);   // no short-circuit branches
        const size_t o = (size_t)(x_pixbase[i] + iy * p.win + ix) * p.pix_stride + c0 + src_q;
        const half_t* s_hi = ok ? p.x_hi + o : p.zero_page;
        const half_t* s_lo = ok ? p.x_lo + o : p.zero_page;
        const int d = (i * 8 + wave) * 1024;
        __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(s_hi), MPX_LDS_PTR(sb + OFF_XHI + d), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(s_lo), MPX_LDS_PTR(sb + OFF_XLO + d), 16, 0, 0);
    };
    auto stage = [&](int buf, int ks, int ky, int kx, int c0, bool live) {
        stage_w(buf, ks, live);
        stage_x(0, buf, ky, kx, c0, live);
        stage_x(1, buf, ky, kx, c0, live);
    };

    f4 acc[CF][PF];
#pragma unroll
    for (int a = 0; a < CF; ++a)
#pragma unroll
        for (int b = 0; b < PF; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.ktot >> 5;
    int ky = 0, kx = 0, c0 = 0;     // coordinates of the NEXT step to stage
    auto advance = [&]() {      // branch-free (the K step must stay one basic block for the scheduler)
        c0 += 32;
        const bool wc0 = (c0 == p.k_per_tap);
        c0 = wc0 ? 0 : c0;
        kx += wc0 ? 1 : 0;
        const bool wkx = (kx == p.kw);
        kx = wkx ? 0 : kx;
        ky += wkx ? 1 : 0;
    };

    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = (wr * (TC / 2) + lrow) * 64 + qsw;
    const int b_off = (wc * (TP / 4) + lrow) * 64 + qsw;

    struct Frags {
        h8 a_hi[CF], a_lo[CF], b_hi[PF], b_lo[PF];
    };
    auto load_frags = [&](int slot, Frags& f) {
        const char* sb = smem + slot * STAGE;
#pragma unroll
        for (int a = 0; a < CF; ++a) {
            f.a_hi[a] = *(const h8*)(sb + OFF_WHI + a_off + a * 1024);
            f.a_lo[a] = *(const h8*)(sb + OFF_WLO + a_off + a * 1024);
        }
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            f.b_hi[b] = *(const h8*)(sb + OFF_XHI + b_off + b * 1024);
            f.b_lo[b] = *(const h8*)(sb + OFF_XLO + b_off + b * 1024);
        }
    };
    auto mfma_row = [&](const Frags& f, int a) {        // PF*3 MFMAs: one 16-channel row of fragments
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_lo[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_lo[a], f.b_hi[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_hi[b], acc[a][b], 0, 0, 0);
        }
    };
    auto mfma_all = [&](const Frags& f) {
#pragma unroll
        for (int a = 0; a < CF; ++a) mfma_row(f, a);
    };

    // Pipeline: the fragment registers are a 4th stage.  While the MFMAs of step ks run from registers,
    // the fragments of step ks+1 are read from the ring and the DMAs of steps ks+2, ks+3 are in flight.
    // prologue: stages 0..NS-1 issued (dummies past the end of K), fragments of step 0 in registers
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        stage(s, s, ky, kx, c0, s < nk);
        advance();
    }
    wait_vmcnt<2 * LPT>();
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    load_frags(0, fa);

    int slot = 0;                   // ring slot of step ks
    // a step that has a successor: frags of ks are in `cur`; leaves frags of ks+1 in `nxt`.  Branch-free.
    // The issue order inside a step is pinned with sched_barrier(0) fences (hipcc otherwise clumps the DMA
    // address arithmetic ahead of the MFMAs and the fragment reads behind them, which exposes both):
    //   the MFMAs of step ks in rows of PF*3; after row 0 the fragment reads of step ks+1 (their LDS latency
    //   hides under the later rows) and between rows one piece of the DMA issue for step ks+3 -- while this wave
    //   does address arithmetic, its SIMD partner (waves w and w+4 share a SIMD) has the matrix pipe.
    auto full_step = [&](int ks, const Frags& cur, Frags& nxt) {
        // own pieces of stage ks+1 landed (stage ks+2 stays in flight), and -- lgkmcnt(0) -- this wave's
        // reads of slot(ks) returned; the barrier then frees slot(ks) for stage ks+3
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<LPT>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nslot = (slot + 1 == NS) ? 0 : slot + 1;
        const bool live = ks + NS < nk;
        // row 0 first: hipcc cannot see the inline-asm wait above and guards the first use of `cur` with its
        // own lgkmcnt(0); placed here it finds nothing outstanding (behind the reads it would wait for them)
        mfma_row(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(nslot, nxt);
        stage_w(slot, ks + NS, live);
        if (CF == 2) {
            stage_x(0, slot, ky, kx, c0, live);
            stage_x(1, slot, ky, kx, c0, live);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(cur, 1);
        } else {
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            stage_x(0, slot, ky, kx, c0, live);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            stage_x(1, slot, ky, kx, c0, live);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(cur, CF - 1);
        }
        advance();
        slot = nslot;
    };
    int ks = 0;
    for (; ks + 2 < nk; ks += 2) {
        full_step(ks, fa, fb);
        full_step(ks + 1, fb, fa);
    }
    if (ks + 2 == nk) {
        full_step(ks, fa, fb);
        mfma_all(fb);
    } else {
        mfma_all(fa);
    }
    wait_vmcnt<0>();                // the trailing dummy DMAs must land before the epilogue reuses the LDS

    // ---- epilogue ----
    // Phase 0: prefetch the residual rows this thread will own in phase 2 (whole 16-B chunks).
    constexpr int GPP = TC / 8;                 // threads per pixel row (8 channels each)
    constexpr int PPI = CONV_THREADS / GPP;     // pixels per phase-2 iteration
    constexpr int ITERS = TP / PPI;
    constexpr int RP = TC * 4;                  // fp32 tile row pitch in bytes
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
                rh[it] = *(const h8*)(p.r_hi + o);
                rl[it] = *(const h8*)(p.r_lo + o);
            }
        }
    }
    __syncthreads();                            // all fragment reads of the last stage are done
    // Phase 1: acc*scale+shift -> fp32 tile [pixel][cout] in LDS; D row = cout (lane>>4)*4+reg, col = pixel.
#pragma unroll
    for (int a = 0; a < CF; ++a) {
        const int col = wr * (TC / 2) + a * 16 + (lane >> 4) * 4;       // cout within the tile
        const f4 sc = *(const f4*)(p.scale + n0 + col);
        const f4 sh = *(const f4*)(p.shift + n0 + col);
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            const int pl = wc * (TP / 4) + b * 16 + lrow;               // pixel within the tile
            const f4 v = acc[a][b] * sc + sh;
            *(f4*)(smem + pl * RP + (((col >> 2) ^ (pl & 7)) << 4)) = v;
        }
    }
    __syncthreads();
    // Phase 2: one thread = 8 consecutive channels of one pixel.
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
        if (p.y_f32) {
            *(f4*)(p.y_f32 + o) = (f4){v[0], v[1], v[2], v[3]};
            *(f4*)(p.y_f32 + o + 4) = (f4){v[4], v[5], v[6], v[7]};
        } else {
            h8 oh, ol;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t hi, lo;
                split_f32(v[j], hi, lo);
                oh[j] = hi;
                ol[j] = lo;
            }
            *(h8*)(p.y_hi + o) = oh;
            *(h8*)(p.y_lo + o) = ol;
        }
    }
}

}  // namespace mpx
