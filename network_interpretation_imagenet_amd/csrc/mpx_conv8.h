// mpx_conv8.h -- conv + BN + residual + ReLU with the cross terms on the block-scaled fp8 MFMA ("f16f8" precision).
//
//   D = W_hi*X_hi                      2 x v_mfma_f32_32x32x16_f16 per 32x32 block and K step of 32
//     + W_h8*X_l8 + W_l8*X_h8          1 x v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3), the two products K-concatenated:
//                                       K block 0 = W_h8 . X_l8, K block 1 = W_l8 . X_h8, each with its E8M0 scale
// where X = X_hi (fp16) + X_l8 (e4m3 of the fp16 remainder * 2^11), X_h8 = e4m3(X_hi), and the same for W (with its
// own power-of-two factors).  The correction terms are ~2^-11 of the main term, so 3 mantissa bits on them leave a
// relative error of ~2^-16 per product: scores move by ~2e-5 (tolerance 1e-4) while the MFMA work per K step drops
// from 48 x 16 to 4 x (32 + 32 + 64) = 512 cycles per wave (tools/probes/mfma_scale32_probe.hip: 1.55x in a bare loop).
//
// Activation format: plane "hi" fp16 [M][C]; plane "8" bytes [M][C/32][64] = per 32 channels [l8 x 32 | h8 x 32].  Both
// planes have 64 B per 32 channels, so tiles, rings, DMA pieces and the byte traffic are those of mpx_conv.h.  The 7x7
// stem reads a staging whose byte plane is split ([l8 plane][h8 plane], x_lo_split bytes apart) because its 32-element
// runs start at arbitrary pixels.
//
// Lane maps (probed, tools/probes/mfma_scale32_probe.hip): f16 32x32x16: lane l holds A[row l&31][k = 8(l>>5) + j];
// fp8 32x32x64: lane l holds 32 bytes, bytes 0..15 = K block 0 elements 16(l>>5) + t, bytes 16..31 = K block 1
// elements 16(l>>5) + t; the scale byte of lane l scales (row l&31, K block l>>5).  D: col = lane&31,
// row = (reg&3) + 8(reg>>2) + 4(lane>>5).  With 64-B LDS rows a 32-row ds_read_b128 is 4-way bank conflicted unless
// the 16-B chunk index is XORed with (row>>2)&3 (on the DMA source address and on the read).
#pragma once
#include "mpx_conv.h"

namespace mpx {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr float F8_LO_SCALE = 2048.f;       // X_l8 = e4m3(X_lo * 2^11)
constexpr float F8_MAX = 448.f;             // e4m3fn largest finite
// Weights (after the per-cout power-of-two scaling to max|w| in [512, 1024)):  W_h8 = e4m3(W_hi / 4), W_l8 = e4m3(W_lo * 512).
// K block 0 = W_h8 . X_l8 = (W_hi / 4)(X_lo * 2^11) -> scale 2^-9;  K block 1 = W_l8 . X_h8 = (W_lo * 2^9) X_hi -> 2^-9.
constexpr float F8_WHI_SCALE = 0.25f, F8_WLO_SCALE = 512.f;
constexpr int F8_E8M0_BLOCK0 = 127 - 9, F8_E8M0_BLOCK1 = 127 - 9;

__device__ __forceinline__ float clamp448(float v) { return __builtin_fminf(__builtin_fmaxf(v, -F8_MAX), F8_MAX); }

// 8 fp32 values -> 8 e4m3 bytes (two dwords)
__device__ __forceinline__ i2v pack_fp8x8(const float* v) {
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[0]), clamp448(v[1]), w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[2]), clamp448(v[3]), w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[4]), clamp448(v[5]), w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[6]), clamp448(v[7]), w1, true);
    return (i2v){w0, w1};
}

__device__ __forceinline__ void unpack_fp8x8(i2v w, float* v) {
    f2v a = __builtin_amdgcn_cvt_pk_f32_fp8(w[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w[0], true);
    f2v c = __builtin_amdgcn_cvt_pk_f32_fp8(w[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8(w[1], true);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1]; v[6] = d[0]; v[7] = d[1];
}

// value -> (hi fp16, l8, h8) for 8 consecutive channels
__device__ __forceinline__ void split_f16f8(const float* v, h8& oh, i2v& l8, i2v& h8b) {
    float lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const half_t h = (half_t)v[j];
        oh[j] = h;
        hi[j] = (float)h;
        lo[j] = (v[j] - (float)h) * F8_LO_SCALE;
    }
    l8 = pack_fp8x8(lo);
    h8b = pack_fp8x8(hi);
}

// byte offset of channel c (multiple of 8) of pixel pix in the 8-bit plane: [pix][c/32][l8 x 32 | h8 x 32]
__device__ __forceinline__ size_t plane8_off(size_t pix, int cstride, int c) {
    return pix * (size_t)cstride * 2 + (size_t)(c >> 5) * 64 + (c & 31);
}

template <class C>
__global__ __launch_bounds__(C::NT, C::MINB) void conv_f16f8_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TC = C::TC, TP = C::TP, NSW = C::NSW, NSX = C::NSX, NW = C::NW, NT = C::NT;
    constexpr int WJ = C::WJ, XJ = C::XJ, WSTAGE = C::WSTAGE, XSTAGE = C::XSTAGE, XBASE = C::XBASE;
    constexpr int CB = TC / C::NWR / 32, PB = TP / C::NWC / 32;       // 32x32 blocks per wave
    static_assert(CB >= 1 && PB >= 1 && (TC / C::NWR) % 32 == 0 && (TP / C::NWC) % 32 == 0, "wave tile must be made of 32x32 blocks");
    constexpr int OFF_WHI = 0, OFF_W8 = TC * 64, OFF_XHI = 0, OFF_X8 = TP * 64;

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

    // ---- DMA addressing (as mpx_conv.h; the chunk swizzle is (row>>2)&3) ---------------------------------------
    const int nk = p.ktot >> 5;
    const int prow = lane >> 2;
    const int sq = (lane & 3) ^ ((prow >> 2) & 3);                   // source chunk of this lane's LDS chunk
    const int howo = p.ho * p.wo;
    const int n_first = m0 / howo;
    const int img_elems = p.hin * p.win * p.pix_stride;
    constexpr unsigned OOB = 0x80000000u;
    const int split8 = p.x_lo_split;                                 // 0: interleaved byte plane; else stem staging
    __amdgpu_buffer_rsrc_t x_rs_hi, x_rs_8, w_rs_hi, w_rs_8;
    {
        const int n_img = p.M / howo;
        const size_t rem = (size_t)(n_img - n_first) * img_elems * 2;
        const int nrec = rem > 0x7fffffffu ? 0x7fffffff : (int)rem;
        x_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_hi + (size_t)n_first * img_elems), 0, nrec, 0x00020000);
        if (split8) {
            // split staging: [l8 plane][h8 plane], 1 byte per element each; the window covers both planes
            const char* base8 = (const char*)p.x_lo + (size_t)n_first * img_elems;
            x_rs_8 = __builtin_amdgcn_make_buffer_rsrc((void*)base8, 0, 0x7fffffff, 0x00020000);
        } else {
            x_rs_8 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_lo + (size_t)n_first * img_elems), 0, nrec, 0x00020000);
        }
        const int wrec = TC * p.ktot * 2;
        w_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
        w_rs_8 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
    }
    int x_off0[XJ], x_off8[XJ], x_iy0[XJ], x_ix0[XJ];
#pragma unroll
    for (int i = 0; i < XJ; ++i) {
        const int m = m0 + (i * NW + wave) * 16 + prow;
        const int n = m / howo;
        const int rem = m - n * howo;
        const int oy = rem / p.wo;
        const int ox = rem - oy * p.wo;
        x_iy0[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
        x_ix0[i] = ox * p.stride - p.pad;
        const int e0 = (((n - n_first) * p.hin + x_iy0[i]) * p.win + x_ix0[i]) * p.pix_stride;     // element offset
        x_off0[i] = e0 * 2 + sq * 16;
        x_off8[i] = split8 ? e0 + (sq & 1) * 16 + (sq >> 1) * split8 : x_off0[i];
    }
    int w_off[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int piece = C::HALF_W ? (wave % (NW / 2)) : (j * NW + wave);
        w_off[j] = (piece * 16 + prow) * p.ktot * 2 + sq * 16;
    }
    auto stage_w = [&](int buf, int ks) {
        char* sb = smem + buf * WSTAGE;
        const int soff = ks * 64;
        const int dead = ks < nk ? 0 : (int)OOB;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (C::HALF_W) {
                if (wave < NW / 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + wave * 1024), 16, w_off[j] | dead, soff, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_8, MPX_LDS_PTR(sb + wave * 1024), 16, w_off[j] | dead, soff, 0, 0);
            } else {
                const int d = (j * NW + wave) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + OFF_WHI + d), 16, w_off[j] | dead, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_8, MPX_LDS_PTR(sb + OFF_W8 + d), 16, w_off[j] | dead, soff, 0, 0);
            }
        }
    };
    auto stage_x = [&](int i, int buf, int ky, int kx, int c0, bool live) {
        char* sb = smem + XBASE + buf * XSTAGE;
        const int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
        const int de = (ky * p.win + kx) * p.pix_stride + c0;              // wave-uniform, in elements
        const int oob = ((iy | (p.hin - 1 - iy) | ix | (p.win - 1 - ix)) & (int)OOB) | (live ? 0 : (int)OOB);
        const int voff = (x_off0[i] + de * 2) | oob;
        const int voff8 = (x_off8[i] + (split8 ? de : de * 2)) | oob;
        const int d = (i * NW + wave) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_hi, MPX_LDS_PTR(sb + OFF_XHI + d), 16, voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_8, MPX_LDS_PTR(sb + OFF_X8 + d), 16, voff8, 0, 0, 0);
    };

    f16v acc[CB][PB];
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    int ky = 0, kx = 0, c0 = 0;
    auto advance = [&]() {
        c0 += 32;
        const bool wc0 = (c0 == p.k_per_tap);
        c0 = wc0 ? 0 : c0;
        kx += wc0 ? 1 : 0;
        const bool wkx = (kx == p.kw);
        kx = wkx ? 0 : kx;
        ky += wkx ? 1 : 0;
    };

    // fragment addressing: row = lane&31 (+32 per block: same swizzle), half = lane>>5
    const int lrow = lane & 31, lh = lane >> 5;
    const int swz = (lrow >> 2) & 3;
    const int a_row = (wr * (TC / C::NWR) + lrow) * 64;
    const int b_row = (wc * (TP / C::NWC) + lrow) * 64;
    const int ch_k0 = ((0 + lh) ^ swz) * 16, ch_k1 = ((2 + lh) ^ swz) * 16;     // hi plane: K half 0 / 1; byte plane: block 0 / 1
    // E8M0 scale of (row, K block lane>>5): byte 0 of the A-side scale register; the B side is 2^0
    const int scale_a = lh ? F8_E8M0_BLOCK1 : F8_E8M0_BLOCK0;
    const int scale_b = 127;

    struct Frags {
        h8 a_hi[CB][2], b_hi[PB][2];
        i8v a8[CB], b8[PB];      // bytes 0..15: K block 0 (ds_read_b128 #1), bytes 16..31: K block 1 (#2)
    };
    constexpr int NF = 4 * (CB + PB);                   // fragment reads per step
    constexpr int NM = 3 * CB * PB;                     // MFMAs per step
    auto put_half = [](i8v& dst, int half, i4v v) {
        dst[4 * half] = v[0]; dst[4 * half + 1] = v[1]; dst[4 * half + 2] = v[2]; dst[4 * half + 3] = v[3];
    };
    auto load_frag = [&](int wslot, int xslot, Frags& f, int j) {
        const char* sw = smem + wslot * WSTAGE;
        const char* sx = smem + XBASE + xslot * XSTAGE;
        if (j < 4 * CB) {
            const int blk = j >> 2, what = j & 3;
            const char* base = sw + a_row + blk * 32 * 64;
            if (what == 0) f.a_hi[blk][0] = *(const h8*)(base + OFF_WHI + ch_k0);
            else if (what == 1) f.a_hi[blk][1] = *(const h8*)(base + OFF_WHI + ch_k1);
            else if (what == 2) put_half(f.a8[blk], 0, *(const i4v*)(base + OFF_W8 + ch_k0));
            else put_half(f.a8[blk], 1, *(const i4v*)(base + OFF_W8 + ch_k1));
        } else {
            const int jj = j - 4 * CB;
            const int blk = jj >> 2, what = jj & 3;
            const char* base = sx + b_row + blk * 32 * 64;
            if (what == 0) f.b_hi[blk][0] = *(const h8*)(base + OFF_XHI + ch_k0);
            else if (what == 1) f.b_hi[blk][1] = *(const h8*)(base + OFF_XHI + ch_k1);
            else if (what == 2) put_half(f.b8[blk], 0, *(const i4v*)(base + OFF_X8 + ch_k0));
            else put_half(f.b8[blk], 1, *(const i4v*)(base + OFF_X8 + ch_k1));
        }
    };
    auto load_frags = [&](int wslot, int xslot, Frags& f) {
#pragma unroll
        for (int j = 0; j < NF; ++j) load_frag(wslot, xslot, f, j);
    };
    auto mfma_one = [&](const Frags& f, int i) {        // order (a, b, term): f16 half 0, f16 half 1, fp8 cross terms
        const int a = i / (3 * PB), r = i % (3 * PB), b = r / 3, term = r % 3;
        if (term == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_hi[a][0], f.b_hi[b][0], acc[a][b], 0, 0, 0);
        else if (term == 1) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_hi[a][1], f.b_hi[b][1], acc[a][b], 0, 0, 0);
        else acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.a8[a], f.b8[b], acc[a][b], 0, 0, 0, scale_a, 0, scale_b);
    };
    auto mfma_all = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < NM; ++i) mfma_one(f, i);
    };

#pragma unroll
    for (int s = 0; s < NSX; ++s) {
        if (s < NSW) stage_w(s, s);
#pragma unroll
        for (int i = 0; i < XJ; ++i) stage_x(i, s, ky, kx, c0, s < nk);
        advance();
    }
    wait_vmcnt<C::WAIT_PROLOGUE>();
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    load_frags(0, 0, fa);

    int wslot = 0, xslot = 0;
    // One step: 12 MFMAs (8 f16 of 8 passes, 4 fp8 of 16 passes), 16 fragment reads of the next step, the DMA groups.
    // The NF reads go early (two per MFMA), so that the last MFMAs of the step cover their latency before the lgkmcnt(0) at
    // the next step's top; DMA groups after MFMAs 1, 3, 5, ...
    auto full_step = [&](int ks, const Frags& cur, Frags& nxt) {
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<C::WAIT_STEP>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nw = (wslot + 1 == NSW) ? 0 : wslot + 1;
        const int nx = (xslot + 1 == NSX) ? 0 : xslot + 1;
        const bool live = ks + NSX < nk;
        constexpr int G = 1 + XJ;
        int rd = 0;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma_one(cur, i);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int RPM = (NF + NM - 5) / (NM - 4) > 2 ? (NF + NM - 5) / (NM - 4) : 2;      // reads per MFMA: done >= 4 MFMAs before the step ends
            const int upto = RPM * (i + 1) < NF ? RPM * (i + 1) : NF;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (rd < upto) {
                    load_frag(nw, nx, nxt, rd);
                    ++rd;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (i == 1 + 2 * g) {
                    if (g == 0) stage_w(wslot, ks + NSW);
                    else stage_x(g - 1, xslot, ky, kx, c0, live);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        static_assert(1 + 2 * (G - 1) < NM, "DMA groups must fit into the step");
        static_assert(NF <= 4 * NM, "fragment reads must fit into the step");
        advance();
        wslot = nw;
        xslot = nx;
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
    wait_vmcnt<0>();

    // ---- epilogue ----
    constexpr int GPP = TC / 8;
    constexpr int PPI = NT / GPP;
    constexpr int ITERS = TP / PPI;
    constexpr int RP = TC * 4;
    const int g = tid % GPP;
    const int prow2 = tid / GPP;
    const int co8 = n0 + g * 8;
    const bool co_ok = co8 < p.cout;
    h8 rh[ITERS];
    i2v r8[ITERS];
    const char* r8p = (const char*)p.r_lo;
    if (p.r_hi) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int pix = m0 + it * PPI + prow2;
            if (co_ok && pix < p.M) {
                rh[it] = __builtin_nontemporal_load((const h8*)(p.r_hi + (size_t)pix * p.cout + co8));
                r8[it] = __builtin_nontemporal_load((const i2v*)(r8p + plane8_off(pix, p.cout, co8)));      // l8 of 8 channels
            }
        }
    }
    __syncthreads();
    // Phase 1: D row = cout (reg&3) + 8(reg>>2) + 4(lane>>5), col = pixel lane&31 -> fp32 tile [pixel][cout] in LDS
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wr * (TC / C::NWR) + a * 32 + 8 * q + 4 * lh;      // cout within the tile (4 consecutive)
            const f4 sc = *(const f4*)(p.scale + n0 + col);
            const f4 sh = *(const f4*)(p.shift + n0 + col);
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                const int pl = wc * (TP / C::NWC) + b * 32 + lrow;
                const f4 v = (f4){acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]} * sc + sh;
                *(f4*)(smem + pl * RP + (((col >> 2) ^ (pl & 7)) << 4)) = v;
            }
        }
    __syncthreads();
    char* y8p = (char*)p.y_lo;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int pl = it * PPI + prow2;
        const int pix = m0 + pl;
        if (!(co_ok && pix < p.M)) continue;
        const f4 v0 = *(const f4*)(smem + pl * RP + (((2 * g) ^ (pl & 7)) << 4));
        const f4 v1 = *(const f4*)(smem + pl * RP + (((2 * g + 1) ^ (pl & 7)) << 4));
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.r_hi) {
            float rl[8];
            unpack_fp8x8(r8[it], rl);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rh[it][j] + rl[j] * (1.0f / F8_LO_SCALE);
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
            h8 oh;
            i2v l8, h8b;
            split_f16f8(v, oh, l8, h8b);
            __builtin_nontemporal_store(oh, (h8*)(p.y_hi + o));
            const size_t o8 = plane8_off(pix, p.cout, co8);
            __builtin_nontemporal_store(l8, (i2v*)(y8p + o8));
            __builtin_nontemporal_store(h8b, (i2v*)(y8p + o8 + 32));
        }
    }
#endif
}

}  // namespace mpx
