// mpx_kernels.h -- gfx950 (MI355X, CDNA4) device kernels of the masked-perturbation scorer.
//
// Data format between kernels ("split-fp16"): an activation tensor is two fp16 NHWC planes
// hi, lo with x ~= hi + lo (22 significant bits).  A convolution is the implicit GEMM
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (ky, kx, ci)
// issued on the fp16 MFMA pipe as three products  W_hi*X_lo + W_lo*X_hi + W_hi*X_hi  with
// fp32 accumulation (v_mfma_f32_16x16x32_f16), which reproduces fp32 convolution to ~1e-7
// relative (oracle/precision_study.py) at 1/3 of the fp16 MFMA rate = 5.3x the f32 MFMA rate.
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

// ------------------------------------------------------------------------------------------
// K1/K2: conv (implicit GEMM) + per-channel scale/shift (BatchNorm) + residual + ReLU + split
// ------------------------------------------------------------------------------------------
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
    int cout;                // real output channels (store bound and row pitch of y/r)
    int M;                   // B*ho*wo output pixels
    int n_tiles_c;           // cout_pad / TC
    int relu;
};

// One workgroup = 4 waves (2x2) computes a TC(cout) x TP(pixel) tile; K advances 32 per step.
// LDS stage = [W_hi TCx64B][W_lo TCx64B][X_hi TPx64B][X_lo TPx64B], double buffered and filled by
// global_load_lds_dwordx4 (lane-linear destination).  Rows are 64 B, so a ds_read_b128 of
// MFMA fragments would be 2-way bank conflicted; the 16-B chunk index is XORed with
// ((row>>3)&1)<<1, applied on the SOURCE address of the DMA and on the fragment read.
template <int TC, int TP>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CF = TC / 32;           // 16-row cout fragments per wave
    constexpr int PF = TP / 32;           // 16-col pixel fragments per wave
    constexpr int WCH = TC * 4 / 256;     // 16-B chunks per thread per W plane per step
    constexpr int XCH = TP * 4 / 256;
    constexpr int OFF_WHI = 0, OFF_WLO = TC * 64, OFF_XHI = TC * 128, OFF_XLO = TC * 128 + TP * 64;
    constexpr int STAGE = (TC + TP) * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

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

    // ---- per-thread staging bookkeeping ----
    int x_pixbase[XCH], x_iy0[XCH], x_ix0[XCH], x_q[XCH];
    const int howo = p.ho * p.wo;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int id = i * 256 + tid;
        const int row = id >> 2;
        x_q[i] = ((id & 3) ^ (((row >> 3) & 1) << 1)) * 8;
        const int m = m0 + row;
        const int n = m / howo;
        const int rem = m - n * howo;
        const int oy = rem / p.wo;
        const int ox = rem - oy * p.wo;
        x_pixbase[i] = n * p.hin * p.win;
        x_iy0[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
        x_ix0[i] = ox * p.stride - p.pad;
    }
    size_t w_off[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int id = i * 256 + tid;
        const int row = id >> 2;
        w_off[i] = (size_t)(n0 + row) * p.ktot + ((id & 3) ^ (((row >> 3) & 1) << 1)) * 8;
    }

    auto stage = [&](int buf, int ks, int ky, int kx, int c0) {
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const size_t o = w_off[i] + (size_t)ks * 32;
            const int d = (i * 256 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(p.w_hi + o), MPX_LDS_PTR(sb + OFF_WHI + d), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(p.w_lo + o), MPX_LDS_PTR(sb + OFF_WLO + d), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
            const bool ok = (unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win;
            const size_t o = (size_t)(x_pixbase[i] + iy * p.win + ix) * p.pix_stride + c0 + x_q[i];
            const half_t* s_hi = ok ? p.x_hi + o : p.zero_page;
            const half_t* s_lo = ok ? p.x_lo + o : p.zero_page;
            const int d = (i * 256 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(s_hi), MPX_LDS_PTR(sb + OFF_XHI + d), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(MPX_GLOBAL_PTR(s_lo), MPX_LDS_PTR(sb + OFF_XLO + d), 16, 0, 0);
        }
    };

    f4 acc[CF][PF];
#pragma unroll
    for (int a = 0; a < CF; ++a)
#pragma unroll
        for (int b = 0; b < PF; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.ktot >> 5;
    int ky = 0, kx = 0, c0 = 0;     // coordinates of the NEXT step to stage
    auto advance = [&]() {
        c0 += 32;
        if (c0 == p.k_per_tap) {
            c0 = 0;
            if (++kx == p.kw) { kx = 0; ++ky; }
        }
    };

    stage(0, 0, ky, kx, c0);
    advance();
    __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier

    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = (wr * (TC / 2) + lrow) * 64 + qsw;
    const int b_off = (wc * (TP / 2) + lrow) * 64 + qsw;

    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) {
            stage(cur ^ 1, ks + 1, ky, kx, c0);
            advance();
        }
        const char* sb = smem + cur * STAGE;
        h8 a_hi[CF], a_lo[CF], b_hi[PF], b_lo[PF];
#pragma unroll
        for (int a = 0; a < CF; ++a) {
            a_hi[a] = *(const h8*)(sb + OFF_WHI + a_off + a * 1024);
            a_lo[a] = *(const h8*)(sb + OFF_WLO + a_off + a * 1024);
        }
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            b_hi[b] = *(const h8*)(sb + OFF_XHI + b_off + b * 1024);
            b_lo[b] = *(const h8*)(sb + OFF_XLO + b_off + b * 1024);
        }
#pragma unroll
        for (int a = 0; a < CF; ++a)
#pragma unroll
            for (int b = 0; b < PF; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[a], b_lo[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[a], b_hi[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[a], b_hi[b], acc[a][b], 0, 0, 0);
            }
        __syncthreads();
    }

    // ---- epilogue: D row = cout (lane>>4)*4 + reg, D col = pixel lane&15 ----
#pragma unroll
    for (int a = 0; a < CF; ++a) {
        const int co = n0 + wr * (TC / 2) + a * 16 + (lane >> 4) * 4;
        if (co >= p.cout) continue;
        const f4 sc = *(const f4*)(p.scale + co);
        const f4 sh = *(const f4*)(p.shift + co);
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            const int pix = m0 + wc * (TP / 2) + b * 16 + lrow;
            if (pix >= p.M) continue;
            const size_t o = (size_t)pix * p.cout + co;
            f4 v = acc[a][b] * sc + sh;
            if (p.r_hi) {
                const h4 rh = *(const h4*)(p.r_hi + o);
                const h4 rl = *(const h4*)(p.r_lo + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)rh[j] + (float)rl[j];
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (p.y_f32) {
                *(f4*)(p.y_f32 + o) = v;
            } else {
                h4 oh, ol;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half_t hi, lo;
                    split_f32(v[j], hi, lo);
                    oh[j] = hi;
                    ol[j] = lo;
                }
                *(h4*)(p.y_hi + o) = oh;
                *(h4*)(p.y_lo + o) = ol;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K0: mask-apply + normalise.  One thread = one pixel; a block stages the on/off rows of MT masks
// in LDS ([MT][S] bytes) and loops over them, so the image and the label map are read once per
// MT masks and every store is a coalesced 512-B (planes) / 256-B (f32) wave row.
// ------------------------------------------------------------------------------------------
constexpr int K0_MT = 32;

struct MaskParams {
    const uint8_t* img_u8;   // [H][W][3] or null
    const float* img_f32;    // [3][H][W] or null
    const int32_t* seg;      // [H][W]
    const uint8_t* onoff;    // [M][S]
    half_t* out_hi;          // engine staging, slot 0 (padded NHWC4)
    half_t* out_lo;
    float* out_f32;          // [M][3][H][W] or null
    float mean[3], std[3];
    int M, S, slot0;
    int size, pad_size, border;   // 224, 230, 3
};

__global__ __launch_bounds__(256) void mask_apply_normalize_kernel(const MaskParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* s_onoff = (uint8_t*)smem;
    const int tid = threadIdx.x;
    const int m_base = blockIdx.y * K0_MT;
    const int mt = min(K0_MT, p.M - m_base);
    {   // stage mt*S bytes (4-byte words when aligned, bytes otherwise)
        const uint8_t* src = p.onoff + (size_t)m_base * p.S;
        const int nbytes = mt * p.S;
        for (int i = tid; i < nbytes; i += 256) s_onoff[i] = src[i];
    }
    __syncthreads();
    const int hw = p.size * p.size;
    const int pix = blockIdx.x * 256 + tid;
    if (pix >= hw) return;
    const int y = pix / p.size, x = pix - y * p.size;
    int s = p.seg[pix];
    const bool s_ok = (unsigned)s < (unsigned)p.S;
    if (!s_ok) s = 0;
    float v[3];
    if (p.img_u8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // ToTensor: u8 -> f32, .div(255); Normalize: .sub_(mean).div_(std); each op rounds in fp32
            const float t = __fdiv_rn((float)p.img_u8[(size_t)pix * 3 + c], 255.0f);
            v[c] = __fdiv_rn(__fsub_rn(t, p.mean[c]), p.std[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = p.img_f32[(size_t)c * hw + pix];
    }
    const size_t plane = (size_t)p.pad_size * p.pad_size * 4;
    const size_t pad_off = ((size_t)(y + p.border) * p.pad_size + (x + p.border)) * 4;
    for (int mi = 0; mi < mt; ++mi) {
        const float keep = (s_ok && s_onoff[mi * p.S + s]) ? 1.0f : 0.0f;
        const int m = m_base + mi;
        float o[3];
        h4 oh, ol;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = v[c] * keep;     // x * mask, as numpy does it (-x*0 = -0.0)
            half_t hi, lo;
            split_f32(o[c], hi, lo);
            oh[c] = hi;
            ol[c] = lo;
        }
        oh[3] = (half_t)0.f;
        ol[3] = (half_t)0.f;
        const size_t so = (size_t)(p.slot0 + m) * plane + pad_off;
        *(h4*)(p.out_hi + so) = oh;
        *(h4*)(p.out_lo + so) = ol;
        if (p.out_f32) {
#pragma unroll
            for (int c = 0; c < 3; ++c) p.out_f32[((size_t)m * 3 + c) * hw + pix] = o[c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// K3: maxpool 3x3 stride 2 pad 1 on split planes (one thread = 8 channels of one output pixel)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const half_t* __restrict__ in_hi,
                                                            const half_t* __restrict__ in_lo,
                                                            half_t* __restrict__ out_hi,
                                                            half_t* __restrict__ out_lo, int B, int hin,
                                                            int c) {
    const int ho = hin / 2;
    const int cg = c / 8;
    const size_t total = (size_t)B * ho * ho * cg;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int g = (int)(t % cg);
        size_t r = t / cg;
        const int ox = (int)(r % ho);
        r /= ho;
        const int oy = (int)(r % ho);
        const int n = (int)(r / ho);
        float best[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)hin) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)hin) continue;
                const size_t o = (((size_t)n * hin + iy) * hin + ix) * c + g * 8;
                const h8 vh = *(const h8*)(in_hi + o);
                const h8 vl = *(const h8*)(in_lo + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) best[j] = fmaxf(best[j], (float)vh[j] + (float)vl[j]);
            }
        }
        h8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t hi, lo;
            split_f32(best[j], hi, lo);
            oh[j] = hi;
            ol[j] = lo;
        }
        const size_t o = (((size_t)n * ho + oy) * ho + ox) * c + g * 8;
        *(h8*)(out_hi + o) = oh;
        *(h8*)(out_lo + o) = ol;
    }
}

// ------------------------------------------------------------------------------------------
// K4a: global average pool [B][hw][c] -> [B][c] (one thread = 8 channels of one image)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void global_avgpool_kernel(const half_t* __restrict__ in_hi,
                                                              const half_t* __restrict__ in_lo,
                                                              half_t* __restrict__ out_hi,
                                                              half_t* __restrict__ out_lo, int B, int hw,
                                                              int c) {
    const int cg = c / 8;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * cg) return;
    const int g = t % cg, n = t / cg;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = 0.f;
    for (int i = 0; i < hw; ++i) {
        const size_t o = ((size_t)n * hw + i) * c + g * 8;
        const h8 vh = *(const h8*)(in_hi + o);
        const h8 vl = *(const h8*)(in_lo + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] += (float)vh[j] + (float)vl[j];
    }
    h8 oh, ol;
    const float denom = (float)hw;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        half_t hi, lo;
        split_f32(__fdiv_rn(sum[j], denom), hi, lo);
        oh[j] = hi;
        ol[j] = lo;
    }
    const size_t o = (size_t)n * c + g * 8;
    *(h8*)(out_hi + o) = oh;
    *(h8*)(out_lo + o) = ol;
}

// ------------------------------------------------------------------------------------------
// K4b: softmax + gather(label) + argmax, one wave per image
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_softmax_gather_kernel(const float* __restrict__ logits,
                                                                  const int32_t* __restrict__ label,
                                                                  float* __restrict__ score,
                                                                  int32_t* __restrict__ pred, int B,
                                                                  int ncls) {
    const int lane = threadIdx.x & 63;
    const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (img >= B) return;
    const float* row = logits + (size_t)img * ncls;
    float mx = -INFINITY;
    int arg = 0;
    for (int i = lane; i < ncls; i += 64) {
        const float v = row[i];
        if (v > mx) { mx = v; arg = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float omx = __shfl_xor(mx, off);
        const int oarg = __shfl_xor(arg, off);
        if (omx > mx || (omx == mx && oarg < arg)) { mx = omx; arg = oarg; }
    }
    float sum = 0.f;
    for (int i = lane; i < ncls; i += 64) sum += expf(row[i] - mx);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) {
        int lb = label[img];
        const bool ok = (unsigned)lb < (unsigned)ncls;
        score[img] = ok ? __fdiv_rn(expf(row[lb] - mx), sum) : 0.f;
        pred[img] = arg;
    }
}

}  // namespace mpx
