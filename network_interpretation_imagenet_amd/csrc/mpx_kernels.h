// mpx_kernels.h -- gfx950 (MI355X, CDNA4) device kernels of the masked-perturbation scorer.
//
// Data format between kernels ("split-fp16"): an activation tensor is two fp16 NHWC planes
// hi, lo with x ~= hi + lo (22 significant bits).  A convolution is the implicit GEMM
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (ky, kx, ci)
// issued on the fp16 MFMA pipe as three products  W_hi*X_lo + W_lo*X_hi + W_hi*X_hi  with
// fp32 accumulation (v_mfma_f32_16x16x32_f16), which reproduces fp32 convolution to ~1e-7
// relative (oracle/precision_study.py) at 1/3 of the fp16 MFMA rate = 5.3x the f32 MFMA rate.
#pragma once
#include "mpx_conv.h"

namespace mpx {

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
        *(h4*)(p.out_hi + so) = oh;       // plain stores: the stem reads these planes next, out of L2 / Infinity Cache
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
                                                                  int ncls, int pitch) {
    const int lane = threadIdx.x & 63;
    const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (img >= B) return;
    const float* row = logits + (size_t)img * pitch;
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

// ------------------------------------------------------------------------------------------
// DownsampleB of the reference's CIFAR ResNet (models/resnet.py:64-74): AvgPool2d(2) on the identity, then zero
// channels up to the block's width.  Planes [B][hin][hin][cin_p] -> [B][hin/2][hin/2][cout_p]; one thread = 8 channels
// of one output pixel; the four taps are summed in torch's order (row-major) and divided by 4 in fp32.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_pad_kernel(const half_t* __restrict__ in_hi, const half_t* __restrict__ in_lo,
                                                           half_t* __restrict__ out_hi, half_t* __restrict__ out_lo, int B,
                                                           int hin, int cin_p, int cout_p) {
    const int ho = hin / 2, cg = cout_p / 8;
    const size_t total = (size_t)B * ho * ho * cg;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int g = (int)(t % cg);
        size_t r = t / cg;
        const int ox = (int)(r % ho);
        r /= ho;
        const int oy = (int)(r % ho);
        const int n = (int)(r / ho);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (g * 8 < cin_p) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t o = (((size_t)n * hin + oy * 2 + dy) * hin + ox * 2 + dx) * cin_p + g * 8;
                    const h8 vh = *(const h8*)(in_hi + o);
                    const h8 vl = *(const h8*)(in_lo + o);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = __fadd_rn(acc[j], (float)vh[j] + (float)vl[j]);
                }
        }
        h8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t hi, lo;
            split_f32(__fdiv_rn(acc[j], 4.0f), hi, lo);
            oh[j] = hi;
            ol[j] = lo;
        }
        const size_t o = (((size_t)n * ho + oy) * ho + ox) * cout_p + g * 8;
        *(h8*)(out_hi + o) = oh;
        *(h8*)(out_lo + o) = ol;
    }
}

// ------------------------------------------------------------------------------------------
// K0 of the small networks: the CIFAR / MNIST scorers' mask convention (generate_gp_training_data_cifar.py:274-321,
// generate_gp_training_data_mnist.py:167-242): the picture is min-max scaled to [0,255] in place (x255), the SELECTED
// superpixels are switched OFF by a {0,255} mask, the product is min-max scaled to [0,255] again and multiplied by
// f32(1/255).  min(x255 * mask) is exactly 0 (x255 contains an exact 0, and 0 * anything = 0), so the second rescale
// only needs max over the KEPT pixels of fl(x255 * 255) = fl(255 * max kept x255): per-superpixel maxima once per image,
// then a max over each mask's kept superpixels.  Every fp32 operation is rounded in the reference's order.
//   stats[0] = min(x), stats[1] = fl(max(x) - min(x)), stats[2 .. 2+S) = per-superpixel max of x255, then per-mask max.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float minmax255(float x, float mn, float range) {
    return __fmul_rn(__fdiv_rn(__fsub_rn(x, mn), range), 255.0f);
}

__global__ __launch_bounds__(256) void smallnet_image_stats_kernel(const float* __restrict__ img, const int32_t* __restrict__ seg,
                                                                   int C, int hw, int S, float* __restrict__ stats) {
    __shared__ float s_mn[256], s_mx[256];
    __shared__ unsigned s_seg[4096];
    const int tid = threadIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < C * hw; i += 256) {
        const float v = img[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    s_mn[tid] = mn;
    s_mx[tid] = mx;
    for (int i = tid; i < S; i += 256) s_seg[i] = 0u;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            s_mn[tid] = fminf(s_mn[tid], s_mn[tid + o]);
            s_mx[tid] = fmaxf(s_mx[tid], s_mx[tid + o]);
        }
        __syncthreads();
    }
    const float gmn = s_mn[0];
    const float range = __fsub_rn(s_mx[0], gmn);       // = max(x - min): subtraction is monotonic
    for (int i = tid; i < C * hw; i += 256) {
        const int sgm = seg[i % hw];
        if ((unsigned)sgm < (unsigned)S) atomicMax(&s_seg[sgm], __float_as_uint(minmax255(img[i], gmn, range)));   // values >= 0
    }
    __syncthreads();
    if (tid == 0) {
        stats[0] = gmn;
        stats[1] = range;
    }
    for (int i = tid; i < S; i += 256) stats[2 + i] = __uint_as_float(s_seg[i]);
}

__global__ __launch_bounds__(256) void smallnet_mask_max_kernel(const uint8_t* __restrict__ removed, int M, int S,
                                                                const float* __restrict__ stats, float* __restrict__ mask_max) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float mx = 0.f;                                     // removed pixels contribute 0 * x = 0
    for (int sgm = 0; sgm < S; ++sgm)
        if (!removed[(size_t)m * S + sgm]) mx = fmaxf(mx, stats[2 + sgm]);
    mask_max[m] = __fmul_rn(mx, 255.0f);                // max over pixels of fl(x255 * 255) (the product is monotonic)
}

struct SmallMaskParams {
    const float* img;        // [C][H][W] as the loader yields it
    const int32_t* seg;      // [H][W] ranks
    const uint8_t* removed;  // [M][S], 1 = superpixel switched off
    const float* stats;      // smallnet_image_stats_kernel
    const float* mask_max;   // [M]
    half_t* out_hi;          // staging [slot][H][W][32]
    half_t* out_lo;
    float* out_f32;          // [M][C][H][W] or null
    int C, hw, M, S, slot0;
};

__global__ __launch_bounds__(256) void smallnet_mask_apply_kernel(const SmallMaskParams p) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (pix >= p.hw) return;
    const int sgm = p.seg[pix];
    const bool keep = (unsigned)sgm < (unsigned)p.S && !p.removed[(size_t)m * p.S + sgm];
    const float mmax = p.mask_max[m];
    h8 oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        oh[j] = (half_t)0.f;
        ol[j] = (half_t)0.f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c >= p.C) break;
        const float x255 = minmax255(p.img[(size_t)c * p.hw + pix], p.stats[0], p.stats[1]);
        const float a = __fmul_rn(x255, keep ? 255.0f : 0.0f);          // org_img * mask
        const float b = __fsub_rn(a, 0.0f);                             // masked -= masked.min()  (the minimum is exactly 0)
        const float d = __fmul_rn(__fdiv_rn(b, mmax), 255.0f);          // /= max; *= 255
        const float v = __fmul_rn(d, 0.003921568859368563f);            // normalize_image: * f32(1/255)
        half_t hi, lo;
        split_f32(v, hi, lo);
        oh[c] = hi;
        ol[c] = lo;
        if (p.out_f32) p.out_f32[((size_t)m * p.C + c) * p.hw + pix] = v;
    }
    const size_t so = ((size_t)(p.slot0 + m) * p.hw + pix) * 32;
    *(h8*)(p.out_hi + so) = oh;        // channels [0,8) of the 32 per pixel; the rest stay zero from mpx_create
    *(h8*)(p.out_lo + so) = ol;
}

// ------------------------------------------------------------------------------------------
// K5: heat-map accumulation  y[p] += sum_m [pred[m] == label[m]] * onoff[m][seg[p]]
// (gp_superpixel_data_imagenet.py:322-323 / gp_regression.py:82-94).  Two tiny launches: per-superpixel
// counts (one thread per superpixel, coalesced over s), then a gather over the 50,176 pixels.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heatmap_segment_count_kernel(const uint8_t* __restrict__ onoff,
                                                                    const int32_t* __restrict__ pred,
                                                                    const int32_t* __restrict__ label, int M, int S,
                                                                    float* __restrict__ per_segment) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    int count = 0;
    for (int m = 0; m < M; ++m) count += (pred[m] == label[m] && onoff[(size_t)m * S + s]) ? 1 : 0;
    per_segment[s] = (float)count;
}

__global__ __launch_bounds__(256) void heatmap_gather_kernel(const int32_t* __restrict__ seg,
                                                             const float* __restrict__ per_segment, int S, int npix,
                                                             float* __restrict__ heat) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int s = seg[p];
    if ((unsigned)s < (unsigned)S) heat[p] += per_segment[s];
}

}  // namespace mpx
