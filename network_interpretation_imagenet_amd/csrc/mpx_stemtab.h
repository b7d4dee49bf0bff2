// mpx_stemtab.h -- the ImageNet stem for the M masked copies of ONE image by SUPERPOSITION (round 4).
//
// The reference multiplies the normalised image by a {0,1} pixel mask that is a union of superpixels and runs the network on each
// copy (generate_gp_training_data_imagenet.py:234-246).  The first layer is linear, so for a conv output pixel q
//     conv1(x * mask_m)[q] = sum over the superpixels s that the 7x7 window of q touches of  onoff[m][s] * R[q][s],
//     R[q][s] = sum over the taps of that window that lie in superpixel s of  w[tap] * x[tap]          (mask-independent)
// A window touches 1 .. 4 superpixels on a 16-pixel grid (1.9 on average) and a handful on felzenszwalb maps.  So per IMAGE the
// table R is built once (one fp32 conv of the image, its taps bucketed by label: stemtab_*_kernel), and per MASK the stem is a gather:
// y = relu(bn(sum of the kept entries)), 3x3 stride-2 max pool, split into the hi + lo planes (stem_apply_kernel) -- no MFMA, no
// staged input: K0's 846 KB per masked image are neither written nor read, and the 118 MMAC of the stem conv per mask become ~20 adds
// per pooled pixel and channel.  Arithmetic: the table entries are fp32 FMA chains over the taps in raster order (the reference's own
// arithmetic type), the mask sum adds them in the order of first occurrence of their label in the window; the result differs from the
// MFMA stem (22-bit operands, another summation order) by rounding only.
//
// Table layout (CSR over the 112 x 112 conv output pixels, row-major): off[q] .. off[q+1] index the entries of pixel q;
// lab[e] = superpixel rank of entry e, vec[e][64] = its 64 output channels (pre-BatchNorm).  Labels outside [0, S) and taps in the zero
// padding contribute nothing (K0 treats an out-of-range label as "removed" too).  Capacity is the worst case, 49 entries per pixel.
#pragma once
#include "mpx_conv.h"

namespace mpx {

constexpr int ST_IN = 224, ST_CONV = 112, ST_POOLED = 56, ST_TAPS = 49, ST_C = 64;
constexpr int ST_NPIX = ST_CONV * ST_CONV;              // 12544 conv output pixels
constexpr int ST_MAX_ENTRIES = ST_NPIX * ST_TAPS;

struct StemTabParams {
    const uint8_t* img_u8;   // [224][224][3] or null
    const float* img_f32;    // [3][224][224] or null
    const int32_t* seg;      // [224][224] ranks
    float mean[3], std[3];
    int S;
    const float* w;          // [147][64] fp32: w[(tap * 3 + ch) * 64 + cout], tap = ky * 7 + kx
    int* cnt;                // [ST_NPIX]
    int* off;                // [ST_NPIX + 1]
    int* lab;                // [ST_MAX_ENTRIES]
    float* vec;              // [ST_MAX_ENTRIES][64]
};

// label of tap t (0..48) of conv output pixel (oy, ox), or -1 for a tap in the padding / with a label outside [0, S)
__device__ __forceinline__ int st_tap_label(const StemTabParams& p, int oy, int ox, int t, int* pix_out) {
    const int ky = t / 7, kx = t - ky * 7;
    const int iy = 2 * oy - 3 + ky, ix = 2 * ox - 3 + kx;
    if ((unsigned)iy >= (unsigned)ST_IN || (unsigned)ix >= (unsigned)ST_IN) return -1;
    const int pix = iy * ST_IN + ix;
    *pix_out = pix;
    const int l = p.seg[pix];
    return (unsigned)l < (unsigned)p.S ? l : -1;
}

// One wave = one conv output pixel.  Lane t < 49 owns tap t: `first` = its label occurs at no earlier tap.
__device__ __forceinline__ unsigned long long st_first_occurrences(const int* s_lab, int lane, int my_label) {
    bool first = lane < ST_TAPS && my_label >= 0;
    for (int t = 0; t < lane && t < ST_TAPS; ++t) first = first && (s_lab[t] != my_label);
    return __ballot(first);
}

__global__ __launch_bounds__(256) void stemtab_count_kernel(const StemTabParams p) {
    __shared__ int s_lab[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const int oy = q / ST_CONV, ox = q - oy * ST_CONV;
    int pix = 0;
    const int l = lane < ST_TAPS ? st_tap_label(p, oy, ox, lane, &pix) : -1;
    s_lab[wave][lane] = l;
    __syncthreads();
    const unsigned long long firsts = st_first_occurrences(s_lab[wave], lane, l);
    if (lane == 0) p.cnt[q] = __popcll(firsts);
}

// exclusive scan of the 12544 counts (one workgroup)
__global__ __launch_bounds__(1024) void stemtab_scan_kernel(const int* __restrict__ cnt, int* __restrict__ off) {
    __shared__ int s[1024];
    constexpr int PER = (ST_NPIX + 1023) / 1024;
    const int t = threadIdx.x, base = t * PER;
    int sum = 0;
    for (int i = 0; i < PER; ++i)
        if (base + i < ST_NPIX) sum += cnt[base + i];
    s[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? s[t - d] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    int run = s[t] - sum;
    for (int i = 0; i < PER; ++i)
        if (base + i < ST_NPIX) {
            off[base + i] = run;
            run += cnt[base + i];
        }
    if (t == 1023) off[ST_NPIX] = s[1023];
}

__global__ __launch_bounds__(256) void stemtab_fill_kernel(const StemTabParams p) {
    __shared__ int s_lab[4][64];
    __shared__ float s_x[4][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const int oy = q / ST_CONV, ox = q - oy * ST_CONV;
    int pix = 0;
    const int l = lane < ST_TAPS ? st_tap_label(p, oy, ox, lane, &pix) : -1;
    float x[3] = {0.f, 0.f, 0.f};
    if (l >= 0) {
        if (p.img_u8) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {       // K0's arithmetic (ToTensor .div(255), Normalize .sub_(mean).div_(std), each rounded in fp32)
                const float t = __fdiv_rn((float)p.img_u8[(size_t)pix * 3 + c], 255.0f);
                x[c] = __fdiv_rn(__fsub_rn(t, p.mean[c]), p.std[c]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = p.img_f32[(size_t)c * ST_IN * ST_IN + pix];
        }
    }
    s_lab[wave][lane] = l;
#pragma unroll
    for (int c = 0; c < 3; ++c) s_x[wave][c][lane] = x[c];
    __syncthreads();
    unsigned long long firsts = st_first_occurrences(s_lab[wave], lane, l);
    int e = p.off[q];
    const float* w = p.w + lane;                // lane = output channel
    while (firsts) {
        const int t0 = __ffsll((long long)firsts) - 1;
        firsts &= firsts - 1;
        const int u = s_lab[wave][t0];          // wave-uniform: the label of this entry
        float acc = 0.f;
        for (int t = t0; t < ST_TAPS; ++t) {    // taps in raster order; earlier taps cannot carry a label that first occurs at t0
            if (s_lab[wave][t] != u) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc = fmaf(w[(t * 3 + c) * ST_C], s_x[wave][c][t], acc);
        }
        p.vec[(size_t)e * ST_C + lane] = acc;
        if (lane == 0) p.lab[e] = u;
        ++e;
    }
}

// bits[s * nmb + mb] bit j = onoff[(mb * 32 + j)][s] != 0   (mask block mb of 32 masks; missing masks read as 0)
__global__ __launch_bounds__(256) void onoff_bitplanes_kernel(const uint8_t* __restrict__ onoff, int M, int S, int nmb,
                                                              unsigned* __restrict__ bits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * nmb) return;
    const int mb = i / S, s = i - mb * S;       // consecutive threads = consecutive s: the byte loads of one mask row coalesce
    unsigned w = 0;
    for (int j = 0; j < 32; ++j) {
        const int m = mb * 32 + j;
        if (m < M && onoff[(size_t)m * S + s]) w |= 1u << j;
    }
    bits[(size_t)s * nmb + mb] = w;
}

struct StemApplyParams {
    const int* off;
    const int* lab;
    const float* vec;
    const unsigned* bits;    // [S][nmb]
    const float* s;          // BatchNorm scale gamma / sqrt(var + eps), [64]
    const float* t;          // BatchNorm shift beta - mean * scale, [64]
    half_t* out_hi;          // pooled planes [max_batch][56][56][64]
    half_t* out_lo;
    int nmb, M, slot0;
};

// One wave = one pooled pixel x one block of 32 masks; lane = channel.  The entries of the (up to) 3 x 3 conv pixels under the pooled
// pixel -- per conv row ONE contiguous CSR range -- are loaded once into registers (NE of them; a pooled pixel with more runs the slow
// path that re-reads them per mask), their keep bits of the 32 masks are one word each, and the mask loop is adds, one fma + two max
// per conv pixel, one split and two 128-B stores.
template <int NE>
__device__ __forceinline__ void stem_apply_body(const StemApplyParams& p, int lane, int mcount, size_t out0, const int (&rb)[3][4],
                                                const int (&rlen)[3], int n_e, int n_empty, float sc, float sh, int mb) {
    float R[NE];
    unsigned kb[NE];
    unsigned long long starts = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        // entry i of the pooled pixel = entry g of the table: rows in order, each row one contiguous range
        int g = -1, r = 0, j = i;
        if (j < rlen[0]) { g = rb[0][0] + j; r = 0; }
        else if ((j -= rlen[0]) < rlen[1]) { g = rb[1][0] + j; r = 1; }
        else if ((j -= rlen[1]) < rlen[2]) { g = rb[2][0] + j; r = 2; }
        const bool ok = i < n_e;
        R[i] = ok ? p.vec[(size_t)g * ST_C + lane] : 0.f;
        const int label = ok ? p.lab[g] : 0;
        kb[i] = ok ? p.bits[(size_t)label * p.nmb + mb] : 0u;
        if (ok && (g == rb[r][0] || g == rb[r][1] || g == rb[r][2])) starts |= 1ull << i;     // first entry of a conv pixel
    }
    const float y_empty = fmaxf(sh, 0.f);       // a conv pixel none of whose taps lies in a kept superpixel: relu(bn(0))
    for (int m = 0; m < mcount; ++m) {
        float best = n_empty > 0 ? y_empty : -INFINITY;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (i < n_e) {          // wave-uniform
                if (i > 0 && (starts >> i & 1)) {
                    best = fmaxf(best, fmaxf(fmaf(v, sc, sh), 0.f));
                    v = 0.f;
                }
                v += (kb[i] >> m & 1) ? R[i] : 0.f;
            }
        }
        if (n_e > 0) best = fmaxf(best, fmaxf(fmaf(v, sc, sh), 0.f));
        half_t hi, lo;
        split_f32(best, hi, lo);
        const size_t o = out0 + (size_t)m * ST_POOLED * ST_POOLED * ST_C + lane;
        p.out_hi[o] = hi;
        p.out_lo[o] = lo;
    }
}

__global__ __launch_bounds__(256) void stem_apply_kernel(const StemApplyParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = blockIdx.x * 4 + wave;                        // pooled pixel (3136 = 784 * 4)
    const int mb = blockIdx.y;
    const int m0 = mb * 32;
    const int mcount = min(32, p.M - m0);
    const int py = P / ST_POOLED, px = P - py * ST_POOLED;
    // valid conv columns / rows under the pooled pixel (3x3 window, stride 2, pad 1)
    const int cx0 = max(2 * px - 1, 0), cx1 = min(2 * px + 1, ST_CONV - 1);
    const int ncol = cx1 - cx0 + 1;
    int rb[3][4], rlen[3];                                      // per conv row: the CSR boundaries of its (up to 3) pixels, entries in the row
    int n_e = 0, n_empty = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int cy = 2 * py - 1 + dy;
        const bool row_ok = (unsigned)cy < (unsigned)ST_CONV;
        const int q0 = (row_ok ? cy : 0) * ST_CONV + cx0;
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[dy][j] = p.off[q0 + min(j, ncol)];
        if (!row_ok) {
#pragma unroll
            for (int j = 1; j < 4; ++j) rb[dy][j] = rb[dy][0];
        }
        rlen[dy] = rb[dy][3] - rb[dy][0];
        n_e += rlen[dy];
        if (row_ok)
            for (int j = 0; j < ncol; ++j) n_empty += rb[dy][j + 1] == rb[dy][j];
        // a boundary that is not the start of a pixel of this row must not be taken for one: pixels beyond ncol repeat the end offset,
        // which no entry of the row equals (entries are < the end offset)
    }
    const float sc = p.s[lane], sh = p.t[lane];
    const size_t out0 = (((size_t)(p.slot0 + m0) * ST_POOLED + py) * ST_POOLED + px) * ST_C;
    if (n_e <= 20) {
        stem_apply_body<20>(p, lane, mcount, out0, rb, rlen, n_e, n_empty, sc, sh, mb);
    } else if (n_e <= 44) {
        stem_apply_body<44>(p, lane, mcount, out0, rb, rlen, n_e, n_empty, sc, sh, mb);
    } else {
        // slow path (label maps with many superpixels under one window): the entries are re-read for every mask
        const float y_empty = fmaxf(sh, 0.f);
        for (int m = 0; m < mcount; ++m) {
            float best = n_empty > 0 ? y_empty : -INFINITY;
            for (int dy = 0; dy < 3; ++dy)
                for (int j = 0; j < 3; ++j) {
                    const int e0 = rb[dy][j], e1 = rb[dy][j + 1];
                    if (e1 == e0) continue;
                    float v = 0.f;
                    for (int e = e0; e < e1; ++e) {
                        const unsigned k = p.bits[(size_t)p.lab[e] * p.nmb + mb];
                        v += (k >> m & 1) ? p.vec[(size_t)e * ST_C + lane] : 0.f;
                    }
                    best = fmaxf(best, fmaxf(fmaf(v, sc, sh), 0.f));
                }
            half_t hi, lo;
            split_f32(best, hi, lo);
            const size_t o = out0 + (size_t)m * ST_POOLED * ST_POOLED * ST_C + lane;
            p.out_hi[o] = hi;
            p.out_lo[o] = lo;
        }
    }
}

}  // namespace mpx
