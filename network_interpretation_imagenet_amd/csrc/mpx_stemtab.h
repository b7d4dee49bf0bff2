// mpx_stemtab.h -- the ImageNet stem for the M masked copies of ONE image by SUPERPOSITION (round 4).
//
// The reference multiplies the normalised image by a {0,1} pixel mask that is a union of superpixels and runs the network on each
// copy (generate_gp_training_data_imagenet.py:234-246).  The first layer is linear, so for a conv output pixel q
//     conv1(x * mask_m)[q] = sum over the superpixels s that the 7x7 window of q touches of  onoff[m][s] * R[q][s],
//     R[q][s] = sum over the taps of that window that lie in superpixel s of  w[tap] * x[tap]          (mask-independent)
// A window touches 1 .. 4 superpixels on a 16-pixel grid (1.9 on average) and a handful on felzenszwalb maps.  So per IMAGE the
// table R is built once (one fp32 conv of the image, its taps bucketed by label: stemtab_*_kernel), and per MASK the stem is a gather:
// y = relu(bn(sum of the kept entries)), 3x3 stride-2 max pool, split into the hi + lo planes (stem_apply_heavy_kernel for the few pooled
// pixels with many superpixels under one window, then stem_apply_kernel for all) -- no MFMA, no
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
    const int* __restrict__ off;
    const int* __restrict__ lab;
    const float* __restrict__ vec;
    const unsigned* __restrict__ bits;    // [S][nmb]
    const float* __restrict__ s;          // BatchNorm scale gamma / sqrt(var + eps), [64]
    const float* __restrict__ t;          // BatchNorm shift beta - mean * scale, [64]
    half_t* __restrict__ out_hi;          // pooled planes [max_batch][56][56][64]
    half_t* __restrict__ out_lo;
    const int* __restrict__ heavy_list;   // pooled pixels of the table in place that take the streaming path, and how many
    const int* __restrict__ heavy_count;
    int nmb, M, slot0;
};

// The four waves of a workgroup hold four neighbouring pooled pixels: their values of FOUR masks meet in LDS, then wave w writes mask
// 4g + w -- each plane's 512 bytes (four whole lines) with one store.  (A wave's own 128 bytes per plane and mask are 14.7 M
// two-byte-per-lane stores per forward batch, which the texture addresser serialises: 3.1 ms for the launch, as long as the MFMA stem.)
// One barrier per four masks, two buffers: a buffer is rewritten two groups later, and the barrier in between needs every wave to have
// read it.  Every wave of the workgroup calls this once per mask, in order.
__device__ __forceinline__ void stem_store4_packed(const StemApplyParams& p, unsigned (*s_out)[4][4][ST_C], int m, int mcount, int wave, int lane,
                                                   unsigned packed, size_t out0_row) {
    s_out[(m >> 2) & 1][m & 3][wave][lane] = packed;
    if ((m & 3) != 3 && m != mcount - 1) return;
    __syncthreads();
    const int ms = (m & ~3) + wave;             // the mask this wave writes
    if (ms < mcount) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u4 q = *(const u4*)&s_out[(m >> 2) & 1][ms & 3][lane >> 4][(lane & 15) * 4];
        u2 vh, vl;
        vh[0] = (q[0] & 0xffffu) | (q[1] << 16);
        vh[1] = (q[2] & 0xffffu) | (q[3] << 16);
        vl[0] = (q[0] >> 16) | (q[1] & 0xffff0000u);
        vl[1] = (q[2] >> 16) | (q[3] & 0xffff0000u);
        const size_t o = out0_row + (size_t)ms * ST_POOLED * ST_POOLED * ST_C + (size_t)(lane >> 4) * ST_C + (lane & 15) * 4;
        __builtin_nontemporal_store(vh, (u2*)(p.out_hi + o));
        __builtin_nontemporal_store(vl, (u2*)(p.out_lo + o));
    }
}

__device__ __forceinline__ void stem_store4(const StemApplyParams& p, unsigned (*s_out)[4][4][ST_C], int m, int mcount, int wave, int lane,
                                            float best, size_t out0_row) {
    half_t hi, lo;
    split_f32(best, hi, lo);
    stem_store4_packed(p, s_out, m, mcount, wave, lane,
                       (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16), out0_row);
}

// One wave = one pooled pixel x one block of 32 masks; lane = channel.  The entries of the (up to) 3 x 3 conv pixels under the pooled
// pixel are per conv row ONE contiguous CSR range.
struct StemRows {           // the three conv rows under a pooled pixel: CSR range [a, a + l) and the starts of its (up to) three pixels
    int a0, l0, a1, l1, a2, l2;
    int o00, o01, o02, o10, o11, o12, o20, o21, o22;
};

__device__ __forceinline__ void stem_entry(const StemRows& w, int i, int& g, bool& start) {
    if (i < w.l0) {
        g = w.a0 + i;
        start = g == w.o00 || g == w.o01 || g == w.o02;
    } else if (i < w.l0 + w.l1) {
        g = w.a1 + (i - w.l0);
        start = g == w.o10 || g == w.o11 || g == w.o12;
    } else {
        g = w.a2 + (i - w.l0 - w.l1);
        start = g == w.o20 || g == w.o21 || g == w.o22;
    }
}

// Streaming path (the heavy pixels: more than six entries under one conv pixel, any label map; stem_apply_heavy_kernel): the entries are
// streamed through registers 32 at a time and a wave's EIGHT masks are carried at once -- per mask the running sum of the open conv pixel
// and the running pool maximum -- so a pooled pixel with any number of entries runs at register speed (an earlier version re-read the table
// per mask for the long ones: three dependent loads per entry, and the few waves that took that path set the launch's duration).  Lane i of
// a chunk holds the keep bits of its entry i and lays the eight masks' factors out in LDS; a conv pixel's end is a branch on the scalar
// unit (see the asm there).
constexpr int SA_CH = 32, SA_MG = 8;

__device__ __forceinline__ void stem_load_chunk(const StemApplyParams& p, const StemRows& w, int cb, int n_e, int lane, int mb,
                                                float (&R)[SA_CH], unsigned& starts, unsigned& kbv) {
    const int cnt = min(SA_CH, n_e - cb);
    starts = 0;
#pragma unroll
    for (int i = 0; i < SA_CH; ++i) {
        int g = 0;
        bool st = false;
        stem_entry(w, cb + i, g, st);
        const bool ok = i < cnt;
        R[i] = ok ? p.vec[(size_t)g * ST_C + lane] : 0.f;
        if (ok && st && cb + i > 0) starts |= 1u << i;
    }
    kbv = 0;                                    // lane i: the keep bits (32 masks) of entry cb + i
    if (lane < cnt) {
        int g = 0;
        bool st = false;
        stem_entry(w, cb + lane, g, st);
        kbv = p.bits[(size_t)p.lab[g] * p.nmb + mb];
    }
}

// Static paths: a pooled pixel whose conv pixels have at most KPP entries each (BORDER: first pooled row / column, where conv pixels
// outside the map take no part in the pool)
// (KPP = 1, 2, 3, 4, 6: the window of a conv pixel inside one superpixel, across one boundary, at a junction, at a corner of a grid, in a
// fragmented part of a felzenszwalb map).  Slot (q, k) = entry k of
// conv pixel q, or a zero: fully static code, one fma per slot, one fma + max3 per conv pixel.  The 0.0 / 1.0 keep factors of the 32 masks
// are laid out once per wave in LDS ([mask][slot]) and come back four slots per broadcast ds_read_b128: extracting them per mask on the
// scalar unit (s_and, s_cmp, s_cselect per slot) made the launch scalar-bound at 3 x the time of its vector instructions.
constexpr int SA_KF_PITCH = 56;                 // floats per mask row: up to 54 slots (KPP = 6), 16-byte aligned rows

template <int KPP, bool BORDER>
__device__ __forceinline__ void stem_apply_fast(const StemApplyParams& p, unsigned (*s_out)[4][4][ST_C], float (*s_kf)[SA_KF_PITCH], int wave,
                                                int lane, int mcount, size_t out0_row, const int (&o)[9], const int (&len)[9],
                                                const bool (&qok)[9], float sc, float sh, int mb) {
    constexpr int NS = 9 * KPP, NS4 = (NS + 3) / 4 * 4;
    float R[NS4];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int k = 0; k < KPP; ++k) {
            const bool ok = k < len[q];
            const int g = o[q] + k;
            R[q * KPP + k] = ok ? p.vec[(size_t)g * ST_C + lane] : 0.f;
            const unsigned kb = ok ? p.bits[(size_t)p.lab[g] * p.nmb + mb] : 0u;       // wave-uniform
            if (lane < 32) s_kf[lane][q * KPP + k] = (kb >> lane & 1u) ? 1.0f : 0.0f;  // lane = mask
        }
#pragma unroll
    for (int i = NS; i < NS4; ++i) {
        R[i] = 0.f;
        if (lane < 32) s_kf[lane][i] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int m = 0; m < mcount; ++m) {
        f4 kf[NS4 / 4];
#pragma unroll
        for (int i = 0; i < NS4 / 4; ++i) kf[i] = *(const f4*)&s_kf[m][4 * i];      // the same address in every lane: a broadcast
        float best = -INFINITY;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < KPP; ++k)       // 1.0 * R + v and 0.0 * R + v are exact for finite R
                v = fmaf(kf[(q * KPP + k) / 4][(q * KPP + k) % 4], R[q * KPP + k], v);
            const float y = fmaxf(fmaf(v, sc, sh), 0.f);
            // BORDER (first pooled row / column): a conv pixel outside the map takes no part in the pool
            best = fmaxf(best, BORDER ? (qok[q] ? y : -INFINITY) : y);
        }
        stem_store4(p, s_out, m, mcount, wave, lane, best, out0_row);
    }
}

// The geometry of one pooled pixel: its three CSR row ranges, the per-conv-pixel view for the static paths, the largest entry count.
struct StemPixel {
    StemRows w;
    int n_e, n_empty, kmax;
    int o9[9], len9[9];
    bool qok[9];
    bool interior;
};

__device__ __forceinline__ void stem_pixel(const StemApplyParams& p, int py, int px, StemPixel& g) {
    // valid conv columns under the pooled pixel (3x3 window, stride 2, pad 1): 2px-1 .. 2px+1 clipped on the left (2 * 55 + 1 = 111 fits)
    const int cx0 = max(2 * px - 1, 0);
    const int ncol = 2 * px + 1 - cx0 + 1;                      // 2 or 3
    StemRows& w = g.w;
    g.n_empty = 0;
    {
        // row r = conv row 2py - 1 + r; rows outside the map (r = 0 for py = 0) are empty
        const int cy0 = 2 * py - 1, cy1 = 2 * py, cy2 = 2 * py + 1;
        const bool ok0 = cy0 >= 0;
        const int q0 = (ok0 ? cy0 : 0) * ST_CONV + cx0, q1 = cy1 * ST_CONV + cx0, q2 = cy2 * ST_CONV + cx0;
        const int e00 = p.off[q0], e01 = p.off[q0 + 1], e02 = p.off[q0 + 2], e03 = p.off[q0 + ncol];
        const int e10 = p.off[q1], e11 = p.off[q1 + 1], e12 = p.off[q1 + 2], e13 = p.off[q1 + ncol];
        const int e20 = p.off[q2], e21 = p.off[q2 + 1], e22 = p.off[q2 + 2], e23 = p.off[q2 + ncol];
        // (with two columns e_2 is the end offset, which no entry of the row equals: never taken for a start)
        w.a0 = e00; w.l0 = ok0 ? e03 - e00 : 0; w.o00 = e00; w.o01 = e01; w.o02 = e02;
        w.a1 = e10; w.l1 = e13 - e10; w.o10 = e10; w.o11 = e11; w.o12 = e12;
        w.a2 = e20; w.l2 = e23 - e20; w.o20 = e20; w.o21 = e21; w.o22 = e22;
        if (ok0) g.n_empty += (e01 == e00) + (e02 == e01) + (ncol == 3 && e03 == e02);
        g.n_empty += (e11 == e10) + (e12 == e11) + (ncol == 3 && e13 == e12);
        g.n_empty += (e21 == e20) + (e22 == e21) + (ncol == 3 && e23 == e22);
    }
    g.n_e = w.l0 + w.l1 + w.l2;
    // the per-pixel view for the static paths: conv pixel q = 3 * r + c at row 2py-1+r, column 2px-1+c; pixels outside the map (r = 0 for
    // py = 0, c = 0 for px = 0) are absent
    const bool col0_ok = px > 0, row0_ok = py > 0;
    const int e0[4] = {w.o00, w.o01, w.o02, w.a0 + w.l0}, e1[4] = {w.o10, w.o11, w.o12, w.a1 + w.l1}, e2[4] = {w.o20, w.o21, w.o22, w.a2 + w.l2};
    g.kmax = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bool cok = col0_ok || c > 0;
        const int j = cok ? (col0_ok ? c : c - 1) : 0;          // with the left column absent the row's ranges start at column 1
        g.qok[c] = cok && row0_ok;
        g.qok[3 + c] = cok;
        g.qok[6 + c] = cok;
        g.o9[c] = e0[j];
        g.o9[3 + c] = e1[j];
        g.o9[6 + c] = e2[j];
        g.len9[c] = g.qok[c] ? e0[j + 1] - e0[j] : 0;
        g.len9[3 + c] = cok ? e1[j + 1] - e1[j] : 0;
        g.len9[6 + c] = cok ? e2[j + 1] - e2[j] : 0;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) g.kmax = max(g.kmax, g.len9[q]);
    g.interior = col0_ok && row0_ok;
}

// a pooled pixel the static paths do not take: more than six (border pixels: four) entries under one conv pixel
__device__ __forceinline__ bool stem_heavy(const StemPixel& g) { return !(g.kmax <= 6 && (g.kmax <= 4 || g.interior)); }

// The list of heavy pooled pixels of the table in place (built once per image, behind the scan: one workgroup, LDS counter).
__global__ __launch_bounds__(1024) void stemtab_heavy_kernel(const int* __restrict__ off, int* __restrict__ heavy_list, int* __restrict__ heavy_count) {
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    StemApplyParams p;
    p.off = off;
    for (int P = threadIdx.x; P < ST_POOLED * ST_POOLED; P += 1024) {
        StemPixel g;
        stem_pixel(p, P / ST_POOLED, P % ST_POOLED, g);
        if (stem_heavy(g)) heavy_list[atomicAdd(&s_n, 1)] = P;
    }
    __syncthreads();
    if (threadIdx.x == 0) *heavy_count = s_n;
}

// First launch: the HEAVY pooled pixels only (a fixed grid walks the table's list of them; none = every workgroup returns at once).  The four waves
// share the pixel's 32 masks, eight each, on the streaming path, and write their 128 bytes per plane and mask themselves -- few pixels, so
// the narrow stores do not matter.  The second launch copies these values into its four-pixel stores.  (One launch, with the heavy pixel's
// wave streaming all 32 masks while the three static waves of its workgroup waited at every store barrier, took 2.6 ms on the felzenszwalb
// fixture against 1.1 on a grid: 11 % of its pixels are heavy, in 38 % of the workgroups.)
__global__ __launch_bounds__(256) void stem_apply_heavy_kernel(const StemApplyParams p) {
    __shared__ __attribute__((aligned(16))) float s_ks[4][SA_MG * SA_CH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mb = blockIdx.y;
    const int mw0 = wave * SA_MG;                               // this wave's first mask within the block
    const int mcount = min(SA_MG, p.M - (mb * 32 + mw0));
    if (mcount <= 0) return;
    const float sc = p.s[lane], sh = p.t[lane];
    const float y_empty = fmaxf(sh, 0.f);
    const int n_heavy = *p.heavy_count;
    float* ks = s_ks[wave];
  for (int hi_ = blockIdx.x; hi_ < n_heavy; hi_ += gridDim.x) {
    const int P = p.heavy_list[hi_];
    const int py = P / ST_POOLED, px = P - py * ST_POOLED;
    StemPixel g;
    stem_pixel(p, py, px, g);
    float v[SA_MG], best[SA_MG];
#pragma unroll
    for (int k = 0; k < SA_MG; ++k) {
        v[k] = 0.f;
        best[k] = g.n_empty > 0 ? y_empty : -INFINITY;
    }
    for (int cb = 0; cb < g.n_e; cb += SA_CH) {
        const int cnt = min(SA_CH, g.n_e - cb);
        float R[SA_CH];
        unsigned starts = 0, kbv = 0;
        stem_load_chunk(p, g.w, cb, g.n_e, lane, mb, R, starts, kbv);
        __builtin_amdgcn_wave_barrier();        // the previous chunk's reads are done
        if (lane < SA_CH) {
#pragma unroll
            for (int k = 0; k < SA_MG; ++k) ks[k * SA_CH + lane] = (kbv >> (mw0 + k) & 1u) ? 1.0f : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < SA_MG; ++k) {
#pragma unroll
            for (int i4 = 0; i4 < SA_CH / 4; ++i4) {
                const f4 kf = *(const f4*)&ks[k * SA_CH + 4 * i4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = 4 * i4 + ii;
                    if (i < cnt) {              // wave-uniform
                        if (starts >> i & 1) {
                            // (the empty asm keeps this a BRANCH on the scalar unit: if-converted, every entry would pay the fma + max3 of a
                            // conv pixel's end and two selects -- six vector instructions per entry instead of two)
                            asm volatile("" ::: "memory");
                            best[k] = fmaxf(best[k], fmaxf(fmaf(v[k], sc, sh), 0.f));
                            v[k] = 0.f;
                        }
                        v[k] = fmaf(kf[ii], R[i], v[k]);       // 1.0 * R + v and 0.0 * R + v are exact for finite R
                    }
                }
            }
        }
    }
    const size_t out0 = (((size_t)(p.slot0 + mb * 32 + mw0) * ST_POOLED + py) * ST_POOLED + px) * ST_C + lane;
#pragma unroll
    for (int k = 0; k < SA_MG; ++k) {
        if (k < mcount) {
            const float b = g.n_e > 0 ? fmaxf(best[k], fmaxf(fmaf(v[k], sc, sh), 0.f)) : best[k];
            half_t hi, lo;
            split_f32(b, hi, lo);
            const size_t o = out0 + (size_t)k * ST_POOLED * ST_POOLED * ST_C;
            p.out_hi[o] = hi;
            p.out_lo[o] = lo;
        }
    }
    __builtin_amdgcn_wave_barrier();            // the next pixel's first chunk overwrites this wave's factors
  }
}

// Second launch: every pooled pixel; a heavy one's wave reads back what the first launch wrote.
__global__ __launch_bounds__(256) void stem_apply_kernel(const StemApplyParams p) {
    __shared__ __attribute__((aligned(16))) unsigned s_out[2][4][4][ST_C];
    __shared__ __attribute__((aligned(16))) float s_kf[4][32][SA_KF_PITCH];         // per wave: [mask][slot] keep factors of the static paths
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = blockIdx.x * 4 + wave;                        // pooled pixel (3136 = 784 * 4): the workgroup's four are neighbours in a row
    const int mb = blockIdx.y;
    const int m0 = mb * 32;
    const int mcount = min(32, p.M - m0);
    const int py = P / ST_POOLED, px = P - py * ST_POOLED;
    StemPixel g;
    stem_pixel(p, py, px, g);
    const float sc = p.s[lane], sh = p.t[lane];
    const size_t out0_row = (((size_t)(p.slot0 + m0) * ST_POOLED + py) * ST_POOLED + (px - wave)) * ST_C;
    if (stem_heavy(g)) {
        const size_t o0 = out0_row + (size_t)wave * ST_C + lane;
        for (int m = 0; m < mcount; ++m) {
            const size_t o = o0 + (size_t)m * ST_POOLED * ST_POOLED * ST_C;
            const unsigned packed = (unsigned)__builtin_bit_cast(unsigned short, p.out_hi[o]) | ((unsigned)__builtin_bit_cast(unsigned short, p.out_lo[o]) << 16);
            stem_store4_packed(p, s_out, m, mcount, wave, lane, packed, out0_row);
        }
    } else if (g.interior) {
        if (g.kmax <= 1) stem_apply_fast<1, false>(p, s_out, s_kf[wave], wave, lane, mcount, out0_row, g.o9, g.len9, g.qok, sc, sh, mb);
        else if (g.kmax <= 2) stem_apply_fast<2, false>(p, s_out, s_kf[wave], wave, lane, mcount, out0_row, g.o9, g.len9, g.qok, sc, sh, mb);
        else if (g.kmax <= 3) stem_apply_fast<3, false>(p, s_out, s_kf[wave], wave, lane, mcount, out0_row, g.o9, g.len9, g.qok, sc, sh, mb);
        else if (g.kmax <= 4) stem_apply_fast<4, false>(p, s_out, s_kf[wave], wave, lane, mcount, out0_row, g.o9, g.len9, g.qok, sc, sh, mb);
        else stem_apply_fast<6, false>(p, s_out, s_kf[wave], wave, lane, mcount, out0_row, g.o9, g.len9, g.qok, sc, sh, mb);
    } else {
        if (g.kmax <= 2) stem_apply_fast<2, true>(p, s_out, s_kf[wave], wave, lane, mcount, out0_row, g.o9, g.len9, g.qok, sc, sh, mb);
        else stem_apply_fast<4, true>(p, s_out, s_kf[wave], wave, lane, mcount, out0_row, g.o9, g.len9, g.qok, sc, sh, mb);
    }
}

}  // namespace mpx
