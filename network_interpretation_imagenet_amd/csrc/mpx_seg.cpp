// mpx_seg.cpp -- libmpxseg.so: Felzenszwalb-Huttenlocher graph segmentation as scikit-image 0.18.3 evaluates
// it for the reference's call felzenszwalb(img_as_float(u8), scale, sigma, min_size)
// (generate_gp_training_data_imagenet.py:183), plus the img_show rescale (:171-178).  Host code, g++ only.
//
// The steps and their arithmetic (all float64, no contraction: build with -ffp-contract=off):
//   1. x = u8 * (1/255)                                   (img_as_float multiplies by the reciprocal)
//   2. Gaussian blur, rows then columns, radius int(4*sigma + 0.5), taps exp(-0.5/sigma^2 * k^2) / sum,
//      'reflect' borders (d c b a | a b c d | d c b a), summed centre first, then tap pairs from the outermost in
//   3. 8-connected edges in the order right, down, down-right, up-right (row-major inside each group), weight
//      sqrt(sum_c d_c^2)
//   4. edges by ascending weight (NumPy's generic introsort, see argsort_introsort); merge a, b when  w < min(f32(int_a + k/|a|), f32(int_b + k/|b|)),
//      k = scale / 255; the merged component keeps int = w, its representative is the smaller pixel index
//   5. second pass in the same order: merge when either side has fewer than min_size pixels
//   6. labels = rank of each pixel's representative (raster order of the component's first pixel)
#include "mpx_seg.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

namespace {

inline int reflect(int i, int n) {          // scipy.ndimage 'reflect' = half-sample symmetric, any distance
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// One 1-D pass along `axis` (0 = down the rows, 1 = along a row) of an [h][w][c] float64 image.
void blur_axis(const std::vector<double>& in, std::vector<double>& out, int h, int w, int c, int axis,
               const std::vector<double>& taps, int radius) {
    const int n = axis == 0 ? h : w;
    const int lines = axis == 0 ? w : h;
    const size_t step = axis == 0 ? (size_t)w * c : (size_t)c;         // between neighbours along the axis
    const size_t line_step = axis == 0 ? (size_t)c : (size_t)w * c;    // between lines
    std::vector<double> buf(n + 2 * radius);
    for (int l = 0; l < lines; ++l)
        for (int ch = 0; ch < c; ++ch) {
            const size_t base = l * line_step + ch;
            for (int i = -radius; i < n + radius; ++i) buf[i + radius] = in[base + reflect(i, n) * step];
            for (int i = 0; i < n; ++i) {
                const double* p = &buf[i + radius];
                double t = p[0] * taps[radius];
                for (int j = -radius; j < 0; ++j) t += (p[j] + p[-j]) * taps[j + radius];
                out[base + i * step] = t;
            }
        }
}

// np.argsort(costs) as NumPy's generic (non-SIMD) kind='quicksort' evaluates it: median-of-3 quicksort on the
// index array, insertion sort below 17 elements, heapsort when the depth budget 2*floor(log2 n) runs out.  The sort
// is not stable, and the order it leaves EQUAL weights in decides which of two equally cheap merges happens first,
// so the same procedure is followed here step by step (it is the one the reference's NumPy generation ran;
// NumPy >= 1.25 on AVX-512 hosts dispatches to a vectorised sort whose tie order differs again).
// The procedure only ever compares keys and swaps / moves elements, so sorting (key, index) records with the same
// steps yields the same permutation as NumPy's indirect sort of the index array -- without a cache miss per compare.
struct KV {
    double v;
    int32_t i;
};

inline void heapsort_kv(KV* start, long n) {
    KV* a = start - 1;        // 1-based
    long i, j, l;
    KV tmp;
    for (l = n >> 1; l > 0; --l) {
        tmp = a[l];
        for (i = l, j = l << 1; j <= n;) {
            if (j < n && a[j].v < a[j + 1].v) j += 1;
            if (tmp.v < a[j].v) { a[i] = a[j]; i = j; j += j; }
            else break;
        }
        a[i] = tmp;
    }
    for (; n > 1;) {
        tmp = a[n];
        a[n] = a[1];
        n -= 1;
        for (i = 1, j = 2; j <= n;) {
            if (j < n && a[j].v < a[j + 1].v) j++;
            if (tmp.v < a[j].v) { a[i] = a[j]; i = j; j += j; }
            else break;
        }
        a[i] = tmp;
    }
}

void sort_introsort(KV* kv, long num) {
    if (num < 2) return;
    constexpr long SMALL = 15;      // partitions of 17 or more elements are split (as NumPy 1.26 does)
    KV *pl = kv, *pr = kv + num - 1;
    KV* stack[128];
    KV** sptr = stack;
    int depth[64];
    int* psdepth = depth;
    int msb = 0;
    for (unsigned long t = (unsigned long)num; t >>= 1;) ++msb;
    int cdepth = msb * 2;
    for (;;) {
        if (cdepth < 0) {
            heapsort_kv(pl, pr - pl + 1);
        } else {
            while (pr - pl > SMALL) {
                KV* pm = pl + ((pr - pl) >> 1);
                if (pm->v < pl->v) std::swap(*pm, *pl);
                if (pr->v < pm->v) std::swap(*pr, *pm);
                if (pm->v < pl->v) std::swap(*pm, *pl);
                const double vp = pm->v;
                KV *pi = pl, *pj = pr - 1;
                std::swap(*pm, *pj);
                for (;;) {
                    do { ++pi; } while (pi->v < vp);
                    do { --pj; } while (vp < pj->v);
                    if (pi >= pj) break;
                    std::swap(*pi, *pj);
                }
                KV* pk = pr - 1;
                std::swap(*pi, *pk);
                if (pi - pl < pr - pi) {       // the larger partition waits on the stack
                    *sptr++ = pi + 1;
                    *sptr++ = pr;
                    pr = pi - 1;
                } else {
                    *sptr++ = pl;
                    *sptr++ = pi - 1;
                    pl = pi + 1;
                }
                *psdepth++ = --cdepth;
            }
            for (KV* pi = pl + 1; pi <= pr; ++pi) {       // insertion sort
                const KV vi = *pi;
                KV *pj = pi, *pk = pi - 1;
                while (pj > pl && vi.v < pk->v) *pj-- = *pk--;
                *pj = vi;
            }
        }
        if (sptr == stack) break;
        pr = *(--sptr);
        pl = *(--sptr);
        cdepth = *(--psdepth);
    }
}

void argsort_introsort(const double* v, int32_t* tosort, long num) {
    std::vector<KV> kv((size_t)num);
    for (long i = 0; i < num; ++i) kv[i] = KV{v[i], (int32_t)i};
    sort_introsort(kv.data(), num);
    for (long i = 0; i < num; ++i) tosort[i] = kv[i].i;
}

struct Forest {
    std::vector<int32_t> parent;
    explicit Forest(int n) : parent(n) { std::iota(parent.begin(), parent.end(), 0); }
    int find(int x) {
        int r = x;
        while (parent[r] != r) r = parent[r];
        while (parent[x] != r) {
            const int nx = parent[x];
            parent[x] = r;
            x = nx;
        }
        return r;
    }
    int join(int a, int b) {        // roots in, smaller index stays the representative
        const int r = a < b ? a : b;
        parent[a] = r;
        parent[b] = r;
        return r;
    }
};

int segment(const uint8_t* img, int h, int w, int c, double scale, double sigma, int min_size, int32_t* labels) {
    if (!img || !labels || h <= 0 || w <= 0 || c < 1 || c > 4 || !(sigma >= 0.0) || !(scale >= 0.0) ||
        (long long)h * w > (1 << 28))
        return MPXSEG_E_ARG;
    try {
        const int np_ = h * w;
        std::vector<double> a((size_t)np_ * c), b((size_t)np_ * c);
        const double inv = 1.0 / 255;
        for (size_t i = 0; i < a.size(); ++i) a[i] = img[i] * inv;
        if (sigma > 1e-15) {        // gaussian_filter skips an axis whose sigma is <= 1e-15
            const int radius = (int)(4.0 * sigma + 0.5);
            std::vector<double> taps(2 * radius + 1);
            const double s2 = sigma * sigma;
            double sum = 0.0;
            for (int k = -radius; k <= radius; ++k) {
                taps[k + radius] = std::exp(-0.5 / s2 * (double)(k * k));
                sum += taps[k + radius];
            }
            for (double& t : taps) t = t / sum;
            blur_axis(a, b, h, w, c, 0, taps, radius);
            blur_axis(b, a, h, w, c, 1, taps, radius);
        }
        // edges: right, down, down-right, up-right
        const size_t n_r = (size_t)h * (w - 1), n_d = (size_t)(h - 1) * w, n_g = (size_t)(h - 1) * (w - 1);
        const size_t ne = n_r + n_d + 2 * n_g;
        std::vector<double> cost(ne);
        std::vector<int32_t> e0(ne), e1(ne);
        auto dist = [&](int p, int q) {
            double s = 0.0;
            for (int ch = 0; ch < c; ++ch) {
                const double d = a[(size_t)p * c + ch] - a[(size_t)q * c + ch];
                s += d * d;
            }
            return std::sqrt(s);
        };
        size_t k = 0;
        for (int y = 0; y < h; ++y)
            for (int x = 1; x < w; ++x, ++k) { e0[k] = y * w + x; e1[k] = y * w + x - 1; cost[k] = dist(e0[k], e1[k]); }
        for (int y = 1; y < h; ++y)
            for (int x = 0; x < w; ++x, ++k) { e0[k] = y * w + x; e1[k] = (y - 1) * w + x; cost[k] = dist(e0[k], e1[k]); }
        for (int y = 1; y < h; ++y)
            for (int x = 1; x < w; ++x, ++k) { e0[k] = y * w + x; e1[k] = (y - 1) * w + x - 1; cost[k] = dist(e0[k], e1[k]); }
        for (int y = 1; y < h; ++y)
            for (int x = 1; x < w; ++x, ++k) { e0[k] = (y - 1) * w + x; e1[k] = y * w + x - 1; cost[k] = dist(e0[k], e1[k]); }
        std::vector<int32_t> order(ne);
        argsort_introsort(cost.data(), order.data(), (long)ne);

        Forest f(np_);
        std::vector<int32_t> size(np_, 1);
        std::vector<double> cint(np_, 0.0);
        const double kk = scale / 255.0;
        for (size_t i = 0; i < ne; ++i) {
            const int32_t e = order[i];
            const int s0 = f.find(e0[e]), s1 = f.find(e1[e]);
            if (s0 == s1) continue;
            // scikit-image keeps the two thresholds in SINGLE precision (found by bisecting its merge decision on
            // two-pixel images: the flip sits on a float32 rounding midpoint); weight and int stay double
            const float i0 = (float)(cint[s0] + kk / size[s0]), i1 = (float)(cint[s1] + kk / size[s1]);
            if (cost[e] < (double)(i0 < i1 ? i0 : i1)) {
                const int r = f.join(s0, s1);
                size[r] = size[s0] + size[s1];
                cint[r] = cost[e];
            }
        }
        for (size_t i = 0; i < ne; ++i) {
            const int32_t e = order[i];
            const int s0 = f.find(e0[e]), s1 = f.find(e1[e]);
            if (s0 == s1) continue;
            if (size[s0] < min_size || size[s1] < min_size) {
                const int r = f.join(s0, s1);
                size[r] = size[s0] + size[s1];
            }
        }
        // representative = smallest pixel index of the component, so raster order of roots = label order
        std::vector<int32_t> rank(np_, -1);
        int S = 0;
        for (int p = 0; p < np_; ++p)
            if (f.find(p) == p) rank[p] = S++;
        for (int p = 0; p < np_; ++p) labels[p] = rank[f.find(p)];
        return S;
    } catch (const std::bad_alloc&) {
        return MPXSEG_E_NOMEM;
    }
}

}  // namespace

extern "C" {

int mpxseg_felzenszwalb(const uint8_t* img, int h, int w, int channels, double scale, double sigma, int min_size,
                        int32_t* labels) {
    return segment(img, h, w, channels, scale, sigma, min_size, labels);
}

int mpxseg_felzenszwalb_batch(const uint8_t* imgs, int n, int h, int w, int channels, double scale, double sigma,
                              int min_size, int32_t* labels, int32_t* counts, int threads) {
    if (n < 0 || !counts || (n > 0 && (!imgs || !labels)) || h <= 0 || w <= 0 || channels < 1 || channels > 4)
        return MPXSEG_E_ARG;
    if (n == 0) return 0;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > n) nt = n;
    const size_t img_bytes = (size_t)h * w * channels, lab = (size_t)h * w;
    std::atomic<int> next(0), err(0);
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            const int s = segment(imgs + i * img_bytes, h, w, channels, scale, sigma, min_size, labels + i * lab);
            counts[i] = s;
            if (s < 0) {
                int zero = 0;
                err.compare_exchange_strong(zero, s);
            }
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve(nt);
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    } catch (...) {
        // fewer threads than asked for: the ones that started and this one still drain the queue
    }
    work();
    for (auto& t : pool) t.join();
    return err.load();
}

int mpxseg_argsort_f64(const double* v, long n, int32_t* order) {
    if (n < 0 || n > 0x7fffffffL || (n > 0 && (!v || !order))) return MPXSEG_E_ARG;
    try {
        argsort_introsort(v, order, n);
    } catch (const std::bad_alloc&) {
        return MPXSEG_E_NOMEM;
    }
    return 0;
}

int mpxseg_minmax_u8(const float* chw, int c, int h, int w, uint8_t* hwc) {
    if (!chw || !hwc || c < 1 || h <= 0 || w <= 0) return MPXSEG_E_ARG;
    const size_t plane = (size_t)h * w, n = plane * c;
    float lo = chw[0];
    for (size_t i = 1; i < n; ++i) lo = chw[i] < lo ? chw[i] : lo;
    float hi = chw[0] - lo;
    for (size_t i = 1; i < n; ++i) {
        const float v = chw[i] - lo;
        hi = v > hi ? v : hi;
    }
    if (!(hi > 0.0f)) {     // constant (or NaN) picture: upstream divides 0 by 0; there is no picture to segment
        std::memset(hwc, 0, n);
        return 0;
    }
    for (int ch = 0; ch < c; ++ch)
        for (size_t p = 0; p < plane; ++p) {
            float v = chw[ch * plane + p] - lo;     // img -= min; img /= max; img *= 255; astype(uint8), all fp32
            v = v / hi;
            v = v * 255.0f;
            hwc[p * c + ch] = (uint8_t)(int)v;
        }
    return 0;
}

}  // extern "C"
