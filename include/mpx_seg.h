/* mpx_seg.h -- C-ABI of libmpxseg.so: the CPU segmentation front-end.
 *
 * The reference segments every image on the CPU with a third-party call,
 *     segments = felzenszwalb(img_as_float(img_show), scale=100, sigma=0.5, min_size=50)
 *     (generate_gp_training_data_imagenet.py:183; gp_superpixel_data_imagenet.py:218;
 *      bayesian_active_learning_imagenet.py:150,263,463; generate_superpixels.py:10,15)
 * from scikit-image (requirements.txt, un-pinned).  The label map is an INPUT of the GPU engine (mpx.h); this
 * library provides it without the Python dependency and for a batch of images on a pool of host threads, so
 * segmentation of the next images overlaps the masked forward passes of the current ones.
 *
 * It is host code on purpose: the reference's step is CPU code too, and Felzenszwalb-Huttenlocher is a serial
 * greedy pass over sorted edges.  It is not a fallback of anything in mpx.h.
 *
 * Pinned bit-exactly against scikit-image 0.18.3 on the committed vectors (tests/golden/felzenszwalb_skimage0183.npz,
 * written by tests/golden/make_felzenszwalb_golden.py).  Edges are built in the order right, down, down-right,
 * up-right (row-major within each) and sorted by weight; edges of EXACTLY equal weight come out in the order NumPy's
 * generic (unstable) quicksort leaves them, which mpx_seg.cpp's argsort_introsort reproduces step by step
 * (partitions of >= 17 elements, median of 3, insertion sort below) -- not in index order.
 */
#ifndef MPX_SEG_H
#define MPX_SEG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPXSEG_E_ARG (-1)
#define MPXSEG_E_NOMEM (-2)

/* One image.  img: u8[h][w][channels] (channels 1..4; the reference passes HWC RGB after its min-max rescale,
 * generate_gp_training_data_imagenet.py:171-178).  labels: i32[h][w], contiguous 0..S-1 in raster order of each
 * segment's first pixel (what np.unique(..., return_inverse=True) yields upstream).
 * Returns S (>= 1) or MPXSEG_E_*. */
int mpxseg_felzenszwalb(const uint8_t* img, int h, int w, int channels, double scale, double sigma, int min_size,
                        int32_t* labels);

/* n images of one shape on `threads` host threads (<= 0: hardware concurrency).  counts[i] = S of image i.
 * Returns 0 or the first MPXSEG_E_*. */
int mpxseg_felzenszwalb_batch(const uint8_t* imgs, int n, int h, int w, int channels, double scale, double sigma,
                              int min_size, int32_t* labels, int32_t* counts, int threads);

/* Test hook: the edge ordering used above, np.argsort(v) of NumPy's generic quicksort (unstable; ties as that
 * procedure leaves them).  order: i32[n].  Returns 0 or MPXSEG_E_*. */
int mpxseg_argsort_f64(const double* v, long n, int32_t* order);

/* img_show (generate_gp_training_data_imagenet.py:171-178): x f32[c][h][w] -> u8[h][w][c],
 * (x - min) / max(x - min) * 255 in fp32, truncated.  Returns 0 or MPXSEG_E_*. */
int mpxseg_minmax_u8(const float* chw, int c, int h, int w, uint8_t* hwc);

#ifdef __cplusplus
}
#endif
#endif
