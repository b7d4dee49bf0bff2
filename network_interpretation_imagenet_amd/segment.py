"""CPU segmentation front-end: ctypes binding of libmpxseg.so (include/mpx_seg.h).

`felzenszwalb(img_u8_hwc)` is the reference's third-party call
`felzenszwalb(img_as_float(img_show), scale=100, sigma=0.5, min_size=50)`
(generate_gp_training_data_imagenet.py:183) without the scikit-image dependency, pinned bit-exactly
against scikit-image 0.18.3 on tests/golden/felzenszwalb_skimage0183.npz.  `SegmenterPool` runs it
for the next images on host threads while the GPU scores the current one (the C call releases the
GIL).  This module needs neither torch nor a GPU."""
import ctypes as C
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpxseg.so")

_u8p, _i32p, _f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
# name -> (restype, argtypes); must list every symbol include/mpx_seg.h declares
SIGNATURES = {
    "mpxseg_felzenszwalb": (C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _i32p]),
    "mpxseg_felzenszwalb_batch": (C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                            _i32p, _i32p, C.c_int]),
    "mpxseg_argsort_f64": (C.c_int, [C.POINTER(C.c_double), C.c_long, _i32p]),
    "mpxseg_minmax_u8": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, _u8p]),
}

_lib = None
_lock = threading.Lock()


class SegError(RuntimeError):
    pass


def load():
    """Load libmpxseg.so (built by __graft_entry__.build() with g++)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SegError("%s not built; run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def _u8_image(img):
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8:
        raise ValueError("image must be uint8 (the reference segments its min-max rescaled u8 picture), got %s" % a.dtype)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or not 1 <= a.shape[2] <= 4:
        raise ValueError("image must be u8[h,w] or u8[h,w,c<=4], got shape %s" % (a.shape,))
    return a


def felzenszwalb(img_u8_hwc, scale=100, sigma=0.5, min_size=50):
    """u8[h,w,c] -> int64[h,w] labels 0..S-1 (dtype and label order as scikit-image returns them)."""
    a = _u8_image(img_u8_hwc)
    h, w, c = a.shape
    out = np.empty((h, w), np.int32)
    rc = load().mpxseg_felzenszwalb(a.ctypes.data_as(_u8p), h, w, c, float(scale), float(sigma), int(min_size),
                                    out.ctypes.data_as(_i32p))
    if rc < 0:
        raise SegError("mpxseg_felzenszwalb failed (rc=%d)" % rc)
    return out.astype(np.int64)


def felzenszwalb_batch(imgs_u8, scale=100, sigma=0.5, min_size=50, threads=0):
    """u8[n,h,w,c] -> (labels int32[n,h,w], counts int32[n]) on `threads` host threads (0 = all)."""
    a = np.ascontiguousarray(imgs_u8)
    if a.dtype != np.uint8 or a.ndim != 4 or not 1 <= a.shape[3] <= 4:
        raise ValueError("images must be u8[n,h,w,c<=4], got %s %s" % (a.dtype, a.shape))
    n, h, w, c = a.shape
    out = np.empty((n, h, w), np.int32)
    counts = np.zeros(n, np.int32)
    rc = load().mpxseg_felzenszwalb_batch(a.ctypes.data_as(_u8p), n, h, w, c, float(scale), float(sigma), int(min_size),
                                          out.ctypes.data_as(_i32p), counts.ctypes.data_as(_i32p), int(threads))
    if rc < 0:
        raise SegError("mpxseg_felzenszwalb_batch failed (rc=%d)" % rc)
    return out, counts


def argsort_f64(v):
    """np.argsort(v) as NumPy's generic (non-SIMD) quicksort orders it, ties included (test hook)."""
    a = np.ascontiguousarray(v, dtype=np.float64).ravel()
    out = np.empty(a.size, np.int32)
    rc = load().mpxseg_argsort_f64(a.ctypes.data_as(C.POINTER(C.c_double)), a.size, out.ctypes.data_as(_i32p))
    if rc < 0:
        raise SegError("mpxseg_argsort_f64 failed (rc=%d)" % rc)
    return out


def minmax_u8(x_chw):
    """img_show (generate_gp_training_data_imagenet.py:171-178): f32[c,h,w] -> u8[h,w,c], truncating."""
    x = np.ascontiguousarray(x_chw, dtype=np.float32)
    if x.ndim != 3:
        raise ValueError("expected f32[c,h,w], got shape %s" % (x.shape,))
    c, h, w = x.shape
    out = np.empty((h, w, c), np.uint8)
    rc = load().mpxseg_minmax_u8(x.ctypes.data_as(_f32p), c, h, w, out.ctypes.data_as(_u8p))
    if rc < 0:
        raise SegError("mpxseg_minmax_u8 failed (rc=%d)" % rc)
    return out


class SegmenterPool:
    """Segment upcoming images on host threads while the GPU works on the current one.

        pool = SegmenterPool(workers=8)
        futs = [pool.submit(x_chw) for x_chw in normalised_images]     # returns at once
        seg = futs[i].result()                                          # int64[224,224]

    `submit` takes the normalised f32[3,h,w] tensor/array the loader yields and applies the
    reference's img_show rescale first; `submit_u8` takes a ready u8[h,w,3] picture."""

    def __init__(self, workers=4, scale=100, sigma=0.5, min_size=50):
        load()
        self._ex = ThreadPoolExecutor(max_workers=int(workers), thread_name_prefix="mpxseg")
        self._params = (scale, sigma, min_size)

    def submit(self, x_chw):
        x = np.array(x_chw, dtype=np.float32, copy=True)
        return self._ex.submit(lambda: felzenszwalb(minmax_u8(x), *self._params))

    def submit_u8(self, img_u8_hwc):
        a = np.array(img_u8_hwc, copy=True)
        return self._ex.submit(lambda: felzenszwalb(a, *self._params))

    def map(self, images_chw):
        return [f.result() for f in [self.submit(x) for x in images_chw]]

    def close(self):
        self._ex.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
