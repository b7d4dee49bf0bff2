#!/opt/conda/bin/python3.9
"""Golden vectors for the native segmentation front-end.

Run with the image's second interpreter, the only one that has scikit-image (0.18.3):
    /opt/conda/bin/python3.9 tests/golden/make_felzenszwalb_golden.py
It calls the THIRD-PARTY function the reference calls --
    felzenszwalb(img_as_float(img_show), scale=100, sigma=0.5, min_size=50)
    (generate_gp_training_data_imagenet.py:183; gp_superpixel_data_imagenet.py:218; bayesian_active_learning_imagenet.py:150)
-- on seeded synthetic u8 images and stores inputs + label maps in felzenszwalb_skimage0183.npz.
Nothing from /root/reference is imported or read.

Edges of exactly equal weight are ordered by np.argsort's unstable default sort, and NumPy >= 1.25 swaps in an
AVX-512 sort on hosts that have it, so upstream's result on tie-heavy pictures depends on the CPU.  The vectors
are taken with that dispatch disabled (NumPy's generic introsort, the only variant the reference's NumPy
generation had); the *_ties cases are the ones that differ under the AVX-512 sort."""
import os

os.environ["NPY_DISABLE_CPU_FEATURES"] = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"

import numpy as np  # noqa: E402
import skimage  # noqa: E402
from skimage.segmentation import felzenszwalb  # noqa: E402
from skimage.util import img_as_float  # noqa: E402


def blobs(rs, h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(8):
            fy, fx = rs.uniform(0.5, 4.0, 2) * 2 * np.pi / max(h, w)
            img[:, :, c] += rs.uniform(0.3, 1.0) * np.sin(fy * yy + fx * xx + rs.uniform(0, 2 * np.pi))
    img -= img.min()
    img /= img.max()
    return (img * 255).astype(np.uint8)


def blocky(rs, h, w, b):
    cols = rs.randint(0, 256, size=((h + b - 1) // b, (w + b - 1) // b, 3))
    img = np.kron(cols, np.ones((b, b, 1)))[:h, :w].astype(np.int64) + rs.randint(-6, 7, size=(h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    rs = np.random.RandomState(20260101)
    cases = {}
    cases["blobs224"] = (blobs(rs, 224, 224), (100, 0.5, 50))
    cases["noise224"] = (rs.randint(0, 256, size=(224, 224, 3)).astype(np.uint8), (100, 0.5, 50))
    cases["blocky224"] = (blocky(rs, 224, 224, 32), (100, 0.5, 50))
    cases["blobs_64x48"] = (blobs(rs, 64, 48), (100, 0.5, 50))
    cases["constant32"] = (np.full((32, 32, 3), 77, np.uint8), (100, 0.5, 50))
    two = np.zeros((40, 40, 3), np.uint8)
    two[:, 17:] = 200
    cases["twotone40"] = (two, (100, 0.5, 50))
    g = blobs(rs, 100, 100)[:, :, :1].repeat(3, axis=2)
    cases["grey100"] = (g, (100, 0.5, 50))
    cases["blobs224_s50"] = (blobs(rs, 224, 224), (50, 0.8, 20))
    cases["blocky96_sigma1.5"] = (blocky(rs, 96, 96, 12), (300, 1.5, 10))
    cases["quantised96_ties"] = ((rs.randint(0, 4, size=(96, 96, 3)) * 60).astype(np.uint8), (300, 0.0, 20))
    cases["noise_grey_64x80_ties"] = (rs.randint(0, 256, size=(64, 80, 1)).astype(np.uint8), (100, 0.0, 50))
    yy, xx = np.mgrid[0:59, 0:105]
    cases["ramp_59x105_ties"] = (((yy + 2 * xx) % 256)[:, :, None].astype(np.uint8), (1, 0.5, 1))
    cases["noise224_sigma0_ties"] = (rs.randint(0, 256, size=(224, 224, 3)).astype(np.uint8), (100, 0.0, 1))
    out = {"skimage_version": np.array(skimage.__version__)}
    for name, (img, (scale, sigma, min_size)) in cases.items():
        seg = felzenszwalb(img_as_float(img), scale=scale, sigma=sigma, min_size=min_size)
        assert seg.min() == 0 and len(np.unique(seg)) == seg.max() + 1
        out[name + "/image"] = img
        out[name + "/labels"] = seg.astype(np.int32)
        out[name + "/params"] = np.array([scale, sigma, min_size], np.float64)
        print("%-18s %s S=%d" % (name, img.shape, seg.max() + 1))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "felzenszwalb_skimage0183.npz"), **out)


if __name__ == "__main__":
    main()
