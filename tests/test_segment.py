"""CPU segmentation front-end: libmpxseg.so against the committed scikit-image 0.18.3
vectors (tests/golden/felzenszwalb_skimage0183.npz, written by tests/golden/make_felzenszwalb_golden.py with the
image's second interpreter), plus properties of the algorithm.  No GPU, no torch in the code under test."""
import ctypes
import os
import re

import numpy as np
import pytest
from scipy import ndimage

import __graft_entry__ as entry

entry.build_seg()
from network_interpretation_imagenet_amd import segment  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "felzenszwalb_skimage0183.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files if "/" in k})


def test_header_symbols_exported_and_bound():
    text = open(os.path.join(ROOT, "include", "mpx_seg.h")).read()
    declared = set(re.findall(r"^int\s+(mpxseg_\w+)\s*\(", text, flags=re.M))
    assert declared and declared == set(segment.SIGNATURES)
    lib = ctypes.CDLL(segment.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name)


@pytest.mark.parametrize("name", CASES)
def test_bit_exact_against_skimage_vectors(name):
    img, ref = GOLD[name + "/image"], GOLD[name + "/labels"]
    scale, sigma, min_size = GOLD[name + "/params"]
    out = segment.felzenszwalb(img, scale, sigma, int(min_size))
    assert out.dtype == np.int64 and out.shape == ref.shape
    assert np.array_equal(out, ref)


def test_reference_defaults_are_the_call_of_the_generators():
    # felzenszwalb(img_as_float(img_show), scale=100, sigma=0.5, min_size=50)  (generate_gp_training_data_imagenet.py:183)
    img = GOLD["blobs224/image"]
    assert np.array_equal(segment.felzenszwalb(img), GOLD["blobs224/labels"])


def test_batch_equals_single_any_thread_count():
    imgs = np.stack([GOLD[n + "/image"] for n in ("blobs224", "noise224", "blocky224")])
    want = np.stack([GOLD[n + "/labels"] for n in ("blobs224", "noise224", "blocky224")])
    for threads in (1, 2, 0):
        labels, counts = segment.felzenszwalb_batch(imgs, threads=threads)
        assert labels.dtype == np.int32 and np.array_equal(labels, want)
        assert counts.tolist() == [int(w.max()) + 1 for w in want]
    labels, counts = segment.felzenszwalb_batch(np.zeros((0, 8, 8, 3), np.uint8))
    assert labels.shape == (0, 8, 8) and counts.size == 0


def test_pool_matches_direct_and_applies_img_show():
    rs = np.random.RandomState(0)
    xs = [rs.randn(3, 48, 40).astype(np.float32) for _ in range(5)]
    with segment.SegmenterPool(workers=3) as pool:
        got = pool.map(xs)
        got_u8 = pool.submit_u8(GOLD["twotone40/image"]).result()
    for x, g in zip(xs, got):
        assert np.array_equal(g, segment.felzenszwalb(segment.minmax_u8(x)))
    assert np.array_equal(got_u8, GOLD["twotone40/labels"])


@pytest.mark.parametrize("seed", range(6))
def test_properties_contiguous_connected_min_size(seed):
    rs = np.random.RandomState(seed)
    h, w = rs.randint(20, 90, 2)
    img = np.clip(ndimage.gaussian_filter(rs.randn(h, w, 3), (3, 3, 0)) * 400 + 128, 0, 255).astype(np.uint8)
    min_size = int(rs.choice([5, 20, 50]))
    scale = float(rs.choice([10, 100]))
    seg = segment.felzenszwalb(img, scale=scale, sigma=0.5, min_size=min_size)
    S = int(seg.max()) + 1
    assert seg.min() == 0 and np.array_equal(np.unique(seg), np.arange(S))
    first = [int(np.flatnonzero(seg.ravel() == s)[0]) for s in range(S)]
    assert first == sorted(first)                       # labels in raster order of each segment's first pixel
    sizes = np.bincount(seg.ravel())
    assert S == 1 or sizes.min() >= min_size
    eight = np.ones((3, 3), int)
    for s in range(S):
        assert ndimage.label(seg == s, structure=eight)[1] == 1
    assert np.array_equal(seg, segment.felzenszwalb(img.copy(), scale=scale, sigma=0.5, min_size=min_size))    # deterministic


def test_degenerate_shapes_and_errors():
    assert segment.felzenszwalb(np.full((1, 1, 3), 9, np.uint8)).tolist() == [[0]]
    row = segment.felzenszwalb(np.arange(40, dtype=np.uint8).reshape(1, 40) * 6, min_size=1)
    assert row.shape == (1, 40) and row[0, 0] == 0
    col = segment.felzenszwalb(np.zeros((17, 1), np.uint8))
    assert col.max() == 0
    with pytest.raises(ValueError):
        segment.felzenszwalb(np.zeros((4, 4, 3), np.float32))
    with pytest.raises(ValueError):
        segment.felzenszwalb(np.zeros((4, 4, 5), np.uint8))
    with pytest.raises(ValueError):
        segment.felzenszwalb_batch(np.zeros((4, 4, 3), np.uint8))
    lib = segment.load()
    assert lib.mpxseg_felzenszwalb(None, 4, 4, 3, 100.0, 0.5, 50, None) == -1
    assert lib.mpxseg_felzenszwalb_batch(None, 2, 4, 4, 3, 100.0, 0.5, 50, None, None, 1) == -1
    assert lib.mpxseg_minmax_u8(None, 3, 4, 4, None) == -1


def test_argsort_hook_is_numpys_generic_quicksort():
    rs = np.random.RandomState(1)
    v = rs.rand(5000)
    assert np.array_equal(segment.argsort_f64(v), np.argsort(v, kind="stable"))      # no ties: any correct sort
    t = rs.randint(0, 5, 3000).astype(np.float64)
    o = segment.argsort_f64(t)
    assert np.array_equal(np.sort(o), np.arange(t.size)) and np.all(np.diff(t[o]) >= 0)
    # known answers of the unstable procedure (NumPy 1.26.4, generic path): 17 equal keys are partitioned once
    assert segment.argsort_f64(np.zeros(17)).tolist() == [0, 14, 13, 12, 11, 10, 9, 15, 8, 6, 5, 4, 3, 2, 1, 7, 16]
    assert segment.argsort_f64(np.zeros(16)).tolist() == list(range(16))
    assert segment.argsort_f64(np.zeros(0)).size == 0


def test_minmax_u8_is_img_show():
    # img_show -= min; /= max; *= 255; astype(uint8)   (generate_gp_training_data_imagenet.py:171-178)
    rs = np.random.RandomState(2)
    x = (rs.randn(3, 31, 45) * 1.7).astype(np.float32)
    ref = x.copy().transpose(1, 2, 0)
    ref -= ref.min()
    ref /= ref.max()
    ref *= 255
    assert np.array_equal(segment.minmax_u8(x), ref.astype(np.uint8))
    assert segment.minmax_u8(np.full((3, 4, 4), 2.5, np.float32)).max() == 0
