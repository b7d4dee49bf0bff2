#!/opt/conda/bin/python3.9
"""Randomised comparison of libmpxseg.so with scikit-image (needs the interpreter that has skimage:
/opt/conda/bin/python3.9 tools/seg_stress_vs_skimage.py [cases]).  Not part of the test suite: the suite uses
the committed vectors."""
import importlib.util
import os
import sys
import time

import numpy as np
from skimage.segmentation import felzenszwalb
from skimage.util import img_as_float

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("segment", os.path.join(root, "network_interpretation_imagenet_amd", "segment.py"))
segment = importlib.util.module_from_spec(spec)
spec.loader.exec_module(segment)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(7)
bad = 0
t_mine = t_sk = 0.0
for i in range(n):
    h, w = rs.randint(1, 120, 2) if i % 4 else (224, 224)
    c = int(rs.choice([1, 3, 3, 3]))
    kind = i % 5
    if kind == 0:
        img = rs.randint(0, 256, size=(h, w, c))
    elif kind == 1:
        img = rs.randint(0, 4, size=(h, w, c)) * 60                      # heavy ties
    elif kind == 2:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(yy * (k + 1) + xx * 2) % 256 for k in range(c)], -1)
    elif kind == 3:
        b = int(rs.randint(2, 20))
        img = np.kron(rs.randint(0, 256, size=(h // b + 1, w // b + 1, c)), np.ones((b, b, 1)))[:h, :w]
    else:
        img = np.clip(rs.normal(128, 40, size=(h, w, c)), 0, 255)
    img = img.astype(np.uint8)
    scale = float(rs.choice([1, 10, 100, 100, 300]))
    sigma = float(rs.choice([0.0, 0.5, 0.5, 0.8, 2.0]))
    min_size = int(rs.choice([1, 20, 50, 50]))
    t0 = time.perf_counter()
    ref = felzenszwalb(img_as_float(img if c > 1 else img[:, :, 0]), scale=scale, sigma=sigma, min_size=min_size)
    t1 = time.perf_counter()
    out = segment.felzenszwalb(img, scale, sigma, min_size)
    t2 = time.perf_counter()
    t_sk += t1 - t0
    t_mine += t2 - t1
    if not np.array_equal(ref, out):
        bad += 1
        if os.environ.get("SEG_DUMP"):
            np.savez(os.path.join(os.environ["SEG_DUMP"], "case%d.npz" % i), img=img, ref=ref, out=out,
                     params=np.array([scale, sigma, min_size]))
        print("MISMATCH case %d kind %d shape %s scale %g sigma %g min_size %d: S %d vs %d, %d px differ" % (
            i, kind, img.shape, scale, sigma, min_size, ref.max() + 1, out.max() + 1, (ref != out).sum()))
print("%d cases, %d mismatches; skimage %.2f s, libmpxseg %.2f s" % (n, bad, t_sk, t_mine))
