#!/usr/bin/env python3
"""Latency of the single-image cases (BASELINE config 5: one image, every window start scored in one call).
usage: python tools/latency_bench.py [arch] [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import masks, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = MaskedForwardEngine(arch, max_batch=512, device=0).load_state_dict(synth.make_state_dict(arch))
img = synth.make_images(1, kind="noise")[0]
for name, seg in (("felzenszwalb fixture S=46", np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "segments_blobs.npz"))["segments"][0].astype(np.int64)),
                  ("14x14 grid S=196", synth.grid_segments())):
    s = int(len(np.unique(seg)))
    onoff = masks.windows_onoff(s, range(0, masks.bo_upper_bound(s) + 1))     # the whole BO domain
    eng.score_masks(img, seg, onoff, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.score_masks(img, seg, onoff, 0)                                 # host arrays in, host scores out
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%s %s: %d window masks per call, %.2f ms per call end to end (H2D + K0 + forward + D2H) = %.0f masked fwd/s" % (
        arch, name, onoff.shape[0], dt * 1e3, onoff.shape[0] / dt))
