#!/usr/bin/env python3
"""Latency of the single-image cases (BASELINE config 5: one image, the Bayesian-optimisation loop over window starts).
usage: python tools/latency_bench.py [arch] [reps]

Per label map (the committed felzenszwalb fixture and the 14x14 grid): (a) one engine call over the whole BO domain
firstIndex in [0, int(0.6 S)] -- host arrays in, host scores out; (b) what the reference-named entry does: the FIRST
api.sample_loss of an image (segmentation excluded: the label map is configured; unmasked row + every window start in one packed
pass) and every LATER call (a table look-up); (c) bo.bayesian_optimisation end to end (5 pre-samples + 8 iterations, the
reference's setting, bayesian_active_learning_imagenet.py:478-486) and the time per BO round."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import api, bo, masks, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402
from oracle import scorer  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = MaskedForwardEngine(arch, max_batch=512, device=0).load_state_dict(synth.make_state_dict(arch))
img = synth.make_images(1, kind="blobs")[0]
x = scorer.to_tensor_normalize(img)
label, _ = eng.predict(x)
loader = [(x[None], torch.tensor([label]))]
for name, seg in (("felzenszwalb fixture", np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "segments_blobs.npz"))["segments"][0].astype(np.int64)),
                  ("14x14 grid", synth.grid_segments())):
    s = int(len(np.unique(seg)))
    ub = masks.bo_upper_bound(s)
    onoff = masks.windows_onoff(s, range(0, ub + 1))     # the whole BO domain
    for _ in range(20):                                   # warm-up: the clocks of an idle chip take a few calls to come up
        eng.score_masks(x, seg, onoff, label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.score_masks(x, seg, onoff, label)                               # host arrays in, host scores out
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%s, %s S=%d: (a) %d window masks (the BO domain) per engine call: %.2f ms end to end (H2D + K0 + forward + D2H) = %.0f masked fwd/s" % (
        arch, name, s, onoff.shape[0], dt * 1e3, onoff.shape[0] / dt))
    first, later = [], []
    for _ in range(reps):
        api.configure(eval_img_index=1, segmenter=lambda im, seg=seg: seg)        # drops the cached session
        t0 = time.perf_counter()
        api.sample_loss([3.0], loader, eng, None)
        t1 = time.perf_counter()
        for f in range(8):
            api.sample_loss([float(f)], loader, eng, None)
        t2 = time.perf_counter()
        first.append(t1 - t0)
        later.append((t2 - t1) / 8)
    print("    (b) api.sample_loss: first call of an image %.2f ms (base prediction + %d window starts scored), later calls %.1f us (table look-up)" % (
        np.median(first) * 1e3, s + 1, np.median(later) * 1e6))
    api.configure(eval_img_index=1, segmenter=lambda im, seg=seg: seg)
    t0 = time.perf_counter()
    xp, yp = bo.bayesian_optimisation(8, api.sample_loss, loader, eng, None, np.array([[0, ub]]), n_pre_samples=5)
    dt = time.perf_counter() - t0
    print("    (c) bo.bayesian_optimisation, 5 pre-samples + 8 iterations over [0, %d]: %.1f ms in all = %.2f ms per BO round (GP fit + EI on the host included), best score %.4f" % (
        ub, dt * 1e3, dt * 1e3 / 13, float(np.max(yp))))
api.configure(eval_img_index=1, segmenter=None)
eng.close()
