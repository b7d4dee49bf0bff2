#!/usr/bin/env python3
"""Throughput of the reference-named entry (api.validate_summed_many) on N synthetic images: masked forwards per second with
the window tables of consecutive images PACKED into full forward batches (api.fill_tables -> MaskedForwardEngine.score_images)
and, for comparison, scored one image per forward (round 2's path: SaliencySession.table()).
usage: python tools/api_throughput.py [arch] [images] [max_batch] [blobs|noise|mixed]   (felzenszwalb finds ~30 superpixels on the smooth
"blobs" pictures and ~330 on uniform noise; natural images lie between)"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import api, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402
from oracle import scorer  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 32
max_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2340      # engine.whole_round_batch(2400)
kind = sys.argv[4] if len(sys.argv) > 4 else "noise"
eng = MaskedForwardEngine(arch, max_batch=max_batch, device=0).load_state_dict(synth.make_state_dict(arch))
if kind == "mixed":      # noise pictures (S ~ 330: the stem table) and blobs pictures (S ~ 28: K0 + the MFMA stem) in turn: both kinds of staging in one loader
    a, b = synth.make_images(n_img, seed=5, kind="noise"), synth.make_images(n_img, seed=5, kind="blobs")
    imgs = np.stack([(a if i % 2 == 0 else b)[i] for i in range(n_img)])
else:
    imgs = synth.make_images(n_img, seed=5, kind=kind)
xs = [scorer.to_tensor_normalize(im) for im in imgs]
labels = [eng.predict(x)[0] for x in xs]
loader = [(x[None], torch.tensor([l])) for x, l in zip(xs, labels)]
idx = list(range(1, n_img + 1))


def run_packed():
    return api.validate_summed_many(loader, eng, None, idx, num_mask_samples=100, rng=random.Random(1), workers=8)


def run_per_image():
    """round 2's path: one image per forward (batch = S + 1), base prediction as a batch-1 forward"""
    from network_interpretation_imagenet_amd import masks, segment
    out, rng = {}, random.Random(1)
    with segment.SegmenterPool(workers=8) as pool:
        futs = [pool.submit(x.numpy()) for x in xs]
        for i, (x, l, f) in enumerate(zip(xs, labels, futs)):
            s = api.SaliencySession(eng, x, l, segments=f.result())
            firsts = masks.draw_first_indices(s.num_segments, 100, rng)
            pred = s.table()[1]
            out[i + 1] = s.summed_labels(firsts, np.array([pred[q] for q in firsts]) == s.label)
    return out


rows = None
for name, fn in (("per image (round 2)", run_per_image), ("packed (fill_tables)", run_packed), ("per image (round 2)", run_per_image), ("packed (fill_tables)", run_packed)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rows is None:
        from network_interpretation_imagenet_amd import segment
        rows = sum(int(segment.felzenszwalb(api.img_show_u8(x.numpy())).max()) + 3 for x in xs)      # S + 1 windows + the unmasked row
    print("%-22s %s, %d %s images, %d masked forwards (unmasked row + every window start): %.3f s -> %.0f masked forwards/s (segmentation included)"
          % (name, arch, n_img, kind, rows, dt, rows / dt))
    if name.startswith("packed"):
        keep = res
    else:
        base = res
assert all(np.array_equal(keep[i], base[i]) for i in idx), "packed and per-image heat maps differ"
print("heat maps of both paths are identical")
eng.close()
