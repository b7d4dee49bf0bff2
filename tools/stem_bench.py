#!/usr/bin/env python3
"""Time the stem by superposition (mpx_stem_table_build / mpx_stem_table_apply) against K0 + the MFMA stem + max pool, per forward batch.
usage: python tools/stem_bench.py [batch=2340] [masks_per_image=512] [seg=grid|felz|grid8] [reps=5]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2340
mpi = int(sys.argv[2]) if len(sys.argv) > 2 else 512
kind = sys.argv[3] if len(sys.argv) > 3 else "grid"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda", 0)
eng = MaskedForwardEngine("resnet18", max_batch=batch, device=0).load_state_dict(synth.make_state_dict("resnet18"))
n_img = -(-batch // mpi)
imgs = [torch.from_numpy(a).to(dev) for a in synth.make_images(n_img, seed=3, kind="blobs" if kind == "felz" else "noise")]
if kind == "felz":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seg_np = np.load(os.path.join(root, "tests", "golden", "segments_blobs.npz"))["segments"][0]
    seg_np = np.unique(seg_np, return_inverse=True)[1].reshape(224, 224).astype(np.int32)
else:
    seg_np = synth.grid_segments(block=8 if kind == "grid8" else 16)
S = int(seg_np.max()) + 1
seg = torch.from_numpy(seg_np).to(dev)
onoff = torch.from_numpy(synth.random_onoff(batch, S)).to(dev)
p = lambda t: C.c_void_p(t.data_ptr())


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def build_only():
    for i in range(n_img):
        eng.build_stem_table(imgs[i], seg, S)


def table():
    for i in range(n_img):
        eng.build_stem_table(imgs[i], seg, S)
        r0 = i * mpi
        eng.apply_stem_table(onoff[r0:min(r0 + mpi, batch)], r0)


oh = torch.empty(batch, 56, 56, 64, dtype=torch.float16, device=dev)
ol = torch.empty_like(oh)


def conv():
    for i in range(n_img):
        r0 = i * mpi
        eng.stage_masks(imgs[i], seg, onoff[r0:min(r0 + mpi, batch)], r0)
    eng._lib.mpx_stem_conv_maxpool(eng._h, p(oh), p(ol), batch, None)


tb, tt, tc = timed(build_only), timed(table), timed(conv)
print("%s S=%d, %d masks (%d images x %d): table build %.3f ms, build + apply %.3f ms (apply %.3f); K0 + MFMA stem + pool %.3f ms" % (
    kind, S, batch, n_img, mpi, tb, tt, tt - tb, tc))
