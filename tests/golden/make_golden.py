"""Regenerates tests/golden/scores_<arch>.npz with the CPU oracle (run in this container:
`python tests/golden/make_golden.py`).  The reference itself cannot be imported here (see
oracle/resnet_ref.py header), so these vectors pin the oracle restatement against drift and let
the GPU parity tests run without recomputing the slow batch-1 CPU loop; they are not outputs of
the reference's files.

  cfg-1  resnet18 : blob image 0, felzenszwalb segments (S=46), 64 masks =
                    47 reference windows (firstIndex 0..46) + 17 Bernoulli(0.4) mask-vectors
  resnet101       : blob image 1, S=23, 16 masks = 8 windows + 8 Bernoulli
score_f32 = reference-style batch-1 fp32 loop; score_f64 = same arithmetic in fp64 (yardstick).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from network_interpretation_imagenet_amd import synth, masks  # noqa: E402
from oracle import scorer  # noqa: E402

torch.set_num_threads(8)
segs = np.load(os.path.join(HERE, "segments_blobs.npz"))["segments"].astype(np.int64)
imgs = synth.make_images(2, seed=1234, kind="blobs")


def case(arch, img_idx, n_windows, n_random):
    sd = synth.make_state_dict(arch, seed=7)
    seg = segs[img_idx]
    s = len(np.unique(seg))
    x = scorer.to_tensor_normalize(imgs[img_idx])
    label = scorer.base_prediction(sd, arch, x)
    starts = list(range(n_windows))
    onoff = np.concatenate([masks.windows_onoff(s, starts), synth.random_onoff(n_random, s, seed=4321)])
    score32, pred = scorer.score_masks_reference_loop(sd, arch, x, seg, onoff, label)
    score64, pred64 = scorer.score_masks_batched(sd, arch, x, seg, onoff, label, dtype=torch.float64, chunk=8)
    print(arch, "S", s, "label", label, "score range", score32.min(), score32.max(),
          "fp32-vs-fp64", np.abs(score32 - score64).max(), "pred==label", int((pred == label).sum()))
    np.savez_compressed(os.path.join(HERE, "scores_%s.npz" % arch), arch=arch, image_index=img_idx,
                        weight_seed=7, image_seed=1234, label=label, onoff=onoff, score_f32=score32,
                        score_f64=score64, pred=pred.astype(np.int32), pred_f64=pred64.astype(np.int32))


case("resnet18", 0, 47, 17)
case("resnet101", 1, 8, 8)
