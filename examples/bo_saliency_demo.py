#!/usr/bin/env python3
"""BASELINE config 5 in miniature: Bayesian optimisation over the superpixel window start, driven through the
reference-named entry points, on one synthetic image segmented by the native felzenszwalb front-end (libmpxseg.so).

    python examples/bo_saliency_demo.py [arch]

The engine scores every candidate window of the image in ONE batched pass on the first sample_loss call;
the 3 + 10 BO evaluations (bayesian_active_learning_imagenet.py:478-486) are then table look-ups.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import api, bo, masks, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MEAN, STD, MaskedForwardEngine  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
img = synth.make_images(2, seed=1234)[0]
x = (torch.from_numpy(img).permute(2, 0, 1).float().div(255) - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)

model = MaskedForwardEngine(arch, max_batch=256).load_state_dict(synth.make_state_dict(arch))
label, _ = model.predict(x)                      # the reference's correctness gate needs pred == label
val_loader = [(x[None], torch.tensor([label]))]
api.configure(eval_img_index=1, segmenter=None)       # default: segment.felzenszwalb(img_show, 100, 0.5, 50)
t0 = time.perf_counter()
seg = api.default_segmenter(api.img_show_u8(x.numpy()))
print("felzenszwalb on the host: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
s = len(np.unique(seg))
ub = masks.bo_upper_bound(s)
t0 = time.perf_counter()
xp, yp = bo.bayesian_optimisation(n_iters=10, sample_loss=api.sample_loss, val_loader=val_loader, nn_model=model,
                                  criterion=None, bounds=np.array([[0, ub]]), n_pre_samples=3)
dt = time.perf_counter() - t0
best = int(xp[np.argmax(yp), 0])
print("%s: S=%d superpixels, window k=%d, BO domain [0,%d]; 13 evaluations in %.3f s" % (arch, s, masks.window_size(s), ub, dt))
print("best window start %d -> P(label=%d) = %.5f; mask covers %d pixels" % (best, label, yp.max(), int(api.superpixel_mask(best).astype(bool).sum())))
