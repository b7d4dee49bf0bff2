#!/usr/bin/env python3
"""The GP-training-data generators in miniature (generate_gp_training_data_imagenet.py / gp_superpixel_data_imagenet.py), several
images in one pass through the reference-named entry points:

    python examples/gp_training_data_demo.py [arch] [images]

api.validate_many        -> per image the number of random windows whose masked prediction is still the label (and, with a mask_dir,
                            the mask_{i}_{label}.png files gp_regression.py reads)
api.validate_summed_many -> per image the summed heat map of the correctly predicted masks
api.prepare_training_data -> the GP's (train_x, train_y) from one image's PNG folder
The window tables of consecutive images share full forward batches; felzenszwalb runs on host threads meanwhile.
"""
import os
import random
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import api, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MEAN, STD, MaskedForwardEngine, whole_round_batch  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 8
imgs = synth.make_images(n_img, seed=99, kind="noise")
mean, std = torch.tensor(MEAN).view(3, 1, 1), torch.tensor(STD).view(3, 1, 1)
xs = [(torch.from_numpy(im).permute(2, 0, 1).float().div(255) - mean) / std for im in imgs]
model = MaskedForwardEngine(arch, max_batch=whole_round_batch(2400)).load_state_dict(synth.make_state_dict(arch))   # 2340: whole rounds of tiles
labels = [model.predict(x)[0] for x in xs]
labels[1] = (labels[1] + 1) % 1000                    # one image whose unmasked prediction is "wrong": the reference skips it
val_loader = [(x[None], torch.tensor([l])) for x, l in zip(xs, labels)]
idx = list(range(1, n_img + 1))

with tempfile.TemporaryDirectory() as tmp:
    api.configure(num_mask_samples=100, mask_dir=tmp, seed=None)
    t0 = time.perf_counter()
    counts = api.validate_many(val_loader, model, None, idx, rng=random.Random(0))
    dt = time.perf_counter() - t0
    print("%s, %d images: correct-prediction counts of 100 random windows each: %s   (%.2f s, PNGs written)" % (arch, n_img, counts, dt))
    api.configure(mask_dir=None)
    t0 = time.perf_counter()
    maps = api.validate_summed_many(val_loader, model, None, idx, rng=random.Random(0))
    print("summed heat maps: %s   (%.2f s)" % ({k: (None if v is None else int(v.max())) for k, v in maps.items()}, time.perf_counter() - t0))
    first = next(k for k, v in counts.items() if v is not None)
    train_x, train_y = api.prepare_training_data(os.path.join(tmp, "img_%d" % first))
    same = np.array_equal(api.summed_heatmap_from_folder(os.path.join(tmp, "img_%d" % first)), maps[first])
    print("image %d: GP training set from its PNG folder: train_x %s, train_y %s, max y %d; equals the device heat map: %s" % (
        first, tuple(train_x.shape), tuple(train_y.shape), int(train_y.max()), same))
model.close()
