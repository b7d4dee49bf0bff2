#!/usr/bin/env python3
"""Whole TRAINED networks as parity fixtures (8c rule 3).

The reference ships the trained checkpoints of its two small networks:
    saved_checkpoints/mnist/checkpoint.pth.tar                 Classification_Net (generate_gp_training_data_mnist.py:86-105)
    saved_checkpoints/cifar10+-resnet-56/model_best.pth.tar    ResNetCifar(56)     (models/resnet.py:77-146; val err 5.9 %)
This script reads them with the safe loader only (torch.load(weights_only=True), argparse.Namespace allow-listed: nothing
from the files is executed) and writes, per network, ONE .npz under tests/golden/ holding
    * the state_dict tensors (data: the engine needs every weight to run the whole net on the GPU box, where
      /root/reference does not exist),
    * seeded synthetic pictures of the network's input shape (there is no MNIST / CIFAR data offline), their felzenszwalb
      label maps (the native front-end libmpxseg.so, bit-exact against scikit-image 0.18.3 on its own fixtures) and a list of
      removed-superpixel sets drawn the way the scorers draw them,
    * what the CPU oracle (oracle/smallnets_ref.py) computes for them: logits of the unmasked pictures in fp32 and fp64, and
      per mask the network input, logits, softmax score of the label and argmax.
Run here only:  python tests/golden/make_smallnets_golden.py
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build_seg()
from network_interpretation_imagenet_amd import masks, segment  # noqa: E402
from oracle import smallnets_ref as ref  # noqa: E402

torch.set_num_threads(8)
REF = "/root/reference/saved_checkpoints"


def synth_digit(seed):
    """28x28 grey picture in [0,1] with a few bright strokes (what ToTensor() yields for MNIST, ..._mnist.py:57-69)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:28, 0:28].astype(np.float64)
    img = np.zeros((28, 28))
    for _ in range(3):
        x0, y0, x1, y1 = rng.uniform(5, 23, 4)
        t = np.clip(((xx - x0) * (x1 - x0) + (yy - y0) * (y1 - y0)) / ((x1 - x0) ** 2 + (y1 - y0) ** 2 + 1e-9), 0, 1)
        d2 = (xx - (x0 + t * (x1 - x0))) ** 2 + (yy - (y0 + t * (y1 - y0))) ** 2
        img = np.maximum(img, np.exp(-d2 / 2.5))
    img = np.floor(img * 255.999) / 255.0
    return img.astype(np.float32)[None]


def synth_cifar(seed):
    """32x32x3 picture normalised to [-1,1] (ToTensor + Normalize((.5,.5,.5),(.5,.5,.5)), ..._cifar.py:52-54)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:32, 0:32].astype(np.float64)
    out = np.zeros((3, 32, 32), dtype=np.float32)
    for c in range(3):
        acc = np.zeros((32, 32))
        for _ in range(5):
            fx, fy, ph, amp = rng.uniform(0, 1, 4)
            acc += (0.3 + amp) * np.sin(2 * np.pi * ((0.3 + 2.2 * fx) * xx / 32 + (0.3 + 2.2 * fy) * yy / 32) + 2 * np.pi * ph)
        acc = (acc - acc.min()) / (acc.max() - acc.min())
        u8 = np.floor(acc * 255.999)
        out[c] = ((u8 / 255.0).astype(np.float32) - np.float32(0.5)) / np.float32(0.5)
    return out


def img_u8(x_chw):
    img = np.array(x_chw, dtype=np.float32, copy=True).transpose(1, 2, 0)
    img -= img.min()
    img /= img.max()
    img *= 255
    return img.astype(np.uint8)


def case(arch, sd, pictures, min_size, n_removed, n_masks, seed):
    out = {"arch": np.array(arch)}
    for k, v in sd.items():
        out["sd/" + k] = v.numpy()
    sd64 = {k: v.double() for k, v in sd.items()}
    rnd = random.Random(seed)
    out["n_pictures"] = np.array(len(pictures))
    for i, x in enumerate(pictures):
        seg = segment.felzenszwalb(img_u8(x), scale=100, sigma=0.5, min_size=min_size).astype(np.int32)     # ..._cifar.py:284, ..._mnist.py:181
        uniq = np.unique(seg)
        with torch.no_grad():
            l32 = ref.forward(sd, torch.from_numpy(x[None]), arch).numpy()[0]
            l64 = ref.forward(sd64, torch.from_numpy(x[None]).double(), arch).numpy()[0]
        label = int(l32.argmax())
        # random.sample(range(uniq[0], uniq[-1]), n): the last label can never be drawn (..._cifar.py:306, ..._mnist.py:209)
        removed_lists = [sorted(r) for r in masks.draw_removed_sets(uniq, min(n_removed, len(uniq) - 1), n_masks, rnd)]       # the package's sampler
        removed_lists[0] = []                                   # nothing removed: only the double min-max rescale acts
        org = ref.org_img_minmax255(x)
        inputs = np.stack([ref.masked_input(org, ref.removed_mask_u8(seg, r)) for r in removed_lists])
        with torch.no_grad():
            ml32 = ref.forward(sd, torch.from_numpy(inputs), arch).numpy()
            ml64 = ref.forward(sd64, torch.from_numpy(inputs).double(), arch).numpy()
        score, pred = ref.score_removed_loop(sd, arch, x, seg, removed_lists, label)
        p = "pic%d/" % i
        out[p + "x"] = x
        out[p + "segments"] = seg
        out[p + "removed"] = ref.removed_onoff(seg, removed_lists)
        out[p + "label"] = np.array(label)
        out[p + "logits_f32"] = l32
        out[p + "logits_f64"] = l64
        out[p + "masked_inputs"] = inputs.astype(np.float32)
        out[p + "masked_logits_f32"] = ml32
        out[p + "masked_logits_f64"] = ml64
        out[p + "score_f32"] = score
        out[p + "pred"] = pred.astype(np.int32)
        print("%s pic %d: S=%d label=%d logits %.2f..%.2f | masks: %d keep the label, score %.4f..%.4f, |f32-f64| logits %.2e" % (
            arch, i, len(uniq), label, l32.min(), l32.max(), int((pred == label).sum()), score.min(), score.max(),
            np.abs(ml32 - ml64).max()))
    path = os.path.join(HERE, "smallnet_%s.npz" % arch)
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f MB" % (os.path.getsize(path) / 1e6))


def main():
    m = torch.load(os.path.join(REF, "mnist/checkpoint.pth.tar"), map_location="cpu", weights_only=True)
    sd = {k: v.float().contiguous() for k, v in m["model"].items() if "num_batches_tracked" not in k}
    case("mnist_net", sd, [synth_digit(s) for s in (1, 2)], min_size=5, n_removed=1, n_masks=24, seed=11)
    with torch.serialization.safe_globals([argparse.Namespace]):
        c = torch.load(os.path.join(REF, "cifar10+-resnet-56/model_best.pth.tar"), map_location="cpu", weights_only=True)
    sd = {(k[7:] if k.startswith("module.") else k): v.float().contiguous() for k, v in c["state_dict"].items()
          if "num_batches_tracked" not in k}
    case("cifar_resnet56", sd, [synth_cifar(s) for s in (3, 4)], min_size=10, n_removed=5, n_masks=24, seed=12)


if __name__ == "__main__":
    sys.exit(main())
