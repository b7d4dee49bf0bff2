#!/usr/bin/env python3
"""Real trained weights for numerics regression (SURVEY.md 8c rule 3; row f4's "real-weight regression").

The reference ships one trained network whose layers have a shape this engine runs: the CIFAR-10 ResNet-56 of
saved_checkpoints/cifar10+-resnet-56/model_best.pth.tar (5.9 % top-1 error) has eighteen 64->64 3x3 conv + BN
pairs in layer3, the shape of ResNet-18/34's layer1.  This script reads the checkpoint with the safe loader
(torch.load(weights_only=True), argparse.Namespace allow-listed; nothing from the file is executed), copies FOUR
conv+BN pairs (layer3.4 and layer3.8, both convs) into trained_layers_cifar_resnet56.npz, and adds expected
outputs of the oracle's conv+BN+ReLU on a seeded input at sampled positions, so that drift of the oracle is caught.

Run here only (/root/reference does not exist on the GPU box):  python tests/golden/make_trained_layers_golden.py"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = "/root/reference/saved_checkpoints/cifar10+-resnet-56/model_best.pth.tar"
PAIRS = [("layer3.4.conv1", "layer3.4.bn1"), ("layer3.4.conv2", "layer3.4.bn2"),
         ("layer3.8.conv1", "layer3.8.bn1"), ("layer3.8.conv2", "layer3.8.bn2")]


def main():
    with torch.serialization.safe_globals([argparse.Namespace]):
        ck = torch.load(CKPT, map_location="cpu", weights_only=True)
    sd = ck["state_dict"]
    out = {"source": np.array("cifar10+-resnet-56/model_best.pth.tar epoch %d err1 %.1f" % (ck["epoch"], ck["best_err1"]))}
    g = torch.Generator().manual_seed(56)
    x = torch.randn(2, 64, 56, 56, generator=g).clamp_min(-0.5) * 1.5
    pick = torch.randperm(2 * 64 * 56 * 56, generator=g)[:4096]
    out["sample_index"] = pick.numpy().astype(np.int64)
    for n, (conv, bn) in enumerate(PAIRS):
        w = sd["module.%s.weight" % conv].float()
        assert tuple(w.shape) == (64, 64, 3, 3)
        out["w%d" % n] = w.numpy()
        for k in ("weight", "bias", "running_mean", "running_var"):
            out["bn%d_%s" % (n, k)] = sd["module.%s.%s" % (bn, k)].float().numpy()
        y = F.relu(F.batch_norm(F.conv2d(x, w, None, 1, 1), sd["module.%s.running_mean" % bn], sd["module.%s.running_var" % bn],
                                sd["module.%s.weight" % bn], sd["module.%s.bias" % bn], False, 0.0, 1e-5))
        out["expect%d" % n] = y.reshape(-1)[pick].numpy()
        out["name%d" % n] = np.array(conv)
        print(conv, "max|w| %.3f  bn gamma %.3f..%.3f  var %.4f..%.3f  out max %.2f" % (
            w.abs().max(), out["bn%d_weight" % n].min(), out["bn%d_weight" % n].max(),
            out["bn%d_running_var" % n].min(), out["bn%d_running_var" % n].max(), y.max()))
    np.savez_compressed(os.path.join(HERE, "trained_layers_cifar_resnet56.npz"), **out)


if __name__ == "__main__":
    sys.exit(main())
