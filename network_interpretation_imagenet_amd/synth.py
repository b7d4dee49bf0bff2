"""Seeded synthetic inputs for benchmarks, smoke runs and parity tests.

There is no network in this environment, so `pretrained=True`
(reference generate_gp_training_data_imagenet.py:579) cannot be honoured; weights are
random-initialised with torchvision key names and shapes.  The statistics are chosen
analytically (no forward pass is needed to build them) so that activations stay O(1)
through 101 layers and the softmax is peaked but unsaturated -- otherwise a 1e-4 check on
the class probability would be vacuous:

  conv    N(0, sqrt(2 / (k*k*Cin)))        variance-preserving through conv+ReLU
  BN      gamma ~ U(0.9,1.1) (x0.25 / x0.1 on each basic / bottleneck block's last BN), beta ~ N(0,0.05),
          running_mean ~ N(0,0.05), running_var ~ U(0.8,1.25)
  fc      N(0, FC_GAIN/sqrt(C)), bias ~ N(0,0.1)

Images are u8 HWC; `blobs` gives smooth low-frequency content (felzenszwalb-friendly),
`noise` uniform random bytes (content does not affect timing).
"""
from collections import OrderedDict

import numpy as np
import torch

ARCH_DEPTHS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}
FC_GAIN = 2.5


def _bn(sd, prefix, c, g, last=0.0):
    gamma = torch.empty(c).uniform_(0.9, 1.1, generator=g)
    if last:
        gamma *= last
    sd[prefix + ".weight"] = gamma
    sd[prefix + ".bias"] = torch.randn(c, generator=g) * 0.05
    sd[prefix + ".running_mean"] = torch.randn(c, generator=g) * 0.05
    sd[prefix + ".running_var"] = torch.empty(c).uniform_(0.8, 1.25, generator=g)


def _conv(sd, name, cin, cout, k, g):
    std = (2.0 / (k * k * cin)) ** 0.5
    sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std


def make_state_dict(arch, seed=7):
    """OrderedDict of f32 CPU tensors with the torchvision ResNet key set."""
    kind, depths = ARCH_DEPTHS[arch]
    exp = 1 if kind == "basic" else 4
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    _conv(sd, "conv1", 3, 64, 7, g)
    _bn(sd, "bn1", 64, g)
    cin = 64
    for s, (w, d) in enumerate(zip((64, 128, 256, 512), depths)):
        for b in range(d):
            stride = 2 if (b == 0 and s > 0) else 1
            p = "layer%d.%d." % (s + 1, b)
            if kind == "basic":
                _conv(sd, p + "conv1", cin, w, 3, g)
                _bn(sd, p + "bn1", w, g)
                _conv(sd, p + "conv2", w, w, 3, g)
                _bn(sd, p + "bn2", w, g, last=0.25)
            else:
                _conv(sd, p + "conv1", cin, w, 1, g)
                _bn(sd, p + "bn1", w, g)
                _conv(sd, p + "conv2", w, w, 3, g)
                _bn(sd, p + "bn2", w, g)
                _conv(sd, p + "conv3", w, w * exp, 1, g)
                _bn(sd, p + "bn3", w * exp, g, last=0.1)
            if b == 0 and (stride != 1 or cin != w * exp):
                _conv(sd, p + "downsample.0", cin, w * exp, 1, g)
                _bn(sd, p + "downsample.1", w * exp, g)
            cin = w * exp
    sd["fc.weight"] = torch.randn(1000, cin, generator=g) * (FC_GAIN / cin ** 0.5)
    sd["fc.bias"] = torch.randn(1000, generator=g) * 0.1
    return sd


def make_images(n, seed=1234, kind="blobs", size=224):
    """u8[n,size,size,3]."""
    g = torch.Generator().manual_seed(seed)
    if kind == "noise":
        return torch.randint(0, 256, (n, size, size, 3), generator=g, dtype=torch.uint8).numpy()
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64),
                         indexing="ij")
    out = np.zeros((n, size, size, 3), dtype=np.uint8)
    for i in range(n):
        for c in range(3):
            acc = np.zeros((size, size))
            p = torch.rand(8, 4, generator=g, dtype=torch.float64).numpy()
            for fx, fy, ph, amp in p:
                acc += (0.3 + amp) * np.sin(2 * np.pi * ((0.5 + 3.5 * fx) * xx / size
                                                          + (0.5 + 3.5 * fy) * yy / size) + 2 * np.pi * ph)
            acc = (acc - acc.min()) / (acc.max() - acc.min())
            out[i, :, :, c] = np.floor(acc * 255.999).astype(np.uint8)
    return out


def grid_segments(size=224, block=16):
    """Fixed block grid label map i32[size,size]; 16-px blocks -> S = 196."""
    per = size // block
    idx = np.arange(size) // block
    return (idx[:, None] * per + idx[None, :]).astype(np.int32)


def random_onoff(m, s, seed=4321, p=0.4):
    """Bernoulli(p) mask-vectors u8[m,s]."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(m, s, generator=g) < p).to(torch.uint8).numpy()


def transplant_trained_layers(sd, npz_path):
    """ResNet-18/34 state_dict with layer1's four 64->64 3x3 conv + BN pairs replaced by the TRAINED pairs of
    tests/golden/trained_layers_cifar_resnet56.npz (the reference's shipped CIFAR ResNet-56, layer3.4 / layer3.8):
    the only real trained weights available offline that have a shape this engine runs.  Returns a new dict."""
    g = np.load(npz_path)
    out = dict(sd)
    slots = ["layer1.0.conv1", "layer1.0.conv2", "layer1.1.conv1", "layer1.1.conv2"]
    for n, conv in enumerate(slots):
        bn = conv.replace("conv", "bn")
        assert tuple(out[conv + ".weight"].shape) == (64, 64, 3, 3)
        out[conv + ".weight"] = torch.from_numpy(g["w%d" % n].copy())
        for k in ("weight", "bias", "running_mean", "running_var"):
            out["%s.%s" % (bn, k)] = torch.from_numpy(g["bn%d_%s" % (n, k)].copy())
    return out
