"""CPU ORACLE (test infrastructure only) -- torchvision-topology ResNet forward.

PARITY UNPINNED: the reference (`/root/reference`) ships no tests, golden vectors
or fixtures for this path and none of its three hot-path scripts can be imported
here (two are not valid Python >= 3.7: `.cuda(async=True)` at
generate_gp_training_data_imagenet.py:118; the third needs cv2/skimage/torchvision,
bayesian_active_learning_imagenet.py:1,10,32).  This file is therefore a CPU
restatement owned by this repo, composed of the same torch CPU ops the reference
reaches through `torchvision.models.<arch>(pretrained=True)`
(generate_gp_training_data_imagenet.py:579, bayesian_active_learning_imagenet.py:391)
in `model.eval()` mode (generate_gp_training_data_imagenet.py:159).  It is pinned
only by known answers (parameter counts, MAC counts, output shapes, softmax
identities -- see tests/test_oracle_resnet.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

Topology restated (torchvision `models/resnet.py`, un-vendored):
  conv7x7 s2 p3 (3->64, no bias) -> BN -> ReLU -> maxpool 3x3 s2 p1
  -> 4 stages of residual blocks -> global avg-pool -> fc (C->1000, bias).
  BasicBlock : conv3x3(s) BN ReLU conv3x3 BN (+identity | conv1x1(s) BN) ReLU
  Bottleneck : conv1x1 BN ReLU conv3x3(s) BN ReLU conv1x1(x4) BN (+identity | conv1x1(s) BN) ReLU
  ("v1.5": the stride sits on the 3x3).  BN eps = 1e-5, running statistics.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5

ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}
STAGE_WIDTH = (64, 128, 256, 512)


def conv_list(arch):
    """[(name, cin, cout, k, stride, pad, hin)] for every conv, in forward order.
    `name` is the torchvision state_dict prefix of the conv weight; the matching BN
    prefix is returned by bn_name(name)."""
    kind, depths = ARCHS[arch]
    exp = 1 if kind == "basic" else 4
    out = [("conv1", 3, 64, 7, 2, 3, 224)]
    cin, h = 64, 56
    for s, (w, d) in enumerate(zip(STAGE_WIDTH, depths)):
        for b in range(d):
            stride = 2 if (b == 0 and s > 0) else 1
            p = "layer%d.%d." % (s + 1, b)
            if kind == "basic":
                out.append((p + "conv1", cin, w, 3, stride, 1, h))
                out.append((p + "conv2", w, w, 3, 1, 1, h // stride))
            else:
                out.append((p + "conv1", cin, w, 1, 1, 0, h))
                out.append((p + "conv2", w, w, 3, stride, 1, h))
                out.append((p + "conv3", w, w * exp, 1, 1, 0, h // stride))
            if b == 0 and (stride != 1 or cin != w * exp):
                out.append((p + "downsample.0", cin, w * exp, 1, stride, 0, h))
            cin = w * exp
            h //= stride
    return out


def bn_name(conv_name):
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    return conv_name.replace("conv", "bn") if "." in conv_name else "bn1"


def feature_dim(arch):
    return 512 * (1 if ARCHS[arch][0] == "basic" else 4)


def state_dict_shapes(arch):
    """OrderedDict key -> shape, the torchvision key set (minus num_batches_tracked)."""
    sd = OrderedDict()
    for name, cin, cout, k, _s, _p, _h in conv_list(arch):
        sd[name + ".weight"] = (cout, cin, k, k)
        bn = bn_name(name)
        for f in ("weight", "bias", "running_mean", "running_var"):
            sd[bn + "." + f] = (cout,)
    sd["fc.weight"] = (1000, feature_dim(arch))
    sd["fc.bias"] = (1000,)
    return sd


def learnable_param_count(arch):
    n = 0
    for k, shp in state_dict_shapes(arch).items():
        if "running_" in k:
            continue
        c = 1
        for d in shp:
            c *= d
        n += c
    return n


def conv_macs(arch):
    """Multiply-accumulates of all convs for one 224x224 forward."""
    total = 0
    for _n, cin, cout, k, s, _p, h in conv_list(arch):
        ho = h // s
        total += ho * ho * cout * cin * k * k
    return total


def flops_per_forward(arch):
    return 2 * (conv_macs(arch) + 1000 * feature_dim(arch))


def _conv_bn(sd, x, name, stride, pad, relu):
    bn = bn_name(name)
    y = F.conv2d(x, sd[name + ".weight"], None, stride, pad)
    y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"],
                     sd[bn + ".weight"], sd[bn + ".bias"], False, 0.0, BN_EPS)
    return F.relu(y) if relu else y


def forward(sd, x, arch, taps=None):
    """logits[B,1000] = eval-mode forward.  `sd` tensors and `x` must share a dtype
    (float32 = reference-faithful, float64 = error yardstick).  If `taps` is a dict,
    intermediate activations (NCHW) are stored under their producer's name."""
    kind, depths = ARCHS[arch]
    exp = 1 if kind == "basic" else 4

    def tap(k, v):
        if taps is not None:
            taps[k] = v
        return v

    x = tap("conv1", _conv_bn(sd, x, "conv1", 2, 3, True))
    x = tap("maxpool", F.max_pool2d(x, 3, 2, 1))
    cin = 64
    for s, (w, d) in enumerate(zip(STAGE_WIDTH, depths)):
        for b in range(d):
            stride = 2 if (b == 0 and s > 0) else 1
            p = "layer%d.%d." % (s + 1, b)
            identity = x
            if kind == "basic":
                y = _conv_bn(sd, x, p + "conv1", stride, 1, True)
                y = _conv_bn(sd, y, p + "conv2", 1, 1, False)
            else:
                y = _conv_bn(sd, x, p + "conv1", 1, 0, True)
                y = _conv_bn(sd, y, p + "conv2", stride, 1, True)
                y = _conv_bn(sd, y, p + "conv3", 1, 0, False)
            if (p + "downsample.0.weight") in sd:
                identity = _conv_bn(sd, x, p + "downsample.0", stride, 0, False)
            x = tap(p[:-1], F.relu(y + identity))
            cin = w * exp
    x = tap("avgpool", F.adaptive_avg_pool2d(x, 1).flatten(1))
    return tap("fc", F.linear(x, sd["fc.weight"], sd["fc.bias"]))


def cast_state_dict(sd, dtype):
    return OrderedDict((k, v.to(dtype)) for k, v in sd.items())
