#!/usr/bin/env python3
"""Does the DATA decide the speed of an MFMA-bound kernel at the package power limit?  (Round 5's timing-only builds stored NaN patterns and
every later layer ran 4 % faster.)  One 3x3 layer 256 -> 256 on 14x14 maps at batch 2340 (tile 12: MFMA pipe 75 % busy, package at the limit),
the same kernel and weights, pixel operands of different bit content, each looped ~1 s, the variants interleaved twice:
    relu      hi + lo of max(randn, 0): what the network feeds it (half the values are zero)
    lo0       the same hi plane, lo plane all zero
    lo8 / lo6 the lo plane with its 3 / 5 low mantissa bits cleared (8 / 6 significant bits: what oracle/precision_lo8.py's byte format keeps)
    dense     hi + lo of |randn| (no zeros)
    zero      both planes zero
usage: python tools/probes/data_power_probe.py [layer=layer3.5.conv2] [batch=2340] [reps=800]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import _lib, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

layer = sys.argv[1] if len(sys.argv) > 1 else "layer3.5.conv2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2340
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 800
dev = torch.device("cuda", 0)
eng = MaskedForwardEngine("resnet101", max_batch=8, device=0).load_state_dict(synth.make_state_dict("resnet101"))
i = [d.name.decode() for d in eng.layers].index(layer)
d = eng.layers[i]
gen = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(batch, d.hin, d.hin, d.cin, device=dev, generator=gen)


def split(v):
    hi = v.half()
    return hi, (v - hi.float()).half()


def clear_low_bits(t, n):
    return (t.view(torch.int16) & ~((1 << n) - 1)).view(torch.float16)


r_hi, r_lo = split(x.clamp_min(0))
d_hi, d_lo = split(x.abs())
variants = {
    "relu": (r_hi, r_lo),
    "lo0": (r_hi, torch.zeros_like(r_lo)),
    "lo8": (r_hi, clear_low_bits(r_lo, 3)),
    "lo6": (r_hi, clear_low_bits(r_lo, 5)),
    "dense": (d_hi, d_lo),
    "zero": (torch.zeros_like(r_hi), torch.zeros_like(r_lo)),
}
rh = rl = None
if d.residual:
    rh, rl = split(torch.randn(batch, d.hout, d.hout, d.cout, device=dev, generator=gen))
oh = torch.empty(batch, d.hout, d.hout, d.cout, dtype=torch.float16, device=dev)
ol = torch.empty_like(oh)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def loop(xh, xl, n):
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, i, p(xh), p(xl), p(rh), p(rl), p(oh), p(ol), None, batch, None), "conv")
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


print("%s %d->%d k%d out%d, batch %d, tile %d, %d launches per sample" % (layer, d.cin, d.cout, d.ksize, d.hout, batch, eng.conv_tile(i), reps))
loop(*variants["relu"], 200)
res = {k: [] for k in variants}
for _ in range(2):
    for k, (xh, xl) in variants.items():
        res[k].append(loop(xh, xl, reps))
base = sum(res["relu"]) / 2
for k, v in res.items():
    print("  %-6s %.4f / %.4f ms   (%+.1f %% against relu)" % (k, v[0], v[1], (sum(v) / 2 / base - 1) * 100))
