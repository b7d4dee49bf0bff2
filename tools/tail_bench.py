#!/usr/bin/env python3
"""Time one block tail (mpx_bottleneck_tail, csrc/mpx_btail.h) on random planes.  usage: python tools/tail_bench.py [k] [batch] [reps]
(tail 0 = layer1.0 runs as mpx_forward runs it: whole, its own conv1 inside the launch)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import _lib, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = MaskedForwardEngine("resnet101", max_batch=8, device=0).load_state_dict(synth.make_state_dict("resnet101"))
dev = eng.device
c2, c3, ds, n1 = eng.bottleneck_tails()[k]
c1 = eng.layers[n1].cout
gen = torch.Generator(device="cuda").manual_seed(0)
p = lambda t: C.c_void_p(t.data_ptr())


def planes(c):
    x = torch.randn(batch, 56, 56, c, device=dev, generator=gen).clamp_min(0)
    hi = x.half()
    return hi, (x - hi.float()).half()


th, tl = planes(64)
xh, xl = planes(64 if ds >= 0 else 256)
oh = torch.empty(batch, 56, 56, 256, dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
zh = torch.empty(batch, 56, 56, c1, dtype=torch.float16, device=dev); zl = torch.empty_like(zh)
whole = ds >= 0
run = lambda: _lib.check(eng._h, eng._lib.mpx_bottleneck_tail(eng._h, c2, None if whole else p(th), None if whole else p(tl), p(xh), p(xl), p(oh), p(ol), p(zh), p(zl), batch, None), "tail")
for _ in range(2):
    run()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    run()
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / reps
units = (1.43 + (0 if whole else 4) + 4 + c1 / 64) * 64 * 4 * batch * 56 * 56
print("tail %d (%s, next conv1 %d) batch %d: %.3f ms per launch; %.1f GB of HBM traffic by the plan = %.2f TB/s" % (
    k, "downsample branch" if ds >= 0 else "identity", c1, batch, ms, units / 1e9, units / ms / 1e9))
eng.close()
