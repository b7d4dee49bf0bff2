#!/usr/bin/env python3
"""Time ONE conv layer through the C-ABI (mpx_conv_bn_act) with random split-fp16 planes.
usage: python tools/conv_bench.py [arch] [layer-name] [batch] [reps] [tile,...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import _lib, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
layer = sys.argv[2] if len(sys.argv) > 2 else "layer3.5.conv2"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
eng = MaskedForwardEngine(arch, max_batch=8, device=0).load_state_dict(synth.make_state_dict(arch))
i = [d.name.decode() for d in eng.layers].index(layer)
d = eng.layers[i]
gen = torch.Generator(device="cuda").manual_seed(0)


def planes(*shape):
    x = torch.randn(*shape, device=dev, generator=gen).clamp_min(0)
    hi = x.half()
    return hi, (x - hi.float()).half()


stem = (i == 0)
eng2 = None
if stem:       # the stem reads the engine's own padded staging: fill it through K0 and use a matching engine
    eng.close()
    eng = MaskedForwardEngine(arch, max_batch=batch, device=0).load_state_dict(synth.make_state_dict(arch))
    img = torch.from_numpy(synth.make_images(1, kind="noise")[0]).to(dev)
    seg = torch.from_numpy(synth.grid_segments()).to(dev)
    onoff = torch.from_numpy(synth.random_onoff(batch, 196)).to(dev)
    eng.stage_masks(img, seg, onoff, 0)
    xh = xl = None
else:
    xh, xl = planes(batch, d.hin, d.hin, d.cin)
rh, rl = planes(batch, d.hout, d.hout, d.cout) if d.residual else (None, None)
oh = torch.empty(batch, d.hout, d.hout, d.cout, dtype=torch.float16, device=dev)
ol = torch.empty_like(oh)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def run():
    _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, i, p(xh), p(xl), p(rh), p(rl), p(oh), p(ol), None, batch, None), "conv")


tiles = [int(t) for t in sys.argv[5].split(",")] if len(sys.argv) > 5 else [-1]
fl = 2.0 * batch * d.hout * d.hout * d.cout * d.cin * d.ksize * d.ksize
for tile in tiles:
    eng.set_conv_tile(i, tile)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        run()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    print("%s %s B=%d tile=%d: %d->%d k%d s%d out%d res=%d  %.4f ms  %.1f TFLOP/s algorithmic (x3 issued = %.0f)" % (
        arch, layer, batch, eng.conv_tile(i), d.cin, d.cout, d.ksize, d.stride, d.hout, d.residual, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))
