#!/usr/bin/env python3
"""Where does the pixel-stationary kernel (tile 11, tools/probes/experimental/mpx_convs.h) spend a workgroup's life?  -DMPX_DIAG build (never the product
library): per workgroup the cycles spent in the per-step rendezvous (vmcnt wait + barrier), waiting for the epilogue loads, in the
epilogue itself, and waiting for the next pixel tile.   usage: python tools/probes/convs_phases.py [layer] [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import conv_timeline as ct  # noqa: E402  (build_diag)

ct.build_diag(["-DMPX_EXPERIMENTAL"])      # tile 11 lives in tools/probes/experimental/ (never a default: DESIGN.md 5)
from network_interpretation_imagenet_amd import _lib, synth  # noqa: E402
_lib.LIB_PATH = ct.DIAG
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

layer = sys.argv[1] if len(sys.argv) > 1 else "layer3.5.conv3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda", 0)
eng = MaskedForwardEngine("resnet101", max_batch=8, device=0).load_state_dict(synth.make_state_dict("resnet101"))
lib = eng._lib
lib.mpx_debug_set_stamps.restype = C.c_int
lib.mpx_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
i = [d.name.decode() for d in eng.layers].index(layer)
d = eng.layers[i]
gen = torch.Generator(device="cuda").manual_seed(0)


def planes(*shape):
    x = torch.randn(*shape, device=dev, generator=gen).clamp_min(0)
    hi = x.half()
    return hi, (x - hi.float()).half()


xh, xl = planes(batch, d.hin, d.hin, d.cin)
rh, rl = planes(batch, d.hout, d.hout, d.cout)
oh = torch.empty(batch, d.hout, d.hout, d.cout, dtype=torch.float16, device=dev)
ol = torch.empty_like(oh)
p = lambda t: C.c_void_p(t.data_ptr())
stamps = torch.zeros(4 * 4096 * 8, dtype=torch.int64, device=dev)
eng.set_conv_tile(i, 11)
run = lambda: _lib.check(eng._h, lib.mpx_conv_bn_act(eng._h, i, p(xh), p(xl), p(rh), p(rl), p(oh), p(ol), None, batch, None), "conv")
for _ in range(3):
    run()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5):
    run()
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 5
lib.mpx_debug_set_stamps(eng._h, C.c_void_p(stamps.data_ptr()))
run()
torch.cuda.synchronize()
allst = stamps.cpu().numpy().reshape(-1, 8)
st = allst[:4096]
keep = st[:, 7] != 0
wbar = allst[8192:12288][keep]
seg = allst[12288:][keep]
st = st[keep]
life = st[:, 1].astype(np.float64)
rv, ew, epi, xw = st[:, 2], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3], st[:, 5] - st[:, 4]
print("%s B=%d tile 11: %.3f ms; %d workgroups, %d..%d pixel tiles each; life median %.0f cycles (-> ~%.2f GHz)" % (
    layer, batch, ms, len(st), st[:, 7].min(), st[:, 7].max(), np.median(life), np.median(life) / (ms * 1e6)))
for name, v in (("rendezvous (vmcnt + barrier), all steps", rv), ("wait for the epilogue loads", ew), ("epilogue arithmetic + store issue", epi),
                ("wait for the next pixel tile", xw)):
    print("   %-44s %5.1f %% of the life   (%.0f cycles per pixel tile)" % (name, 100 * np.median(v / life), np.median(v / st[:, 7])))
nsteps = st[:, 7] * (d.cout // 256) * (d.cin // 32)
nct = st[:, 7] * (d.cout // 256)
print("   of the rendezvous: barrier part %.1f %% of the life" % (100 * np.median(st[:, 6] / life)))
print("   barrier wait per K step, by wave: " + "  ".join("%d: %.0f" % (w, np.median(wbar[:, w] / nsteps)) for w in range(8)))
for w, o in ((0, 0), (4, 4)):
    print("   wave %d, cycles per K step: first half (18 MFMAs, 4 reads) %.0f | second half %.0f, of which its first 13 MFMAs with the 4 DMA pieces %.0f" % (
        w, np.median(seg[:, o] / nsteps), np.median(seg[:, o + 2] / nsteps), np.median(seg[:, o + 1] / nsteps)))
print("   cycles per K step, everything included: %.0f; of which in the rendezvous %.0f" % (np.median(life / nsteps), np.median(rv / nsteps)))
eng.close()
