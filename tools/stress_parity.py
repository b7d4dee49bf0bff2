#!/usr/bin/env python3
"""One-off stress sweep (GPU box): EVERY conv layer of an arch, every kernel tile variant, ragged random batches,
against torch CPU fp64.  usage: python tools/stress_parity.py [arch] [seed]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import _lib, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sd = synth.make_state_dict(arch)
eng = MaskedForwardEngine(arch, max_batch=16, device=0).load_state_dict(sd)
dev = eng.device
gen = torch.Generator().manual_seed(seed)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
worst = 0.0
n = 0
for i, d in enumerate(eng.layers):
    name, bn = d.name.decode(), d.bn_name.decode()
    if i == 0 or name == "fc":
        continue
    batch = int(torch.randint(1, 14, (1,), generator=gen))
    x = torch.randn(batch, d.hin, d.hin, d.cin, generator=gen).clamp_min(-0.3) * 1.3
    res = torch.randn(batch, d.hout, d.hout, d.cout, generator=gen) if d.residual else None
    xh = x.half(); xl = (x - xh.float()).half()
    xu = (xh.float() + xl.float()).double()
    rh = rl = None
    ru = None
    if res is not None:
        rh = res.half(); rl = (res - rh.float()).half(); ru = (rh.float() + rl.float()).double()
    y = F.conv2d(xu.permute(0, 3, 1, 2), sd[name + ".weight"].double(), None, d.stride, d.pad)
    sc = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
    y = (y - sd[bn + ".running_mean"].double().view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + sd[bn + ".bias"].double().view(1, -1, 1, 1)
    if ru is not None:
        y = y + ru.permute(0, 3, 1, 2)
    if d.relu:
        y = F.relu(y)
    want = y.permute(0, 2, 3, 1)
    dxh, dxl = xh.to(dev), xl.to(dev)
    drh, drl = (rh.to(dev), rl.to(dev)) if rh is not None else (None, None)
    for tile in range(12):
        if eng._lib.mpx_set_conv_tile(eng._h, i, tile) != 0:
            continue                                    # patch / 256x256 / persistent kernels: eligible layers only
        oh = torch.full((batch, d.hout, d.hout, d.cout), float("nan"), dtype=torch.float16, device=dev)
        ol = torch.full_like(oh, float("nan"))
        _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, i, p(dxh), p(dxl), p(drh), p(drl), p(oh), p(ol), None, batch, None), name)
        torch.cuda.synchronize()
        got = (oh.float() + ol.float()).cpu().double()
        err = float((got - want).abs().max()) / max(float(want.abs().max()), 1.0)
        assert err == err and err < 4e-6, "%s tile %d batch %d: rel err %g" % (name, tile, batch, err)
        worst = max(worst, err)
        n += 1
    eng.set_conv_tile(i, -1)
print("%s: %d (layer, tile) cases, worst relative error %.2e" % (arch, n, worst))
