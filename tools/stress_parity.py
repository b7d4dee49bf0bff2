#!/usr/bin/env python3
"""Stress sweep (GPU box): conv layers x every kernel tile the layer is eligible for x ragged random batches, against torch CPU fp64.
    python tools/stress_parity.py [arch] [seed]        every layer of the arch, one random batch each (the round-2 log)
tests/test_gpu_parity.py::test_stress_sweep_distinct_shapes runs sweep() on the distinct layer shapes with three batches each."""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TILE_IDS = (0, 1, 2, 4, 6, 7, 9, 10, 12, 13, 14)          # what mpx_set_conv_tile accepts in the product build


def distinct_shape_layers(eng):
    """One layer index per distinct (cin, cout, k, stride, hin, residual) of the engine's conv stack (stem and fc excluded)."""
    seen, out = set(), []
    for i, d in enumerate(eng.layers):
        key = (d.cin, d.cout, d.ksize, d.stride, d.hin, d.residual)
        if i == 0 or d.name == b"fc" or key in seen:
            continue
        seen.add(key)
        out.append(i)
    return out


def sweep(eng, sd, layers, batches, seed=0, tol=4e-6):
    """-> (cases, worst relative error).  For every layer index and every batch size: random split-fp16 inputs, the fp64 conv + BN
    (+ residual) (+ ReLU) once on the CPU, then EVERY tile id the layer accepts through mpx_conv_bn_act."""
    from network_interpretation_imagenet_amd import _lib
    dev = eng.device
    gen = torch.Generator().manual_seed(seed)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    worst, n = 0.0, 0
    for i in layers:
        d = eng.layers[i]
        name, bn = d.name.decode(), d.bn_name.decode()
        sc = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
        for batch in (batches(gen) if callable(batches) else batches):
            x = torch.randn(batch, d.hin, d.hin, d.cin, generator=gen).clamp_min(-0.3) * 1.3
            res = torch.randn(batch, d.hout, d.hout, d.cout, generator=gen) if d.residual else None
            xh = x.half(); xl = (x - xh.float()).half()
            xu = (xh.float() + xl.float()).double()
            rh = rl = ru = None
            if res is not None:
                rh = res.half(); rl = (res - rh.float()).half(); ru = (rh.float() + rl.float()).double()
            y = F.conv2d(xu.permute(0, 3, 1, 2), sd[name + ".weight"].double(), None, d.stride, d.pad)
            y = (y - sd[bn + ".running_mean"].double().view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + sd[bn + ".bias"].double().view(1, -1, 1, 1)
            if ru is not None:
                y = y + ru.permute(0, 3, 1, 2)
            if d.relu:
                y = F.relu(y)
            want = y.permute(0, 2, 3, 1)
            scale = max(float(want.abs().max()), 1.0)
            dxh, dxl = xh.to(dev), xl.to(dev)
            drh, drl = (rh.to(dev), rl.to(dev)) if rh is not None else (None, None)
            for tile in TILE_IDS:
                if eng._lib.mpx_set_conv_tile(eng._h, i, tile) != 0:
                    continue                                # patch / 256x256 / persistent kernels: eligible layers only
                oh = torch.full((batch, d.hout, d.hout, d.cout), float("nan"), dtype=torch.float16, device=dev)
                ol = torch.full_like(oh, float("nan"))
                _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, i, p(dxh), p(dxl), p(drh), p(drl), p(oh), p(ol), None, batch, None), name)
                torch.cuda.synchronize()
                got = (oh.float() + ol.float()).cpu().double()
                err = float((got - want).abs().max()) / scale
                assert err == err and err < tol, "%s tile %d batch %d: rel err %g" % (name, tile, batch, err)
                worst = max(worst, err)
                n += 1
        eng.set_conv_tile(i, -1)
    return n, worst


if __name__ == "__main__":
    import __graft_entry__ as g
    g.build()
    from network_interpretation_imagenet_amd import synth
    from network_interpretation_imagenet_amd.engine import MaskedForwardEngine
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sd = synth.make_state_dict(arch)
    eng = MaskedForwardEngine(arch, max_batch=16, device=0).load_state_dict(sd)
    layers = [i for i, d in enumerate(eng.layers) if i and d.name != b"fc"]
    n, worst = sweep(eng, sd, layers, lambda gen: [int(torch.randint(1, 14, (1,), generator=gen))], seed)
    print("%s: %d (layer, tile) cases, worst relative error %.2e" % (arch, n, worst))
