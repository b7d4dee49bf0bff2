#!/usr/bin/env python3
"""Per-layer check and timing of the f16f8 conv kernel through mpx_conv_bn_act (GPU box).
usage: python tools/f8_conv_check.py [arch] [batch] [layer ...]"""
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

LO_SCALE = 2048.0


def to_planes(x):
    """f32[M..., C] -> (hi f16, byte plane u8[..., C/32, 64]) in the f16f8 activation format"""
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * LO_SCALE).clamp(-448, 448).to(torch.float8_e4m3fn)
    h8 = hi.float().clamp(-448, 448).to(torch.float8_e4m3fn)
    shp = x.shape[:-1] + (x.shape[-1] // 32, 32)
    p8 = torch.cat([lo.view(torch.uint8).reshape(shp), h8.view(torch.uint8).reshape(shp)], dim=-1)
    return hi.contiguous(), p8.contiguous()


def from_planes(hi, p8):
    c = hi.shape[-1]
    l8 = p8[..., :32].contiguous().view(torch.float8_e4m3fn).float().reshape(hi.shape[:-1] + (c,))
    return hi.float() + l8 / LO_SCALE


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    names = sys.argv[3:] or ["layer1.0.conv1", "layer1.0.conv2", "layer1.0.conv3", "layer2.0.conv2", "layer3.5.conv2", "layer3.5.conv3", "layer4.2.conv3"]
    sd = synth.make_state_dict(arch)
    dev = torch.device("cuda", 0)
    worst = 0.0
    for precision in ("f16f8", "f16x3"):
        eng = MaskedForwardEngine(arch, max_batch=batch, device=0, precision=precision).load_state_dict(sd)
        for name in names:
            i = [d.name.decode() for d in eng.layers].index(name)
            d = eng.layers[i]
            gen = torch.Generator().manual_seed(i)
            x = torch.randn(batch, d.hin, d.hin, d.cin, generator=gen).clamp_min(-0.5) * 1.5
            res = torch.randn(batch, d.hout, d.hout, d.cout, generator=gen) if d.residual else None
            if precision == "f16f8":
                xh, x8 = to_planes(x.to(dev))
                rh, r8 = to_planes(res.to(dev)) if res is not None else (None, None)
                x_used = from_planes(xh, x8).cpu()
                res_used = from_planes(rh, r8).cpu() if res is not None else None
                oh = torch.zeros(batch, d.hout, d.hout, d.cout, dtype=torch.float16, device=dev)
                o8 = torch.zeros(batch, d.hout, d.hout, d.cout // 32, 64, dtype=torch.uint8, device=dev)
            else:
                xh = x.to(dev).half(); x8 = (x.to(dev) - xh.float()).half()
                rh = res.to(dev).half() if res is not None else None
                r8 = (res.to(dev) - rh.float()).half() if res is not None else None
                x_used = (xh.float() + x8.float()).cpu()
                res_used = (rh.float() + r8.float()).cpu() if res is not None else None
                oh = torch.zeros(batch, d.hout, d.hout, d.cout, dtype=torch.float16, device=dev)
                o8 = torch.zeros_like(oh)
            st = eng._stream()
            for _ in range(2):
                rc = eng._lib.mpx_conv_bn_act(eng._h, i, ptr(xh), ptr(x8), ptr(rh), ptr(r8), ptr(oh), ptr(o8), None, batch, st)
                _lib.check(eng._h, rc, "mpx_conv_bn_act")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nrep = int(os.environ.get("F8_SUSTAIN", "5"))       # thousands of launches: steady-state (power-limited) timing
            e0.record()
            for _ in range(nrep):
                eng._lib.mpx_conv_bn_act(eng._h, i, ptr(xh), ptr(x8), ptr(rh), ptr(r8), ptr(oh), ptr(o8), None, batch, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / nrep
            if os.environ.get("F8_NOREF"):
                fl = 2.0 * batch * d.hout * d.hout * d.cout * d.cin * d.ksize * d.ksize
                print("%-6s %-22s %4d->%-4d k%d B=%d: %.4f ms  %.1f TFLOP/s" % (precision, name, d.cin, d.cout, d.ksize, batch, ms, fl / ms / 1e9))
                continue
            got = (from_planes(oh, o8) if precision == "f16f8" else oh.float() + o8.float()).cpu().double()
            bn = d.bn_name.decode()
            w = sd[name + ".weight"].double()
            y = F.conv2d(x_used.double().permute(0, 3, 1, 2), w, None, d.stride, d.pad)
            s = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
            y = (y - sd[bn + ".running_mean"].double().view(1, -1, 1, 1)) * s.view(1, -1, 1, 1) + sd[bn + ".bias"].double().view(1, -1, 1, 1)
            if res_used is not None:
                y = y + res_used.double().permute(0, 3, 1, 2)
            if d.relu:
                y = F.relu(y)
            want = y.permute(0, 2, 3, 1)
            err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
            rms = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
            fl = 2.0 * batch * d.hout * d.hout * d.cout * d.cin * d.ksize * d.ksize
            print("%-6s %-22s %4d->%-4d k%d B=%d: max rel err %.2e  rms rel %.2e   %.4f ms  %.1f TFLOP/s" % (
                precision, name, d.cin, d.cout, d.ksize, batch, err, rms, ms, fl / ms / 1e9))
            if precision == "f16f8":
                worst = max(worst, err)
        eng.close()
    print("worst f16f8 max rel err %.2e" % worst)


if __name__ == "__main__":
    main()
