"""Diagnostic: where does mpx_bottleneck_tail differ from the fp64 chain?  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
import __graft_entry__ as g
g.build()
from network_interpretation_imagenet_amd import synth, _lib
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine
sys.path.insert(0, os.path.join(g.ROOT, "tests"))
import test_gpu_parity as T

eng = MaskedForwardEngine("resnet101", max_batch=8, device=0).load_state_dict(synth.make_state_dict("resnet101"))
sd = synth.make_state_dict("resnet101")
for k, batch in [(1, 1), (0, 1), (2, 1)]:
    c2, c3, ds, n1 = eng.bottleneck_tails()[k]
    d2, d3, dn = eng.layers[c2], eng.layers[c3], eng.layers[n1]
    gen = torch.Generator().manual_seed(1)
    t1 = torch.randn(batch, 56, 56, 64, generator=gen).clamp_min(0) * 1.5
    x = torch.randn(batch, 56, 56, 64 if ds >= 0 else 256, generator=gen).clamp_min(0) * 1.5
    dev = eng.device
    th, tl = T.split(t1.to(dev)); xh, xl = T.split(x.to(dev))
    nan = lambda c: torch.full((batch, 56, 56, c), float("nan"), dtype=torch.float16, device=dev)
    oh, ol, zh, zl = nan(256), nan(256), nan(dn.cout), nan(dn.cout)
    rc = eng._lib.mpx_bottleneck_tail(eng._h, c2, T._p(th), T._p(tl), T._p(xh), T._p(xl), T._p(oh), T._p(ol), T._p(zh), T._p(zl), batch, eng._stream())
    _lib.check(eng._h, rc, "tail"); torch.cuda.synchronize()
    t1u = T.merge(th, tl).cpu().double().permute(0, 3, 1, 2); xu = T.merge(xh, xl).cpu().double().permute(0, 3, 1, 2)
    t2 = T._round_split(F.relu(T._conv_bn_fp64(sd, d2, t1u)))
    ident = T._conv_bn_fp64(sd, eng.layers[ds], xu) if ds >= 0 else xu
    out = F.relu(T._conv_bn_fp64(sd, d3, t2) + ident)
    z = F.relu(T._conv_bn_fp64(sd, dn, T._round_split(out)))
    go = T.merge(oh, ol).cpu().double().permute(0, 3, 1, 2); gz = T.merge(zh, zl).cpu().double().permute(0, 3, 1, 2)
    for name, got, want in (("out", go, out), ("z", gz, z)):
        e = (got - want).abs()
        print("tail", k, name, "nan", int(torch.isnan(got).sum()), "max err %.3e scale %.2f" % (torch.nan_to_num(e).max().item(), want.abs().max().item()))
        e = torch.nan_to_num(e, nan=9.0)
        ey = e.amax(dim=(0, 1, 3)).view(7, 8).amax(0); ex = e.amax(dim=(0, 1, 2)).view(4, 14).amax(0)
        print("  by row in tile:", np.array2string(ey.numpy(), precision=1))
        print("  by col in tile:", np.array2string(ex.numpy(), precision=1))
        ec = e.amax(dim=(0, 2, 3)); print("  by channel/16:", np.array2string(ec.view(-1, 16).amax(1).numpy(), precision=1))
        print("  by channel%16:", np.array2string(ec.view(-1, 16).amax(0).numpy(), precision=1))
        print("  by image row:", np.array2string(e.amax(dim=(0, 1, 3)).numpy()[:16], precision=1))
