#!/usr/bin/env python3
"""Determinism / race stress of the block-tail kernel (GPU box): every tail, a sweep of batch sizes (one tile per workgroup up to
several, ragged grids), each launched repeatedly -- the outputs must be BIT-identical run to run (a stale LDS read or a wait that is
one instruction short shows up as a flicker) and within tolerance of the layer-by-layer kernels on the same inputs."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import _lib, synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

eng = MaskedForwardEngine("resnet101", max_batch=8, device=0).load_state_dict(synth.make_state_dict("resnet101"))
dev = eng.device
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
gen = torch.Generator(device="cuda").manual_seed(1)


def planes(b, c):
    x = torch.randn(b, 56, 56, c, device=dev, generator=gen).clamp_min(0) * 1.5
    hi = x.half()
    return hi, (x - hi.float()).half()


worst = 0.0
# uneven load: a side stream keeps HBM busy with large copies while the tails run (memory latency then varies from wave to wave)
side = torch.cuda.Stream(device=dev)
big_a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
big_b = torch.empty_like(big_a)
for k, (c2, c3, ds, n1) in enumerate(eng.bottleneck_tails()):
    c1 = eng.layers[n1].cout
    for batch in (1, 2, 3, 5, 8, 13, 19, 20, 37, 64, 100, 147):
        th, tl = planes(batch, 64)
        xh, xl = planes(batch, 64 if ds >= 0 else 256)
        outs = []
        for rep in range(4):
            oh = torch.full((batch, 56, 56, 256), float("nan"), dtype=torch.float16, device=dev)
            ol, zh = torch.full_like(oh, float("nan")), torch.full((batch, 56, 56, c1), float("nan"), dtype=torch.float16, device=dev)
            zl = torch.full_like(zh, float("nan"))
            torch.cuda.synchronize()
            if rep >= 2:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        big_b.copy_(big_a, non_blocking=True)
            _lib.check(eng._h, eng._lib.mpx_bottleneck_tail(eng._h, c2, p(th), p(tl), p(xh), p(xl), p(oh), p(ol), p(zh), p(zl), batch, None), "tail")
            torch.cuda.synchronize()
            outs.append((oh, ol, zh, zl))
        for o in outs[1:]:
            assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(outs[0], o)), "tail %d batch %d: run-to-run difference" % (k, batch)
        # layer by layer on the same inputs: conv2, conv3 (+ identity / downsample), next conv1
        t2h = torch.empty(batch, 56, 56, 64, dtype=torch.float16, device=dev); t2l = torch.empty_like(t2h)
        _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, c2, p(th), p(tl), None, None, p(t2h), p(t2l), None, batch, None), "conv2")
        rh = torch.empty(batch, 56, 56, 256, dtype=torch.float16, device=dev); rl = torch.empty_like(rh)
        if ds >= 0:
            _lib.check(eng._h, eng._lib.mpx_conv_dual_bn_act(eng._h, c3, p(t2h), p(t2l), p(xh), p(xl), p(rh), p(rl), batch, None), "dual")
        else:
            _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, c3, p(t2h), p(t2l), p(xh), p(xl), p(rh), p(rl), None, batch, None), "conv3")
        qh = torch.empty(batch, 56, 56, c1, dtype=torch.float16, device=dev); ql = torch.empty_like(qh)
        _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, n1, p(rh), p(rl), None, None, p(qh), p(ql), None, batch, None), "conv1")
        torch.cuda.synchronize()
        oh, ol, zh, zl = outs[0]
        if ds >= 0:
            # the whole block in one launch (t1 = NULL: conv1 runs on the patch of the block input) against conv1 -> the same tail
            t1h = torch.empty(batch, 56, 56, 64, dtype=torch.float16, device=dev); t1l = torch.empty_like(t1h)
            _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, c2 - 1, p(xh), p(xl), None, None, p(t1h), p(t1l), None, batch, None), "conv1")
            a_ = [torch.full_like(t, float("nan")) for t in (oh, ol, zh, zl)]
            b_ = [torch.full_like(t, float("nan")) for t in (oh, ol, zh, zl)]
            _lib.check(eng._h, eng._lib.mpx_bottleneck_tail(eng._h, c2, p(t1h), p(t1l), p(xh), p(xl), *[p(t) for t in a_], batch, None), "tail")
            for rep in range(3):
                _lib.check(eng._h, eng._lib.mpx_bottleneck_tail(eng._h, c2, None, None, p(xh), p(xl), *[p(t) for t in b_], batch, None), "whole")
                torch.cuda.synchronize()
                for u, v in ((a_[0].float() + a_[1].float(), b_[0].float() + b_[1].float()), (a_[2].float() + a_[3].float(), b_[2].float() + b_[3].float())):
                    assert not torch.isnan(v).any()
                    worst = max(worst, float((u - v).abs().max() / u.abs().max().clamp_min(1.0)))
        for got, want in (((oh, ol), (rh, rl)), ((zh, zl), (qh, ql))):
            a, b = got[0].float() + got[1].float(), want[0].float() + want[1].float()
            assert not torch.isnan(a).any()
            worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1.0)))
    print("tail %d: 12 batch sizes x 4 launches (two of them next to a 1-GiB copy stream) bit-identical; worst relative difference to the layer-by-layer kernels so far %.2e" % (k, worst))
assert worst <= 4e-6
eng.close()
