#!/usr/bin/env python3
"""Determinism / race stress of the persistent kernels (GPU box): tile 12 (csrc/mpx_conv3pp.h) against tile 6, tile 13
(csrc/mpx_conv256p.h) against tile 9 and tile 14 (csrc/mpx_convw.h, with and without a residual operand) against tile 10 on the layers
they are defaults for, a sweep of batch sizes (one tile per workgroup up to
fourteen, ragged grids), each launched repeatedly -- half of the launches next to a side stream that keeps HBM busy with 1-GiB copies
(memory latency then varies from wave to wave).  Every output must be BIT-identical to the non-persistent kernel's: a counted vmcnt
that is one instruction short, or a stale LDS read at a tile boundary, shows up as a difference."""
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
gen = torch.Generator(device="cuda").manual_seed(3)
names = [d.name.decode() for d in eng.layers]
side = torch.cuda.Stream(device=dev)
big_a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
big_b = torch.empty_like(big_a)


def run(i, tile, xh, xl, batch, busy, rh=None, rl=None):
    d = eng.layers[i]
    oh = torch.full((batch, d.hout, d.hout, d.cout), float("nan"), dtype=torch.float16, device=dev)
    ol = torch.full_like(oh, float("nan"))
    eng.set_conv_tile(i, tile)
    torch.cuda.synchronize()
    if busy:
        with torch.cuda.stream(side):
            for _ in range(2):
                big_b.copy_(big_a, non_blocking=True)
    _lib.check(eng._h, eng._lib.mpx_conv_bn_act(eng._h, i, p(xh), p(xl), p(rh), p(rl), p(oh), p(ol), None, batch, None), "conv")
    torch.cuda.synchronize()
    eng.set_conv_tile(i, -1)
    return oh, ol


cases = 0
for name, ref_tile, tile, batches in (("layer3.5.conv3", 10, 14, (11, 21, 47, 84, 161, 335, 700, 1003, 2006, 2340)),
                                      ("layer3.22.conv3", 10, 14, (-85, -1171)),        # negative: no residual operand
                                      ("layer3.5.conv2", 6, 12, (41, 83, 335, 700, 1003, 2006, 2340)),
                                      ("layer2.1.conv2", 6, 12, (11, 21, 84, 335, 1171, 2340)),
                                      ("layer3.5.conv1", 9, 13, (41, 83, 335, 700, 1003, 2006, 2340)),
                                      ("layer2.0.conv1" if "layer2.0.conv1" in names else "layer3.0.conv1", 9, 13, (5, 21, 84)),
                                      ("layer4.1.conv1", 9, 13, (84, 335, 1339, 2340))):
    i = names.index(name)
    d = eng.layers[i]
    if eng._lib.mpx_set_conv_tile(eng._h, i, tile) != 0:          # not eligible (e.g. the 256 -> 128 conv of a tail): nothing to compare
        print("%s: tile %d not eligible, skipped" % (name, tile))
        continue
    eng.set_conv_tile(i, -1)
    for batch in batches:
        with_res = bool(d.residual) and batch > 0
        batch = abs(batch)
        x = torch.randn(batch, d.hin, d.hin, d.cin, device=dev, generator=gen).clamp_min(-0.5) * 1.5
        xh = x.half()
        xl = (x - xh.float()).half()
        del x
        rh = rl = None
        if with_res:
            r = torch.randn(batch, d.hout, d.hout, d.cout, device=dev, generator=gen)
            rh = r.half()
            rl = (r - rh.float()).half()
            del r
        want = run(i, ref_tile, xh, xl, batch, False, rh, rl)
        for rep in range(4):
            got = run(i, tile, xh, xl, batch, rep >= 2, rh, rl)
            assert not torch.isnan(got[0].float()).any(), "%s batch %d: unwritten output" % (name, batch)
            assert torch.equal(got[0].view(torch.int16), want[0].view(torch.int16)) and torch.equal(got[1].view(torch.int16), want[1].view(torch.int16)), \
                "%s tile %d batch %d launch %d: differs from tile %d" % (name, tile, batch, rep, ref_tile)
            cases += 1
        del xh, xl, rh, rl, want, got
        torch.cuda.empty_cache()
    print("%s (%d -> %d, %dx%d): tile %d bit-identical to tile %d on batches %s, 4 launches each (2 next to the copy stream)"
          % (name, d.cin, d.cout, d.hout, d.hout, tile, ref_tile, [abs(b) for b in batches]))
print("persistent-kernel stress: %d launches compared, all bit-identical" % cases)
