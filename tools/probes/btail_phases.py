#!/usr/bin/env python3
"""Where does a block-tail workgroup (csrc/mpx_btail.h) spend its time?  -DMPX_DIAG build (never the product .so): wave 0 of
every workgroup adds up the s_memtime cycles of each phase over its tiles.
usage: python tools/probes/btail_phases.py [batch]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
DIAG = os.path.join(HERE, "libmpx_diag.so")


def build_diag():
    """MPX_DIAG_FLAGS="-DBT_AUX_STORE=0 ..." adds compile flags (probe variants of the kernel)"""
    sys.path.insert(0, HERE)
    import conv_timeline
    conv_timeline.build_diag(os.environ.get("MPX_DIAG_FLAGS", "").split())


if __name__ == "__main__":
    build_diag()
    if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
        sys.exit(0)
    from network_interpretation_imagenet_amd import _lib, synth
    _lib.LIB_PATH = DIAG
    from network_interpretation_imagenet_amd.engine import MaskedForwardEngine
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda", 0)
    eng = MaskedForwardEngine("resnet101", max_batch=8, device=0).load_state_dict(synth.make_state_dict("resnet101"))
    lib = eng._lib
    lib.mpx_debug_set_stamps.restype = C.c_int
    lib.mpx_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    gen = torch.Generator(device="cuda").manual_seed(0)

    def planes(c):
        x = torch.randn(batch, 56, 56, c, device=dev, generator=gen).clamp_min(0)
        hi = x.half()
        return hi, (x - hi.float()).half()

    p = lambda t: C.c_void_p(t.data_ptr())
    th, tl = planes(64)
    stamps = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
    names = ["patch wait", "conv2 (18 steps)", "conv3 steps", "chunk epilogues", "conv1' steps", "t1' epilogue + end barrier"]
    for k, (c2, c3, ds, n1) in enumerate(eng.bottleneck_tails()):
        xh, xl = planes(64 if ds >= 0 else 256)
        oh = torch.empty(batch, 56, 56, 256, dtype=torch.float16, device=dev)
        ol = torch.empty_like(oh)
        c1 = eng.layers[n1].cout
        zh = torch.empty(batch, 56, 56, c1, dtype=torch.float16, device=dev)
        zl = torch.empty_like(zh)
        run = lambda: _lib.check(eng._h, lib.mpx_bottleneck_tail(eng._h, c2, p(th), p(tl), p(xh), p(xl), p(oh), p(ol), p(zh), p(zl), batch, None), "tail")
        lib.mpx_debug_set_stamps(eng._h, None)
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
        stamps.zero_()
        lib.mpx_debug_set_stamps(eng._h, p(stamps))
        run()
        torch.cuda.synchronize()
        lib.mpx_debug_set_stamps(eng._h, None)
        raw = stamps.cpu().numpy().reshape(-1, 8)
        nwg = int((raw[:, 7] != 0).sum())
        sub = raw[nwg:2 * nwg, :5].astype(np.float64)
        st = raw[:nwg].astype(np.float64)
        ghz = np.median(st[:, 6]) / (ms * 1e6)
        tiles = st[:, 7]
        print("== tail %d (%s, C1 = %d) B=%d: kernel %.3f ms, %d workgroups x %.0f tiles, clock ~%.2f GHz, %.1f us per tile" % (
            k, "downsample branch" if ds >= 0 else "identity", c1, batch, ms, len(st), np.median(tiles), ghz, np.median(st[:, 6] / tiles) / ghz / 1e3))
        for i, nm in enumerate(names):
            per = st[:, i] / tiles / ghz / 1e3
            print("   %-30s %6.2f us per tile (p10 %5.2f  p90 %5.2f)  %5.1f %%" % (nm, np.median(per), *np.percentile(per, [10, 90]), 100 * st[:, i].sum() / st[:, 6].sum()))
        for i, nm in enumerate(["work between step tops", "counted vmcnt wait (stage landed)", "s_barrier", "tile end: held-back arithmetic + last stores", "tile end: patch DMA issue (10 pieces)"]):
            print("   step tops: %-34s %6.2f us per tile  %5.1f %%" % (nm, np.median(sub[:, i] / tiles) / ghz / 1e3, 100 * sub[:, i].sum() / st[:, 6].sum()))
    eng.close()
