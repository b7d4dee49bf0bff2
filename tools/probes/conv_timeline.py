#!/usr/bin/env python3
"""Where does a conv workgroup spend its life?  Diagnostic build of the library (-DMPX_DIAG: every workgroup writes
s_memtime stamps of its phases and the CU it ran on; never the product .so) run on ONE layer.

usage: python tools/probes/conv_timeline.py [arch] [layer] [batch] [tile,...]

Prints, per tile variant: kernel time, and per-workgroup medians (in ns, via the s_memrealtime/s_memtime ratio) of
prologue (start -> first stage landed), K loop (-> trailing DMAs drained), epilogue part 1 (residual issue, LDS
transposition) and part 2 (LDS read, add, split, stores issued); then per CU: how much of the kernel's span had
0 / 1 / >= 2 workgroups inside their K loop (the MFMA-capable phase) and inside their epilogue.
"""
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


def build_diag(extra_flags=()):
    """-DMPX_DIAG build of the library next to this file.  Rebuilt when the sha256 of its sources + flags differs from the stamp
    written at build time (mtimes mean nothing once the tree has been copied to the GPU box, __graft_entry__._stale)."""
    import __graft_entry__ as g
    flags = g.HIPCC_FLAGS + ["-DMPX_DIAG"] + list(extra_flags)
    srcs = g.lib_sources()
    if "-DMPX_EXPERIMENTAL" in extra_flags:
        import glob
        srcs = srcs + sorted(glob.glob(os.path.join(HERE, "experimental", "*.h")))
    want = g._source_hash(srcs, flags)
    stamp = DIAG + ".sha256"
    if os.path.exists(DIAG) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-o", DIAG, os.path.join(g.CSRC, "mpx_api.hip")], cwd=g.CSRC)
    with open(stamp, "w") as fh:
        fh.write(want + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
        build_diag()
        sys.exit(0)
    if not os.path.exists(DIAG):
        build_diag()
    from network_interpretation_imagenet_amd import _lib, synth
    _lib.LIB_PATH = DIAG                                   # this process binds the diagnostic library
    from network_interpretation_imagenet_amd.engine import MaskedForwardEngine

    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
    layer = sys.argv[2] if len(sys.argv) > 2 else "layer3.5.conv3"
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    tiles = [int(t) for t in sys.argv[4].split(",")] if len(sys.argv) > 4 else [-1]
    dev = torch.device("cuda", 0)
    eng = MaskedForwardEngine(arch, max_batch=8, device=0).load_state_dict(synth.make_state_dict(arch))
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
    rh, rl = planes(batch, d.hout, d.hout, d.cout) if d.residual else (None, None)
    oh = torch.empty(batch, d.hout, d.hout, d.cout, dtype=torch.float16, device=dev)
    ol = torch.empty_like(oh)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    max_wg = 1 << 18
    stamps = torch.zeros(max_wg * 8, dtype=torch.int64, device=dev)

    def run():
        _lib.check(eng._h, lib.mpx_conv_bn_act(eng._h, i, p(xh), p(xl), p(rh), p(rl), p(oh), p(ol), None, batch, None), "conv")

    for tile in tiles:
        eng.set_conv_tile(i, tile)
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
        lib.mpx_debug_set_stamps(eng._h, C.c_void_p(stamps.data_ptr()))
        run()
        torch.cuda.synchronize()
        lib.mpx_debug_set_stamps(eng._h, None)
        st = stamps.cpu().numpy().reshape(-1, 8)
        st = st[st[:, 5] != 0]
        if eng.conv_tile(i) == 10:
            print("   (tile 10 = persistent kernel: per workgroup the SUMS over its %d..%d tiles -- 'prologue' = K loops, 'k-loop' = waits for "
                  "the residual lines, 'epilogue-1' = epilogue arithmetic and store issue)" % (st[:, 7].min(), st[:, 7].max()))
        n = len(st)
        hw, t_start, t_pro, t_kend, t_epi, t_end, rt_end = (st[:, k].astype(np.int64) for k in range(7))
        # clock: every CU is busy for (almost) the whole kernel, and s_memtime is consistent within a CU, so the median over the
        # CUs of (last stamp - first stamp) cycles divided by the kernel's event time is the in-kernel clock
        key0 = ((hw >> 32) << 16) | ((hw & 0xffffffff) >> 8 & 0xff)
        spans = np.array([(t_end[key0 == cu].max() - t_start[key0 == cu].min()) for cu in np.unique(key0)], dtype=np.float64)
        span_cyc = float(np.median(spans))
        ghz = span_cyc / (ms * 1e6)
        ns = lambda c: c / ghz
        print("== %s %s B=%d tile=%d: %d->%d k%d res=%d | kernel %.3f ms, %d workgroups, in-kernel clock ~%.2f GHz (median CU span / kernel time)" % (
            arch, layer, batch, eng.conv_tile(i), d.cin, d.cout, d.ksize, d.residual, ms, n, ghz))
        ph = {"prologue": t_pro - t_start, "k-loop": t_kend - t_pro, "epilogue-1 (residual wait, LDS write)": t_epi - t_kend,
              "epilogue-2 (LDS read, add, stores)": t_end - t_epi, "whole workgroup": t_end - t_start}
        for k, v in ph.items():
            q = np.percentile(v, [10, 50, 90])
            print("   %-40s median %8.0f ns  (p10 %7.0f  p90 %7.0f)   = %5.1f %% of a workgroup's life" % (
                k, ns(q[1]), ns(q[0]), ns(q[2]), 100.0 * v.mean() / (t_end - t_start).mean()))
        nk = d.cin * d.ksize * d.ksize // 32
        print("   K steps %d -> %.0f ns per step (median k-loop / steps)" % (nk, ns(np.median(t_kend - t_pro)) / nk))
        # per-CU concurrency: key = XCC id + HW_ID without wave/simd/pipe/queue/state bits (keep cu, sh, se ids)
        key = ((hw >> 32) << 16) | ((hw & 0xffffffff) >> 8 & 0xff)
        occ_k, occ_e, occ_any = np.zeros(4), np.zeros(4), np.zeros(6)
        for cu in np.unique(key):
            m = key == cu
            ev = []
            for a, b, kind in ((t_pro[m], t_kend[m], 0), (t_kend[m], t_end[m], 1), (t_start[m], t_end[m], 2)):
                ev += [(int(x), +1, kind) for x in a] + [(int(x), -1, kind) for x in b]
            ev.sort()
            cnt = [0, 0, 0]
            last = int(t_start[m].min())
            for t, dlt, kind in ev:
                occ_k[min(cnt[0], 3)] += t - last
                occ_e[min(cnt[1], 3)] += t - last
                occ_any[min(cnt[2], 5)] += t - last
                cnt[kind] += dlt
                last = t
        tot = occ_k.sum()
        print("   CUs seen: %d; share of CU time with N workgroups in the K loop:  0: %.1f %%  1: %.1f %%  2: %.1f %%  3+: %.1f %%" % (
            len(np.unique(key)), *(100 * occ_k / tot)))
        print("                 share with N workgroups in the epilogue:           0: %.1f %%  1: %.1f %%  2: %.1f %%  3+: %.1f %%" % (
            *(100 * occ_e / tot),))
        print("                 share with N workgroups resident:                  " + "  ".join("%d: %.1f %%" % (k, v) for k, v in enumerate(100 * occ_any / occ_any.sum())))
    eng.close()
