#!/usr/bin/env python3
"""Do an HBM-bound and an MFMA-bound layer overlap when they run on DISJOINT halves of the chip?  Two engines, each on a stream
created with hipExtStreamCreateWithCUMask (half the CUs), run their forwards concurrently; compared with one engine on the whole
chip.  Same total work: 2 x 1024 masks against 1 x 2048.

usage: python tools/probes/cu_mask_overlap.py [arch] [pattern ...]     patterns: low (CU bits 0-127 | 128-255), even (even | odd bits),
                                                                       nibble (alternate groups of 4 bits), none (two unmasked streams)"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
patterns = sys.argv[2:] or ["none", "low", "even", "nibble"]
dev = torch.device("cuda", 0)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def masks_for(pattern):
    if pattern == "low":
        a = [i < 128 for i in range(256)]
    elif pattern == "even":
        a = [i % 2 == 0 for i in range(256)]
    elif pattern == "nibble":
        a = [(i // 4) % 2 == 0 for i in range(256)]
    else:
        return None
    return a, [not x for x in a]


SMALL = bool(os.environ.get("MPX_SMALL_TILES"))     # every wide layer on a 64-KB kernel (tiles 2 / 7): two workgroups of DIFFERENT kernels fit one CU


def small_tiles(eng):
    """Round 4: with the default kernels (144-160 KB of LDS) two concurrent forwards can only time-slice a CU.  On the 64-KB kernels a
    workgroup of an HBM-bound layer of one stream and one of an MFMA-bound layer of the other can be resident together."""
    if not SMALL:
        return eng
    eng.set_fusion(1)                                   # layer by layer in layer1 (the tails are 78-KB persistent workgroups)
    for i, d in enumerate(eng.layers):
        if d.cout <= 64 or d.name == b"fc":
            continue
        eng.set_conv_tile(i, 7 if (d.ksize == 1 and d.stride == 1 and d.cout > d.cin) else 2)
    return eng


sd = synth.make_state_dict(arch)
img = torch.from_numpy(synth.make_images(1, kind="noise")[0]).to(dev)
seg = torch.from_numpy(synth.grid_segments()).to(dev)
REPS = 6


def run_single(batch):
    eng = small_tiles(MaskedForwardEngine(arch, max_batch=batch, device=0).load_state_dict(sd))
    onoff = torch.from_numpy(synth.random_onoff(batch, 196)).to(dev)
    labels = torch.zeros(batch, dtype=torch.int32, device=dev)
    for _ in range(2):
        eng.stage_masks(img, seg, onoff, 0)
        eng.forward(batch, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        eng.stage_masks(img, seg, onoff, 0)
        eng.forward(batch, labels)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.close()
    return REPS * batch / dt


def run_pair(pattern, batch, n_eng=2):
    engs = [small_tiles(MaskedForwardEngine(arch, max_batch=batch, device=0).load_state_dict(sd)) for _ in range(n_eng)]
    m = masks_for(pattern)
    streams = [masked_stream(m[0]), masked_stream(m[1])] if m else [torch.cuda.Stream(dev) for _ in range(n_eng)]
    onoff = torch.from_numpy(synth.random_onoff(batch, 196)).to(dev)
    labels = torch.zeros(batch, dtype=torch.int32, device=dev)
    outs = [(torch.empty(batch, device=dev), torch.empty(batch, dtype=torch.int32, device=dev)) for _ in range(n_eng)]
    torch.cuda.synchronize()

    def both():
        for e, s, o in zip(engs, streams, outs):
            with torch.cuda.stream(s):
                e.stage_masks(img, seg, onoff, 0)
                e.forward(batch, labels, score_out=o[0], pred_out=o[1])

    for _ in range(2):
        both()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        both()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = bool(torch.equal(outs[0][0], outs[1][0]))
    for e in engs:
        e.close()
    return n_eng * REPS * batch / dt, same


print("%s%s: one engine, whole chip, batch 2340: %8.0f fwd/s" % (arch, " (64-KB kernels)" if SMALL else "", run_single(2340)), flush=True)
print("%s: one engine, whole chip, batch 1170: %8.0f fwd/s" % (arch, run_single(1170)), flush=True)
for pat in patterns:
    if ":" in pat:                       # "N:batch" = N engines on N unmasked streams
        n, b = (int(v) for v in pat.split(":"))
        r, same = run_pair("none", b, n)
        print("%s: %d engines x batch %d on %d plain streams: %8.0f fwd/s   (identical scores: %s)" % (arch, n, b, n, r, same), flush=True)
        continue
    r, same = run_pair(pat, 1170)
    print("%s: two engines x batch 1170, CU masks '%s': %8.0f fwd/s   (scores of the two engines identical: %s)" % (arch, pat, r, same), flush=True)
