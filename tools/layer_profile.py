#!/usr/bin/env python3
"""Per-layer conv timing (HIP events inside the engine) for one forward batch.
usage: python tools/layer_profile.py [arch] [batch] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from network_interpretation_imagenet_amd import synth  # noqa: E402
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine, MpxError  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
eng = MaskedForwardEngine(arch, max_batch=batch, device=0).load_state_dict(synth.make_state_dict(arch))
if os.environ.get("MPX_NO_FUSION"):     # tool-only: run a stage's first conv3 and its downsample conv as two launches
    eng.set_fusion(False)
if os.environ.get("MPX_FUSION_MASK"):   # tool-only: mpx_set_fusion mask (1 = round 2's plan: no block tails)
    eng.set_fusion(int(os.environ["MPX_FUSION_MASK"]))
img = torch.from_numpy(synth.make_images(1, kind="noise")[0]).to(dev)
seg = torch.from_numpy(synth.grid_segments()).to(dev)
onoff = torch.from_numpy(synth.random_onoff(batch, 196)).to(dev)
labels = torch.zeros(batch, dtype=torch.int32, device=dev)
if os.environ.get("MPX_TILE_PATCH"):      # tool-only override: patch kernel (tile 6) wherever it is eligible
    for i, d in enumerate(eng.layers):
        try:
            eng.set_conv_tile(i, 6)
        except MpxError:
            pass
if os.environ.get("MPX_TILE_RULES"):     # tool-only: "class:tile,..." with classes k1exp k1red k1s2 k3s1 k3s2 c64k1 c64k3 stem k1x (= default tile 10 or 14)
    rules = dict(r.split(":") for r in os.environ["MPX_TILE_RULES"].split(","))
    took, kept = {}, {}
    for i, d in enumerate(eng.layers):
        if d.cin == 3:
            cls = "stem"
        elif d.cout <= 64:
            cls = "c64k1" if d.ksize == 1 else "c64k3"
        elif d.ksize == 3:
            cls = "k3s2" if d.stride == 2 else "k3s1"
        elif d.stride == 2:
            cls = "k1s2"
        else:
            cls = "k1exp" if d.cout > d.cin else "k1red"
        if "k1x" in rules and eng.conv_tile(i) in (10, 14):         # the layers whose default is the persistent expanding kernel (256->1024, 128->512)
            cls = "k1x"
        if cls in rules and d.name != b"fc":
            try:
                eng.set_conv_tile(i, int(rules[cls]))
                took[cls] = took.get(cls, 0) + 1
            except MpxError:            # the layer is not eligible for that kernel: it keeps its default -- and says so
                kept[cls] = kept.get(cls, 0) + 1
    for cls in rules:
        print("rule %s:%s -> %d layers took the tile, %d kept their default (not eligible)" % (cls, rules[cls], took.get(cls, 0), kept.get(cls, 0)))
if os.environ.get("MPX_TILE_1X1"):      # tool-only override: one tile variant on every 1x1 conv with cout >= 128
    for i, d in enumerate(eng.layers):
        if d.ksize == 1 and d.cout >= 128:
            eng.set_conv_tile(i, int(os.environ["MPX_TILE_1X1"]))
if os.environ.get("MPX_TILE_C64"):      # tool-only override: one tile variant on every cout <= 64 conv
    for i, d in enumerate(eng.layers):
        if d.cout <= 64:
            eng.set_conv_tile(i, int(os.environ["MPX_TILE_C64"]))
if os.environ.get("MPX_TILE_ALL"):      # tool-only override: force one tile variant on every conv
    for i in range(len(eng.layers)):
        eng.set_conv_tile(i, int(os.environ["MPX_TILE_ALL"]))
table = eng.stem == "table" and not os.environ.get("MPX_STEM_CONV")      # tool-only: MPX_STEM_CONV=1 = K0 + the MFMA stem (rounds 1-3)


def stage():
    if table:               # the engine's default for an image with this many rows: the stem by superposition (row 0 of the table = its apply launch)
        eng.build_stem_table(img, seg, 196)
        eng.apply_stem_table(onoff, 0)
    else:
        eng.stage_masks(img, seg, onoff, 0)


for _ in range(2):
    stage()
    eng.forward(batch, labels)
torch.cuda.synchronize()
eng.profile(True)
for _ in range(reps):
    stage()
    eng.forward(batch, labels)
eng.profile(False)
prof = eng.collect_profile()
tot = 0.0
groups = {}
print("%-26s %5s %5s %2s %2s %4s %9s %9s %8s" % ("layer", "cin", "cout", "k", "s", "hout", "ms", "GFLOP", "TFLOP/s"))
for li, (d, ms) in enumerate(zip(eng.layers, prof["per_conv_ms"])):
    ms /= reps
    fl = 2.0 * batch * d.hout * d.hout * d.cout * d.cin * d.ksize * d.ksize
    key = (d.cin, d.cout, d.ksize, d.stride, d.hout)
    a = groups.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += ms
    a[2] += fl
    tot += ms
    if os.environ.get("MPX_PER_LAYER"):     # tool-only: one row per conv (a fused launch is booked on its main conv)
        print("%-26s %5d %5d %2d %2d %4d %9.3f %9.1f %8.1f   tile %d" % (d.name.decode(), d.cin, d.cout, d.ksize, d.stride, d.hout, ms, fl / 1e9, fl / max(ms, 1e-9) / 1e9, eng.conv_tile(li)))
print("-- grouped by shape --")
for key, (n, ms, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("%5d->%-5d k%d s%d out%-4d x%-3d %8.3f ms %5.1f%% %8.1f TFLOP/s%s" % (key[0], key[1], key[2], key[3], key[4], n, ms, 100 * ms / tot, fl / max(ms, 1e-9) / 1e9,
                                                                             "   (runs inside its block's conv3 launch)" if ms == 0 else ""))
tails = eng.bottleneck_tails()
if tails and not os.environ.get("MPX_NO_FUSION") and os.environ.get("MPX_FUSION_MASK", "3") == "3":
    names = [d.name.decode() for d in eng.layers]
    print("-- block tails (one launch each: conv2 -> conv3 + identity -> next conv1; the time is booked on the conv2 row; the tail with the"
          " downsample branch runs its block's own conv1 too) --")
    for c2, c3, ds, n1 in tails:
        print("  %s%-16s + %s%s + %-16s %8.3f ms" % ((names[c2 - 1] + " + ") if ds >= 0 else "", names[c2], names[c3],
                                                      (" + " + names[ds]) if ds >= 0 else "", names[n1], prof["per_conv_ms"][c2] / reps))
layer1 = sum(ms for d, ms in zip(eng.layers, prof["per_conv_ms"]) if d.name.startswith(b"layer1.") or d.name == b"layer2.0.conv1") / reps
print("layer1 (+ layer2.0.conv1): %.3f ms/batch" % layer1)
allfl = eng.flops_per_forward * batch
print("conv total %.3f ms/batch -> %.1f TFLOP/s algorithmic; other kinds ms/batch: %s" % (
    tot, allfl / tot / 1e9, {k: round(v / reps, 3) for k, v in prof["ms"].items()}))
