"""Where a BO round's latency goes (GPU box): K0 + forward on device-resident inputs against engine.score_masks with host arrays,
and a cProfile of the host side.  usage: python tools/probes/latency_split.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.build()
from network_interpretation_imagenet_amd import masks, synth
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine, rank_segments
from oracle import scorer
eng = MaskedForwardEngine("resnet101", max_batch=512, device=0).load_state_dict(synth.make_state_dict("resnet101"))
img = synth.make_images(1, kind="blobs")[0]
x = scorer.to_tensor_normalize(img)
label, _ = eng.predict(x)
seg = synth.grid_segments()
s = 196
onoff = masks.windows_onoff(s, range(0, masks.bo_upper_bound(s) + 1))
m = onoff.shape[0]
dev = eng.device
xd = x.to(dev); segd = torch.from_numpy(rank_segments(seg)[0]).to(dev); ond = torch.from_numpy(onoff).to(dev)
labels = torch.full((m,), int(label), dtype=torch.int32, device=dev)
sc = torch.empty(m, dtype=torch.float32, device=dev); pr = torch.empty(m, dtype=torch.int32, device=dev)
def gpu_only():
    eng.stage_masks(xd, segd, ond, 0)
    eng.forward(m, labels, score_out=sc, pred_out=pr)
for _ in range(3): gpu_only()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(10): gpu_only()
e1.record(); torch.cuda.synchronize()
print("device-resident K0 + forward, %d masks, back to back: %.2f ms per call (events), %.2f ms wall" % (m, e0.elapsed_time(e1) / 10, (time.perf_counter() - t0) * 100))
t0 = time.perf_counter()
for _ in range(10):
    gpu_only(); torch.cuda.synchronize()
print("  ... with a synchronise per call: %.2f ms wall" % ((time.perf_counter() - t0) * 100))
t0 = time.perf_counter()
for _ in range(10): eng.score_masks(x, seg, onoff, label)
print("score_masks (host arrays in and out): %.2f ms wall" % ((time.perf_counter() - t0) * 100))
import cProfile, pstats
pr_ = cProfile.Profile(); pr_.enable()
for _ in range(10): eng.score_masks(x, seg, onoff, label)
pr_.disable()
pstats.Stats(pr_).sort_stats("cumulative").print_stats(14)
