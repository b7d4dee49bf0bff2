import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g; g.build()
from network_interpretation_imagenet_amd import synth
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine
eng = MaskedForwardEngine("resnet101", max_batch=128, device=0).load_state_dict(synth.make_state_dict("resnet101"))
dev = eng.device
img = torch.from_numpy(synth.make_images(1, kind="noise")[0]).to(dev)
seg = torch.from_numpy(synth.grid_segments()).to(dev)
for B in (28, 118):
    onoff = torch.from_numpy(synth.random_onoff(B, 196)).to(dev)
    labels = torch.zeros(B, dtype=torch.int32, device=dev)
    def run():
        eng.stage_masks(img, seg, onoff, 0)
        return eng.forward(B, labels)
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): s, p = run()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2): run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        s, p = run()
    torch.cuda.synchronize()
    ref = run()[0].clone(); torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    assert torch.equal(s, ref), "graph replay differs"
    t0 = time.perf_counter()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 20
    print("B=%d eager %.2f ms  graph replay %.2f ms" % (B, eager * 1e3, graph * 1e3))
