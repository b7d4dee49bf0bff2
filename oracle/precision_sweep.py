"""CPU study (test infrastructure, uses the oracle): what does it cost to store t1 -- the tensor between a block's first conv and
its 3x3 conv2 -- as ONE fp16 plane, so that the 3x3 convs run two MFMA products (W_hi*X + W_lo*X) instead of three?

The gate VERDICT r3 item 2 asks for before the kernels are built: >= 5 pictures x 64 masks of each kind (uniform noise, smooth
blobs, felzenszwalb windows of the blobs) for ResNet-18 and ResNet-101 with the synthetic weights, plus the reference's TRAINED
CIFAR ResNet-56 (tests/golden/smallnet_cifar_resnet56.npz: the checkpoint's weights, its two pictures, 24 masks each).  The MNIST
Classification_Net is a plain conv chain without blocks: no tensor of it is "inside a block", the mode leaves it untouched.

Every conv / linear operand is rounded to the format under test, the arithmetic runs in fp64 (as oracle/precision_study.py):
    full      every activation and weight hi + lo (22 bits): what the engine computes by default up to round 3
    t1_hi     as full, but conv2's INPUT is rounded to ONE fp16 (round to nearest) -- the mode under test
The score error is |softmax(logits)[label] - the same through exact fp64 operands|, label = argmax of the UNMASKED picture.

    python oracle/precision_sweep.py [resnet18,resnet101,cifar_resnet56] [pictures=5] [masks=64]

The gate: t1_hi becomes the engine's default only if its worst score error over everything is <= 5e-5 (tolerance 1e-4).
Result of this script in the build container: profiles/r04_precision_sweep.txt.
"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from network_interpretation_imagenet_amd import masks, segment, synth  # noqa: E402
from oracle import resnet_ref as R, scorer as S, smallnets_ref as SN  # noqa: E402

GATE = 5e-5
POLICIES = ("full", "t1_hi") + (("w2_hi",) if os.environ.get("SWEEP_W2") else ())      # SWEEP_W2=1 adds the weight-side two-product form


def rnd(t, fmt):
    if fmt == "f64":
        return t
    hi = t.to(torch.float16).double()
    if fmt == "f16":
        return hi
    return hi + (t - hi).to(torch.float16).double()          # f16x2


class Formats:
    """Patches F.conv2d / F.linear so that a conv named by the forward being run sees operands in the policy's formats.  The oracle's
    forwards call F.conv2d(x, w, ...) once per layer, in a fixed order: conv2's calls are recognised by their weight tensor."""

    def __init__(self, sd64, policy):
        self.policy = policy
        self.t1_consumers = set()
        self.hi_only_weights = set()
        if policy == "t1_hi":
            self.t1_consumers = {id(v) for k, v in sd64.items() if k.endswith(".conv2.weight")}
        if policy == "w2_hi":       # the other two-product form: conv2's WEIGHTS as one fp16 plane, its input hi + lo
            self.hi_only_weights = {id(v) for k, v in sd64.items() if k.endswith(".conv2.weight")}
        self.wcache = {}

    def __enter__(self):
        self.conv, self.lin = F.conv2d, F.linear
        if self.policy == "f64":
            return self
        conv, lin = self.conv, self.lin

        def w_of(w):
            if id(w) not in self.wcache:
                self.wcache[id(w)] = rnd(w, "f16" if id(w) in self.hi_only_weights else "f16x2")
            return self.wcache[id(w)]

        def conv2d(x, w, b=None, stride=1, padding=0):
            return conv(rnd(x, "f16" if id(w) in self.t1_consumers else "f16x2"), w_of(w), b, stride, padding)

        F.conv2d = conv2d
        F.linear = lambda x, w, b=None: lin(rnd(x, "f16x2"), w_of(w), b)
        return self

    def __exit__(self, *a):
        F.conv2d, F.linear = self.conv, self.lin


def run(forward, sd64, xb, policy):
    with Formats(sd64, policy), torch.no_grad():
        return forward(sd64, xb)


def score_err(lg, ref, label):
    return float((torch.softmax(lg, 1)[:, label] - torch.softmax(ref, 1)[:, label]).abs().max())


def imagenet_cases(n_pic, n_mask):
    """[(kind, picture index, u8 image, label map, onoff u8[n_mask, S])]"""
    seg_lib = segment.load() and segment
    out = []
    noise = synth.make_images(n_pic, seed=501, kind="noise")
    blobs = synth.make_images(n_pic, seed=502, kind="blobs")
    grid = synth.grid_segments()
    for i in range(n_pic):
        out.append(("noise", i, noise[i], grid, synth.random_onoff(n_mask, 196, seed=600 + i)))
    for i in range(n_pic):
        out.append(("blobs", i, blobs[i], grid, synth.random_onoff(n_mask, 196, seed=700 + i)))
    for i in range(n_pic):
        # the reference's own masks: windows of int(0.4 * S) consecutive felzenszwalb superpixels
        # (generate_gp_training_data_imagenet.py:183,223-230), drawn as it draws them
        seg = seg_lib.felzenszwalb(blobs[i])
        uniq, inv = np.unique(seg, return_inverse=True)
        s = len(uniq)
        first = masks.draw_first_indices(s, n_mask, random.Random(800 + i))
        out.append(("felz", i, blobs[i], inv.reshape(seg.shape).astype(np.int32), masks.windows_onoff(s, first)))
    return out


def sweep_imagenet(arch, n_pic, n_mask):
    sd64 = R.cast_state_dict(synth.make_state_dict(arch), torch.float64)
    fwd = lambda sd, x: R.forward(sd, x, arch)
    worst = {}
    for kind, i, img, seg, onoff in imagenet_cases(n_pic, n_mask):
        t0 = time.time()
        x = S.to_tensor_normalize(img)
        xb = torch.from_numpy(np.stack([x] + [S.apply_mask(x, S.onoff_mask_u8(seg, onoff[m])) for m in range(len(onoff))])).double()
        ref = run(fwd, sd64, xb, "f64")
        label = int(ref[0].argmax())                # row 0 = the unmasked picture (the reference's base prediction)
        e = {p: score_err(run(fwd, sd64, xb, p)[1:], ref[1:], label) for p in POLICIES}
        pr = torch.softmax(ref[1:], 1)[:, label]
        print("%-10s %-5s pic %d  S=%-4d label %-4d scores %.3f..%.3f   %s   (%.0f s)" % (
            arch, kind, i, onoff.shape[1], label, float(pr.min()), float(pr.max()), "   ".join("%s %.2e" % (p, e[p]) for p in POLICIES), time.time() - t0), flush=True)
        for p, v in e.items():
            worst[(kind, p)] = max(worst.get((kind, p), 0.0), v)
    return worst


def sweep_cifar():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "smallnet_cifar_resnet56.npz"))
    sd64 = {k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith("sd/")}
    fwd = lambda sd, x: SN.forward(sd, x, "cifar_resnet56")
    worst = {}
    for i in range(int(g["n_pictures"])):
        xb = torch.from_numpy(g["pic%d/masked_inputs" % i]).double()
        label = int(g["pic%d/label" % i])
        ref = run(fwd, sd64, xb, "f64")
        e = {p: score_err(run(fwd, sd64, xb, p), ref, label) for p in POLICIES}
        pr = torch.softmax(ref, 1)[:, label]
        print("%-10s %-5s pic %d  label %-4d scores %.3f..%.3f   %s" % (
            "cifar_resnet56", "trained", i, label, float(pr.min()), float(pr.max()), "   ".join("%s %.2e" % (p, e[p]) for p in POLICIES)), flush=True)
        for p, v in e.items():
            worst[("trained", p)] = max(worst.get(("trained", p), 0.0), v)
    return worst


if __name__ == "__main__":
    archs = (sys.argv[1] if len(sys.argv) > 1 else "resnet18,resnet101,cifar_resnet56").split(",")
    n_pic = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n_mask = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    overall = 0.0
    print("# oracle/precision_sweep.py %s %d %d  (fp64 arithmetic, operands rounded to the format; score error vs exact operands)" % (",".join(archs), n_pic, n_mask))
    for arch in archs:
        worst = sweep_cifar() if arch.startswith("cifar") else sweep_imagenet(arch, n_pic, n_mask)
        for (kind, p), v in sorted(worst.items()):
            print("== %-14s %-8s %-6s worst score error %.2e" % (arch, kind, p, v), flush=True)
            if p == "t1_hi":
                overall = max(overall, v)
    print("== t1_hi worst over everything: %.2e -> %s the gate of %.0e (tolerance 1e-4)" % (overall, "PASSES" if overall <= GATE else "FAILS", GATE))
