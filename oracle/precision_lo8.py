"""CPU study (test infrastructure, uses the oracle's cases and weights): can the `lo` half of the split-fp16 operand format live in
EIGHT bits?  The gate VERDICT r4 item 2 asks for before any kernel is touched; it opens or closes the last arithmetic / byte family.

The engine stores every activation as two fp16 planes, hi = fp16(x) and lo = fp16(x - hi): 4 bytes per element, 22 significant bits,
three MFMA products per conv (W_hi.X_hi + W_hi.X_lo + W_lo.X_hi).  Formats under test -- every one keeps hi = fp16(x) and the main
product on the fp16 MFMA:

    full          hi + lo fp16 everywhere (what the engine computes today)
    trunk_e4m3    (i)   the TRUNK tensors (stem output, every block's output: what the residual adds and the next block's conv1 read)
                        store lo as OCP e4m3 with one power-of-two scale per 32 channels of a pixel (3 bytes per element); inner tensors full
    all_e4m3      (ii)  every activation stored that way
    all_e4m3_w8   (iii) (ii), and both correction products on ONE block-scaled fp8 MFMA: e4m3(W_hi).X_lo8 + W_lo8.e4m3(X_hi) with a
                        scale per 32 k (two units of matrix work instead of three; round 1's f16f8, DESIGN.md 5 "measured and rejected")
    trunk_i8      (iv)  the trunk tensors store lo as a SIGNED BYTE in units of ulp(hi) / 256 (the exponent comes from hi: no scale is
                        stored; 19 significant bits, 3 bytes per element); inner tensors full
    all_i8        (v)   every activation stored that way
    all_pi8c      (vi)  the one arithmetic family not priced before: corrections as INTEGER slices.  Every activation stores lo as a signed byte with ONE
                        power-of-two scale per pixel (over all its channels), and both correction products run on the int8 MFMA (twice the fp16 rate, exact
                        int32 accumulation over the whole K): q8(W_hi).X_lo8 + W_lo8.q8(X_hi) with 8-bit fixed point per pixel / per output channel
                        -- two units of matrix work instead of three.  Scales chosen independently per term (the optimistic form: sharing one int32
                        accumulator would tie lo's scale to hi's and cost two more bits)
    act_hi / alt_hi     two-product forms for the MNIST chain, VERDICT r4 item 7 (every / every second conv input as ONE fp16 plane); on the
                        ResNets act_hi is "t1 AND t2 as one plane", the harshest two-product form, for scale

A stored tensor is rounded ONCE, where the producer's epilogue would write it, and every consumer (convs and the residual add) sees the
rounded value; the arithmetic runs in fp64 (BatchNorm, ReLU, pools and adds exact), weights are hi + lo fp16.  The first conv (the
stem: an fp32 table in the engine) and fc always see the full format.  Score error = |softmax(logits)[label] - the same through exact fp64
operands|, label = argmax of the unmasked picture.

Cases: ResNet-18 and ResNet-101 with the synthetic weights on 5 pictures x 64 masks of each kind (uniform noise, blobs, felzenszwalb
windows of the blobs: oracle/precision_sweep.imagenet_cases), and BOTH trained checkpoints of the reference (the weights committed in
tests/golden/smallnet_*.npz) on their 2 committed pictures + 4 more seeded ones, 24 removed-superpixel sets each, drawn and staged as
generate_gp_training_data_cifar.py:271-321 / ..._mnist.py:163-243 do (oracle/smallnets_ref.py).

    python oracle/precision_lo8.py [resnet18,resnet101,cifar_resnet56,mnist_net] [pictures=5] [masks=64] [policies=...]

Round 6 (the hardened gate of the integer-slice family): `resnet18_tl` / `resnet101_tl` are the same topologies with
TRAINED-LIKE BatchNorm statistics on every layer (oracle/trained_like.py: per-channel tuples resampled from the CIFAR checkpoint, conv weights
calibrated so that the statistics hold), and LO8_SMALL_EXTRA=10 puts both shipped checkpoints on 12 pictures x 24 masks:

    LO8_SMALL_EXTRA=10 python oracle/precision_lo8.py resnet18_tl,resnet101_tl,cifar_resnet56,mnist_net 8 64 full,all_pi8c

Gate: a format may become a kernel only if its worst score error over everything is <= 5e-5 (the north-star's tolerance is 1e-4; the
same gate the two-product form failed in round 4).  Result of this script in the build container: profiles/r05_precision_lo8.txt.
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
from oracle.precision_sweep import imagenet_cases  # noqa: E402

GATE = 5e-5
BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------------------------------------------
# formats
# ------------------------------------------------------------------------------------------------------------------------------------
def f16(t):
    return t.to(torch.float16).double()


def e4m3_round(v):
    """Round fp64 values to the OCP e4m3 grid (4 exponent bits, bias 7, 3 mantissa bits, subnormals, largest finite 448), ties to
    even, saturating."""
    a = v.abs()
    e = torch.floor(torch.log2(torch.clamp(a, min=2.0 ** -30)))
    e = torch.clamp(e, min=-6.0)                        # below the smallest normal 2^-6 the grid is 2^-9
    ulp = torch.pow(2.0, e - 3.0)
    q = torch.round(v / ulp) * ulp                      # torch.round: half to even
    return torch.clamp(q, -448.0, 448.0)


def e4m3_block(t, dim, block=32):
    """Block-scaled e4m3 (MX-style): along `dim`, every `block` consecutive elements share the power-of-two scale that puts the block's
    largest magnitude into e4m3's top binade [256, 512) -> values rounded onto scale * e4m3 grid."""
    t = t.movedim(dim, -1)
    shp = t.shape
    n = shp[-1]
    pad = (-n) % block
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(shp[:-1] + (-1, block))
    amax = b.abs().amax(-1, keepdim=True)
    scale = torch.pow(2.0, torch.floor(torch.log2(torch.clamp(amax, min=2.0 ** -60))) - 8.0)
    q = e4m3_round(b / scale) * scale
    q = q.reshape(shp[:-1] + (-1,))[..., :n]
    return q.movedim(-1, dim)


def lo_i8(x, hi):
    """lo as a signed byte in units of ulp(hi) / 256: ulp(hi) = 2^(e - 10) with e = hi's binary exponent (>= -14: fp16's subnormal
    range has the spacing of its first binade), so lo = q * 2^(e - 18), q in [-128, 127].  The consumer rebuilds an fp16 operand from
    it: what fp16 cannot hold (below 2^-24) is lost as in the two-plane format."""
    e = torch.floor(torch.log2(torch.clamp(hi.abs(), min=2.0 ** -14)))
    unit = torch.pow(2.0, e - 18.0)
    q = torch.clamp(torch.round((x - hi) / unit), -128.0, 127.0)
    return f16(q * unit)


def q8_fixed(t, dims):
    """Symmetric 8-bit fixed point with one power-of-two scale per slice: the largest magnitude over `dims` maps into [64, 127]."""
    amax = t.abs().amax(dims, keepdim=True)
    scale = torch.pow(2.0, torch.ceil(torch.log2(torch.clamp(amax, min=2.0 ** -60) / 127.0)))
    return torch.clamp(torch.round(t / scale), -127.0, 127.0) * scale


class Stored:
    """An activation as it sits in memory: hi (fp16 values) + lo (whatever the format keeps of x - hi), both held as fp64."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo

    def value(self):
        return self.hi if self.lo is None else self.hi + self.lo


POLICIES = {
    # name: (format of trunk tensors, format of inner tensors, correction products on the fp8 MFMA)
    "f64": ("f64", "f64", False),
    "full": ("x2", "x2", False),
    "trunk_e4m3": ("e4m3", "x2", False),
    "all_e4m3": ("e4m3", "e4m3", False),
    "all_e4m3_w8": ("e4m3", "e4m3", True),
    "trunk_i8": ("i8", "x2", False),
    "all_i8": ("i8", "i8", False),
    "all_pi8c": ("pi8", "pi8", "int8"),
    "act_hi": ("x2", "hi", False),
    "alt_hi": ("x2", "alt", False),          # inner tensors alternate one plane / two planes (the MNIST chain: conv2, conv4, conv6 inputs)
}


class Policy:
    def __init__(self, name, sd64):
        self.name = name
        self.trunk, self.inner, self.w8 = POLICIES[name]
        self.sd = sd64
        self.wc = {}
        self.n_inner = 0

    def store(self, x, kind):
        """kind: 'trunk' | 'inner' | 'full' (network input, pooled features)."""
        fmt = {"trunk": self.trunk, "inner": self.inner, "full": "x2"}[kind]
        if self.name == "f64":
            return Stored(x, None)
        if fmt == "alt":
            fmt = "hi" if self.n_inner % 2 == 0 else "x2"
            self.n_inner += 1
        hi = f16(x)
        if fmt == "hi":
            return Stored(hi, None)
        if fmt == "x2":
            return Stored(hi, f16(x - hi))
        if fmt == "e4m3":
            return Stored(hi, e4m3_block(x - hi, 1))
        if fmt == "i8":
            return Stored(hi, lo_i8(x, hi))
        if fmt == "pi8":
            return Stored(hi, q8_fixed(x - hi, (1,)) if x.dim() == 4 else f16(x - hi))
        raise ValueError(fmt)

    def weights(self, key):
        if key not in self.wc:
            w = self.sd[key]
            if self.name == "f64":
                self.wc[key] = (w, None, None, None)
            else:
                # as the engine packs them (mpx_pack_conv_weights): every output channel is multiplied by the power of two that puts its largest
                # magnitude into [512, 1024) before the split, so that `lo` stays out of fp16's subnormal range; the epilogue's scale takes the
                # 2^-e back.  (Up to round 5 this script split the raw weights: on calibrated / trained-like networks whose channels carry
                # weights of 1e-3 the `lo` plane fell under 2^-24 and the "full" format looked like 13 bits -- 1.0e-3 on resnet101_tl.)
                amax = w.abs().flatten(1).amax(1).clamp(min=2.0 ** -60)
                e = (9.0 - torch.floor(torch.log2(amax))).reshape((-1,) + (1,) * (w.dim() - 1))
                ws = w * torch.pow(2.0, e)
                hi = f16(ws)
                lo = f16(ws - hi) * torch.pow(2.0, -e)
                hi = hi * torch.pow(2.0, -e)
                if self.w8 == "int8":
                    dims = tuple(range(1, w.dim()))
                    self.wc[key] = (hi, lo, q8_fixed(hi, dims), q8_fixed(lo, dims))
                elif self.w8:
                    self.wc[key] = (hi, lo, e4m3_block(hi, 1), e4m3_block(lo, 1))
                else:
                    self.wc[key] = (hi, lo, None, None)
        return self.wc[key]

    def conv(self, st, key, stride=1, pad=0, bias=None, plain=False):
        """The conv of a stored activation with the weights `key`.  Default: (W_hi + W_lo).(X_hi + X_lo) -- the engine drops the
        lo.lo term, 2^-22 of the result, below every format studied here.  With w8 (and not `plain`): the main product exact, the
        two correction products with BOTH operands on the block-scaled e4m3 grid."""
        w_hi, w_lo, w_hi8, w_lo8 = self.weights(key)
        if w_lo is None:
            return F.conv2d(st.value(), w_hi, bias, stride, pad)
        if self.w8 and not plain:
            y = F.conv2d(st.hi, w_hi, bias, stride, pad)
            if st.lo is not None:
                y = y + F.conv2d(st.lo, w_hi8, None, stride, pad)       # st.lo is already on its e4m3 grid (format 'e4m3')
            return y + F.conv2d(q8_fixed(st.hi, (1,)) if self.w8 == "int8" else e4m3_block(st.hi, 1), w_lo8, None, stride, pad)
        return F.conv2d(st.value(), w_hi + w_lo, bias, stride, pad)

    def linear(self, st, wkey, bkey):
        w_hi, w_lo, _a, _b = self.weights(wkey)
        return F.linear(st.value(), w_hi if w_lo is None else w_hi + w_lo, self.sd[bkey])


def bn(sd, y, prefix):
    return F.batch_norm(y, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"], sd[prefix + ".bias"],
                        False, 0.0, BN_EPS)


# ------------------------------------------------------------------------------------------------------------------------------------
# the three forwards with storage points (topologies: oracle/resnet_ref.forward, oracle/smallnets_ref.*_forward; checked against them
# under the f64 policy by check_forwards())
# ------------------------------------------------------------------------------------------------------------------------------------
def resnet_forward(P, x, arch):
    sd = P.sd
    kind, depths = R.ARCHS[arch]
    y = F.relu(bn(sd, P.conv(P.store(x, "full"), "conv1.weight", 2, 3, plain=True), "bn1"))
    t = P.store(F.max_pool2d(y, 3, 2, 1), "trunk")
    for s, d in enumerate(depths):
        for b in range(d):
            stride = 2 if (b == 0 and s > 0) else 1
            p = "layer%d.%d." % (s + 1, b)
            if kind == "basic":
                a = P.store(F.relu(bn(sd, P.conv(t, p + "conv1.weight", stride, 1), p + "bn1")), "inner")
                y = bn(sd, P.conv(a, p + "conv2.weight", 1, 1), p + "bn2")
            else:
                a = P.store(F.relu(bn(sd, P.conv(t, p + "conv1.weight", 1, 0), p + "bn1")), "inner")
                a = P.store(F.relu(bn(sd, P.conv(a, p + "conv2.weight", stride, 1), p + "bn2")), "inner")
                y = bn(sd, P.conv(a, p + "conv3.weight", 1, 0), p + "bn3")
            if (p + "downsample.0.weight") in sd:
                identity = bn(sd, P.conv(t, p + "downsample.0.weight", stride, 0), p + "downsample.1")
            else:
                identity = t.value()
            t = P.store(F.relu(y + identity), "trunk")
    f = P.store(F.adaptive_avg_pool2d(t.value(), 1).flatten(1), "full")
    return P.linear(f, "fc.weight", "fc.bias")


def cifar_forward(P, x, depth=56):
    sd = P.sd
    t = P.store(F.relu(bn(sd, P.conv(P.store(x, "full"), "conv1.weight", 1, 1, plain=True), "bn1")), "trunk")
    for name, inplanes, planes, stride in SN.cifar_resnet_blocks(depth):
        identity = t.value()
        if stride != 1 or inplanes != planes:
            identity = SN._downsample_b(identity, inplanes, planes, stride)
        a = P.store(F.relu(bn(sd, P.conv(t, name + ".conv1.weight", stride, 1), name + ".bn1")), "inner")
        y = bn(sd, P.conv(a, name + ".conv2.weight", 1, 1), name + ".bn2")
        t = P.store(F.relu(identity + y), "trunk")
    f = P.store(F.avg_pool2d(t.value(), 8).flatten(1), "full")
    return P.linear(f, "fc.weight", "fc.bias")


def mnist_forward(P, x):
    """A plain chain: no tensor is a trunk, every activation between two convs is 'inner'."""
    sd = P.sd
    st = P.store(x, "full")
    for i, (name, _cin, _cout, stride) in enumerate(SN.MNIST_CONVS):
        y = F.relu(bn(sd, P.conv(st, name + ".0.weight", stride, 1, bias=sd[name + ".0.bias"], plain=(i == 0)), name + ".1"))
        st = P.store(y, "inner")
    y = P.conv(st, "conv6.weight", 1, 1, bias=sd["conv6.bias"])
    f = P.store(y.mean(3).mean(2), "full")
    return P.linear(f, "fc1.weight", "fc1.bias")


def forward(arch, sd64, xb, policy):
    P = Policy(policy, sd64)
    with torch.no_grad():
        if arch == "mnist_net":
            return mnist_forward(P, xb)
        if arch.startswith("cifar_resnet"):
            return cifar_forward(P, xb, int(arch[len("cifar_resnet"):]))
        return resnet_forward(P, xb, arch)


def check_forwards():
    """The forwards above restate the oracle's; under the f64 policy they must reproduce it."""
    g = torch.Generator().manual_seed(5)
    sd = R.cast_state_dict(synth.make_state_dict("resnet18"), torch.float64)
    x = torch.randn(2, 3, 224, 224, generator=g, dtype=torch.float64)
    with torch.no_grad():
        d = float((forward("resnet18", sd, x, "f64") - R.forward(sd, x, "resnet18")).abs().max())
    assert d < 1e-10, d
    for arch, shape in (("cifar_resnet56", (2, 3, 32, 32)), ("mnist_net", (2, 1, 28, 28))):
        sd = load_small(arch)[0]
        x = torch.rand(*shape, generator=g, dtype=torch.float64)
        with torch.no_grad():
            d = float((forward(arch, sd, x, "f64") - SN.forward(sd, x, arch)).abs().max())
        assert d < 1e-10, (arch, d)
    # the formats: e4m3 values survive, the grid is 3 mantissa bits, the byte format carries 19 bits
    v = torch.tensor([448.0, 0.875, 2.0 ** -9, 17.0, 0.3], dtype=torch.float64)
    assert torch.equal(e4m3_round(v), torch.tensor([448.0, 0.875, 2.0 ** -9, 16.0, 0.3125], dtype=torch.float64))
    x = torch.rand(4, 64, 3, 3, generator=g, dtype=torch.float64) * 8 - 4
    hi = f16(x)
    assert float(((hi + lo_i8(x, hi)) - x).abs().max() / 4) <= 2.0 ** -18
    assert float((e4m3_block(x - hi, 1) - (x - hi)).abs().max()) <= 2.0 ** -11 * 4 * 2.0 ** -4


# ------------------------------------------------------------------------------------------------------------------------------------
# cases
# ------------------------------------------------------------------------------------------------------------------------------------
def score_err(lg, ref, label):
    return float((torch.softmax(lg, 1)[:, label] - torch.softmax(ref, 1)[:, label]).abs().max())


def sweep_imagenet(arch, n_pic, n_mask, policies):
    if arch.endswith("_tl"):        # trained-like BatchNorm statistics on every layer (oracle/trained_like.py)
        from oracle import trained_like
        arch_name, arch = arch, arch[:-3]
        sd64 = R.cast_state_dict(trained_like.make_trained_like_state_dict(arch), torch.float64)
    else:
        arch_name = arch
        sd64 = R.cast_state_dict(synth.make_state_dict(arch), torch.float64)
    worst = {}
    for kind, i, img, seg, onoff in imagenet_cases(n_pic, n_mask):
        t0 = time.time()
        x = S.to_tensor_normalize(img)
        xb = torch.from_numpy(np.stack([x] + [S.apply_mask(x, S.onoff_mask_u8(seg, onoff[m])) for m in range(len(onoff))])).double()
        ref = forward(arch, sd64, xb, "f64")
        label = int(ref[0].argmax())                # row 0 = the unmasked picture (the reference's base prediction)
        e = {p: score_err(forward(arch, sd64, xb, p)[1:], ref[1:], label) for p in policies}
        pr = torch.softmax(ref[1:], 1)[:, label]
        print("%-14s %-7s pic %d  S=%-4d label %-4d scores %.3f..%.3f   %s   (%.0f s)" % (
            arch_name, kind, i, onoff.shape[1], label, float(pr.min()), float(pr.max()), "   ".join("%s %.2e" % (p, e[p]) for p in policies),
            time.time() - t0), flush=True)
        for p, v in e.items():
            worst[(kind, p)] = max(worst.get((kind, p), 0.0), v)
    return worst


def load_small(arch):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "smallnet_%s.npz" % arch))
    return {k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith("sd/")}, g


def synth_small_picture(arch, seed):
    """Seeded pictures of the small networks' input shapes, in the value ranges their loaders yield (there is no MNIST / CIFAR data
    offline): strokes on black in [0, 1] for MNIST (ToTensor, generate_gp_training_data_mnist.py:57-69), smooth colour fields in
    [-1, 1] for CIFAR (ToTensor + Normalize(.5, .5), generate_gp_training_data_cifar.py:52-54)."""
    rng = np.random.default_rng(seed)
    if arch == "mnist_net":
        yy, xx = np.mgrid[0:28, 0:28].astype(np.float64)
        img = np.zeros((28, 28))
        for _ in range(3):
            x0, y0, x1, y1 = rng.uniform(5, 23, 4)
            t = np.clip(((xx - x0) * (x1 - x0) + (yy - y0) * (y1 - y0)) / ((x1 - x0) ** 2 + (y1 - y0) ** 2 + 1e-9), 0, 1)
            d2 = (xx - (x0 + t * (x1 - x0))) ** 2 + (yy - (y0 + t * (y1 - y0))) ** 2
            img = np.maximum(img, np.exp(-d2 / 2.5))
        return (np.floor(img * 255.999) / 255.0).astype(np.float32)[None]
    yy, xx = np.mgrid[0:32, 0:32].astype(np.float64)
    out = np.zeros((3, 32, 32), dtype=np.float32)
    for c in range(3):
        acc = np.zeros((32, 32))
        for _ in range(5):
            fx, fy, ph, amp = rng.uniform(0, 1, 4)
            acc += (0.3 + amp) * np.sin(2 * np.pi * ((0.3 + 2.2 * fx) * xx / 32 + (0.3 + 2.2 * fy) * yy / 32) + 2 * np.pi * ph)
        acc = (acc - acc.min()) / (acc.max() - acc.min())
        out[c] = ((np.floor(acc * 255.999) / 255.0).astype(np.float32) - np.float32(0.5)) / np.float32(0.5)
    return out


def small_cases(arch, g, extra, n_mask):
    """[(tag, masked inputs f32[M,C,H,W], label)]: the committed pictures, then `extra` seeded ones staged the scorers' way."""
    out = []
    for i in range(int(g["n_pictures"])):
        out.append(("golden%d" % i, g["pic%d/masked_inputs" % i], int(g["pic%d/label" % i])))
    sd32 = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    min_size, n_removed = (5, 1) if arch == "mnist_net" else (10, 5)
    for j in range(extra):
        x = synth_small_picture(arch, 100 + j)
        u8 = np.array(x, dtype=np.float32, copy=True).transpose(1, 2, 0)
        u8 -= u8.min()
        u8 /= u8.max()
        u8 *= 255
        seg = segment.felzenszwalb(u8.astype(np.uint8), scale=100, sigma=0.5, min_size=min_size).astype(np.int32)
        uniq = np.unique(seg)
        removed = [sorted(r) for r in masks.draw_removed_sets(uniq, min(n_removed, len(uniq) - 1), n_mask, random.Random(900 + j))]
        org = SN.org_img_minmax255(x)
        inputs = np.stack([SN.masked_input(org, SN.removed_mask_u8(seg, r)) for r in removed]).astype(np.float32)
        with torch.no_grad():
            label = int(SN.forward(sd32, torch.from_numpy(x[None]), arch)[0].argmax())
        out.append(("seed%d" % (100 + j), inputs, label))
    return out


def sweep_small(arch, policies, extra=4, n_mask=24):
    sd64, g = load_small(arch)
    worst = {}
    for tag, inputs, label in small_cases(arch, g, extra, n_mask):
        xb = torch.from_numpy(inputs).double()
        ref = forward(arch, sd64, xb, "f64")
        e = {p: score_err(forward(arch, sd64, xb, p), ref, label) for p in policies}
        pr = torch.softmax(ref, 1)[:, label]
        print("%-14s trained %-8s label %-2d scores %.3f..%.3f   %s" % (
            arch, tag, label, float(pr.min()), float(pr.max()), "   ".join("%s %.2e" % (p, e[p]) for p in policies)), flush=True)
        for p, v in e.items():
            worst[("trained", p)] = max(worst.get(("trained", p), 0.0), v)
    return worst


if __name__ == "__main__":
    archs = (sys.argv[1] if len(sys.argv) > 1 else "resnet18,resnet101,cifar_resnet56,mnist_net").split(",")
    n_pic = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n_mask = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    policies = (sys.argv[4].split(",") if len(sys.argv) > 4 else
                ["full", "trunk_e4m3", "all_e4m3", "all_e4m3_w8", "trunk_i8", "all_i8", "act_hi"])
    segment.load()
    check_forwards()
    print("# oracle/precision_lo8.py %s %d %d %s  (fp64 arithmetic, tensors rounded where they are stored; score error vs exact operands)" % (
        ",".join(archs), n_pic, n_mask, ",".join(policies)), flush=True)
    overall = {}
    for arch in archs:
        pol = list(policies)
        if arch == "mnist_net" and "alt_hi" not in pol:
            pol.append("alt_hi")
        worst = (sweep_small(arch, pol, extra=int(os.environ.get("LO8_SMALL_EXTRA", "4"))) if (arch == "mnist_net" or arch.startswith("cifar"))
                 else sweep_imagenet(arch, n_pic, n_mask, pol))
        for (kind, p), v in sorted(worst.items()):
            print("== %-14s %-8s %-12s worst score error %.2e" % (arch, kind, p, v), flush=True)
            overall[p] = max(overall.get(p, 0.0), v)
    for p, v in overall.items():
        print("== %-12s worst over everything: %.2e -> %s the gate of %.0e (tolerance 1e-4)" % (p, v, "PASSES" if v <= GATE else "FAILS", GATE))
