"""The ImageNet stem by superposition (csrc/mpx_stemtab.h: mpx_stem_table_build / mpx_stem_table_apply) against the two things it
replaces -- K0 + the MFMA stem conv + max pool of the engine, and the oracle's normalise -> mask -> conv1 -> bn1 -> relu -> maxpool
(generate_gp_training_data_imagenet.py:598-599,234-246 through torchvision's ResNet.forward) in fp64 -- on grid, felzenszwalb,
single-segment and pathological (pixel-noise) label maps, and end to end against the conv-stem engine and the CPU loop."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from network_interpretation_imagenet_amd import synth
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine, MpxError
from oracle import scorer

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def eng(mpx_lib):
    e = MaskedForwardEngine("resnet18", max_batch=96, device=0).load_state_dict(synth.make_state_dict("resnet18"))
    yield e
    e.close()


def _segments(kind, golden_dir):
    if kind == "grid":
        return synth.grid_segments()
    if kind == "single":
        return np.zeros((224, 224), dtype=np.int32)
    if kind == "felz":
        seg = np.load(os.path.join(golden_dir, "segments_blobs.npz"))["segments"][0]           # scikit-image 0.18.3 labels of a blobs picture
        return np.unique(seg, return_inverse=True)[1].reshape(224, 224).astype(np.int32)
    if kind == "noise":           # a new label at almost every pixel: up to 49 superpixels under one window (the slow path)
        return np.random.default_rng(5).integers(0, 2000, size=(224, 224)).astype(np.int32)
    if kind == "stripes":         # 2-pixel stripes: up to 4 x ... labels per window, more than 20 entries per pooled pixel
        return (np.arange(224)[None, :] // 2 + 0 * np.arange(224)[:, None]).astype(np.int32)
    raise ValueError(kind)


def _oracle_pooled(sd, img, seg, onoff):
    """fp64: normalise, mask, conv1 + bn1 + relu + maxpool(3, 2, 1)  -> [M,56,56,64]"""
    x = torch.as_tensor(scorer.to_tensor_normalize(img)).double()
    keep = torch.from_numpy(onoff.astype(np.float64))[:, torch.from_numpy(seg.astype(np.int64))]          # [M,224,224]
    xb = x[None] * keep[:, None]
    y = F.conv2d(xb, sd["conv1.weight"].double(), None, 2, 3)
    s = sd["bn1.weight"].double() / torch.sqrt(sd["bn1.running_var"].double() + 1e-5)
    y = (y - sd["bn1.running_mean"].double().view(1, -1, 1, 1)) * s.view(1, -1, 1, 1) + sd["bn1.bias"].double().view(1, -1, 1, 1)
    return F.max_pool2d(F.relu(y), 3, 2, 1).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("kind,m", [("grid", 37), ("felz", 64), ("single", 3), ("stripes", 33), ("noise", 5)])
def test_stem_table_vs_oracle_and_vs_the_mfma_stem(eng, golden_dir, kind, m):
    dev = eng.device
    sd = synth.make_state_dict("resnet18")
    img = synth.make_images(1, seed=41, kind="blobs" if kind == "felz" else "noise")[0]
    seg = _segments(kind, golden_dir)
    s = int(seg.max()) + 1
    onoff = synth.random_onoff(m, s, seed=43)
    onoff[0] = 1
    if m > 1:
        onoff[1] = 0
    img_d, seg_d, onoff_d = (torch.from_numpy(a).to(dev) for a in (img, seg, onoff))
    slot0 = 7
    # (a) the superposition path into the engine's pooled planes
    hi, lo = eng.stem_planes(slot0 + m)
    hi.fill_(float("nan"))
    lo.fill_(float("nan"))
    eng.build_stem_table(img_d, seg_d, s)
    eng.apply_stem_table(onoff_d, slot0)
    torch.cuda.synchronize()
    got = (hi[slot0:].float() + lo[slot0:].float()).cpu().double()
    assert not torch.isnan(got).any() and bool(torch.isnan(hi[:slot0].float()).all())       # exactly the slots asked for
    # (b) K0 + the MFMA stem + max pool on the same masks
    eng.stage_masks(img_d, seg_d, onoff_d, 0)
    oh = torch.empty(m, 56, 56, 64, dtype=torch.float16, device=dev)
    ol = torch.empty_like(oh)
    rc = eng._lib.mpx_stem_conv_maxpool(eng._h, _p(oh), _p(ol), m, eng._stream())
    assert rc == 0
    torch.cuda.synchronize()
    mfma = (oh.float() + ol.float()).cpu().double()
    want = _oracle_pooled(sd, img, seg, onoff)
    scale = float(want.abs().max())
    e_tab, e_mfma = float((got - want).abs().max()) / scale, float((mfma - want).abs().max()) / scale
    print("stem %s S=%d M=%d: table %.2e, MFMA stem %.2e relative to the fp64 oracle" % (kind, s, m, e_tab, e_mfma))
    assert e_tab <= 1e-6 and e_mfma <= 4e-6
    assert float((got - mfma).abs().max()) / scale <= 4e-6


def test_out_of_range_labels_count_as_removed(eng):
    """K0 keeps nothing of a pixel whose label is outside [0, S); the table drops its taps."""
    dev = eng.device
    img = synth.make_images(1, seed=44, kind="noise")[0]
    seg = synth.grid_segments().copy()
    seg[40:90, 100:160] = 500                   # outside [0, 196)
    seg[0:5, 0:5] = -3
    onoff = synth.random_onoff(9, 196, seed=45)
    onoff[0] = 1
    img_d, seg_d, onoff_d = (torch.from_numpy(a).to(dev) for a in (img, seg, onoff))
    eng.build_stem_table(img_d, seg_d, 196)
    eng.apply_stem_table(onoff_d, 0)
    hi, lo = eng.stem_planes(9)
    got = (hi.float() + lo.float()).cpu().double()
    eng.stage_masks(img_d, seg_d, onoff_d, 0)
    oh = torch.empty(9, 56, 56, 64, dtype=torch.float16, device=dev)
    ol = torch.empty_like(oh)
    assert eng._lib.mpx_stem_conv_maxpool(eng._h, _p(oh), _p(ol), 9, eng._stream()) == 0
    torch.cuda.synchronize()
    mfma = (oh.float() + ol.float()).cpu().double()
    assert float((got - mfma).abs().max()) <= 4e-6 * float(mfma.abs().max())


def test_stem_table_errors_and_mixed_staging(eng):
    dev = eng.device
    lib, h = eng._lib, eng._h
    img = torch.from_numpy(synth.make_images(1, seed=46, kind="noise")[0]).to(dev)
    seg = torch.from_numpy(synth.grid_segments()).to(dev)
    onoff = torch.from_numpy(synth.random_onoff(8, 196, seed=47)).to(dev)
    labels = torch.zeros(8, dtype=torch.int32, device=dev)
    eng.build_stem_table(img, seg, 196)
    with pytest.raises(MpxError):
        eng.apply_stem_table(onoff[:, :195].contiguous(), 0)                # S differs from the table's
    with pytest.raises(MpxError):
        eng.apply_stem_table(onoff, eng.max_batch - 3)                      # slots beyond max_batch
    assert lib.mpx_stem_table_build(h, _p(img), _p(img), _p(seg), 196, eng._mean, eng._std, None) == -1       # both image forms
    assert lib.mpx_stem_table_build(h, _p(img), None, _p(seg), 0, eng._mean, eng._std, None) == -1
    assert lib.mpx_stem_table_build(h, _p(img), None, _p(seg), 4097, eng._mean, eng._std, None) == -1
    # a forward takes slots staged one way: 4 by the table, 4 by K0 -> MPX_E_STATE, nothing guessed
    eng.apply_stem_table(onoff[:4].contiguous(), 0)
    eng.stage_masks(img, seg, onoff[4:].contiguous(), 4)
    with pytest.raises(MpxError, match="one forward takes one kind"):
        eng.forward(8, labels)
    eng.apply_stem_table(onoff[4:].contiguous(), 4)
    s_tab, p_tab = eng.forward(8, labels)
    eng.stage_masks(img, seg, onoff, 0)
    s_k0, p_k0 = eng.forward(8, labels)
    torch.cuda.synchronize()
    assert float((s_tab - s_k0).abs().max()) <= 2e-6 and bool((p_tab == p_k0).all())
    # loading layer 0 again discards the table (it was built from the old weights)
    eng.load_state_dict(synth.make_state_dict("resnet18"), only=["conv1"])
    with pytest.raises(MpxError, match="no table in place"):
        eng.apply_stem_table(onoff, 0)
    small = MaskedForwardEngine("mnist_net", max_batch=2, device=0)
    try:
        assert small.stem == "conv"
        assert small._lib.mpx_stem_table_apply(small._h, _p(onoff), 1, 196, 0, None) == -2
    finally:
        small.close()
    with pytest.raises(ValueError):
        MaskedForwardEngine("mnist_net", max_batch=2, device=0, stem="table")


@pytest.mark.parametrize("arch,tight", [("resnet18", 2e-5), ("resnet50", 2e-5)])
def test_table_stem_engine_vs_conv_stem_engine_vs_cpu_loop(mpx_lib, arch, tight):
    """End to end: the default engine (stem by superposition) against an engine that stages through K0 and the MFMA stem, and both
    against the reference-style CPU loop; rows straddle forwards (max_batch 24 < 40 rows), felzenszwalb-like irregular map."""
    sd = synth.make_state_dict(arch)
    img = synth.make_images(1, seed=51, kind="blobs")[0]
    rng = np.random.default_rng(52)
    seg = (synth.grid_segments(block=28) + 64 * (rng.random((224, 224)) < 0.02)).astype(np.int32)       # 8x8 blocks + speckles
    seg = np.unique(seg, return_inverse=True)[1].reshape(224, 224).astype(np.int32)
    s = int(seg.max()) + 1
    onoff = synth.random_onoff(40, s, seed=53)
    tab = MaskedForwardEngine(arch, max_batch=24, device=0).load_state_dict(sd)
    conv = MaskedForwardEngine(arch, max_batch=24, device=0, stem="conv").load_state_dict(sd)
    try:
        assert tab.stem == "table" and conv.stem == "conv"
        tab.stem_table_min_rows = 1              # 40 rows: below the default threshold (256 rows per image)
        label = tab.predict(img)[0]
        assert conv.predict(img)[0] == label
        _o, s_t, p_t = tab.score_masks(img, seg, onoff, label)
        _o, s_c, p_c = conv.score_masks(img, seg, onoff, label)
        assert np.abs(s_t - s_c).max() <= 2e-6 and (p_t == p_c).all()
        ref, ref_pred = scorer.score_masks_reference_loop(sd, arch, scorer.to_tensor_normalize(img), seg, onoff[:12], label)
        worst = float(np.abs(s_t[:12] - ref).max())
        print("%s, stem by superposition: max|d| vs CPU loop %.3e (conv stem %.3e)" % (arch, worst, float(np.abs(s_c[:12] - ref).max())))
        assert worst <= tight and (p_t[:12] == ref_pred).all()
        _o, s_l, _p2, lg = tab.score_masks(img, seg, onoff[:5], label, return_logits=True)      # the logits path stages the same way
        assert (s_l == s_t[:5]).all()
        # the table's 160 MB belong to engines that use it: allocated by the first mpx_stem_table_build, never by a K0-only engine ...
        assert 140e6 < tab.workspace_bytes - conv.workspace_bytes < 200e6
        # ... and a call may ask for either staging whatever the engine's default is (shard.job_stem hands the job's choice to every rank)
        _o, s_ct, p_ct = conv.score_masks(img, seg, onoff, label, stem="table")
        _o, s_tc, p_tc = tab.score_masks(img, seg, onoff, label, stem="conv")
        assert np.array_equal(s_ct, s_t) and np.array_equal(s_tc, s_c) and (p_ct == p_t).all() and (p_tc == p_c).all()
        assert tab.workspace_bytes == conv.workspace_bytes
        with pytest.raises(ValueError):
            tab.score_masks(img, seg, onoff, label, stem="mfma")
    finally:
        tab.close()
        conv.close()


def test_stem_table_apply_is_deterministic_and_block_independent(mpx_lib, golden_dir):
    """The apply launch meets its four waves' values in LDS (two buffers, one barrier per four masks) and takes one of four code paths
    per pooled pixel: the same rows must give the same bits (a) launched again next to a side stream that keeps HBM busy, (b) as one
    call of 515 rows (17 mask blocks, the last of 3) and as calls of 1, 2, 33 and 479 rows, (c) at another slot offset."""
    dev = torch.device("cuda", 0)
    eng = MaskedForwardEngine("resnet18", max_batch=600, device=0).load_state_dict(synth.make_state_dict("resnet18"))
    try:
        img = torch.from_numpy(synth.make_images(1, seed=61, kind="blobs")[0]).to(dev)
        seg_np = _segments("felz", golden_dir)
        s = int(seg_np.max()) + 1
        seg = torch.from_numpy(seg_np).to(dev)
        onoff = torch.from_numpy(synth.random_onoff(515, s, seed=62)).to(dev)
        eng.build_stem_table(img, seg, s)
        hi, lo = eng.stem_planes(600)

        def run(parts, slot0):
            hi.fill_(float("nan"))
            lo.fill_(float("nan"))
            at = 0
            for n in parts:
                eng.apply_stem_table(onoff[at:at + n].contiguous(), slot0 + at)
                at += n
            torch.cuda.synchronize()
            return hi[slot0:slot0 + 515].clone(), lo[slot0:slot0 + 515].clone()

        want = run([515], 0)
        assert not torch.isnan(want[0].float()).any()
        side = torch.cuda.Stream(device=dev)
        big_a = torch.empty(1 << 27, dtype=torch.float32, device=dev).normal_()
        big_b = torch.empty_like(big_a)
        for _ in range(3):
            with torch.cuda.stream(side):
                for _c in range(4):
                    big_b.copy_(big_a)
            got = run([515], 0)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        torch.cuda.synchronize()
        got = run([1, 2, 33, 479], 0)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        got = run([515], 85)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    finally:
        eng.close()
