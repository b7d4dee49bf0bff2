"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Every check goes through the C-ABI of
libmpx.so; the CPU oracle (oracle/) and torch CPU fp64 ops are the checkers.

Tolerances: integer/bit work (K0 fp32 output, maxpool, argmax) is bit-exact; floating-point scores
must be within 1e-4 of the reference-style CPU loop (BASELINE.json north_star) -- the split-fp16
MFMA path is expected to land ~1e-6, so the tests also assert the tighter 2e-5.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from network_interpretation_imagenet_amd import _lib, masks, synth
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine, MpxError
from oracle import scorer

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4        # north_star tolerance
SCORE_TOL_TIGHT = 2e-5  # what split-fp16 should achieve


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def split(x):
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return hi.contiguous(), lo.contiguous()


def merge(hi, lo):
    return hi.float() + lo.float()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def eng18(mpx_lib, dev):
    e = MaskedForwardEngine("resnet18", max_batch=64, device=0).load_state_dict(synth.make_state_dict("resnet18"))
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng101(mpx_lib, dev):
    e = MaskedForwardEngine("resnet101", max_batch=32, device=0).load_state_dict(synth.make_state_dict("resnet101"))
    yield e
    e.close()


def _input_planes(eng, n):
    hi, lo = eng.input_planes(n)
    return hi.clone(), lo.clone()


# ------------------------------------------------------------------------------------------------
# K0
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,seg_kind", [(5, "grid"), (37, "felz"), (64, "single")])
def test_k0_u8_bit_exact(eng18, dev, golden_dir, m, seg_kind):
    img = synth.make_images(2, kind="noise")[1]
    if seg_kind == "grid":
        seg = synth.grid_segments()
    elif seg_kind == "felz":
        seg = np.load(os.path.join(golden_dir, "segments_blobs.npz"))["segments"][0].astype(np.int32)
    else:
        seg = np.zeros((224, 224), dtype=np.int32)
    s = int(seg.max()) + 1
    onoff = synth.random_onoff(m, s, seed=m)
    if seg_kind == "single":
        onoff[:, 0] = np.arange(m) % 2
    out = torch.empty(m, 3, 224, 224, dtype=torch.float32, device=dev)
    eng18.stage_masks(torch.from_numpy(img).to(dev), torch.from_numpy(seg).to(dev), torch.from_numpy(onoff).to(dev), 0, out)
    torch.cuda.synchronize()
    x = scorer.to_tensor_normalize(img)
    want = np.stack([scorer.apply_mask(x, scorer.onoff_mask_u8(seg, onoff[i])) for i in range(m)])
    got = out.cpu().numpy()
    assert (got.view(np.int32) == want.view(np.int32)).all()          # bit-exact, including -0.0
    hi, lo = _input_planes(eng18, m)
    staged = merge(hi, lo).cpu().numpy()
    assert (staged[:, :3] == 0).all() and (staged[:, -3:] == 0).all() and (staged[:, :, :3] == 0).all() \
        and (staged[:, :, -3:] == 0).all() and (staged[..., 3] == 0).all()      # border + 4th channel stay zero
    inner = staged[:, 3:227, 3:227, :3].transpose(0, 3, 1, 2)
    assert np.abs(inner - want).max() <= 2.0 ** -21 * np.abs(want).max()       # 22-bit split of the same values


def test_k0_f32_input_and_slots(eng18, dev):
    img = synth.make_images(1)[0]
    x = scorer.to_tensor_normalize(img)
    seg = synth.grid_segments()
    onoff = synth.random_onoff(3, 196, seed=9)
    out = torch.empty(3, 3, 224, 224, dtype=torch.float32, device=dev)
    eng18.stage_masks(x.to(dev), torch.from_numpy(seg).to(dev), torch.from_numpy(onoff).to(dev), 7, out)
    torch.cuda.synchronize()
    want = np.stack([scorer.apply_mask(x, scorer.onoff_mask_u8(seg, onoff[i])) for i in range(3)])
    assert (out.cpu().numpy().view(np.int32) == want.view(np.int32)).all()
    hi, lo = _input_planes(eng18, 10)
    got = merge(hi, lo)[7:10, 3:227, 3:227, :3].permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - want).max() <= 2.0 ** -21 * np.abs(want).max()


def test_k0_argument_errors(eng18, dev):
    img = torch.zeros(224, 224, 3, dtype=torch.uint8, device=dev)
    seg = torch.zeros(224, 224, dtype=torch.int32, device=dev)
    with pytest.raises(MpxError):
        eng18.stage_masks(img, seg, torch.ones(65, 1, dtype=torch.uint8, device=dev), 0)      # > max_batch
    with pytest.raises(MpxError):
        eng18.stage_masks(img, seg, torch.ones(1, 4096, dtype=torch.uint8, device=dev), 0)    # S too large
    with pytest.raises(ValueError):
        eng18.stage_masks(img.float(), seg, torch.ones(1, 1, dtype=torch.uint8, device=dev), 0)
    with pytest.raises(ValueError):
        eng18.stage_masks(img, seg.long(), torch.ones(1, 1, dtype=torch.uint8, device=dev), 0)


# ------------------------------------------------------------------------------------------------
# conv + BN (+ residual) (+ ReLU): every distinct kernel path, against torch CPU fp64
# ------------------------------------------------------------------------------------------------
def _conv_reference(sd, d, x_nchw64, res_nchw64):
    name, bn = d.name.decode(), d.bn_name.decode()
    w = sd[name + ".weight"].double()
    y = F.conv2d(x_nchw64, w, None, d.stride, d.pad)
    scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
    y = (y - sd[bn + ".running_mean"].double().view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1) + sd[bn + ".bias"].double().view(1, -1, 1, 1)
    if res_nchw64 is not None:
        y = y + res_nchw64
    return F.relu(y) if d.relu else y


def _run_conv(eng, i, x_nhwc, res_nhwc, batch, on_device=False):
    """on_device: return only the merged output, left on the GPU (kernel-against-kernel comparisons of large batches)"""
    d = eng.layers[i]
    dev = eng.device
    xh, xl = split(x_nhwc.to(dev))
    rh = rl = None
    if res_nhwc is not None:
        rh, rl = split(res_nhwc.to(dev))
    oh = torch.full((batch, d.hout, d.hout, d.cout), float("nan"), dtype=torch.float16, device=dev)
    ol = torch.full_like(oh, float("nan"))
    rc = eng._lib.mpx_conv_bn_act(eng._h, i, _p(xh), _p(xl), _p(rh), _p(rl), _p(oh), _p(ol), None, batch, eng._stream())
    _lib.check(eng._h, rc, "mpx_conv_bn_act")
    torch.cuda.synchronize()
    if on_device:
        return (merge(oh, ol),)
    return merge(oh, ol).cpu(), merge(xh, xl).cpu(), (merge(rh, rl).cpu() if rh is not None else None)


def _layer_index(eng, name):
    return [d.name.decode() for d in eng.layers].index(name)


R18_LAYERS = ["layer1.0.conv1", "layer1.0.conv2", "layer2.0.conv1", "layer2.0.downsample.0", "layer2.0.conv2",
              "layer3.1.conv2", "layer4.0.conv1", "layer4.1.conv2"]
R101_LAYERS = ["layer1.0.conv1", "layer1.0.conv2", "layer1.0.conv3", "layer1.0.downsample.0", "layer1.1.conv1",
               "layer2.0.conv2", "layer2.0.downsample.0", "layer3.5.conv1", "layer3.5.conv2", "layer3.5.conv3",
               "layer4.0.conv2", "layer4.2.conv3"]


def _check_layer(eng, sd, name, batch=3, tile=-1):
    i = _layer_index(eng, name)
    d = eng.layers[i]
    eng.set_conv_tile(i, tile)
    try:
        _check_layer_body(eng, sd, i, d, name, batch)
    finally:
        eng.set_conv_tile(i, -1)


def _check_layer_body(eng, sd, i, d, name, batch):
    g = torch.Generator().manual_seed(i)
    x = torch.randn(batch, d.hin, d.hin, d.cin, generator=g).clamp_min(-0.5) * 1.5     # mostly post-ReLU-like
    res = torch.randn(batch, d.hout, d.hout, d.cout, generator=g) if d.residual else None
    got, x_used, res_used = _run_conv(eng, i, x, res, batch)
    want = _conv_reference(sd, d, x_used.double().permute(0, 3, 1, 2),
                           res_used.double().permute(0, 3, 1, 2) if res_used is not None else None)
    want = want.permute(0, 2, 3, 1)
    assert not torch.isnan(got).any()
    err = (got.double() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= 4e-6 * max(scale, 1.0), "%s: max err %.3e (scale %.2f)" % (name, err, scale)


@pytest.mark.parametrize("name", R18_LAYERS)
def test_conv_layers_resnet18(eng18, name):
    _check_layer(eng18, synth.make_state_dict("resnet18"), name)


@pytest.mark.parametrize("name", R101_LAYERS)
def test_conv_layers_resnet101(eng101, name):
    _check_layer(eng101, synth.make_state_dict("resnet101"), name)


@pytest.mark.parametrize("tile", [0, 1, 2, 4, 7])
@pytest.mark.parametrize("name", ["layer1.0.conv1", "layer1.0.conv3", "layer2.0.conv2", "layer3.5.conv2", "layer4.2.conv3"])
def test_conv_every_tile_variant(eng101, name, tile):
    """Each shape of the generic kernel (mpx_set_conv_tile) on 1x1 / 3x3 / strided / residual layers, odd batch (ragged tiles)."""
    _check_layer(eng101, synth.make_state_dict("resnet101"), name, batch=5, tile=tile)


def test_conv_tile_ids_are_the_default_kernels(eng101):
    """The documented tile ids are exactly what default_tile hands out; the ids of kernels that never became a default (3, 5, 8, 11:
    probe builds only) and anything else are refused."""
    i = _layer_index(eng101, "layer3.5.conv3")
    for tile in (3, 5, 8, 11, 15):
        assert eng101._lib.mpx_set_conv_tile(eng101._h, i, tile) == -1
    assert b"product ids" in eng101._lib.mpx_last_error(eng101._h)
    assert {eng101.conv_tile(j) for j in range(len(eng101.layers))} <= {0, 1, 2, 4, 6, 7, 9, 10, 12, 13, 14}
    # the defaults of the expanding 1x1 layers: K = 256 on the weights-in-registers kernel, the other K >= 128 ones (7x7 maps included) on tile 10
    for name, tile in (("layer3.5.conv3", 14), ("layer3.22.conv3", 14), ("layer2.1.conv3", 10), ("layer4.1.conv3", 10), ("layer1.1.conv3", 7)):
        assert eng101.conv_tile(_layer_index(eng101, name)) == tile, name


@pytest.mark.parametrize("name,batch", [("layer3.5.conv3", 5), ("layer3.5.conv3", 47), ("layer3.5.conv1", 13), ("layer2.1.conv3", 3),
                                        ("layer1.1.conv3", 2), ("layer4.1.conv1", 21), ("layer4.2.conv3", 9), ("layer1.1.conv1", 1),
                                        ("layer3.5.conv1", 340), ("layer4.1.conv1", 700)])      # whole rounds + small remainder: split launch
def test_conv256_kernel(eng101, name, batch):
    """Tile id 9 = the 256x256-tile kernel for 1x1 stride-1 layers (csrc/mpx_conv256.h): quadrant-snaked K step on a
    two-stage 128-KB ring; ragged pixel counts (last tile partial), K from 64 (one loop iteration) to 2048, with and without
    residual.  layer1.1.conv1 (cout 64) is not eligible.  The two large batches give 261 / 268 tiles on 256 CUs: the
    engine then runs the 256 whole-round tiles on this kernel and the remaining pixels on the 128x128 kernel."""
    i = _layer_index(eng101, name)
    if eng101.layers[i].cout % 256:
        assert eng101._lib.mpx_set_conv_tile(eng101._h, i, 9) == -1
        return
    _check_layer(eng101, synth.make_state_dict("resnet101"), name, batch=batch, tile=9)


@pytest.mark.parametrize("name,batch", [("layer3.5.conv3", 5), ("layer3.5.conv3", 47), ("layer3.5.conv3", 161), ("layer2.1.conv3", 25),
                                        ("layer4.2.conv3", 200), ("layer3.5.conv1", 201), ("layer4.1.conv1", 9), ("layer1.1.conv3", 2),
                                        ("layer3.5.conv1", 1)])
def test_convx_persistent_expanding_kernel(eng101, name, batch):
    """Tile id 10 = the persistent pipelined kernel for expanding 1x1 layers (csrc/mpx_convx.h): three-stage ring that runs on
    across the tiles of a workgroup, register epilogue with residual lines requested two K steps ahead, position-dependent
    counted vmcnt waits.  Batches give from one tile per workgroup (no tile boundary) up to four (boundaries, ragged last
    tile), K = 128 (two step pairs) to 2048, with and without residual.  layer1.1.conv3 (K = 64) is not eligible; a launch with
    fewer tiles than CUs (layer3.5.conv3 at 5 images = 32 tiles, layer3.5.conv1 at 1) runs on the 128x128 8-wave kernel, which sums
    in the same order -- the test asserts which kernel ran."""
    i = _layer_index(eng101, name)
    d = eng101.layers[i]
    if d.cin < 128:
        assert eng101._lib.mpx_set_conv_tile(eng101._h, i, 10) == -1
        return
    _check_layer(eng101, synth.make_state_dict("resnet101"), name, batch=batch, tile=10)
    mask = eng101._lib.mpx_last_conv_kernels(eng101._h)
    tiles = -(-batch * d.hout * d.hout // 128) * (d.cout // 256)
    assert mask == (1 << 10 if tiles >= eng101.num_cus else 1 << 7), (mask, tiles)       # under one round of tiles: the 128x128 8-wave kernel


@pytest.mark.parametrize("name,batch", [("layer3.5.conv3", 47), ("layer3.5.conv3", 161), ("layer3.5.conv3", 5), ("layer3.22.conv3", 84),
                                        ("layer2.1.conv3", 25), ("layer3.5.conv1", 9)])
def test_convw_weights_in_registers_kernel(eng101, name, batch):
    """Tile id 14 = the persistent expanding-1x1 kernel whose weights live in registers (csrc/mpx_convw.h; since round 5 with a column-major
    K loop and the epilogue slices between its MFMAs): K = 256 only (128 -> 512 and 1024 -> 256 are refused), 64-pixel tiles, one barrier per
    tile.  Batches give a ragged last tile (47 images = 143.9 pixel tiles, 161 = 493.06) and from 2 to 8 tiles per workgroup; 84 images are
    257.25 pixel tiles: workgroups with 4 and with 5; under two rounds of tiles (5 images) the launch runs on the 128x128 8-wave kernel.
    (The kernel is a template over K; K = 128 was built, passed this test bit-equal to tile 10 and tied with it: not instantiated.)  Against the
    fp64 conv + BN + residual + ReLU, and bit-equal to tile 10 (the same order of summation and the same epilogue arithmetic) -- the test
    asserts which kernel ran."""
    i = _layer_index(eng101, name)
    d = eng101.layers[i]
    if not (d.cin == 256 and d.cout % 256 == 0 and d.ksize == 1):
        assert eng101._lib.mpx_set_conv_tile(eng101._h, i, 14) == -1
        return
    _check_layer(eng101, synth.make_state_dict("resnet101"), name, batch=batch, tile=14)
    mask = eng101._lib.mpx_last_conv_kernels(eng101._h)
    tiles = -(-batch * d.hout * d.hout // 64) * (d.cout // 256)
    assert mask == (1 << 14 if tiles >= 2 * eng101.num_cus else 1 << 7), (mask, tiles)
    g = torch.Generator().manual_seed(100 + batch)
    x = torch.randn(batch, d.hin, d.hin, d.cin, generator=g).clamp_min(-0.5) * 1.5
    res = torch.randn(batch, d.hout, d.hout, d.cout, generator=g)
    outs = []
    for tile in (14, 10):
        eng101.set_conv_tile(i, tile)
        try:
            outs.append(_run_conv(eng101, i, x, res, batch, on_device=True)[0])
        finally:
            eng101.set_conv_tile(i, -1)
    assert torch.equal(outs[0], outs[1])
    # without a residual the weights-in-registers kernel is not what runs (round 6, ADVICE r5: only the residual + ReLU form every default layer
    # asks for is instantiated and tested): tile 14 hands the call to tile 10's kernel -- same bits, signed zeros included
    outs = []
    for tile in (14, 10):
        eng101.set_conv_tile(i, tile)
        try:
            outs.append(_run_conv(eng101, i, x, None, batch, on_device=True)[0])
            ran = eng101._lib.mpx_last_conv_kernels(eng101._h)
            assert not ran & (1 << 14), ran
        finally:
            eng101.set_conv_tile(i, -1)
    assert torch.equal(outs[0], outs[1]) and torch.equal(torch.signbit(outs[0]), torch.signbit(outs[1]))


def test_input_planes_getter_leaves_the_staging_record_alone(eng18):
    """mpx_input_planes used to mark every slot as K0-staged, so a diagnostic call between mpx_stem_table_apply and mpx_forward turned
    the forward onto stale input planes.  It is a pure getter now; a caller that writes the planes by hand says so with mpx_mark_input_staged."""
    img = synth.make_images(1, seed=31)[0]
    seg = synth.grid_segments(block=32)
    onoff = synth.random_onoff(6, 49, seed=9)
    label, _ = eng18.predict(img)
    _o, want, want_pred = eng18.score_masks(img, seg, onoff, label, stem="table")
    dev = eng18.device
    img_d, seg_d = torch.from_numpy(img).to(dev), torch.from_numpy(seg).to(dev)
    onoff_d = torch.from_numpy(onoff).to(dev)
    labels = torch.full((6,), int(label), dtype=torch.int32, device=dev)
    # stale planes: another picture staged through K0 first
    other = torch.from_numpy(synth.make_images(1, seed=32)[0]).to(dev)
    eng18.stage_masks(other, seg_d, onoff_d, 0)
    k0_hi = eng18.input_planes(6)[0].clone()
    eng18.build_stem_table(img_d, seg_d, 49)
    eng18.apply_stem_table(onoff_d, 0)
    hi, _lo = eng18.input_planes(6)                     # the diagnostic call in between
    assert torch.equal(hi, k0_hi)
    score, pred = eng18.forward(6, labels)
    assert np.array_equal(score.cpu().numpy(), want) and np.array_equal(pred.cpu().numpy(), want_pred)
    # the explicit marker: the same slots, declared hand-written -> the forward runs the stem on the (other picture's) input planes
    eng18.mark_input_staged(0, 6)
    score2, _p2 = eng18.forward(6, labels)
    _o, other_want, _p = eng18.score_masks(other.cpu().numpy(), seg, onoff, label, stem="conv")
    assert np.array_equal(score2.cpu().numpy(), other_want)
    assert eng18._lib.mpx_mark_input_staged(eng18._h, 60, 8) == -1 and eng18._lib.mpx_mark_input_staged(eng18._h, 0, 0) == -1
    # a mixed batch is still refused
    eng18.apply_stem_table(onoff_d[:3], 0)
    eng18.mark_input_staged(3, 3)
    with pytest.raises(MpxError):
        eng18.forward(6, labels)
    eng18.stage_masks(img_d, seg_d, onoff_d, 0)         # leave the engine in a clean state for the tests that follow
    eng18.forward(6, labels)


def test_stress_sweep_distinct_shapes(eng101):
    """A bounded slice of tools/stress_parity.py where the driver runs it: the 23 distinct conv shapes of ResNet-50 / 101 / 152
    (they share them; stem and fc have tests of their own) x three ragged batches x EVERY kernel tile id the layer is eligible for, each against the fp64 conv + BN
    (+ residual) (+ ReLU) on the same split inputs, <= 4e-6 relative."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    layers = sp.distinct_shape_layers(eng101)
    assert len(layers) == 23                              # (cin, cout, k, stride, map, residual) without the stem and fc
    n, worst = sp.sweep(eng101, synth.make_state_dict("resnet101"), layers, [1, 5, 9], seed=3)
    print("stress sweep: %d (layer, batch, tile) cases, worst relative error %.2e" % (n, worst))
    assert n >= 3 * 60 and worst <= 4e-6


def test_stress_sweep_trained_like_weights(mpx_lib, dev):
    """The same sweep with the conv + BatchNorm pairs of the trained-like ResNet-101 (oracle/trained_like.py: running variances 4e-6 .. 100,
    means up to 4 sigma, gammas -0.2 .. 1.6, calibrated weights with a common-mode component): every distinct shape x every eligible kernel
    tile against the fp64 conv + BN, the same 4e-6 relative bound as on the synthetic initialisation."""
    import importlib.util
    from oracle import trained_like
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    sd = trained_like.make_trained_like_state_dict("resnet101")
    eng = MaskedForwardEngine("resnet101", max_batch=8, device=0).load_state_dict(sd)
    try:
        layers = sp.distinct_shape_layers(eng)
        n, worst = sp.sweep(eng, sd, layers, [3, 7], seed=5)
        print("stress sweep, trained-like weights: %d (layer, batch, tile) cases, worst relative error %.2e" % (n, worst))
        assert n >= 2 * 60 and worst <= 4e-6
    finally:
        eng.close()


def test_heatmap_device_and_one_buffer_layout(eng18):
    """engine.heatmap_device (what shard.heatmap_sharded runs per rank): scores + K5 into a device buffer f32[224*224 + 1] whose last
    element counts the correct masks -- equal to the host-array path exactly; a second image accumulates on top."""
    from network_interpretation_imagenet_amd import shard
    from network_interpretation_imagenet_amd.engine import rank_segments
    img = synth.make_images(1, seed=12)[0]
    seg, s = rank_segments(synth.grid_segments(block=28))
    onoff = synth.random_onoff(37, s, seed=2)
    label, _ = eng18.predict(img)
    _o, score, pred = eng18.score_masks(img, seg, onoff, label)
    want = eng18.heatmap(seg, onoff, pred, label)
    buf = torch.zeros(224 * 224 + 1, dtype=torch.float32, device=eng18.device)
    s_d, p_d = eng18.heatmap_device(img, seg, onoff, label, buf)
    assert np.array_equal(s_d.cpu().numpy(), score) and np.array_equal(p_d.cpu().numpy(), pred)
    got = buf.cpu().numpy()
    assert np.array_equal(got[:-1].reshape(224, 224).astype(np.float64), want) and int(got[-1]) == int((pred == label).sum())
    heat, n_ok = shard.heatmap_sharded(eng18, img, seg, onoff, label)              # single process: same buffer, no collective
    assert heat.device == eng18.device and np.array_equal(heat.cpu().numpy().astype(np.float64), want) and n_ok == int(got[-1])
    eng18.heatmap_device(img, seg, onoff[:5], label, buf)
    assert int(buf[-1].item()) == int(got[-1]) + int((pred[:5] == label).sum())


@pytest.mark.parametrize("tile", [-1, 2, 7])
@pytest.mark.parametrize("stage", [1, 2, 3, 4])
def test_conv_with_fused_downsample(eng101, stage, tile):
    """mpx_conv_dual_bn_act: layerN.0.conv3 + layerN.0.downsample K-concatenated in one launch (the default path of
    mpx_forward) against relu(bn3(conv3(t2)) + bn_ds(conv_ds(x))) in fp64 on the same split inputs; ragged batch."""
    sd = synth.make_state_dict("resnet101")
    i = _layer_index(eng101, "layer%d.0.conv3" % stage)
    j = _layer_index(eng101, "layer%d.0.downsample.0" % stage)
    d3, dd = eng101.layers[i], eng101.layers[j]
    batch = 5
    g = torch.Generator().manual_seed(100 + stage)
    t2 = torch.randn(batch, d3.hin, d3.hin, d3.cin, generator=g).clamp_min(0) * 1.5
    x = torch.randn(batch, dd.hin, dd.hin, dd.cin, generator=g).clamp_min(0) * 1.5
    dev = eng101.device
    th, tl = split(t2.to(dev))
    xh, xl = split(x.to(dev))
    oh = torch.full((batch, d3.hout, d3.hout, d3.cout), float("nan"), dtype=torch.float16, device=dev)
    ol = torch.full_like(oh, float("nan"))
    eng101.set_conv_tile(i, tile)
    try:
        rc = eng101._lib.mpx_conv_dual_bn_act(eng101._h, i, _p(th), _p(tl), _p(xh), _p(xl), _p(oh), _p(ol), batch, eng101._stream())
        _lib.check(eng101._h, rc, "mpx_conv_dual_bn_act")
        torch.cuda.synchronize()
    finally:
        eng101.set_conv_tile(i, -1)

    def branch(d, inp):
        name, bn = d.name.decode(), d.bn_name.decode()
        y = F.conv2d(inp.double().permute(0, 3, 1, 2), sd[name + ".weight"].double(), None, d.stride, d.pad)
        sc = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
        return (y - sd[bn + ".running_mean"].double().view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + sd[bn + ".bias"].double().view(1, -1, 1, 1)

    want = F.relu(branch(d3, merge(th, tl).cpu()) + branch(dd, merge(xh, xl).cpu())).permute(0, 2, 3, 1)
    got = merge(oh, ol).cpu().double()
    assert not torch.isnan(got).any()
    err = (got - want).abs().max().item()
    assert err <= 4e-6 * max(want.abs().max().item(), 1.0), "stage %d: %.3e" % (stage, err)
    assert eng101._lib.mpx_conv_dual_bn_act(eng101._h, i + 1, _p(th), _p(tl), _p(xh), _p(xl), _p(oh), _p(ol), batch, None) == -1


def _bn_fp64(sd, d, y):
    bn = d.bn_name.decode()
    sc = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
    return (y - sd[bn + ".running_mean"].double().view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + sd[bn + ".bias"].double().view(1, -1, 1, 1)


def _conv_bn_fp64(sd, d, x_nchw):
    return _bn_fp64(sd, d, F.conv2d(x_nchw, sd[d.name.decode() + ".weight"].double(), None, d.stride, d.pad))


def _round_split(x64):
    """What a tensor becomes when it is stored as hi + lo fp16 planes (from its fp32 value)."""
    hi, lo = split(x64.float())
    return merge(hi, lo).double()


@pytest.mark.parametrize("k,batch,whole", [(0, 2, False), (1, 1, False), (1, 3, False), (2, 2, False), (1, 41, False), (0, 23, False), (2, 37, False),
                                           (0, 1, True), (0, 3, True), (0, 41, True)])
def test_bottleneck_tail_vs_fp64_chain(eng101, k, batch, whole):
    """mpx_bottleneck_tail (csrc/mpx_btail.h): conv2 -> conv3 + identity (k = 0: + the K-concatenated downsample branch) -> the next
    block's conv1 (k = 2: layer2.0.conv1, 128 channels) in ONE launch, against the fp64 chain of the same three (four) layers on
    the same split inputs, with t2 and the block output rounded to hi + lo where the layer-by-layer path stores them.  One image
    is 28 tiles; 23 .. 41 images are 644 .. 1148 tiles on 512 resident workgroups: tile boundaries, the weight ring running on
    across tiles, workgroups with one, two and three tiles.  whole (layer1.0 only): t1 planes = NULL -- the launch also runs the
    block's own conv1 on the patch of the block input (what mpx_forward does), checked against the chain with that conv in front."""
    sd = synth.make_state_dict("resnet101")
    tails = eng101.bottleneck_tails()
    assert len(tails) == 3
    c2, c3, ds, n1 = tails[k]
    d2, d3, dn = eng101.layers[c2], eng101.layers[c3], eng101.layers[n1]
    assert d2.name.decode() == "layer1.%d.conv2" % k and (ds >= 0) == (k == 0) and dn.cout == (128 if k == 2 else 64)
    g = torch.Generator().manual_seed(500 + 10 * k + batch)
    t1 = torch.randn(batch, 56, 56, 64, generator=g).clamp_min(0) * 1.5
    xc = 64 if ds >= 0 else 256
    x = torch.randn(batch, 56, 56, xc, generator=g).clamp_min(0) * 1.5
    dev = eng101.device
    th, tl = split(t1.to(dev))
    xh, xl = split(x.to(dev))
    nan = lambda c: torch.full((batch + 1, 56, 56, c), float("nan"), dtype=torch.float16, device=dev)
    oh, ol, zh, zl = nan(256), nan(256), nan(dn.cout), nan(dn.cout)
    rc = eng101._lib.mpx_bottleneck_tail(eng101._h, c2, None if whole else _p(th), None if whole else _p(tl), _p(xh), _p(xl), _p(oh), _p(ol),
                                         _p(zh), _p(zl), batch, eng101._stream())
    _lib.check(eng101._h, rc, "mpx_bottleneck_tail")
    torch.cuda.synchronize()
    assert torch.isnan(oh[batch]).all() and torch.isnan(zl[batch]).all()           # nothing written past the batch

    t1u, xu = merge(th, tl).cpu().double().permute(0, 3, 1, 2), merge(xh, xl).cpu().double().permute(0, 3, 1, 2)
    if whole:
        d1 = eng101.layers[c2 - 1]
        assert d1.name.decode() == "layer1.0.conv1"
        t1u = _round_split(F.relu(_conv_bn_fp64(sd, d1, xu)))
    t2 = _round_split(F.relu(_conv_bn_fp64(sd, d2, t1u)))
    ident = _conv_bn_fp64(sd, eng101.layers[ds], xu) if ds >= 0 else xu
    out = F.relu(_conv_bn_fp64(sd, d3, t2) + ident)
    z = F.relu(_conv_bn_fp64(sd, dn, _round_split(out)))
    got_out = merge(oh[:batch], ol[:batch]).cpu().double().permute(0, 3, 1, 2)
    got_z = merge(zh[:batch], zl[:batch]).cpu().double().permute(0, 3, 1, 2)
    assert not torch.isnan(got_out).any() and not torch.isnan(got_z).any()
    e_out = (got_out - out).abs().max().item() / max(out.abs().max().item(), 1.0)
    e_z = (got_z - z).abs().max().item() / max(z.abs().max().item(), 1.0)
    assert e_out <= 4e-6 and e_z <= 4e-6, "tail %d batch %d: out %.3e, next conv1 %.3e" % (k, batch, e_out, e_z)


def test_bottleneck_tail_argument_errors(eng101, eng18, dev):
    t = torch.zeros(8, dtype=torch.float16, device=dev)
    lib, h = eng101._lib, eng101._h
    c2 = eng101.bottleneck_tails()[1][0]
    assert lib.mpx_bottleneck_tail(h, c2 + 1, _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), 1, None) == -1     # not a conv2 of a tail
    assert lib.mpx_bottleneck_tail(h, c2, _p(t), _p(t), _p(t), None, _p(t), _p(t), _p(t), _p(t), 1, None) == -1
    assert lib.mpx_bottleneck_tail(h, c2, None, None, _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), 1, None) == -1       # no t1: only layer1.0 can compute it
    assert lib.mpx_bottleneck_tail(h, c2, _p(t), None, _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), 1, None) == -1       # half a plane pair
    assert lib.mpx_bottleneck_tail(h, c2, _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), _p(t), 0, None) == -1
    assert b"bottleneck_tail" in lib.mpx_last_error(h)
    # aliased planes: the layer-by-layer API lets a caller reuse a dead buffer, this launch does not (workgroups read t1's halo and
    # the identity while others write) -- rejected, not raced
    n = 56 * 56
    bufs = [torch.zeros(n * c, dtype=torch.float16, device=dev) for c in (64, 64, 256, 256, 256, 256, 64, 64)]
    ptrs = [_p(b) for b in bufs]
    for a, b in ((4, 2), (5, 3), (6, 0), (4, 5), (7, 1)):          # out = x, out_lo = x_lo, next = t1, out_hi = out_lo, next_lo = t1_lo
        q = list(ptrs)
        q[a] = q[b]
        assert lib.mpx_bottleneck_tail(h, c2, *q, 1, None) == -1 and b"overlap" in lib.mpx_last_error(h)
    q = list(ptrs)
    q[4] = C.c_void_p(bufs[2].data_ptr() + 256)                     # partial overlap of out_hi with x_hi
    assert lib.mpx_bottleneck_tail(h, c2, *q, 1, None) == -1 and b"x_hi and out_hi overlap" in lib.mpx_last_error(h)
    assert lib.mpx_bottleneck_tail(h, c2, *ptrs, 1, None) == 0     # distinct buffers: accepted
    torch.cuda.synchronize()
    assert eng18.bottleneck_tails() == []              # basic blocks have no such tail
    assert lib.mpx_bottleneck_tail_info(h, 3, None, None, None, None) == -1


def test_forward_block_tails_vs_layer_by_layer(eng101):
    """The same masks with layer1's block tails as single launches (default) and with round 2's launch plan (mask 1): two summation
    orders of the same arithmetic -- the scores agree to rounding, not bit for bit."""
    img = synth.make_images(1, seed=9, kind="noise")[0]
    seg = synth.grid_segments()
    onoff = synth.random_onoff(24, 196, seed=6)
    _o, s_bt, p_bt = eng101.score_masks(img, seg, onoff, 17)
    eng101.set_fusion(1)
    try:
        _o, s_plain, p_plain = eng101.score_masks(img, seg, onoff, 17)
    finally:
        eng101.set_fusion(True)
    assert np.abs(s_bt - s_plain).max() <= 2e-6 and (p_bt == p_plain).all()
    assert not (s_bt == s_plain).all()


def test_forward_fused_vs_unfused_downsample(eng101):
    """The same masks with the downsample fusion on (default) and off: two different summation orders of the same
    arithmetic, so the scores agree to rounding, not bit for bit."""
    img = synth.make_images(1, seed=8, kind="noise")[0]
    seg = synth.grid_segments()
    onoff = synth.random_onoff(24, 196, seed=5)
    _o, s_fused, p_fused = eng101.score_masks(img, seg, onoff, 17)
    eng101.set_fusion(False)
    try:
        _o, s_plain, p_plain = eng101.score_masks(img, seg, onoff, 17)
    finally:
        eng101.set_fusion(True)
    assert np.abs(s_fused - s_plain).max() <= 2e-6 and (p_fused == p_plain).all()
    assert not (s_fused == s_plain).all()          # they really are two code paths


@pytest.mark.parametrize("name,unfused_mask", [("layer1.1.conv3", 1), ("layer1.0.conv3", 1), ("layer1.0.downsample.0", 0),
                                               ("layer3.0.conv3", 0), ("layer2.0.conv1", 1)])
def test_reloading_one_layer_rebuilds_what_was_derived_from_it(mpx_lib, dev, name, unfused_mask):
    """mpx_set_conv_weights on ONE layer after the engine is complete (mpx_api.hip: the tail's permuted conv3 copy, the
    K-concatenated conv3 | downsample planes and the tail copy of THOSE are derived tensors): the fused forward must follow the
    new weights exactly as the layer-by-layer forward does -- within 2e-6 of it before and after, different from before, and
    bit-equal to an engine that was loaded with the new state_dict from the start."""
    arch = "resnet50"
    sd = synth.make_state_dict(arch)
    rng = np.random.default_rng(3)
    sd2 = dict(sd)
    w = sd[name + ".weight"]
    sd2[name + ".weight"] = (w * torch.from_numpy(rng.uniform(0.5, 1.5, size=tuple(w.shape)).astype(np.float32))).contiguous()
    img = synth.make_images(1, seed=31, kind="noise")[0]
    seg = synth.grid_segments()
    onoff = synth.random_onoff(12, 196, seed=32)
    eng = MaskedForwardEngine(arch, max_batch=12, device=0).load_state_dict(sd)
    fresh = MaskedForwardEngine(arch, max_batch=12, device=0).load_state_dict(sd2)
    try:
        def both():
            _o, s_f, p_f = eng.score_masks(img, seg, onoff, 17)
            eng.set_fusion(unfused_mask)
            try:
                _o, s_u, p_u = eng.score_masks(img, seg, onoff, 17)
            finally:
                eng.set_fusion(True)
            assert np.abs(s_f - s_u).max() <= 2e-6 and (p_f == p_u).all()
            return s_f, s_u
        before_f, before_u = both()
        eng.load_state_dict(sd2, only=[name])
        after_f, after_u = both()
        assert not (after_f == before_f).all() and not (after_u == before_u).all()       # the reload is visible on both paths
        _o, want, _p2 = fresh.score_masks(img, seg, onoff, 17)
        assert (after_f == want).all(), "fused planes were not rebuilt from the reloaded %s" % name
        eng.load_state_dict(sd, only=[name])                                               # and back
        _o, again, _p3 = eng.score_masks(img, seg, onoff, 17)
        assert (again == before_f).all()
    finally:
        eng.close()
        fresh.close()


@pytest.mark.parametrize("which", ["resnet18", "resnet101"])
def test_score_images_packed_equals_per_image(eng18, eng101, golden_dir, which):
    """MaskedForwardEngine.score_images / score_packed: the mask rows of consecutive images share forward batches (rows of one
    image straddle two batches; images with different label maps, one of them felzenszwalb's; u8 and normalised f32 inputs).
    The packed scores are BIT-equal to scoring every image on its own, and sampled slots are within 2e-5 of the oracle's
    batch-1 loop (generate_gp_training_data_imagenet.py:221-266)."""
    eng = eng18 if which == "resnet18" else eng101
    sd = synth.make_state_dict(which)
    imgs = synth.make_images(4, seed=31, kind="blobs")
    felz = np.load(os.path.join(golden_dir, "segments_blobs.npz"))["segments"][0].astype(np.int32)
    segs = [synth.grid_segments(), felz, synth.grid_segments(block=32), synth.grid_segments()]
    sizes = [eng.max_batch // 2 + 3, eng.max_batch // 2 + 5, 7, eng.max_batch + 2]          # straddles, a small one, more than one batch
    onoffs = [synth.random_onoff(m, int(sg.max()) + 1, seed=40 + i) for i, (m, sg) in enumerate(zip(sizes, segs))]
    labels = [3, 17, 999, 500]
    inputs = [imgs[0], scorer.to_tensor_normalize(imgs[1]), imgs[2], imgs[3]]
    packed = eng.score_images(inputs, segs, onoffs, labels)
    for i in range(4):
        _o, s1, p1 = eng.score_masks(inputs[i], segs[i], onoffs[i], labels[i])
        assert np.array_equal(packed[i][0], s1) and np.array_equal(packed[i][1], p1), "image %d" % i
    for i, rows in ((0, [0, sizes[0] - 1]), (1, [4]), (3, [eng.max_batch + 1])):
        x = scorer.to_tensor_normalize(imgs[i])
        ref_s, ref_p = scorer.score_masks_reference_loop(sd, which, x, segs[i], onoffs[i][rows], labels[i])
        assert np.abs(packed[i][0][rows] - ref_s).max() <= SCORE_TOL_TIGHT and (packed[i][1][rows] == ref_p).all()
    assert eng.score_images([], [], [], []) == []
    with pytest.raises(ValueError):
        eng.score_images(inputs[:1], segs[:1], [onoffs[1]], labels[:1])            # S of the label map and of the rows differ


def test_api_fill_tables_on_the_engine(eng18):
    """api.fill_tables (what validate_many / validate_summed_many run): unmasked row + every window of several images in packed
    batches -- the tables are bit-equal to SaliencySession.table() one image at a time, the base predictions to predict()."""
    from network_interpretation_imagenet_amd import api
    imgs = synth.make_images(3, seed=77, kind="blobs")
    xs = [scorer.to_tensor_normalize(im) for im in imgs]
    segs = [synth.grid_segments(block=32), synth.grid_segments(block=16), synth.grid_segments(block=56)]
    base = [eng18.predict(x)[0] for x in xs]
    targets = [base[0], (base[1] + 1) % 1000, base[2]]
    packed = [api.SaliencySession(eng18, x, t, segments=sg, check_base=False) for x, t, sg in zip(xs, targets, segs)]
    assert api.fill_tables(eng18, packed) == [True, False, True]
    for s, x, sg, b in zip(packed, xs, segs, base):
        one = api.SaliencySession(eng18, x, b, segments=sg)
        assert s.base_pred == b and np.array_equal(s.table()[1], one.table()[1])         # argmax does not depend on the label
        assert s.label != b or np.array_equal(s.table()[0], one.table()[0])               # softmax[label] does


@pytest.mark.parametrize("arch,n_masks", [("resnet18", 24), ("resnet101", 12)])
def test_trained_like_batchnorm_statistics_end_to_end(mpx_lib, dev, arch, n_masks):
    """Round 6: ImageNet-depth networks whose EVERY BatchNorm carries trained-like statistics (oracle/trained_like.py: per-channel
    running_mean / running_var / gamma / beta resampled from the reference's shipped CIFAR checkpoint -- variances 4e-6 .. 11, non-zero means,
    nearly dead channels -- with the conv weights calibrated so that the statistics are true for the network).  The synthetic initialisation
    (mean ~ 0, var ~ 1) never decided a precision gate, trained statistics did: the engine's f16x3 arithmetic must hold the
    north-star tolerance on them too, through both stagings, against the batch-1 fp32 CPU loop and the fp64 yardstick."""
    from oracle import trained_like
    sd = trained_like.make_trained_like_state_dict(arch)
    eng = MaskedForwardEngine(arch, max_batch=32, device=0).load_state_dict(sd)
    try:
        seg = synth.grid_segments()
        worst = 0.0
        for kind, seed in (("blobs", 61), ("noise", 62)):
            img = synth.make_images(1, seed=seed, kind=kind)[0]
            onoff = synth.random_onoff(n_masks, 196, seed=seed, p=0.7)
            x = scorer.to_tensor_normalize(img)
            label, _ = eng.predict(img)
            ref_score, ref_pred = scorer.score_masks_reference_loop(sd, arch, x, seg, onoff, label)
            ref64, pred64 = scorer.score_masks_batched(sd, arch, x, seg, onoff, label, dtype=torch.float64, chunk=8)
            for stem in ("conv", "table"):
                _o, score, pred = eng.score_masks(img, seg, onoff, label, stem=stem)
                d32 = float(np.abs(score.astype(np.float64) - ref_score.astype(np.float64)).max())
                d64 = float(np.abs(score.astype(np.float64) - ref64).max())
                print("trained-like %s %s stem=%s: scores %.3f..%.3f  max|d| vs fp32 loop %.2e, vs fp64 %.2e (fp32 loop vs fp64 %.2e)" % (
                    arch, kind, stem, ref64.min(), ref64.max(), d32, d64, float(np.abs(ref_score - ref64).max())))
                assert d32 <= SCORE_TOL and d64 <= SCORE_TOL
                # argmax: equal wherever the fp64 top-2 margin is not a rounding tie
                assert (pred == pred64).all() or d64 < 1e-6
                worst = max(worst, d64)
        assert worst <= SCORE_TOL_TIGHT, worst
    finally:
        eng.close()


def _stripes(s):
    """i32[224,224] label map of exactly s raster stripes (S is free, unlike a block grid's)."""
    return (np.arange(224 * 224, dtype=np.int64) * s // (224 * 224)).reshape(224, 224).astype(np.int32)


def test_an_image_gets_the_same_bits_alone_and_packed_across_the_256_row_line(eng18):
    """staging (stem table from 256 rows on, K0 + the MFMA stem below; they round differently) is a property of the IMAGE.
    S = 300 (302 rows -> table) packed with S = 60 (62 rows -> conv) and S = 254 (256 rows: on the line) through api.fill_tables and
    engine.score_images must carry the bits each image gets alone -- the binary labels (generate_gp_training_data_imagenet.py:248,257) and the
    heat-map sum (gp_superpixel_data_imagenet.py:322-323) must not depend on lookahead or grouping."""
    from network_interpretation_imagenet_amd import api
    sizes = [300, 60, 254]
    imgs = synth.make_images(3, seed=91, kind="blobs")
    xs = [scorer.to_tensor_normalize(im) for im in imgs]
    segs = [_stripes(s) for s in sizes]
    base = [eng18.predict(x)[0] for x in xs]
    assert [eng18.stem_for_rows(s + 2) for s in sizes] == ["table", "conv", "table"]
    alone = [api.SaliencySession(eng18, x, b, segments=sg) for x, b, sg in zip(xs, base, segs)]          # check_base: S + 2 rows
    packed = [api.SaliencySession(eng18, x, b, segments=sg, check_base=False) for x, b, sg in zip(xs, base, segs)]
    assert api.fill_tables(eng18, packed) == [True, True, True]
    for a, p in zip(alone, packed):
        assert a.stem == p.stem and p.base_pred == a.base_pred
        assert np.array_equal(a.table()[0], p.table()[0]) and np.array_equal(a.table()[1], p.table()[1])
    # the other packing orders and the raw packed entry: still each image's own bits
    rows = [np.concatenate([np.ones((1, s), np.uint8), masks.windows_onoff(s, range(0, s + 1))]) for s in sizes]
    for order in ([1, 0, 2], [2, 1, 0], [0, 2, 1]):
        res = eng18.score_images([xs[k] for k in order], [segs[k] for k in order], [rows[k] for k in order], [base[k] for k in order])
        for k, (score, pred) in zip(order, res):
            assert np.array_equal(score[1:], alone[k].table()[0]) and np.array_equal(pred[1:], alone[k].table()[1])
    # S = 254: S + 1 = 255 rows without the unmasked row, 256 with it -- one staging either way, and for a window outside the table
    lazy = api.SaliencySession(eng18, xs[2], base[2], segments=segs[2], check_base=False)
    assert lazy.stem == "table" and np.array_equal(lazy.table()[0], alone[2].table()[0])
    out_of_table = lazy.score(-3)
    _o, want, _p = eng18.score_masks(xs[2], segs[2], masks.windows_onoff(254, [-3]), base[2], stem="table")
    assert out_of_table[0] == want[0]
    # and the two stagings DO differ in bits somewhere on these rows (otherwise this test would not guard anything)
    _o, conv_bits, _p = eng18.score_masks(xs[0], segs[0], rows[0][1:], base[0], stem="conv")
    assert not np.array_equal(conv_bits, alone[0].table()[0]) and np.abs(conv_bits - alone[0].table()[0]).max() < 1e-5


TRAINED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_layers_cifar_resnet56.npz")


@pytest.fixture(scope="module")
def eng18_trained(mpx_lib, dev):
    sd = synth.transplant_trained_layers(synth.make_state_dict("resnet18"), TRAINED)
    e = MaskedForwardEngine("resnet18", max_batch=64, device=0).load_state_dict(sd)
    yield e, sd
    e.close()


@pytest.mark.parametrize("tile", [-1, 0, 2, 4])
@pytest.mark.parametrize("name", ["layer1.0.conv1", "layer1.0.conv2", "layer1.1.conv1", "layer1.1.conv2"])
def test_conv_trained_cifar_weights(eng18_trained, name, tile):
    """Real trained conv + BN pairs (reference checkpoint cifar10+-resnet-56, layer3.4 / layer3.8; BN variances from
    0.005 to 11) through the conv kernel, against the fp64 conv on the same split inputs."""
    eng, sd = eng18_trained
    _check_layer(eng, sd, name, batch=3, tile=tile)


def test_trained_layers_end_to_end_scores(eng18_trained):
    """ResNet-18 with its layer1 replaced by the trained CIFAR pairs: masked scores against the CPU loop."""
    eng, sd = eng18_trained
    img = synth.make_images(1, seed=5)[0]
    seg = synth.grid_segments()
    onoff = synth.random_onoff(12, 196, seed=2)
    label, _ = eng.predict(img)
    _o, score, pred = eng.score_masks(img, seg, onoff, label)
    ref_score, ref_pred = scorer.score_masks_reference_loop(sd, "resnet18", scorer.to_tensor_normalize(img), seg, onoff, label)
    assert np.abs(score - ref_score).max() <= SCORE_TOL_TIGHT
    assert (pred == ref_pred).all()


@pytest.mark.parametrize("batch", [1, 3, 11])
@pytest.mark.parametrize("name", ["layer1.0.conv2", "layer1.2.conv2", "layer2.1.conv2", "layer2.3.conv2", "layer3.5.conv2", "layer3.22.conv2",
                                  "layer4.1.conv2", "layer4.2.conv2"])
def test_conv_patch_kernel(eng101, name, batch):
    """Tile id 6 = the 3x3 patch kernel (csrc/mpx_conv3p.h: input patch staged once per 32-channel chunk, taps are LDS
    row offsets), ragged batches so that tiles start anywhere in an image and the last tile is partial."""
    if batch > eng101.max_batch:
        pytest.skip("engine too small")
    _check_layer(eng101, synth.make_state_dict("resnet101"), name, batch=batch, tile=6)


def _kernels_ran(eng):
    """Tile ids of the kernels the last mpx_conv_bn_act call launched (mpx_last_conv_kernels)."""
    mask = eng._lib.mpx_last_conv_kernels(eng._h)
    return {t for t in range(32) if mask >> t & 1}


def _tiles_of(eng, d, batch, tp, tc):
    """Tiles of a launch: pixel tiles of `tp` x cout tiles of `tc`."""
    return -(-batch * d.hout * d.hout // tp) * (d.cout // tc)


@pytest.mark.parametrize("batch", [1, 11, 41, 335, 392, 523])
@pytest.mark.parametrize("name", ["layer2.1.conv2", "layer3.5.conv2", "layer3.22.conv2", "layer4.1.conv2"])
def test_conv_persistent_patch_kernel(eng101, name, batch):
    """Tile id 12 = the patch kernel as one persistent workgroup per CU (csrc/mpx_conv3pp.h): weight ring and patch buffers run on
    across the tiles of a workgroup, register epilogue, the next tile's geometry computed inside the K loop with float-reciprocal
    divisions.  A launch with fewer tiles than CUs runs on tile 6 itself (launch_conv_patchp: batches 1 .. 41 on 14x14 / 7x7 maps,
    1 .. 11 on 28x28) -- those cases check the fallback rule; the larger batches stay on the persistent walk with UNEVEN tile counts
    per workgroup: 256 -> 256 on 14x14 maps (two cout tiles) 335 images = 257 x 2 tiles, 392 = 301 x 2, 523 = 401 x 2;
    128 -> 128 on 28x28 maps 335 images = 1026 tiles; 512 -> 512 on 7x7 maps (192-pixel tiles, four cout tiles) 41 images = 44 and
    335 = 344 tiles.  Every case asserts which kernel ran.  Against the fp64 conv + BN for the small batches, and
    BIT-identical to tile 6 for all of them (mpx_conv_bn_act takes any batch: the planes are the caller's)."""
    sd = synth.make_state_dict("resnet101")
    if batch <= 11:
        _check_layer(eng101, sd, name, batch=batch, tile=12)
    i = _layer_index(eng101, name)
    d = eng101.layers[i]
    if d.hout == 28 and batch > 347:
        pytest.skip("28x28 maps: 347 images are 1063 tiles already")
    # (drawn on the device: the two kernels are compared with each other, and 523 images of a CPU randn cost more than the launches)
    x = torch.randn(batch, d.hin, d.hin, d.cin, device=eng101.device, generator=torch.Generator(device=eng101.device).manual_seed(7)).clamp_min(-0.5)
    outs, ran = [], []
    for tile in (6, 12):
        eng101.set_conv_tile(i, tile)
        try:
            outs.append(_run_conv(eng101, i, x, None, batch, on_device=True)[0])
            ran.append(_kernels_ran(eng101))
        finally:
            eng101.set_conv_tile(i, -1)
    tp, tc = (192, 128) if d.hout == 7 else (256, 128)          # PatchTile2 on 7x7 maps, PatchTile0 elsewhere
    persistent = _tiles_of(eng101, d, batch, tp, tc) >= eng101.num_cus
    assert ran[0] == {6} and ran[1] == ({12} if persistent else {6}), (ran, _tiles_of(eng101, d, batch, tp, tc))
    if batch >= 335:
        assert persistent, "this batch is meant to stay on the persistent walk"
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("batch", [1, 41, 335, 523, 700, 1013])
@pytest.mark.parametrize("name", ["layer2.1.conv1", "layer3.5.conv1", "layer4.1.conv1"])
def test_conv_persistent_256_kernel(eng101, name, batch):
    """Tile id 13 = the 256x256 kernel as one persistent workgroup per CU (csrc/mpx_conv256p.h): the two-stage ring runs on across
    the tiles of a workgroup (pixel descriptor switched when the fill wraps), register epilogue whose stores retire under the next
    tile's first step (vmcnt(32) at its rendezvous).  512 -> 128 is not eligible (cout % 256).  A launch with fewer tiles than CUs
    runs on the 128x128 kernel, which sums in the same order (launch_conv256p); from one round on the persistent walk runs, and when
    a small last round is left (launch_conv: rest <= half the CUs) the images behind the whole rounds go to the 128x128 kernel too:
    1024 -> 256 on 14x14 at 335 / 700 images = 257 / 536 tiles -> the whole rounds persistent + the rest on tile 2; 523
    images = 401 tiles (rest 145 > half the CUs) -> all on the persistent walk with uneven tile counts per workgroup; 2048 -> 512 on
    7x7 (two cout tiles) at 700 images = 268 tiles (split) and 1013 = 388 (all persistent).  Every case asserts which kernels ran.  Against the fp64 conv + BN for the small batches, BIT-identical to tile 9 for all."""
    sd = synth.make_state_dict("resnet101")
    i = _layer_index(eng101, name)
    d = eng101.layers[i]
    if d.cout % 256:
        assert eng101._lib.mpx_set_conv_tile(eng101._h, i, 13) == -1
        return
    if batch <= 11:
        _check_layer(eng101, sd, name, batch=batch, tile=13)
    x = torch.randn(batch, d.hin, d.hin, d.cin, device=eng101.device, generator=torch.Generator(device=eng101.device).manual_seed(11)).clamp_min(-0.5)
    outs, ran = [], []
    for tile in (9, 13):
        eng101.set_conv_tile(i, tile)
        try:
            outs.append(_run_conv(eng101, i, x, None, batch, on_device=True)[0])
            ran.append(_kernels_ran(eng101))
        finally:
            eng101.set_conv_tile(i, -1)
    total, cus = _tiles_of(eng101, d, batch, 256, 256), eng101.num_cus
    rest = total % cus
    split = total >= cus and 0 < rest <= cus // 2
    if total < cus:
        want9, want13 = {9}, {2}
    else:
        want9, want13 = ({9, 2}, {13, 2}) if split else ({9}, {13})
    assert ran[0] == want9 and ran[1] == want13, (ran, total)
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[0], outs[1])


def test_conv_persistent_patch_kernel_eligibility(eng101, eng18):
    for name in ("layer1.0.conv2", "layer1.0.conv1", "layer2.0.conv2", "conv1"):       # cout 64, 1x1, stride 2, stem
        i = _layer_index(eng101, name)
        assert eng101._lib.mpx_set_conv_tile(eng101._h, i, 12) == -1
    i = _layer_index(eng18, "layer3.1.conv2")          # BasicBlock conv2: residual operand
    assert eng18._lib.mpx_set_conv_tile(eng18._h, i, 12) == -1
    i = _layer_index(eng18, "layer3.1.conv1")
    eng18.set_conv_tile(i, 12)
    eng18.set_conv_tile(i, -1)


def test_conv_patch_kernel_eligibility(eng101):
    for name in ("layer1.0.conv1", "layer2.0.conv2", "layer4.0.conv2", "conv1"):       # 1x1, stride 2, stride 2, stem
        i = _layer_index(eng101, name)
        assert eng101._lib.mpx_set_conv_tile(eng101._h, i, 6) == -1
        assert eng101.conv_tile(i) != 6


def test_set_conv_tile_errors(eng18):
    assert eng18._lib.mpx_set_conv_tile(eng18._h, 999, 0) == -1
    assert eng18._lib.mpx_set_conv_tile(eng18._h, 1, 17) == -1
    assert eng18._lib.mpx_get_conv_tile(eng18._h, 999) == -1
    default = eng18.conv_tile(1)
    eng18.set_conv_tile(1, 4)
    assert eng18.conv_tile(1) == 4
    eng18.set_conv_tile(1, -1)
    assert eng18.conv_tile(1) == default


def test_stem_conv_from_staged_input(eng18, dev):
    sd = synth.make_state_dict("resnet18")
    img = synth.make_images(1, kind="noise")[0]
    seg = synth.grid_segments()
    onoff = synth.random_onoff(3, 196, seed=2)
    eng18.stage_masks(torch.from_numpy(img).to(dev), torch.from_numpy(seg).to(dev), torch.from_numpy(onoff).to(dev), 0)
    d = eng18.layers[0]
    oh = torch.full((3, 112, 112, 64), float("nan"), dtype=torch.float16, device=dev)
    ol = torch.full_like(oh, float("nan"))
    rc = eng18._lib.mpx_conv_bn_act(eng18._h, 0, None, None, None, None, _p(oh), _p(ol), None, 3, eng18._stream())
    _lib.check(eng18._h, rc, "stem")
    torch.cuda.synchronize()
    x = scorer.to_tensor_normalize(img)
    xin = torch.from_numpy(np.stack([scorer.apply_mask(x, scorer.onoff_mask_u8(seg, onoff[i])) for i in range(3)]))
    want = _conv_reference(sd, d, xin.double(), None).permute(0, 2, 3, 1)
    got = merge(oh, ol).cpu().double()
    assert (got - want).abs().max().item() <= 4e-6 * max(want.abs().max().item(), 1.0)


def test_stem_conv_with_max_pool_in_one_launch(eng18, dev):
    """mpx_stem_conv_maxpool: every workgroup computes the 15 x 17 conv outputs under a 7 x 8 block of pooled pixels and pools them
    from its fp32 tile -- bit-identical to mpx_conv_bn_act (layer 0) followed by mpx_maxpool3x3s2, on every pixel of a ragged batch
    (image borders = max-pool padding, tile borders = recomputed conv rows / columns)."""
    img = synth.make_images(1, seed=3, kind="noise")[0]
    seg = synth.grid_segments()
    batch = 5
    onoff = synth.random_onoff(batch, 196, seed=4)
    eng18.stage_masks(torch.from_numpy(img).to(dev), torch.from_numpy(seg).to(dev), torch.from_numpy(onoff).to(dev), 0)
    lib, h = eng18._lib, eng18._h
    ch = torch.full((batch, 112, 112, 64), float("nan"), dtype=torch.float16, device=dev)
    cl = torch.full_like(ch, float("nan"))
    _lib.check(h, lib.mpx_conv_bn_act(h, 0, None, None, None, None, _p(ch), _p(cl), None, batch, eng18._stream()), "stem")
    ph = torch.full((batch, 56, 56, 64), float("nan"), dtype=torch.float16, device=dev)
    pl = torch.full_like(ph, float("nan"))
    _lib.check(h, lib.mpx_maxpool3x3s2(h, _p(ch), _p(cl), _p(ph), _p(pl), batch, 112, 64, eng18._stream()), "maxpool")
    fh = torch.full((batch + 1, 56, 56, 64), float("nan"), dtype=torch.float16, device=dev)      # one image of slack: nothing may be written there
    fl = torch.full_like(fh, float("nan"))
    _lib.check(h, lib.mpx_stem_conv_maxpool(h, _p(fh), _p(fl), batch, eng18._stream()), "stem + pool")
    torch.cuda.synchronize()
    assert torch.equal(fh[:batch], ph) and torch.equal(fl[:batch], pl)
    assert torch.isnan(fh[batch]).all() and torch.isnan(fl[batch]).all()
    assert lib.mpx_stem_conv_maxpool(h, None, _p(fl), batch, None) == -1
    assert lib.mpx_stem_conv_maxpool(h, _p(fh), _p(fl), 10 ** 6, None) == -2


def test_forward_with_and_without_fusions(mpx_lib, dev):
    """mpx_set_fusion(0) runs stem / max pool and conv3 / downsample as separate launches: the stem fusion is bit-identical, the
    downsample fusion changes the summation order (scores agree to rounding).  ResNet-18 has no 1x1-downsample fusion partner
    eligible for K-concatenation with a 3x3 main conv, so there the whole forward must be bit-identical."""
    eng = MaskedForwardEngine("resnet18", max_batch=64, device=0).load_state_dict(synth.make_state_dict("resnet18"))
    try:
        img = synth.make_images(1, seed=21, kind="noise")[0]
        seg = synth.grid_segments()
        onoff = synth.random_onoff(64, 196, seed=9)
        _o, s_on, p_on = eng.score_masks(img, seg, onoff, 5)
        eng.set_fusion(False)
        _o, s_off, p_off = eng.score_masks(img, seg, onoff, 5)
    finally:
        eng.close()
    assert np.array_equal(s_on, s_off) and np.array_equal(p_on, p_off)
    assert np.isfinite(s_on).all() and s_on.std() > 0


def test_conv_argument_errors(eng18, dev):
    t = torch.zeros(8, dtype=torch.float16, device=dev)
    lib, h = eng18._lib, eng18._h
    assert lib.mpx_conv_bn_act(h, 999, _p(t), _p(t), None, None, _p(t), _p(t), None, 1, None) == -1
    assert lib.mpx_conv_bn_act(h, 1, None, None, None, None, _p(t), _p(t), None, 1, None) == -1     # missing input
    assert lib.mpx_conv_bn_act(h, 1, _p(t), _p(t), _p(t), None, _p(t), _p(t), None, 1, None) == -1  # half a residual
    assert lib.mpx_conv_bn_act(h, 0, _p(t), _p(t), None, None, _p(t), _p(t), None, 1, None) == -1   # stem takes staging
    assert b"conv_bn_act" in lib.mpx_last_error(h)


# ------------------------------------------------------------------------------------------------
# pools and head
# ------------------------------------------------------------------------------------------------
def test_maxpool_exact(eng18, dev):
    x = torch.randn(3, 112, 112, 64, generator=torch.Generator().manual_seed(0)).clamp_min(0)
    xh, xl = split(x.to(dev))
    oh = torch.empty(3, 56, 56, 64, dtype=torch.float16, device=dev)
    ol = torch.empty_like(oh)
    _lib.check(eng18._h, eng18._lib.mpx_maxpool3x3s2(eng18._h, _p(xh), _p(xl), _p(oh), _p(ol), 3, 112, 64, eng18._stream()), "maxpool")
    torch.cuda.synchronize()
    want = F.max_pool2d(merge(xh, xl).cpu().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(merge(oh, ol).cpu(), want)


def test_global_avgpool(eng18, dev):
    x = torch.randn(5, 7, 7, 512, generator=torch.Generator().manual_seed(1)).clamp_min(0)
    xh, xl = split(x.to(dev))
    oh = torch.empty(5, 512, dtype=torch.float16, device=dev)
    ol = torch.empty_like(oh)
    _lib.check(eng18._h, eng18._lib.mpx_global_avgpool(eng18._h, _p(xh), _p(xl), _p(oh), _p(ol), 5, 49, 512, eng18._stream()), "avgpool")
    torch.cuda.synchronize()
    want = merge(xh, xl).cpu().double().mean(dim=(1, 2))
    assert (merge(oh, ol).cpu().double() - want).abs().max().item() <= 1e-6


def test_head_softmax_gather(eng18, dev):
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(9, 1000, generator=g) * 4
    logits[2, 17] = logits[2].max() + 1
    label = torch.randint(0, 1000, (9,), generator=g, dtype=torch.int32)
    label[2] = 17
    score = torch.empty(9, dtype=torch.float32, device=dev)
    pred = torch.empty(9, dtype=torch.int32, device=dev)
    ld, lb = logits.to(dev), label.to(dev)
    _lib.check(eng18._h, eng18._lib.mpx_head_softmax_gather(eng18._h, _p(ld), _p(lb), _p(score), _p(pred), 9, eng18._stream()), "head")
    torch.cuda.synchronize()
    want = torch.softmax(logits.double(), 1)[torch.arange(9), label.long()]
    assert (score.cpu().double() - want).abs().max().item() <= 2e-7
    assert torch.equal(pred.cpu().long(), logits.argmax(1))


# ------------------------------------------------------------------------------------------------
# end to end: golden vectors, live oracle, size-independent properties
# ------------------------------------------------------------------------------------------------
def _golden_case(arch, golden_dir):
    g = np.load(os.path.join(golden_dir, "scores_%s.npz" % arch))
    seg = np.load(os.path.join(golden_dir, "segments_blobs.npz"))["segments"][int(g["image_index"])].astype(np.int64)
    img = synth.make_images(2, seed=int(g["image_seed"]))[int(g["image_index"])]
    return g, seg, img


def test_cfg1_resnet18_64_masks_vs_golden(eng18, golden_dir):
    """BASELINE cfg-1: ResNet-18, 1 image, 64 superpixel masks."""
    g, seg, img = _golden_case("resnet18", golden_dir)
    label = int(g["label"])
    assert eng18.predict(img)[0] == label
    onoff, score, pred = eng18.score_masks(img, seg, g["onoff"], label)
    err = np.abs(score - g["score_f32"]).max()
    err64 = np.abs(score.astype(np.float64) - g["score_f64"]).max()
    print("resnet18 cfg-1: max|d| vs fp32 loop %.3e, vs fp64 %.3e" % (err, err64))
    assert err <= SCORE_TOL and err <= SCORE_TOL_TIGHT
    assert (pred == g["pred"]).all()
    assert score.dtype == np.float32 and pred.dtype == np.int32 and onoff is not None


def test_resnet101_vs_golden(eng101, golden_dir):
    g, seg, img = _golden_case("resnet101", golden_dir)
    label = int(g["label"])
    assert eng101.predict(img)[0] == label
    _o, score, pred = eng101.score_masks(img, seg, g["onoff"], label)
    err = np.abs(score - g["score_f32"]).max()
    print("resnet101: max|d| vs fp32 loop %.3e, vs fp64 %.3e" % (err, np.abs(score - g["score_f64"]).max()))
    assert err <= SCORE_TOL and err <= SCORE_TOL_TIGHT
    assert (pred == g["pred"]).all()


def test_resnet18_vs_live_oracle_noise_image(eng18):
    """Same masks through the normalised-f32 entry (what the reference's val_loader yields)."""
    sd = synth.make_state_dict("resnet18")
    img = synth.make_images(3, seed=77, kind="noise")[2]
    x = scorer.to_tensor_normalize(img)
    seg = synth.grid_segments()
    onoff = masks.windows_onoff(196, [0, 1, 60, 118, 196])
    label = scorer.base_prediction(sd, "resnet18", x)
    ref, ref_pred = scorer.score_masks_reference_loop(sd, "resnet18", x, seg, onoff, label)
    _o, s_u8, p_u8 = eng18.score_masks(img, seg, onoff, label)
    _o, s_f32, p_f32 = eng18.score_masks(x, seg, onoff, label)
    assert np.abs(s_u8 - ref).max() <= SCORE_TOL_TIGHT and np.abs(s_f32 - ref).max() <= SCORE_TOL_TIGHT
    assert (s_u8 == s_f32).all()                                   # both entries stage identical bits
    assert (p_u8 == ref_pred).all() and (p_f32 == ref_pred).all()


@pytest.mark.parametrize("arch", ["resnet34", "resnet50", "resnet152"])
def test_other_depths_vs_live_oracle(mpx_lib, arch):
    """The remaining torchvision ResNet topologies the reference's `-a` flag can name (BasicBlock 3-4-6-3 and
    Bottleneck 3-4-6-3): 6 masks against the reference-style CPU loop."""
    sd = synth.make_state_dict(arch)
    eng = MaskedForwardEngine(arch, max_batch=8, device=0).load_state_dict(sd)
    try:
        img = synth.make_images(1, seed=3)[0]
        x = scorer.to_tensor_normalize(img)
        seg = synth.grid_segments(block=32)                       # 49 superpixels
        onoff = np.concatenate([masks.windows_onoff(49, [0, 7, 30]), synth.random_onoff(3, 49, seed=1)])
        label = scorer.base_prediction(sd, arch, x)
        assert eng.predict(img)[0] == label
        ref, ref_pred = scorer.score_masks_reference_loop(sd, arch, x, seg, onoff, label)
        _o, score, pred = eng.score_masks(img, seg, onoff, label)
        assert np.abs(score - ref).max() <= SCORE_TOL_TIGHT and (pred == ref_pred).all()
    finally:
        eng.close()


def test_properties_full_batch(eng18):
    """Size-independent properties at the engine's full batch (64) and beyond it (chunking)."""
    imgs = synth.make_images(2, seed=5, kind="noise")
    seg = synth.grid_segments()
    label = 123
    onoff = synth.random_onoff(150, 196, seed=11)          # > max_batch: three chunks
    onoff[0] = 1
    onoff[1] = 0
    onoff[70] = onoff[3]                                    # duplicates in different slots / chunks
    onoff[149] = onoff[3]
    _o, s0, p0, lg0 = eng18.score_masks(imgs[0], seg, onoff, label, return_logits=True)
    _o, s1, _p1 = eng18.score_masks(imgs[1], seg, onoff, label)
    base_pred, base_prob = eng18.predict(imgs[0])
    assert abs(s0[0] - base_prob[label]) < 1e-5     # all-ones mask == unmasked (one row through the MFMA stem, 150 through the stem table)
    assert p0[0] == base_pred
    assert s0[1] == s1[1]                                   # all-zeros mask: independent of the image
    assert s0[3] == s0[70] == s0[149] and p0[3] == p0[70]   # batch/slot invariance, bit for bit
    assert np.isfinite(lg0).all() and (s0 >= 0).all() and (s0 <= 1).all()
    sm = np.exp(lg0 - lg0.max(1, keepdims=True))
    sm /= sm.sum(1, keepdims=True)
    assert np.abs(sm[np.arange(150), label] - s0).max() < 1e-6
    assert (lg0.argmax(1) == p0).all()
    _o, s_again, _ = eng18.score_masks(imgs[0], seg, onoff, label)
    assert (s_again == s0).all()                            # deterministic run to run


def test_resnet101_full_batch_properties(mpx_lib, dev):
    """BASELINE configs[2] batch shape (ResNet-101, 512 masks of one image in one forward batch): the oracle
    cannot cover this size on the CPU, so check size-independent properties instead."""
    eng = MaskedForwardEngine("resnet101", max_batch=512, device=0).load_state_dict(synth.make_state_dict("resnet101"))
    try:
        imgs = synth.make_images(2, seed=21, kind="noise")
        seg = synth.grid_segments()
        onoff = synth.random_onoff(512, 196, seed=3)
        onoff[0] = 1
        onoff[1] = 0
        onoff[511] = onoff[7]
        onoff[300] = onoff[7]
        base_pred, base_prob = eng.predict(imgs[0])
        _o, s0, p0, lg = eng.score_masks(imgs[0], seg, onoff, base_pred, return_logits=True)
        _o, s1, _p = eng.score_masks(imgs[1], seg, onoff, base_pred)
        assert abs(s0[0] - base_prob[base_pred]) < 1e-5 and p0[0] == base_pred     # all-ones == unmasked (predict stages one row through K0 and
                                                                                   # the MFMA stem, 512 rows go through the stem table: rounding)
        assert s0[1] == s1[1]                                                      # all-zeros: image-independent
        assert s0[7] == s0[300] == s0[511]                                         # slot invariance, bit for bit
        assert np.isfinite(lg).all() and (lg.argmax(1) == p0).all()
        sm = np.exp(lg - lg.max(1, keepdims=True))
        sm /= sm.sum(1, keepdims=True)
        assert np.abs(sm[np.arange(512), base_pred] - s0).max() < 1e-6
        # the same masks through a smaller engine (different batch -> different tile rounds) agree bit for bit
        small = MaskedForwardEngine("resnet101", max_batch=24, device=0).load_state_dict(synth.make_state_dict("resnet101"))
        small.stem_table_min_rows = 1           # 48 rows: stage them the way the 512 were staged (bits are compared)
        _o, s_small, p_small = small.score_masks(imgs[0], seg, onoff[:48], base_pred)
        small.close()
        assert (s_small == s0[:48]).all() and (p_small == p0[:48]).all()
    finally:
        eng.close()


def _bench_shape_forward(eng, imgs_u8, seg, onoff, labels, dev):
    """Stage n_img x n_mask masks by hand (image j -> slots [j*n_mask, (j+1)*n_mask), every slot offset a multiple of n_mask)
    and run ONE forward over the whole batch -- how bench.py's step was written up to round 2.  The bench now runs
    MaskedForwardEngine.score_packed at forward batch 2340 (images straddle forwards): tests/test_gpu_benched_entry.py covers that
    entry.  -> (score f32[n_img, n_mask], pred i32[n_img, n_mask]) numpy."""
    n_img, n_mask = onoff.shape[:2]
    seg_d = torch.from_numpy(seg).to(dev)
    for j in range(n_img):
        eng.stage_masks(torch.from_numpy(imgs_u8[j]).to(dev), seg_d, torch.from_numpy(onoff[j]).to(dev), j * n_mask)
    label_rows = torch.from_numpy(np.repeat(np.asarray(labels, dtype=np.int32), n_mask)).to(dev)
    score, pred = eng.forward(n_img * n_mask, label_rows)
    torch.cuda.synchronize()
    return score.view(n_img, n_mask).cpu().numpy(), pred.view(n_img, n_mask).cpu().numpy()


def test_cfg3_benched_shape_resnet101_2048_vs_oracle(mpx_lib, dev):
    """BASELINE configs[2] at the shape bench.py times: ResNet-101, forward batch 2048 = 4 different images x 512
    masks in one launch sequence (n_first descriptor rebasing, the 2-GiB num_records clamp, the patch kernel's
    q0 / PIMG arithmetic over thousands of images and the multi-image slot staging all run at this size only).
    First and last mask of every image against the reference-style batch-1 CPU loop (<= 2e-5); every one of the
    2048 scores bit-equal to the same masks through a max_batch=32 engine (different tile rounds and descriptors)."""
    arch, n_img, n_mask = "resnet101", 4, 512
    sd = synth.make_state_dict(arch)
    imgs = synth.make_images(n_img, seed=1234, kind="noise")
    seg = synth.grid_segments()
    onoff = synth.random_onoff(n_img * n_mask, 196, seed=4321).reshape(n_img, n_mask, 196)
    # (staged by hand through K0: both engines on the MFMA stem -- the stem by superposition rounds differently, its bits are compared
    # with its own in tests/test_gpu_benched_entry.py)
    small = MaskedForwardEngine(arch, max_batch=32, device=0, stem="conv").load_state_dict(sd)
    big = MaskedForwardEngine(arch, max_batch=n_img * n_mask, device=0, stem="conv").load_state_dict(sd)
    try:
        labels = [small.predict(imgs[j])[0] for j in range(n_img)]
        score, pred = _bench_shape_forward(big, imgs, seg, onoff, labels, dev)
        assert np.isfinite(score).all()
        worst = 0.0
        for j in range(n_img):
            x = scorer.to_tensor_normalize(imgs[j])
            assert scorer.base_prediction(sd, arch, x) == labels[j]
            pick = [0, n_mask - 1]
            ref, ref_pred = scorer.score_masks_reference_loop(sd, arch, x, seg, onoff[j][pick], labels[j])
            worst = max(worst, float(np.abs(score[j][pick] - ref).max()))
            assert (pred[j][pick] == ref_pred).all()
            _o, s_small, p_small = small.score_masks(imgs[j], seg, onoff[j], labels[j])     # 16 forwards of 32
            assert (s_small == score[j]).all() and (p_small == pred[j]).all(), "image %d differs between batch 2048 and batch 32" % j
        print("cfg-3 benched shape: max|d| vs CPU loop on 8 slots %.3e" % worst)
        assert worst <= SCORE_TOL_TIGHT
    finally:
        big.close()
        small.close()


def test_cfg2_resnet18_256_masks_per_image_in_one_batch(mpx_lib, dev):
    """BASELINE configs[1]: ResNet-18, 256 masks per image, 8 of the 32 images in one forward batch of 2048 (the
    bench's `--arch resnet18 --masks 256 --images-per-forward 8`).  16 slots (first and last mask of each image)
    against the CPU loop; the rest by slot invariance: an engine of max_batch 64 must give the same bits."""
    arch, n_img, n_mask = "resnet18", 8, 256
    sd = synth.make_state_dict(arch)
    imgs = synth.make_images(n_img, seed=99, kind="noise")
    seg = synth.grid_segments()
    onoff = synth.random_onoff(n_img * n_mask, 196, seed=17).reshape(n_img, n_mask, 196)
    onoff[3, 100] = onoff[3, 7]                    # duplicate rows inside an image
    small = MaskedForwardEngine(arch, max_batch=64, device=0, stem="conv").load_state_dict(sd)       # hand staging through K0: the MFMA stem on both
    big = MaskedForwardEngine(arch, max_batch=n_img * n_mask, device=0, stem="conv").load_state_dict(sd)
    try:
        labels = [small.predict(imgs[j])[0] for j in range(n_img)]
        score, pred = _bench_shape_forward(big, imgs, seg, onoff, labels, dev)
        assert score[3, 100] == score[3, 7]
        worst = 0.0
        for j in range(n_img):
            x = scorer.to_tensor_normalize(imgs[j])
            pick = [0, n_mask - 1]
            ref, ref_pred = scorer.score_masks_reference_loop(sd, arch, x, seg, onoff[j][pick], labels[j])
            worst = max(worst, float(np.abs(score[j][pick] - ref).max()))
            assert (pred[j][pick] == ref_pred).all()
            _o, s_small, p_small = small.score_masks(imgs[j], seg, onoff[j], labels[j])
            assert (s_small == score[j]).all() and (p_small == pred[j]).all()
        print("cfg-2 shape: max|d| vs CPU loop on 16 slots %.3e" % worst)
        assert worst <= SCORE_TOL_TIGHT
    finally:
        big.close()
        small.close()


def _cfg5_case(kind, golden_dir):
    """(u8 picture, label map as the segmentation library returned it): the felzenszwalb fixture (what the reference's scorers call),
    or one of the two SLIC maps of tests/golden/segments_slic.npz (what BASELINE configs[4] names; generate_superpixels.py:2 imports
    slic) -- "slic1" carries labels 1 .. S, a map that does not start at 0."""
    if kind == "felzenszwalb":
        g = np.load(os.path.join(golden_dir, "felzenszwalb_skimage0183.npz"))
        return g["blobs224/image"], g["blobs224/labels"].astype(np.int64)
    g = np.load(os.path.join(golden_dir, "segments_slic.npz"))
    i = int(kind[-1])
    return synth.make_images(2, seed=int(g["image_seed"]), kind="blobs")[i], g["segments"][i].astype(np.int64)


@pytest.mark.parametrize("kind", ["felzenszwalb", "slic0", "slic1"])
def test_cfg5_bo_window_sweep_on_hip_engine_vs_oracle_loop(mpx_lib, dev, golden_dir, kind):
    """BASELINE configs[4] / SURVEY 8 f1: the reference-named BO objective on the HIP engine.  On the committed
    felzenszwalb fixture (scikit-image 0.18.3 labels of the blobs224 picture) and on the two SLIC label maps (scikit-image 0.18.3,
    labels from 0 and from 1) api.sample_loss([f], ...) for EVERY
    f in [0, int(0.6*S)] must agree with the oracle's literal loop (bayesian_active_learning_imagenet.py:173-198:
    np.unique(segments)[f:f+k] -> mask[segments == v] = 1 -> input*mask -> batch-1 forward -> softmax[label]);
    then bo.bayesian_optimisation runs end to end (BayesianOptimization.py:99-192 signature) and every y it
    collected is the table entry of its x.  The SLIC cases also put random on/off vectors through engine.score_masks against the
    oracle loop (the generators' entry) with the label map exactly as the library returned it."""
    import random
    from network_interpretation_imagenet_amd import api, bo
    arch = "resnet101"
    sd = synth.make_state_dict(arch)
    img, seg = _cfg5_case(kind, golden_dir)
    S = len(np.unique(seg))
    assert int(seg.min()) == (1 if kind == "slic1" else 0)
    ub = scorer.bo_upper_bound(S)
    x = scorer.to_tensor_normalize(img)
    label = scorer.base_prediction(sd, arch, x)
    eng = MaskedForwardEngine(arch, max_batch=256, device=0).load_state_dict(sd)
    api.configure(eval_img_index=1, segmenter=lambda _img_show: seg, mask_dir=None, seed=None)
    try:
        loader = [(x[None], torch.tensor([label]))]
        got = np.array([api.sample_loss([f], loader, eng, None) for f in range(ub + 1)], dtype=np.float32)
        assert got.dtype == np.float32
        want = np.zeros(ub + 1, dtype=np.float32)
        for f in range(ub + 1):
            mask = scorer.window_mask_u8(seg, f)
            want[f], _pred = scorer.score_one(sd, arch, scorer.apply_mask(x, mask), label)
            assert (api.superpixel_mask(f) == mask * 255).all()                     # bayesian...:224-276
        err = float(np.abs(got - want).max())
        print("cfg-5: S=%d, %d windows, max|d| vs literal loop %.3e, score range %.4f..%.4f" % (S, ub + 1, err, want.min(), want.max()))
        assert err <= SCORE_TOL_TIGHT
        assert api.sample_loss(np.array([12.9]), loader, eng, None) == got[12]      # L-BFGS-B floats truncate (:283)
        xp, yp = bo.bayesian_optimisation(n_iters=10, sample_loss=api.sample_loss, val_loader=loader, nn_model=eng,
                                          criterion=None, bounds=np.array([[0, ub]]), n_pre_samples=3, rng=random.Random(5))
        assert xp.shape == (13, 1) and yp.shape == (13,)
        assert ((xp >= 0) & (xp <= ub)).all()
        assert all(np.float32(y) == got[int(f)] for f, y in zip(xp[:, 0], yp))
        if kind != "felzenszwalb":
            onoff = synth.random_onoff(6, S, seed=17)
            _o, g_score, g_pred = eng.score_masks(img, seg, onoff, label)              # raw labels in, ranked inside (np.unique order)
            w_score, w_pred = scorer.score_masks_reference_loop(sd, arch, x, seg, onoff, label)
            assert float(np.abs(g_score - w_score).max()) <= SCORE_TOL_TIGHT and (g_pred == w_pred).all()
    finally:
        api.configure(eval_img_index=1, segmenter=None, mask_dir=None, seed=None)
        eng.close()


def test_heatmap_accumulate_exact(eng18, dev, golden_dir):
    """K5 vs the oracle's literal accumulation (integer-valued, so exact)."""
    seg = np.load(os.path.join(golden_dir, "segments_blobs.npz"))["segments"][0].astype(np.int32)
    s = int(seg.max()) + 1
    onoff = np.concatenate([masks.windows_onoff(s, range(0, s + 1)), synth.random_onoff(30, s, seed=8)])
    m = onoff.shape[0]
    rng = np.random.default_rng(0)
    label = np.full(m, 7, dtype=np.int32)
    pred = np.where(rng.random(m) < 0.5, 7, 9).astype(np.int32)
    heat = torch.zeros(224, 224, dtype=torch.float32, device=dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    eng18.heatmap_accumulate(t(seg), t(onoff), t(pred), t(label), heat)
    eng18.heatmap_accumulate(t(seg), t(onoff), t(pred), t(label), heat)          # accumulates in place
    want = scorer.summed_superpixel_labels(seg, onoff, pred == label)
    assert (heat.cpu().numpy().astype(np.float64) == 2 * want).all()
    assert eng18._lib.mpx_heatmap_accumulate(eng18._h, None, None, None, None, 1, 1, None, None) == -1


def test_pipelined_heat_maps_with_native_segmentation(eng18):
    """Row f3 + f2 end to end: api.validate_summed_many segments three images on the host pool (libmpxseg.so)
    while the engine scores; image 2's heat map is checked against the oracle's literal accumulation."""
    import random
    from network_interpretation_imagenet_amd import api, segment
    imgs = synth.make_images(3, seed=77, kind="blobs")
    xs = [scorer.to_tensor_normalize(im) for im in imgs]
    labels = [eng18.predict(x)[0] for x in xs]
    loader = [(x[None], torch.tensor([lab])) for x, lab in zip(xs, labels)]
    api.configure(segmenter=None, num_mask_samples=10, mask_dir=None, seed=None)
    try:
        maps = api.validate_summed_many(loader, eng18, None, [1, 2, 3], rng=random.Random(3), workers=3)
        assert sorted(maps) == [1, 2, 3] and all(m.shape == (224, 224) for m in maps.values())
        only2 = api.validate_summed_many(loader, eng18, None, [2], rng=random.Random(4), workers=1)[2]
        seg = segment.felzenszwalb(api.img_show_u8(xs[1].numpy()))
        S = int(seg.max()) + 1
        firsts = masks.draw_first_indices(S, 10, random.Random(4))
        onoff = masks.windows_onoff(S, firsts)
        _s, pred = scorer.score_masks_batched(synth.make_state_dict("resnet18"), "resnet18", xs[1], seg, onoff, labels[1], chunk=10)
        assert (only2 == scorer.summed_superpixel_labels(seg, onoff, pred == labels[1])).all()
    finally:
        api.configure(segmenter=None, num_mask_samples=100, mask_dir=None, seed=None)


def test_engine_errors(eng18, dev):
    img = synth.make_images(1)[0]
    seg = synth.grid_segments()
    with pytest.raises(ValueError):
        eng18.score_masks(img, seg, np.ones((2, 195), dtype=np.uint8), 0)      # S mismatch
    with pytest.raises(ValueError):
        eng18.score_masks(img[:100], seg, np.ones((2, 196), dtype=np.uint8), 0)
    with pytest.raises(ValueError):
        eng18.score_masks(img, seg, np.ones((2, 196), dtype=np.uint8), 1000)
    o, s, p = eng18.score_masks(img, seg, np.zeros((0, 196), dtype=np.uint8), 0)   # empty batch
    assert s.shape == (0,) and p.shape == (0,)
    # engine.heatmap_device (the per-rank step of shard.heatmap_sharded) validates like score_masks: a mismatched S or label used to
    # come back as a silently wrong heat map (the kernels only guard their indices)
    buf = torch.zeros(224 * 224 + 1, dtype=torch.float32, device=dev)
    with pytest.raises(ValueError):
        eng18.heatmap_device(img, seg, np.ones((2, 195), dtype=np.uint8), 0, buf)       # S mismatch
    with pytest.raises(ValueError):
        eng18.heatmap_device(img, seg, np.ones((2, 196), dtype=np.uint8), 1000, buf)    # label out of range
    with pytest.raises(ValueError):
        eng18.heatmap_device(img, seg, np.ones((2, 196), dtype=np.float32), 0, buf)     # not a u8 table
    with pytest.raises(ValueError):
        eng18.heatmap_device(img, seg[:100], np.ones((2, 196), dtype=np.uint8), 0, buf)
    with pytest.raises(ValueError):
        eng18.heatmap_device(img, seg, np.ones((2, 196), dtype=np.uint8), 0, buf[:-1])
    assert float(buf.abs().sum()) == 0.0                                                 # nothing ran
    s_d, p_d = eng18.heatmap_device(img, seg * 3 + 5, np.ones((2, 196), dtype=np.uint8), 0, buf)     # any integer label map is ranked
    assert s_d.shape == (2,) and float(buf[:-1].max()) <= 2.0
    with pytest.raises(KeyError):
        eng18.load_state_dict(synth.make_state_dict("resnet18"), only=["layer9.0.conv1"])
    labels = torch.zeros(65, dtype=torch.int32, device=dev)
    with pytest.raises(MpxError):
        eng18.forward(65, labels)                                               # > max_batch
    with pytest.raises(ValueError):
        MaskedForwardEngine("vgg16")
    fresh = MaskedForwardEngine("resnet18", max_batch=2, device=0)
    with pytest.raises(MpxError):
        fresh.forward(1, labels[:1])                                            # weights not loaded
    with pytest.raises(KeyError):
        fresh.load_state_dict({})
    fresh.close()


# ------------------------------------------------------------------------------------------------
# the reference's two small networks, WHOLE TRAINED nets (shipped checkpoints) on the HIP kernels
# ------------------------------------------------------------------------------------------------
def _smallnet_case(arch, golden_dir):
    g = np.load(os.path.join(golden_dir, "smallnet_%s.npz" % arch))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd


@pytest.mark.parametrize("arch", ["mnist_net", "cifar_resnet56"])
def test_trained_small_network_end_to_end(mpx_lib, dev, golden_dir, arch):
    """Classification_Net (generate_gp_training_data_mnist.py:86-105) and ResNetCifar(56) (models/resnet.py:77-146) with
    the weights of the reference's own checkpoints, scored under the CIFAR / MNIST scorers' mask convention
    (generate_gp_training_data_cifar.py:274-321): the staged network inputs must equal the oracle's NumPy arithmetic bit for
    bit, logits / scores agree with the CPU oracle within the score tolerance, argmax exactly."""
    from oracle import smallnets_ref
    g, sd = _smallnet_case(arch, golden_dir)
    eng = MaskedForwardEngine(arch, max_batch=16, device=0).load_state_dict(sd)      # 24 masks: two chunks
    try:
        assert (eng.image_size, eng.in_channels, eng.num_classes) == ((28, 1, 10) if arch == "mnist_net" else (32, 3, 10))
        worst_logit = worst_score = 0.0
        for i in range(int(g["n_pictures"])):
            p = "pic%d/" % i
            x, seg, removed, label = g[p + "x"], g[p + "segments"], g[p + "removed"], int(g[p + "label"])
            _r, score, pred, logits, inputs = eng.score_masks_removed(x, seg, removed, label, return_logits=True, return_inputs=True)
            assert (inputs.view(np.int32) == g[p + "masked_inputs"].view(np.int32)).all(), "K0 (min-max mask convention) is not bit-exact"
            want = g[p + "masked_logits_f32"]
            worst_logit = max(worst_logit, float(np.abs(logits - want).max() / max(1.0, np.abs(want).max())))
            worst_score = max(worst_score, float(np.abs(score - g[p + "score_f32"]).max()))
            assert (pred == g[p + "pred"]).all()
            assert np.abs(logits.astype(np.float64) - g[p + "masked_logits_f64"]).max() <= 1e-4
            # live oracle on three masks (the fixture pins the oracle; this pins the test's reading of it)
            uniq = np.unique(seg)
            lists = [[int(uniq[j]) for j in np.nonzero(row)[0]] for row in removed[:3]]
            ref_score, ref_pred = smallnets_ref.score_removed_loop(sd, arch, x, seg, lists, label)
            assert np.abs(score[:3] - ref_score).max() <= SCORE_TOL_TIGHT and (pred[:3] == ref_pred).all()
        print("%s trained: max rel logit err %.3e, max |d score| %.3e" % (arch, worst_logit, worst_score))
        assert worst_logit <= 1e-5 and worst_score <= SCORE_TOL_TIGHT
        with pytest.raises(ValueError):
            eng.score_masks(np.zeros((224, 224, 3), dtype=np.uint8), np.zeros((224, 224), dtype=np.int32), np.ones((1, 1), dtype=np.uint8), 0)
    finally:
        eng.close()


@pytest.mark.parametrize("arch", ["mnist_net", "cifar_resnet56"])
def test_eval_superpixel_png_labels_follow_the_oracle_loop(mpx_lib, dev, golden_dir, tmp_path, arch):
    """api.eval_superpixel = eval_superpixel() of generate_gp_training_data_cifar.py:236-342 / generate_gp_training_data_mnist.py:
    153-269 on the engine: the package's sampler (masks.draw_removed_sets), all masks in batched passes, mask_{i}_{0|1}.png =
    the {0,255} mask as the scripts write it.  The PNG labels and the returned count equal `pred == label` of the oracle's
    one-mask-at-a-time loop on the same draws, the PNG pixels the oracle's mask picture."""
    import random
    from PIL import Image
    from network_interpretation_imagenet_amd import api
    from oracle import smallnets_ref
    g, sd = _smallnet_case(arch, golden_dir)
    eng = MaskedForwardEngine(arch, max_batch=16, device=0).load_state_dict(sd)
    k, burn = (1, True) if arch == "mnist_net" else (5, False)
    try:
        x, seg, label = g["pic1/x"], g["pic1/segments"], int(g["pic1/label"])
        loader = [(torch.zeros(1, *x.shape), torch.tensor([0])), (torch.from_numpy(x)[None], torch.tensor([label]))]
        api.configure(mask_dir=str(tmp_path / "masks"))
        n_ok = api.eval_superpixel(loader, eng, eval_img_index=2, num_mask_samples=40, rng=random.Random(5), segments=seg)
        sets = masks.draw_removed_sets(np.unique(seg), k, 40, random.Random(5), burn_window_draw=burn)
        _score, ref_pred = smallnets_ref.score_removed_loop(sd, arch, x, seg, sets, label)
        files, labels = api.load_images_from_folder(str(tmp_path / "masks"))
        assert len(files) == 40
        by_i = {int(os.path.basename(f).split("_")[1]): (f, int(l)) for f, l in zip(files, labels)}
        assert [by_i[i][1] for i in range(40)] == [int(p == label) for p in ref_pred] and n_ok == int((ref_pred == label).sum())
        for i in (0, 7, 39):
            assert np.array_equal(np.asarray(Image.open(by_i[i][0])), smallnets_ref.removed_mask_u8(seg, sets[i]))
        # the script's own defaults: image index, superpixels per mask and min_size follow the engine's architecture; native segmentation
        api.configure(mask_dir=None)
        idx = 2 if arch == "mnist_net" else 5
        loader5 = [loader[0]] * (idx - 1) + [loader[1]]
        n2 = api.eval_superpixel(loader5, eng, num_mask_samples=30, rng=random.Random(6))
        assert 0 <= n2 <= 30 and api.eval_superpixel(loader5[:idx - 1], eng, num_mask_samples=3) == 0
    finally:
        api.configure(mask_dir=None)
        eng.close()


def test_small_network_ops(mpx_lib, dev, golden_dir):
    """DownsampleB (mpx_avgpool2_pad) exactly, and a mask that removes every superpixel gives NaN inputs as 0/0 does upstream."""
    g, sd = _smallnet_case("cifar_resnet56", golden_dir)
    eng = MaskedForwardEngine("cifar_resnet56", max_batch=4, device=0).load_state_dict(sd)
    try:
        x = torch.randn(3, 16, 16, 32, generator=torch.Generator().manual_seed(4))
        x[..., 16:] = 0
        xh, xl = split(x.to(dev))
        oh = torch.full((3, 8, 8, 32), float("nan"), dtype=torch.float16, device=dev)
        ol = torch.full_like(oh, float("nan"))
        _lib.check(eng._h, eng._lib.mpx_avgpool2_pad(eng._h, _p(xh), _p(xl), _p(oh), _p(ol), 3, 16, 32, 32, eng._stream()), "avgpool2_pad")
        torch.cuda.synchronize()
        want = F.avg_pool2d(merge(xh, xl).cpu().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        got = merge(oh, ol).cpu()          # the fp32 average, stored as hi + lo (22 bits)
        assert (got - want).abs().max().item() <= 2.0 ** -21 * want.abs().max().item() and torch.equal(got[..., 16:], torch.zeros(3, 8, 8, 16))
        assert eng._lib.mpx_avgpool2_pad(eng._h, _p(xh), _p(xl), _p(oh), _p(ol), 3, 15, 32, 32, None) == -1
        p = "pic0/"
        seg = g[p + "segments"]
        S = len(np.unique(seg))
        _r, score, pred, inputs = eng.score_masks_removed(g[p + "x"], seg, np.ones((1, S), dtype=np.uint8), 0, return_inputs=True)
        assert np.isnan(inputs).all()
        big = MaskedForwardEngine("resnet18", max_batch=1, device=0)
        assert big._lib.mpx_mask_apply_minmax(big._h, None, None, None, 1, 1, 0, None, None) == -2      # ImageNet engines refuse it
        assert big.input_plane_shape(1) == (1, 230, 230, 4)
        big.close()
        # the zero-copy staging view follows the engine's geometry: [n][32][32][32] here, not the ImageNet [n][230][230][4] (which
        # would run far past the small arena); the staged values are the masked inputs in the first 3 of 32 channels
        assert eng.input_plane_shape(3) == (3, 32, 32, 32)
        _r, _s, _pr, inputs = eng.score_masks_removed(g[p + "x"], seg, g[p + "removed"][:3], int(g[p + "label"]), return_inputs=True)
        hi, lo = eng.input_planes(3)
        assert tuple(hi.shape) == (3, 32, 32, 32) and hi.numel() == 3 * 32 * 32 * 32
        staged = merge(hi, lo).cpu().numpy()
        assert np.abs(staged[..., :3].transpose(0, 3, 1, 2) - inputs).max() <= 2.0 ** -21 * np.abs(inputs).max() and (staged[..., 3:] == 0).all()
        with pytest.raises(ValueError):
            eng.input_planes(5)                       # more slots than max_batch = 4
    finally:
        eng.close()
