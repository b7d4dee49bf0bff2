"""Scorer semantics of the oracle (SURVEY.md appendix A "must match") on hand-computable cases,
plus the committed golden vectors (tests/golden, produced by tests/golden/make_golden.py)."""
import os
import random

import numpy as np
import pytest
import torch

from network_interpretation_imagenet_amd import synth, masks
from oracle import scorer as S

SEG4 = np.array([[0, 0, 1, 1],
                 [0, 2, 2, 1],
                 [3, 3, 4, 4],
                 [5, 5, 5, 4]])   # 6 superpixels


def test_window_size_and_bounds():
    assert S.num_conse_superpixels(6) == 2 and S.num_conse_superpixels(196) == 78
    assert S.bo_upper_bound(196) == 117
    rng = random.Random(0)
    draws = [S.draw_first_index(rng, 196) for _ in range(2000)]
    assert min(draws) == 1 and max(draws) == 196 - 78          # inclusive both ends, never 0


def test_window_mask_hand_case():
    m = S.window_mask_u8(SEG4, 1)       # labels 1,2 kept
    want = np.array([[0, 0, 1, 1], [0, 1, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]], dtype=np.uint8)
    assert m.dtype == np.uint8 and (m == want).all()
    # slice running off the end truncates silently
    assert (S.window_mask_u8(SEG4, 5) == (SEG4 == 5)).all()
    assert S.window_mask_u8(SEG4, 6).sum() == 0


def test_window_uses_unique_order_not_label_values():
    seg = SEG4 * 10 + 7                  # labels 7,17,27,...
    assert (S.window_mask_u8(seg, 1) == S.window_mask_u8(SEG4, 1)).all()
    assert (S.onoff_mask_u8(seg, S.window_onoff(6, 1)) == S.window_mask_u8(seg, 1)).all()


def test_normalise_then_mask():
    img = np.zeros((4, 4, 3), dtype=np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 255, 128, 0
    x = S.to_tensor_normalize(img)
    assert x.dtype == torch.float32 and x.shape == (3, 4, 4)
    np.testing.assert_allclose(x[:, 0, 0].numpy(), [(1 - 0.485) / 0.229, (128 / 255 - 0.456) / 0.224, -0.406 / 0.225], rtol=1e-6)
    masked = S.apply_mask(x, S.window_mask_u8(SEG4, 1))
    assert masked.dtype == np.float32
    assert (masked[:, 3, 3] == 0.0).all()                 # removed pixel is exactly 0.0 in normalised space
    assert (masked[:, 0, 2] == x[:, 0, 2].numpy()).all()   # kept pixel untouched, same mask for R,G,B


def test_img_show_truncates():
    x = torch.tensor([[[0.0, 1.0]], [[0.5, 0.25]], [[0.999, 0.0]]])
    out = S.img_show_u8(x)
    assert out.dtype == np.uint8 and out.shape == (1, 2, 3)
    assert out[0, 0].tolist() == [0, 127, 254] and out[0, 1].tolist() == [255, 63, 0]


def test_summed_labels():
    onoff = np.stack([S.window_onoff(6, 1), S.window_onoff(6, 2), S.window_onoff(6, 4)])
    acc = S.summed_superpixel_labels(SEG4, onoff, np.array([1, 0, 1]))
    want = (np.isin(SEG4, [1, 2]).astype(float) + np.isin(SEG4, [4, 5]).astype(float))
    assert (acc == want).all()


def test_host_masks_module_agrees_with_oracle():
    for s in (6, 23, 46, 196):
        assert masks.window_size(s) == S.num_conse_superpixels(s)
        assert masks.bo_upper_bound(s) == S.bo_upper_bound(s)
        for f in (0, 1, s // 2, s - 1, s):
            assert (masks.window_onoff(s, f) == S.window_onoff(s, f)).all()
    rank = np.unique(SEG4 * 3, return_inverse=True)[1].reshape(4, 4)
    assert (masks.expand_pixel_mask(rank, S.window_onoff(6, 1)) == S.window_mask_u8(SEG4 * 3, 1)).all()


def test_all_ones_and_all_zeros_masks():
    arch = "resnet18"
    sd = synth.make_state_dict(arch)
    imgs = synth.make_images(2)
    seg = synth.grid_segments()
    x0, x1 = S.to_tensor_normalize(imgs[0]), S.to_tensor_normalize(imgs[1])
    label = S.base_prediction(sd, arch, x0)
    ones = np.ones((1, 196), dtype=np.uint8)
    zeros = np.zeros((1, 196), dtype=np.uint8)
    s_all, p_all = S.score_masks_reference_loop(sd, arch, x0, seg, ones, label)
    with torch.no_grad():
        base = torch.softmax(S.resnet_ref.forward(sd, x0[None], arch), 1)[0, label].item()
    assert abs(s_all[0] - base) < 1e-7 and p_all[0] == label
    z0, _ = S.score_masks_reference_loop(sd, arch, x0, seg, zeros, label)
    z1, _ = S.score_masks_reference_loop(sd, arch, x1, seg, zeros, label)
    assert z0[0] == z1[0]                                   # all-zeros mask: score independent of the image


@pytest.mark.parametrize("arch", ["resnet18", "resnet101"])
def test_golden_scores_reproduce(arch, golden_dir):
    """Oracle vs committed golden vectors (a subset of masks, to stay within the CPU budget)."""
    g = np.load(os.path.join(golden_dir, "scores_%s.npz" % arch))
    segs = np.load(os.path.join(golden_dir, "segments_blobs.npz"))["segments"].astype(np.int64)
    sd = synth.make_state_dict(arch, seed=int(g["weight_seed"]))
    idx = int(g["image_index"])
    x = S.to_tensor_normalize(synth.make_images(2, seed=int(g["image_seed"]))[idx])
    label = int(g["label"])
    assert S.base_prediction(sd, arch, x) == label
    pick = [0, 5, len(g["onoff"]) - 1]
    score, pred = S.score_masks_reference_loop(sd, arch, x, segs[idx], g["onoff"][pick], label)
    np.testing.assert_allclose(score, g["score_f32"][pick], atol=2e-6)
    assert (pred == g["pred"][pick]).all()
    assert np.abs(g["score_f32"] - g["score_f64"]).max() < 5e-6
