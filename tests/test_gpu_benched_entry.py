"""The launch sequence the driver's bench number comes from, under an oracle test ("what's missing" 1).

bench.py's step is MaskedForwardEngine.score_packed on an engine of max_batch = engine.whole_round_batch(2400) (2340 slots on
256 CUs): the 512-row mask tables of consecutive images are packed into forward batches of 2340 slots, so an image straddles two
forwards, K0's slot offsets are not multiples of 512 and the last forward of a step is small.  Here the same entry runs on
5 (ResNet-101) / 10 (ResNet-18) images -- forwards of 2340 + 220 rows -- and is compared with the reference's loop
(generate_gp_training_data_imagenet.py:221-266: one mask, one batch-1 fp32 forward, softmax[label] as in
bayesian_active_learning_imagenet.py:196-198) on the first and last row of every image and the two rows either side of the
forward-batch boundary; every row must be bit-equal to the same rows through a max_batch=32 engine (other tile rounds, other
kernels for the small launches, other descriptors)."""
import numpy as np
import pytest
import torch

from network_interpretation_imagenet_amd import synth
from network_interpretation_imagenet_amd.engine import MaskedForwardEngine, whole_round_batch
from oracle import scorer

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4        # north_star tolerance (BASELINE.json)


def _packed(eng, imgs, seg, onoff, labels, dev):
    """bench.py's step(): device-resident inputs, ONE score_packed call -> (score f32[n_img, n_mask], pred i32[...]) numpy."""
    n_img, n_mask = onoff.shape[:2]
    img_d = [torch.from_numpy(imgs[j]).to(dev) for j in range(n_img)]
    seg_d = torch.from_numpy(seg).to(dev)
    onoff_d = [torch.from_numpy(onoff[j]).to(dev) for j in range(n_img)]
    rows = torch.from_numpy(np.repeat(np.asarray(labels, dtype=np.int32), n_mask)).to(dev)
    score = torch.empty(n_img * n_mask, dtype=torch.float32, device=dev)
    pred = torch.empty(n_img * n_mask, dtype=torch.int32, device=dev)
    eng.score_packed(img_d, seg_d, onoff_d, rows, score, pred)
    torch.cuda.synchronize()
    return score.view(n_img, n_mask).cpu().numpy(), pred.view(n_img, n_mask).cpu().numpy()


def _benched_entry_case(arch, n_img, n_mask, tight, seed):
    dev = torch.device("cuda", 0)
    sd = synth.make_state_dict(arch)
    imgs = synth.make_images(n_img, seed=seed, kind="noise")
    seg = synth.grid_segments()
    onoff = synth.random_onoff(n_img * n_mask, 196, seed=seed + 1).reshape(n_img, n_mask, 196)
    small = MaskedForwardEngine(arch, max_batch=32, device=0).load_state_dict(sd)
    batch = whole_round_batch(2400, num_cus=small.num_cus)
    big = MaskedForwardEngine(arch, max_batch=batch, device=0).load_state_dict(sd)
    try:
        assert big.num_cus == small.num_cus and (big.num_cus != 256 or batch == 2340)
        total = n_img * n_mask
        assert batch < total < 2 * batch and total % batch != 0, "the case must straddle a forward-batch boundary with a small last forward"
        labels = [small.predict(imgs[j])[0] for j in range(n_img)]
        score, pred = _packed(big, imgs, seg, onoff, labels, dev)
        assert np.isfinite(score).all()
        # rows to check against the CPU loop: first / last of every image, and the two rows either side of the boundary
        picks = {j: {0, n_mask - 1} for j in range(n_img)}
        for w in (batch - 2, batch - 1, batch, batch + 1):
            picks[w // n_mask].add(w % n_mask)
        j_b = batch // n_mask
        assert (batch - 1) // n_mask == j_b == batch // n_mask and 0 < batch % n_mask, "image %d must straddle the boundary" % j_b
        worst, n_checked = 0.0, 0
        for j in range(n_img):
            x = scorer.to_tensor_normalize(imgs[j])
            assert scorer.base_prediction(sd, arch, x) == labels[j]
            pick = sorted(picks[j])
            ref, ref_pred = scorer.score_masks_reference_loop(sd, arch, x, seg, onoff[j][pick], labels[j])
            worst = max(worst, float(np.abs(score[j][pick].astype(np.float64) - ref).max()))
            assert (pred[j][pick] == ref_pred).all(), "argmax differs from the CPU loop on image %d rows %s" % (j, pick)
            n_checked += len(pick)
        s_small, p_small = _packed(small, imgs, seg, onoff, labels, dev)            # 80 forwards of 32 slots
        diff = np.argwhere(s_small != score)
        assert diff.size == 0 and (p_small == pred).all(), "rows %s differ between forward batch %d and 32" % (diff[:4].tolist(), batch)
        print("%s benched entry (score_packed, forwards of %d + %d rows): max|d| vs CPU loop on %d rows %.3e"
              % (arch, batch, total - batch, n_checked, worst))
        assert worst <= SCORE_TOL
        assert worst <= tight, "the engine's default arithmetic is expected to stay within %g of the fp32 loop" % tight
    finally:
        big.close()
        small.close()


def test_cfg3_benched_entry_resnet101_2340_packed_vs_oracle(mpx_lib):
    """BASELINE configs[2] as bench.py runs it: ResNet-101, 512 masks per image, forward batch 2340."""
    _benched_entry_case("resnet101", 5, 512, tight=5e-5, seed=1234)


def test_cfg2_benched_entry_resnet18_2340_packed_vs_oracle(mpx_lib):
    """BASELINE configs[1] through the same entry: ResNet-18, 256 masks per image, 2340 slots (`bench.py --arch resnet18 --masks 256
    --images 32` packs it this way)."""
    _benched_entry_case("resnet18", 10, 256, tight=2e-5, seed=99)


def test_cfg3_full_size_properties_resnet101_128_images_x_512_masks(mpx_lib):
    """BASELINE configs[2] at its FULL size through the benched entry: 128 images x 512 masks = 65,536 rows in one score_packed
    call (28 forwards of 2340 + one of 16).  The CPU loop cannot cover this size, so the checks are the size-independent ones:
    the all-ones row of every image equals its unmasked prediction; the all-zeros row is the same bits for every image; an image
    that appears twice (tables at different positions of different forward batches, one of them straddling a boundary) gets the
    same bits; duplicate rows inside a table get the same bits; every score is a probability and every prediction a class; a
    second run reproduces the first bit for bit; and 64 rows spread over the range equal a max_batch=32 engine's bits."""
    dev = torch.device("cuda", 0)
    arch, n_img, n_mask = "resnet101", 128, 512
    sd = synth.make_state_dict(arch)
    imgs = synth.make_images(n_img, seed=2024, kind="noise")
    imgs[77] = imgs[5]                                     # the same picture twice, 72 tables apart
    imgs[100] = imgs[4]                                    # image 4's table straddles the first forward boundary (rows 2048 .. 2559)
    seg = synth.grid_segments()
    onoff = synth.random_onoff(n_img * n_mask, 196, seed=2025).reshape(n_img, n_mask, 196)
    onoff[:, 0] = 1
    onoff[:, 1] = 0
    onoff[:, 300] = onoff[:, 7]
    onoff[77] = onoff[5]
    onoff[100] = onoff[4]
    small = MaskedForwardEngine(arch, max_batch=32, device=0).load_state_dict(sd)
    big = MaskedForwardEngine(arch, max_batch=whole_round_batch(2400, num_cus=small.num_cus), device=0).load_state_dict(sd)
    try:
        base = [small.predict(imgs[j]) for j in range(n_img)]
        labels = [b[0] for b in base]
        labels[77], labels[100] = labels[5], labels[4]
        score, pred = _packed(big, imgs, seg, onoff, labels, dev)
        assert score.shape == (n_img, n_mask) and np.isfinite(score).all() and (score >= 0).all() and (score <= 1).all()
        assert (pred >= 0).all() and (pred < 1000).all()
        for j in range(n_img):
            # all-ones == unmasked (predict stages its one row through K0 and the MFMA stem, the tables go through the stem table: rounding)
            assert abs(float(score[j, 0]) - float(base[j][1][labels[j]])) < 1e-5 and pred[j, 0] == base[j][0]
        assert (pred[:, 1] == pred[0, 1]).all()                                  # all-zeros: the input is image-independent ...
        for lab in set(labels):                                                   # ... and so are the bits of the images scored for the same class
            same = [j for j in range(n_img) if labels[j] == lab]
            assert (score[same, 1] == score[same[0], 1]).all()
        assert (score[:, 300] == score[:, 7]).all() and (pred[:, 300] == pred[:, 7]).all()
        assert (score[77] == score[5]).all() and (pred[77] == pred[5]).all()
        assert (score[100] == score[4]).all() and (pred[100] == pred[4]).all()
        again_s, again_p = _packed(big, imgs, seg, onoff, labels, dev)
        assert (again_s == score).all() and (again_p == pred).all()
        rows = [(j, m) for j in range(0, n_img, 8) for m in (2, 129, 383, 511)]                  # 64 rows over the whole range
        small.stem_table_min_rows = 1           # four rows per call: stage them the way the tables were staged (bits are compared)
        for j in sorted({r[0] for r in rows}):
            ms = [m for jj, m in rows if jj == j]
            _o, s_small, p_small = small.score_masks(imgs[j], seg, onoff[j][ms], labels[j])
            assert (s_small == score[j, ms]).all() and (p_small == pred[j, ms]).all(), "image %d differs from the batch-32 engine" % j
    finally:
        big.close()
        small.close()
