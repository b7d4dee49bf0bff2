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
