"""CPU checks of the small-network oracle (oracle/smallnets_ref.py) against the committed whole-network fixtures
(tests/golden/smallnet_*.npz: the reference's TRAINED checkpoints + seeded pictures + the oracle's outputs) and against
hand-computable answers of the scorers' mask convention."""
import os

import numpy as np
import pytest
import torch

from oracle import smallnets_ref as ref

HERE = os.path.dirname(os.path.abspath(__file__))


def load_case(arch):
    g = np.load(os.path.join(HERE, "golden", "smallnet_%s.npz" % arch))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd


def _mnist_params():
    convs = [(1, 32), (32, 32), (32, 64), (64, 64), (64, 128)]
    return sum(ci * co * 9 + co + 2 * co for ci, co in convs) + (128 * 128 * 9 + 128) + (128 * 10 + 10)


def _cifar_params(depth):
    n = (depth - 2) // 6
    total = 3 * 16 * 9 + 2 * 16
    inp = 16
    for planes in (16, 32, 64):
        for b in range(n):
            total += inp * planes * 9 + 2 * planes + planes * planes * 9 + 2 * planes
            inp = planes
    return total + 64 * 10 + 10


@pytest.mark.parametrize("arch,n_params", [("mnist_net", _mnist_params()), ("cifar_resnet56", _cifar_params(56))])
def test_trained_checkpoints_known_answers(arch, n_params):
    g, sd = load_case(arch)
    learnable = sum(v.numel() for k, v in sd.items() if "running_" not in k)
    assert learnable == n_params                 # counted by hand from the layer lists (288,362; 853,018 = the published 0.85 M of CIFAR ResNet-56)
    assert n_params == {"mnist_net": 288362, "cifar_resnet56": 853018}[arch]
    assert int(g["n_pictures"]) == 2
    if arch == "cifar_resnet56":
        assert len(ref.cifar_resnet_blocks(56)) == 27
        assert [b for b in ref.cifar_resnet_blocks(56) if b[3] == 2] == [("layer2.0", 16, 32, 2), ("layer3.0", 32, 64, 2)]


@pytest.mark.parametrize("arch", ["mnist_net", "cifar_resnet56"])
def test_oracle_reproduces_golden_logits(arch):
    """Drift check of the restatement on real trained weights: unmasked logits and every masked input / logit / score."""
    g, sd = load_case(arch)
    for i in range(int(g["n_pictures"])):
        p = "pic%d/" % i
        x = g[p + "x"]
        with torch.no_grad():
            l32 = ref.forward(sd, torch.from_numpy(x[None]), arch).numpy()[0]
        assert np.abs(l32 - g[p + "logits_f32"]).max() <= 2e-5 and np.abs(l32 - g[p + "logits_f64"]).max() <= 1e-4
        assert l32.shape == (10,) and int(l32.argmax()) == int(g[p + "label"])
        seg, removed = g[p + "segments"], g[p + "removed"]
        uniq = np.unique(seg)
        lists = [[int(uniq[j]) for j in np.nonzero(row)[0]] for row in removed]
        org = ref.org_img_minmax255(x)
        inputs = np.stack([ref.masked_input(org, ref.removed_mask_u8(seg, r)) for r in lists])
        assert (inputs.view(np.int32) == g[p + "masked_inputs"].view(np.int32)).all()          # bit-exact: pure fp32 NumPy arithmetic
        score, pred = ref.score_removed_loop(sd, arch, x, seg, lists, int(g[p + "label"]))
        assert np.abs(score - g[p + "score_f32"]).max() <= 2e-6 and (pred == g[p + "pred"]).all()
        assert (ref.removed_onoff(seg, lists) == removed).all()


def test_mask_convention_by_hand():
    """generate_gp_training_data_cifar.py:274-321 on a 2x2 picture: min-max to [0,255], {0,255} mask with the SELECTED superpixel
    switched OFF, min-max again, * f32(1/255)."""
    x = np.array([[[-1.0, 0.0], [0.5, 1.0]]], dtype=np.float32)             # [1,2,2]
    seg = np.array([[0, 0], [1, 2]])
    org = ref.org_img_minmax255(x)
    assert np.allclose(org, [[[0.0, 127.5], [191.25, 255.0]]])
    assert (x == np.array([[[-1.0, 0.0], [0.5, 1.0]]], dtype=np.float32)).all()      # the caller's tensor is not touched here
    mask = ref.removed_mask_u8(seg, [2])
    assert mask.tolist() == [[255, 255], [255, 0]]
    inp = ref.masked_input(org, mask)
    # kept pixels: x255*255 / (191.25*255) * 255 / 255 ; removed pixel 0
    assert np.allclose(inp, [[[0.0, 127.5 / 191.25], [1.0, 0.0]]], atol=1e-6)
    none = ref.masked_input(org, ref.removed_mask_u8(seg, []))
    assert np.allclose(none, org / 255.0, atol=1e-6)                          # nothing removed: the picture in [0,1]
    with np.errstate(invalid="ignore", divide="ignore"):
        assert np.isnan(ref.masked_input(org, ref.removed_mask_u8(seg, [0, 1, 2]))).all()    # everything removed: 0/0 upstream too


def test_downsample_b_and_block_structure():
    x = torch.arange(2 * 16 * 4 * 4, dtype=torch.float32).view(2, 16, 4, 4)
    y = ref._downsample_b(x, 16, 32, 2)
    assert y.shape == (2, 32, 2, 2) and (y[:, 16:] == 0).all()
    assert torch.equal(y[:, :16], torch.nn.functional.avg_pool2d(x, 2))
