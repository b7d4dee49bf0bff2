"""An independent check of the oracle's TOPOLOGY restatement (oracle/resnet_ref.py restates torchvision's models/resnet.py from its
published structure; torchvision itself is not installed here and the reference cannot be imported -- SURVEY.md 8c).  Hugging Face
`transformers` ships its own implementation of the same ResNet v1.5 family (ResNetForImageClassification: 7x7 stem + max pool, basic /
bottleneck layers with the stride on the 3x3 conv, projection shortcuts, global average pool, linear head).  With the SAME seeded weights
under both key sets the two implementations must produce the same logits.  This pins block order, strides, paddings, shortcut placement,
BatchNorm epsilon and the pooling of the oracle against code written by third parties; it does not pin the oracle against a run of the
reference (nothing can, offline): the parity grade stays "unpinned"."""
import numpy as np
import pytest
import torch

from network_interpretation_imagenet_amd import synth
from oracle import resnet_ref as R

transformers = pytest.importorskip("transformers")


def _hf_key(name):
    """torchvision state_dict prefix -> transformers' module path."""
    if name == "conv1":
        return "resnet.embedder.embedder.convolution"
    if name == "bn1":
        return "resnet.embedder.embedder.normalization"
    if name == "fc":
        return "classifier.1"
    stage, block, rest = name.split(".", 2)
    base = "resnet.encoder.stages.%d.layers.%s" % (int(stage[len("layer"):]) - 1, block)
    if rest.startswith("downsample"):
        return base + ".shortcut." + ("convolution" if rest.endswith("0") else "normalization")
    kind, idx = rest[:-1], int(rest[-1]) - 1                  # conv1 / bn2 ...
    return base + ".layer.%d.%s" % (idx, "convolution" if kind == "conv" else "normalization")


@pytest.mark.parametrize("arch", ["resnet18", "resnet50", "resnet101"])
def test_oracle_forward_equals_the_transformers_implementation(arch):
    from transformers import ResNetConfig, ResNetForImageClassification
    kind, depths = R.ARCHS[arch]
    exp = 1 if kind == "basic" else 4
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[w * exp for w in R.STAGE_WIDTH], depths=list(depths),
                       layer_type=kind, hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False, num_labels=1000)
    hf = ResNetForImageClassification(cfg).eval()
    sd = synth.make_state_dict(arch)
    mapped = {}
    for k, v in sd.items():
        prefix, field = k.rsplit(".", 1)
        mapped[_hf_key(prefix) + "." + field] = v
    want_keys = {k for k in hf.state_dict() if not k.endswith("num_batches_tracked")}
    assert set(mapped) == want_keys                             # every tensor of either implementation has a partner, none is left over
    missing, unexpected = hf.load_state_dict(mapped, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    x = torch.from_numpy(np.stack([np.asarray(synth.make_images(2, seed=5, kind="blobs")[i], dtype=np.float32).transpose(2, 0, 1) / 255.0
                                   for i in range(2)]))
    with torch.no_grad():
        got = R.forward(sd, x, arch)
        want = hf(pixel_values=x).logits
    scale = float(want.abs().max())
    assert got.shape == want.shape == (2, 1000)
    assert float((got - want).abs().max()) <= 1e-5 * scale      # the same torch ops in the same order: equal up to the BatchNorm formulation
    assert (got.argmax(1) == want.argmax(1)).all()
