"""CPU ORACLE (test infrastructure only) -- the reference's two small networks and their superpixel scorer
(SURVEY.md 8 row f4).

PARITY: pinned by REAL TRAINED WEIGHTS only.  The reference ships the checkpoints of both networks
(saved_checkpoints/mnist/checkpoint.pth.tar, saved_checkpoints/cifar10+-resnet-56/model_best.pth.tar); they are read
with torch.load(weights_only=True) by tests/golden/make_smallnets_golden.py, which stores the weights, seeded inputs and
the logits THIS restatement computes.  The reference scripts themselves cannot be imported (cv2, torchvision absent;
generate_gp_training_data_mnist.py runs argparse and builds data loaders at import), so no output of the reference's
own run exists: the networks below restate
    Classification_Net            generate_gp_training_data_mnist.py:72-105
    ResNetCifar + BasicBlock...    models/resnet.py:10-41,64-74,77-146   (eval mode: death rates play no role, :31)
and the scorer restates the masking convention of
    generate_gp_training_data_cifar.py:271-321 and generate_gp_training_data_mnist.py:163-243
with the same torch / numpy CPU ops, file:line cited per function.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5      # nn.BatchNorm2d default in both files


def _bn(sd, x, prefix):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], False, 0.0, BN_EPS)


# ---------------------------------------------------------------------------------------------------------------
# MNIST: Classification_Net (generate_gp_training_data_mnist.py:86-105); conv() = Conv2d(3x3, pad 1, bias) + BN + ReLU (:72-77)
# ---------------------------------------------------------------------------------------------------------------
MNIST_CONVS = [("conv1", 1, 32, 1), ("conv2", 32, 32, 1), ("conv3", 32, 64, 2), ("conv4", 64, 64, 1), ("conv5", 64, 128, 2)]


def mnist_net_forward(sd, x, taps=None):
    """x f32/f64[B,1,28,28] -> pred0 [B,10]  (forward(), :97-105; x0/x1/x2 are returned upstream too and unused by the scorer)."""
    for name, _cin, _cout, stride in MNIST_CONVS:
        x = F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride, 1)
        x = F.relu(_bn(sd, x, name + ".1"))
        if taps is not None:
            taps[name] = x
    x2 = F.conv2d(x, sd["conv6.weight"], sd["conv6.bias"], 1, 1)           # plain nn.Conv2d(128, 128, 3, padding=1), :94
    if taps is not None:
        taps["conv6"] = x2
    f = x2.mean(3).mean(2)                                                  # :101
    return F.linear(f, sd["fc1.weight"], sd["fc1.bias"])


# ---------------------------------------------------------------------------------------------------------------
# CIFAR: ResNetCifar(depth = 6n+2) with BasicBlockWithDeathRate and DownsampleB (models/resnet.py)
# ---------------------------------------------------------------------------------------------------------------
def cifar_resnet_blocks(depth):
    assert (depth - 2) % 6 == 0
    n = (depth - 2) // 6
    out = []
    inplanes = 16
    for stage, planes in enumerate((16, 32, 64)):
        for b in range(n):
            stride = 2 if (stage > 0 and b == 0) else 1
            out.append(("layer%d.%d" % (stage + 1, b), inplanes, planes, stride))
            inplanes = planes
    return out


def _downsample_b(x, n_in, n_out, stride):
    """DownsampleB.forward (models/resnet.py:64-74): AvgPool2d(stride), then zero channels appended."""
    x = F.avg_pool2d(x, stride)
    return torch.cat([x] + [x.mul(0)] * (n_out // n_in - 1), 1)


def cifar_resnet_forward(sd, x, depth=56, taps=None):
    """x [B,3,32,32] -> logits [B,10]  (ResNetCifar.forward, models/resnet.py:131-146; block :26-41 in eval mode)."""
    x = F.relu(_bn(sd, F.conv2d(x, sd["conv1.weight"], None, 1, 1), "bn1"))
    if taps is not None:
        taps["conv1"] = x
    for name, inplanes, planes, stride in cifar_resnet_blocks(depth):
        residual = x
        if stride != 1 or inplanes != planes:
            x = _downsample_b(x, inplanes, planes, stride)
        residual = F.relu(_bn(sd, F.conv2d(residual, sd[name + ".conv1.weight"], None, stride, 1), name + ".bn1"))
        residual = _bn(sd, F.conv2d(residual, sd[name + ".conv2.weight"], None, 1, 1), name + ".bn2")
        x = F.relu(x + residual)
        if taps is not None:
            taps[name] = x
    x = F.avg_pool2d(x, 8)
    return F.linear(x.view(x.size(0), -1), sd["fc.weight"], sd["fc.bias"])


def forward(sd, x, arch, taps=None):
    if arch == "mnist_net":
        return mnist_net_forward(sd, x, taps)
    if arch.startswith("cifar_resnet"):
        return cifar_resnet_forward(sd, x, int(arch[len("cifar_resnet"):]), taps)
    raise ValueError(arch)


# ---------------------------------------------------------------------------------------------------------------
# The scorer's masking convention (selected superpixels are switched OFF; re-min-max; /255)
# ---------------------------------------------------------------------------------------------------------------
def org_img_minmax255(x_chw):
    """The picture the scorer multiplies with the mask.  Upstream `img = org_img.transpose(1,2,0)` is a VIEW and
    `img -= img.min(); img /= img.max(); img *= 255` run in place (generate_gp_training_data_cifar.py:274-279,
    ..._mnist.py:167-171): org_img itself ends up min-max scaled to [0, 255] in fp32."""
    org = np.array(x_chw, dtype=np.float32, copy=True)
    img = org.transpose(1, 2, 0)
    img -= img.min()
    img /= img.max()
    img *= 255
    return org


def removed_mask_u8(segments, removed_values):
    """mask.fill(255); mask[segments == segVal] = 0 for the sampled superpixels (..._cifar.py:310-313, ..._mnist.py:213-217)."""
    mask = np.zeros(segments.shape[:2], dtype="uint8")
    mask.fill(255)
    for seg_val in removed_values:
        mask[segments == seg_val] = 0
    return mask


def masked_input(org255_chw, mask_u8):
    """`masked_img = org_img * mask` (f32 * u8{0,255}), then in place `-= min; /= max; *= 255` and normalize_image
    (utils.py:92-94: np.multiply(image.astype(np.float32), 1.0 / 255.0))  -- ..._cifar.py:315-321; the MNIST script reaches the
    same arithmetic through its `pic` view (..._mnist.py:220-242).  -> f32[C,H,W], the network's input."""
    masked = org255_chw * mask_u8
    masked -= masked.min()
    masked /= masked.max()
    masked *= 255
    return np.multiply(masked.astype(np.float32), 1.0 / 255.0)


def score_removed_loop(sd, arch, x_chw, segments, removed_lists, label):
    """The reference loop, one mask at a time (..._cifar.py:305-335, ..._mnist.py:196-262): build the mask, batch-1 forward,
    argmax (`mask_output.data.max(1, keepdim=True)[1]`) and the softmax probability of `label`.
    returns (score f32[M], pred i64[M])."""
    org = org_img_minmax255(x_chw)
    m = len(removed_lists)
    score = np.zeros(m, dtype=np.float32)
    pred = np.zeros(m, dtype=np.int64)
    for i, removed in enumerate(removed_lists):
        inp = masked_input(org, removed_mask_u8(segments, removed))
        with torch.no_grad():
            logits = forward(sd, torch.from_numpy(inp[None]), arch)
            prob = F.softmax(logits, dim=1)
        score[i] = prob.numpy()[0][label]
        pred[i] = int(logits.max(1, keepdim=True)[1][0, 0])
    return score, pred


def removed_onoff(segments, removed_lists):
    """u8[M,S] over np.unique(segments) order: 1 = superpixel REMOVED (the batched API's mask-vector for this convention)."""
    uniq = np.unique(segments)
    out = np.zeros((len(removed_lists), len(uniq)), dtype=np.uint8)
    for i, removed in enumerate(removed_lists):
        for v in removed:
            j = np.searchsorted(uniq, v)
            if j < len(uniq) and uniq[j] == v:
                out[i, j] = 1
    return out
