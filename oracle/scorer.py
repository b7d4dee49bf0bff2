"""CPU ORACLE (test infrastructure only) -- masked-perturbation scorer semantics.

PARITY UNPINNED (see oracle/resnet_ref.py header: the reference holds no tests or
fixtures for this path and cannot be imported in this image).  Every function
restates specific reference lines; file:line citations are relative to
/root/reference.  Deliberately literal and slow: one mask at a time, batch = 1,
exactly as the reference runs it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import resnet_ref

# transforms.Normalize constants, generate_gp_training_data_imagenet.py:590-591
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
SUPERPIXEL_FRACTION = 0.4   # generate_gp_training_data_imagenet.py:224
BO_UPPER_FRACTION = 0.6     # bayesian_active_learning_imagenet.py:467


def to_tensor_normalize(img_u8_hwc):
    """ToTensor + Normalize (generate_gp_training_data_imagenet.py:598-599):
    u8[H,W,3] -> f32[3,H,W]; x = (u8/255 - mean_c)/std_c, each step rounded in fp32
    the way torchvision does it (`.div(255)`, `.sub_(mean).div_(std)`)."""
    x = torch.from_numpy(np.ascontiguousarray(img_u8_hwc)).permute(2, 0, 1).contiguous()
    x = x.to(torch.float32).div(255)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)
    return x.sub_(mean).div_(std)


def img_show_u8(x_chw):
    """min-max rescale to uint8 HWC, the image felzenszwalb sees
    (generate_gp_training_data_imagenet.py:171-178)."""
    img = x_chw.numpy().copy().transpose(1, 2, 0)
    img -= img.min()
    img /= img.max()
    img *= 255
    return img.astype(np.uint8)


def num_conse_superpixels(total_num_segments):
    """generate_gp_training_data_imagenet.py:224 -- int(0.4*S)."""
    return int(SUPERPIXEL_FRACTION * total_num_segments)


def draw_first_index(rng, total_num_segments):
    """generate_gp_training_data_imagenet.py:227 -- randint(1, S-k), inclusive both ends.
    `rng` is a random.Random (the reference uses the unseeded module-level one)."""
    return rng.randint(1, total_num_segments - num_conse_superpixels(total_num_segments))


def bo_upper_bound(total_num_segments):
    """bayesian_active_learning_imagenet.py:467 -- ub = int(0.6*S)."""
    return int(BO_UPPER_FRACTION * total_num_segments)


def window_mask_u8(segments, first_index):
    """Pixel mask of one window, restating generate_gp_training_data_imagenet.py:223-237
    (= bayesian_active_learning_imagenet.py:173-185): the k consecutive entries of
    np.unique(segments) starting at first_index are switched ON (1), the rest stay 0.
    A slice running off the end silently truncates, as in the reference."""
    total = len(np.unique(segments))
    k = num_conse_superpixels(total)
    random_sampled_list = np.unique(segments)[first_index:(first_index + k)]
    mask = np.zeros(segments.shape[:2], dtype="uint8")
    for seg_val in random_sampled_list:
        mask[segments == seg_val] = 1
    return mask


def onoff_mask_u8(segments, onoff_row):
    """Generalisation used by the batched API: onoff_row[j] says whether the j-th entry
    of np.unique(segments) is kept.  window_mask_u8(seg, f) ==
    onoff_mask_u8(seg, window_onoff(S, f))."""
    uniq = np.unique(segments)
    mask = np.zeros(segments.shape[:2], dtype="uint8")
    for j, seg_val in enumerate(uniq):
        if onoff_row[j]:
            mask[segments == seg_val] = 1
    return mask


def window_onoff(total_num_segments, first_index):
    """mask-vector in {0,1}^S of the window [first_index, first_index+k)."""
    k = num_conse_superpixels(total_num_segments)
    row = np.zeros(total_num_segments, dtype=np.uint8)
    row[first_index:first_index + k] = 1
    return row


def apply_mask(x_chw, mask_u8):
    """generate_gp_training_data_imagenet.py:240 -- normalise THEN mask:
    `input[0].numpy().copy() * mask` (f32[3,H,W] * u8[H,W] -> f32, broadcast over C)."""
    return x_chw.numpy().copy() * mask_u8


def score_one(sd, arch, masked_chw, label):
    """One batch-1 forward + score extraction.
    returns (class_prob_score np.float32, pred int):
      bayesian_active_learning_imagenet.py:189-198  softmax(logits)[0][label]
      generate_gp_training_data_imagenet.py:248      logits.max(1)[1]"""
    t = torch.from_numpy(masked_chw[None, :, :, :])
    with torch.no_grad():
        logits = resnet_ref.forward(sd, t, arch)
        prob = F.softmax(logits, dim=1)
    return prob.numpy()[0][label], int(logits.max(1, keepdim=True)[1][0, 0])


def base_prediction(sd, arch, x_chw):
    """Unmasked forward + argmax (generate_gp_training_data_imagenet.py:193,202)."""
    with torch.no_grad():
        logits = resnet_ref.forward(sd, x_chw[None], arch)
    return int(logits.max(1, keepdim=True)[1][0, 0])


def score_masks_reference_loop(sd, arch, x_chw, segments, onoff, label):
    """The reference hot loop (generate_gp_training_data_imagenet.py:221-266) with the BO
    script's score (bayesian_active_learning_imagenet.py:196-198): for each mask-vector,
    build the pixel mask, multiply into the normalised image, run ONE batch-1 forward.
    returns (score f32[M], pred i64[M]).  No PNG writes, no visualisation copies."""
    m = onoff.shape[0]
    score = np.zeros(m, dtype=np.float32)
    pred = np.zeros(m, dtype=np.int64)
    for i in range(m):
        mask = onoff_mask_u8(segments, onoff[i])
        masked = apply_mask(x_chw, mask)
        score[i], pred[i] = score_one(sd, arch, masked, label)
    return score, pred


def score_masks_batched(sd, arch, x_chw, segments, onoff, label, dtype=torch.float32, chunk=16):
    """Same result as score_masks_reference_loop, forwards run `chunk` at a time (torch CPU
    kernels are batch-invariant up to accumulation order; used for fp64 yardsticks and to
    keep CPU test time down)."""
    sd = resnet_ref.cast_state_dict(sd, dtype)
    m = onoff.shape[0]
    score = np.zeros(m, dtype=np.float64)
    pred = np.zeros(m, dtype=np.int64)
    for s in range(0, m, chunk):
        batch = np.stack([apply_mask(x_chw, onoff_mask_u8(segments, onoff[i]))
                          for i in range(s, min(m, s + chunk))])
        with torch.no_grad():
            logits = resnet_ref.forward(sd, torch.from_numpy(batch).to(dtype), arch)
            prob = F.softmax(logits, dim=1)
        score[s:s + len(batch)] = prob[:, label].double().numpy()
        pred[s:s + len(batch)] = logits.argmax(1).numpy()
    return score, pred


def summed_superpixel_labels(segments, onoff, correct):
    """gp_superpixel_data_imagenet.py:322-323 / gp_regression.py:82-94:
    y[p] = sum_i label_i * mask_i[p] (f64[H,W])."""
    acc = np.zeros(segments.shape[:2], dtype=np.float64)
    for i in range(onoff.shape[0]):
        if correct[i]:
            acc += onoff_mask_u8(segments, onoff[i])
    return acc
