"""Mask-vector construction (host logic, integer only).

A mask-vector is onoff in {0,1}^S over the S distinct superpixel labels in np.unique(segments)
order; the reference only ever uses contiguous windows of k = int(0.4*S) labels
(generate_gp_training_data_imagenet.py:223-230, bayesian_active_learning_imagenet.py:173-178).
"""
import random

import numpy as np

SUPERPIXEL_FRACTION = 0.4   # generate_gp_training_data_imagenet.py:224
BO_UPPER_FRACTION = 0.6     # bayesian_active_learning_imagenet.py:467


def window_size(num_segments):
    """num_conse_superpixels = int(0.4*total_num_segments)."""
    return int(SUPERPIXEL_FRACTION * num_segments)


def bo_upper_bound(num_segments):
    """ub of the BO domain firstIndex in [0, ub] (bayesian_active_learning_imagenet.py:467,478)."""
    return int(BO_UPPER_FRACTION * num_segments)


def window_onoff(num_segments, first_index):
    """u8[S]: 1 on [first_index, first_index+k).  Like the reference's slice
    `np.unique(segments)[firstIndex:firstIndex+k]` a window running off the end truncates and a
    negative first_index follows Python slice semantics."""
    row = np.zeros(num_segments, dtype=np.uint8)
    k = window_size(num_segments)
    row[int(first_index):int(first_index) + k] = 1
    return row


def windows_onoff(num_segments, first_indices):
    """u8[M,S] for a list of window starts."""
    idx = list(first_indices)
    out = np.zeros((len(idx), num_segments), dtype=np.uint8)
    for i, f in enumerate(idx):
        out[i] = window_onoff(num_segments, f)
    return out


def draw_first_indices(num_segments, count, rng=None):
    """firstIndex = randint(1, S-k), both ends inclusive
    (generate_gp_training_data_imagenet.py:227).  rng: random.Random (the reference uses the
    unseeded module-level generator)."""
    rng = rng or random
    k = window_size(num_segments)
    if num_segments - k < 1:
        raise ValueError("too few superpixels (%d) to draw a window start" % num_segments)
    return [rng.randint(1, num_segments - k) for _ in range(count)]


def expand_pixel_mask(seg_rank, onoff_row):
    """u8[H,W] in {0,1}: mask[y,x] = onoff[seg[y,x]] (host mirror of what K0 does on the device;
    used for the PNG wire format and heat-map reduction)."""
    return np.asarray(onoff_row, dtype=np.uint8)[seg_rank]


# ---- the small networks' sampler ------------------------------------------------------------------------
def draw_removed_sets(uniq, k, count, rng=None, burn_window_draw=False):
    """The CIFAR / MNIST scorers' draw: `random.sample(range(np.unique(segments)[0], np.unique(segments)[-1]), k)` per mask
    (generate_gp_training_data_cifar.py:308 with k = 5, generate_gp_training_data_mnist.py:215 with k = 1) -> `count` lists of k
    superpixel LABEL VALUES to switch off.  The population is range(first, last): the LAST superpixel can never be drawn, and with
    gaps in the labels a drawn value may name no superpixel (then `mask[segments == v] = 0` removes nothing).  Like
    random.sample it raises ValueError when k exceeds the population.  burn_window_draw: the MNIST script first draws an unused
    `firstIndex = randint(1, S - 1)` per mask (:210) -- consume it too, so that a seeded generator stays in step with upstream.
    rng: random.Random (the reference uses the unseeded module-level generator)."""
    rng = rng or random
    uniq = np.asarray(uniq)
    lo, hi = int(uniq[0]), int(uniq[-1])
    out = []
    for _ in range(int(count)):
        if burn_window_draw:
            rng.randint(1, len(uniq) - 1)
        out.append(rng.sample(range(lo, hi), int(k)))
    return out


def removed_onoff(uniq, removed_sets):
    """u8[M,S]: removed[m][s] = 1 when the s-th label of np.unique(segments) is in removed_sets[m] (the layout
    MaskedForwardEngine.score_masks_removed / mpx_mask_apply_minmax take)."""
    uniq = np.asarray(uniq)
    out = np.zeros((len(removed_sets), len(uniq)), dtype=np.uint8)
    for m, vals in enumerate(removed_sets):
        if len(vals):
            out[m] = np.isin(uniq, np.asarray(list(vals)))
    return out


def removed_mask_u8(segments, removed_set):
    """u8[H,W] in {0,255}: `mask.fill(255); mask[segments == segVal] = 0` for the drawn values
    (generate_gp_training_data_cifar.py:310-313) -- the picture the scorers write as mask_{i}_{label}.png."""
    seg = np.asarray(segments)
    mask = np.full(seg.shape, 255, dtype=np.uint8)
    if len(removed_set):
        mask[np.isin(seg, np.asarray(list(removed_set)))] = 0
    return mask
