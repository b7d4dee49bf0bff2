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
