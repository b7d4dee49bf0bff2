"""Property tests (hypothesis) of the integer host logic: mask-vectors, segment ranking, sharding."""
import numpy as np
from hypothesis import given, settings, strategies as st

from network_interpretation_imagenet_amd import masks, shard
from network_interpretation_imagenet_amd.engine import rank_segments
from oracle import scorer


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 400), st.integers(-5, 420))
def test_window_onoff_matches_reference_slice(s, first):
    row = masks.window_onoff(s, first)
    want = np.zeros(s, dtype=np.uint8)
    want[np.arange(s)[first:first + int(0.4 * s)]] = 1       # the reference's slice np.unique(seg)[f:f+k]
    assert (row == want).all() and (row == scorer.window_onoff(s, first)).all()
    assert row.sum() <= masks.window_size(s)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_rank_segments_is_order_preserving_relabelling(seed):
    rng = np.random.default_rng(seed)
    labels = np.sort(rng.choice(10_000, size=rng.integers(1, 40), replace=False)) - 5000
    seg = labels[rng.integers(0, len(labels), size=(224, 224))]
    rank, s = rank_segments(seg)
    uniq = np.unique(seg)
    assert s == len(uniq) and rank.min() == 0 and rank.max() == s - 1
    assert (uniq[rank] == seg).all()                           # rank r <-> r-th entry of np.unique(segments)
    onoff = (rng.random(s) < 0.5).astype(np.uint8)
    assert (masks.expand_pixel_mask(rank, onoff) == scorer.onoff_mask_u8(seg, onoff)).all()


@settings(max_examples=100, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 16), st.integers(1, 64))
def test_shard_blocks_partition_the_range(total, world, per_image):
    covered = []
    for r in range(world):
        lo, hi = shard.block(total, r, world)
        assert 0 <= lo <= hi <= total
        pieces = shard.image_ranges(lo, hi, per_image)
        flat = [img * per_image + m for img, m_lo, m_hi in pieces for m in range(m_lo, m_hi)]
        assert flat == list(range(lo, hi))
        assert all(0 <= m_lo < m_hi <= per_image for _i, m_lo, m_hi in pieces)
        covered += flat
    assert covered == list(range(total))
