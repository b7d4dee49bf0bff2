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


def test_slic_label_maps_rank_like_np_unique():
    """BASELINE configs[4] names SLIC label maps (generate_superpixels.py:2): the two committed scikit-image 0.18.3 maps -- labels 0 .. 89
    and labels 1 .. 94 (a map that does not start at 0) -- rank like np.unique, and every window of the BO domain expands to the pixel mask
    the reference's literal loop builds (bayesian_active_learning_imagenet.py:173-185)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "segments_slic.npz"))
    assert [int(s.min()) for s in g["segments"]] == [0, 1]
    for seg in g["segments"].astype(np.int64):
        rank, s = rank_segments(seg)
        uniq = np.unique(seg)
        assert s == len(uniq) and rank.dtype == np.int32 and (uniq[rank] == seg).all()
        for f in (0, 1, s // 3, masks.bo_upper_bound(s)):
            assert (masks.expand_pixel_mask(rank, masks.window_onoff(s, f)) == scorer.window_mask_u8(seg, f)).all()


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(0, 700), min_size=1, max_size=9), st.sampled_from([64, 257, 512, 1000]), st.sampled_from([None, "conv", "table"]))
def test_score_images_packs_any_job_with_one_kind_per_forward(sizes, max_batch, stem):
    """MaskedForwardEngine.score_images / score_packed over recorded kernel calls (tests/test_host_logic._StagingProbe: no library, no GPU):
    for ANY mix of per-image row counts, batch size and `stem` argument every row comes back where it belongs, every image is staged by its
    own row count (or by the caller's `stem`), no forward mixes the two kinds, no forward exceeds max_batch, and the number of forwards is
    the least possible for one change of kind."""
    from test_host_logic import _StagingProbe, _probe_job
    eng = _StagingProbe(max_batch=max_batch)
    out = eng.score_images(*_probe_job(sizes), **({} if stem is None else {"stem": stem}))
    assert [len(s) for s, _p in out] == sizes
    for i, (score, pred) in enumerate(out):
        assert np.array_equal(score, i * 1000 + np.arange(sizes[i], dtype=np.float32))
        kind = stem or ("table" if sizes[i] >= 256 else "conv")
        assert (pred == (2 if kind == "table" else 1)).all()
    assert all(len(k) == 1 and 0 < b <= max_batch for b, k in eng.forwards)
    rows = {"table": 0, "conv": 0}
    for m in sizes:
        rows[stem or ("table" if m >= 256 else "conv")] += m
    assert len(eng.forwards) == sum(-(-r // max_batch) for r in rows.values())
    assert sorted(eng.builds) == sorted(i for i, m in enumerate(sizes) if m and (stem or ("table" if m >= 256 else "conv")) == "table")
