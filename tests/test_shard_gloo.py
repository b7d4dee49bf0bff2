"""N>1 path on CPU: two gloo ranks shard a flattened (image, mask) range and all-gather scores;
the result must equal the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from network_interpretation_imagenet_amd import shard


def _fake_score(img, m_lo, m_hi):
    """Deterministic stand-in for the engine: a pure function of (image, mask index)."""
    m = torch.arange(m_lo, m_hi, dtype=torch.float64)
    return (torch.sin(m * 0.37 + img * 1.3) * 0.5 + 0.5).to(torch.float32)


def _worker(rank, world, port, n_img, n_mask, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def fn(img, lo, hi):
            calls.append((img, lo, hi))
            return _fake_score(img, lo, hi)

        full = shard.score_sharded(fn, n_img, n_mask, torch.device("cpu"))
        np.save(os.path.join(out_dir, "r%d.npy" % rank), full.numpy())
        np.save(os.path.join(out_dir, "calls%d.npy" % rank), np.array(calls, dtype=np.int64).reshape(-1, 3))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_img,n_mask", [(4, 16), (3, 7), (1, 5)])
def test_two_rank_gather_matches_single_process(tmp_path, n_img, n_mask):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_img, n_mask, str(tmp_path)), nprocs=world, join=True)
    want = torch.cat([_fake_score(i, 0, n_mask) for i in range(n_img)]).numpy()
    single = shard.score_sharded(_fake_score, n_img, n_mask, torch.device("cpu")).numpy()
    assert (single == want).all()
    covered = 0
    for r in range(world):
        got = np.load(tmp_path / ("r%d.npy" % r))
        assert got.shape == want.shape and (got == want).all()      # bit-identical on every rank
        calls = np.load(tmp_path / ("calls%d.npy" % r))
        covered += int((calls[:, 2] - calls[:, 1]).sum())
    assert covered == n_img * n_mask                                  # disjoint blocks, nothing scored twice


class _TableEngine:
    """Engine stand-in whose score is a pure function of the mask-vector (no GPU, no oracle)."""
    device = torch.device("cpu")

    def score_masks(self, image, segments, onoff, label):
        w = np.arange(1, onoff.shape[1] + 1, dtype=np.float64)
        v = (onoff.astype(np.float64) * w).sum(1)
        return onoff, (np.sin(v) * 0.5 + 0.5).astype(np.float32), (v.astype(np.int64) % 7).astype(np.int32)

    def heatmap(self, seg_rank, onoff, pred, label):
        per_segment = (onoff * (np.asarray(pred) == label)[:, None]).sum(0).astype(np.float64)
        return per_segment[seg_rank]


def _mask_worker(rank, world, port, m, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        onoff = (rng.random((m, 23)) < 0.4).astype(np.uint8)
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda out, t, *a, **k: (calls.append(int(t.numel())), real(out, t, *a, **k))[1]
        try:
            score, pred = shard.score_masks_sharded(_TableEngine(), None, None, onoff, 0)
        finally:
            dist.all_gather_into_tensor = real
        np.savez(os.path.join(out_dir, "m%d.npz" % rank), score=score, pred=pred, collectives=np.array(calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m", [24, 7, 1])
def test_mask_axis_sharding_single_image(tmp_path, m):
    """Config-5 shape: one image, the mask axis split over two ranks, ONE all-gather carrying scores and predictions
    (SURVEY.md 8e), counted."""
    mp.spawn(_mask_worker, args=(2, _free_port(), m, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(5)
    onoff = (rng.random((m, 23)) < 0.4).astype(np.uint8)
    _o, want_s, want_p = _TableEngine().score_masks(None, None, onoff, 0)
    single = shard.score_masks_sharded(_TableEngine(), None, None, onoff, 0)
    assert (single[0] == want_s).all() and (single[1] == want_p).all()
    for r in range(2):
        got = np.load(tmp_path / ("m%d.npz" % r))
        assert got["score"].dtype == np.float32 and got["pred"].dtype == np.int32
        assert (got["score"] == want_s).all() and (got["pred"] == want_p).all()
        assert got["collectives"].tolist() == [2 * (-(-m // 2))]            # one collective: score bits + preds of the widest block


class _StagingEngine(_TableEngine):
    """Stand-in that has the engine's staging rule (stem_for_rows, threshold 256) and records the `stem` every call was handed."""
    stem_table_min_rows = 256

    def __init__(self):
        self.seen = []

    def stem_for_rows(self, rows):
        return "table" if rows >= self.stem_table_min_rows else "conv"

    def score_masks(self, image, segments, onoff, label, stem=None):
        self.seen.append((int(onoff.shape[0]), stem))
        return _TableEngine.score_masks(self, image, segments, onoff, label)


def _staging_worker(rank, world, port, m, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        onoff = (np.random.default_rng(5).random((m, 23)) < 0.4).astype(np.uint8)
        eng = _StagingEngine()
        shard.score_masks_sharded(eng, None, None, onoff, 0)
        shard.heatmap_sharded(eng, None, _seg23(), onoff, 3)
        with open(os.path.join(out_dir, "s%d.txt" % rank), "w") as fh:
            fh.write(repr(eng.seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m,want", [(400, "table"), (512, "table"), (255, "conv"), (40, "conv")])
def test_staging_is_chosen_from_the_global_row_count(tmp_path, m, want):
    """SURVEY 8(e) "bit-identical to the 1-GPU run": the stem table and K0 + the MFMA stem round differently, so the choice must not
    depend on how many rows a rank holds.  400 rows are table-staged on one engine; two ranks of 200 (< 256 each) must be handed
    stem="table" too -- and 255 rows stay on "conv" however they are cut."""
    mp.spawn(_staging_worker, args=(2, _free_port(), m, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        seen = eval(open(tmp_path / ("s%d.txt" % r)).read())
        lo, hi = shard.block(m, r, 2)
        assert seen == [(hi - lo, want), (hi - lo, want)]           # the mask-axis split and the heat map's scoring call
    single = _StagingEngine()
    shard.score_masks_sharded(single, None, None, (np.random.default_rng(5).random((m, 23)) < 0.4).astype(np.uint8), 0)
    assert single.seen == [(m, want)]
    assert shard.job_stem(_TableEngine(), 400) is None                # stand-ins without the rule are called without `stem`


def _seg23():
    idx = np.arange(224) // 10
    return ((idx[:, None] + idx[None, :]) % 23).astype(np.int32)


def _heat_worker(rank, world, port, m, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        onoff = (rng.random((m, 23)) < 0.4).astype(np.uint8)
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(int(t.numel())), real(t, *a, **k))[1]
        try:
            heat, n_ok = shard.heatmap_sharded(_TableEngine(), None, _seg23(), onoff, 3)
        finally:
            dist.all_reduce = real
        np.savez(os.path.join(out_dir, "h%d.npz" % rank), heat=heat.numpy(), n_ok=n_ok, collectives=np.array(calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m", [40, 5, 1])
def test_heat_map_all_reduce_two_ranks(tmp_path, m):
    """per-rank partial heat maps (K5) + ONE all_reduce of f32[224,224] per image = the single-engine
    heat map, exactly, on every rank."""
    mp.spawn(_heat_worker, args=(2, _free_port(), m, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(11)
    onoff = (rng.random((m, 23)) < 0.4).astype(np.uint8)
    eng = _TableEngine()
    _o, _s, pred = eng.score_masks(None, None, onoff, 3)
    want = eng.heatmap(_seg23(), onoff, pred, 3)
    single, n_single = shard.heatmap_sharded(eng, None, _seg23(), onoff, 3)
    assert (single.numpy().astype(np.float64) == want).all() and n_single == int((pred == 3).sum())
    for r in range(2):
        got = np.load(tmp_path / ("h%d.npz" % r))
        assert got["heat"].dtype == np.float32 and (got["heat"].astype(np.float64) == want).all()
        assert int(got["n_ok"]) == n_single
        assert got["collectives"].tolist() == [224 * 224 + 1]          # ONE all_reduce per image: the map with the count as its last element
