"""N>1 path with the REAL engine: two processes, one MaskedForwardEngine each (both on the one GPU of the test box), gloo for the
collectives -- the sharding code of shard.py (BASELINE cfg-5's mask-axis split, the image-first split of cfg-4 and the heat-map
all-reduce) end to end on hardware results.  RCCL itself needs one GPU per rank; the driver's 8-GPU run covers it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_IMG, N_MASK, S = 3, 37, 64
N_WIDE = 400        # one image's rows in the wide case: >= stem_table_min_rows (256) as a whole, 200 < 256 per rank


def _inputs():
    from network_interpretation_imagenet_amd import synth
    imgs = synth.make_images(N_IMG, seed=77, kind="blobs")
    seg = synth.grid_segments(block=28)
    onoff = [synth.random_onoff(N_MASK, S, seed=90 + i) for i in range(N_IMG)]
    return imgs, seg, onoff


def _wide():
    from network_interpretation_imagenet_amd import synth
    return synth.random_onoff(N_WIDE, S, seed=321)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from network_interpretation_imagenet_amd import shard, synth
        from network_interpretation_imagenet_amd.engine import MaskedForwardEngine
        eng = MaskedForwardEngine("resnet18", max_batch=32, device=0).load_state_dict(synth.make_state_dict("resnet18"))
        imgs, seg, onoff = _inputs()
        # cfg-5: one image, the mask axis split over the ranks, one all-gather
        score, pred = shard.score_masks_sharded(eng, imgs[0], seg, onoff[0], 5)
        # heat map of one image: per-rank K5 into the device buffer, ONE all-reduce (map + count)
        heat, n_ok = shard.heatmap_sharded(eng, imgs[1], seg, onoff[1], int(pred[0]))
        # cfg-4 shape: the flat (image, mask) range cut image-first, one all-gather of the scores
        def fn(i, lo, hi):
            return torch.from_numpy(eng.score_masks(imgs[i], seg, onoff[i][lo:hi], 5, stem=shard.job_stem(eng, N_MASK))[1])
        flat = shard.score_sharded(fn, N_IMG, N_MASK, torch.device("cpu"))
        # the staging is a property of the JOB: 400 rows of one image are table-staged as a whole, and so must the 200 of each rank be
        # (a per-call choice gave the shards K0 + the MFMA stem, 2.5e-6 away from the single engine)
        wide = _wide()
        assert shard.job_stem(eng, N_WIDE) == "table" and eng.stem_for_rows(N_WIDE // world) == "conv"
        w_score, w_pred = shard.score_masks_sharded(eng, imgs[2], seg, wide, 5)
        w_heat, w_ok = shard.heatmap_sharded(eng, imgs[2], seg, wide, int(w_pred[0]))
        np.savez(os.path.join(out_dir, "g%d.npz" % rank), score=score, pred=pred, heat=heat.cpu().numpy(), n_ok=n_ok, flat=flat.numpy(),
                 w_score=w_score, w_pred=w_pred, w_heat=w_heat.cpu().numpy(), w_ok=w_ok)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_engine_processes_shard_like_one(tmp_path, mpx_lib):
    from network_interpretation_imagenet_amd import synth
    from network_interpretation_imagenet_amd.engine import MaskedForwardEngine
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    eng = MaskedForwardEngine("resnet18", max_batch=32, device=0).load_state_dict(synth.make_state_dict("resnet18"))
    try:
        imgs, seg, onoff = _inputs()
        _o, want_s, want_p = eng.score_masks(imgs[0], seg, onoff[0], 5)
        label = int(want_p[0])
        _o, _s, p1 = eng.score_masks(imgs[1], seg, onoff[1], label)
        want_heat = eng.heatmap(seg, onoff[1], p1, label)
        want_flat = np.concatenate([eng.score_masks(imgs[i], seg, onoff[i], 5)[1] for i in range(N_IMG)])
        wide = _wide()
        _o, want_ws, want_wp = eng.score_masks(imgs[2], seg, wide, 5)                  # 400 rows in one call: the stem table
        w_label = int(want_wp[0])
        _o, _s, wp1 = eng.score_masks(imgs[2], seg, wide, w_label)
        want_wheat = eng.heatmap(seg, wide, wp1, w_label)
        _o, k0_ws, _p = eng.score_masks(imgs[2], seg, wide, 5, stem="conv")            # the other staging: close, not equal
    finally:
        eng.close()
    for r in range(2):
        got = np.load(tmp_path / ("g%d.npz" % r))
        assert np.array_equal(got["score"], want_s) and np.array_equal(got["pred"], want_p)            # bit-identical on every rank
        assert np.array_equal(got["heat"].astype(np.float64), want_heat) and int(got["n_ok"]) == int((p1 == label).sum())
        assert np.array_equal(got["flat"], want_flat)
        assert np.array_equal(got["w_score"], want_ws) and np.array_equal(got["w_pred"], want_wp)
        assert np.array_equal(got["w_heat"].astype(np.float64), want_wheat) and int(got["w_ok"]) == int((wp1 == w_label).sum())
    # the comparison above has teeth: the two stagings do differ in bits on these rows (and agree within the engine's tolerance)
    assert not np.array_equal(k0_ws, want_ws) and float(np.abs(k0_ws - want_ws).max()) <= 2e-5
