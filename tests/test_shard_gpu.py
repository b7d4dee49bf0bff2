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


def _inputs():
    from network_interpretation_imagenet_amd import synth
    imgs = synth.make_images(N_IMG, seed=77, kind="blobs")
    seg = synth.grid_segments(block=28)
    onoff = [synth.random_onoff(N_MASK, S, seed=90 + i) for i in range(N_IMG)]
    return imgs, seg, onoff


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
            return torch.from_numpy(eng.score_masks(imgs[i], seg, onoff[i][lo:hi], 5)[1])
        flat = shard.score_sharded(fn, N_IMG, N_MASK, torch.device("cpu"))
        np.savez(os.path.join(out_dir, "g%d.npz" % rank), score=score, pred=pred, heat=heat.cpu().numpy(), n_ok=n_ok, flat=flat.numpy())
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
    finally:
        eng.close()
    for r in range(2):
        got = np.load(tmp_path / ("g%d.npz" % r))
        assert np.array_equal(got["score"], want_s) and np.array_equal(got["pred"], want_p)            # bit-identical on every rank
        assert np.array_equal(got["heat"].astype(np.float64), want_heat) and int(got["n_ok"]) == int((p1 == label).sum())
        assert np.array_equal(got["flat"], want_flat)
