#!/usr/bin/env python3
"""masked-forward-passes/sec benchmark (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the workload of BASELINE configs[2]
("ResNet-101, 512 masks/image x 128 images, 1 MI355X"): per image, K0 stages its 512 masked
copies, the network scores them, 512 scores stay on the device.  With N GPUs every rank runs its
own 128 images (weak scaling; N=8 is BASELINE configs[3], 1024 images) and ONE RCCL all-gather of
the per-mask scores closes the step.  Inputs (images, label map, mask-vectors, weights) are
resident in HBM before the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_EVERY = 4               # HIP-event bracketing on every 4th forward batch of the timed region


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--images", type=int, default=128, help="images per GPU per step")
    ap.add_argument("--masks", type=int, default=512, help="masks per image")
    ap.add_argument("--images-per-forward", type=int, default=4,
                    help="images whose masks share one forward batch (batch = this x masks; larger batches fill "
                         "256 CUs with fewer partial rounds of tiles)")
    ap.add_argument("--cpu-masks", type=int, default=16, help="masks of the CPU baseline sample (0 = skip)")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent forward batches in flight (one engine + HIP stream each): lets the HBM-bound "
                         "kernels of one batch overlap the MFMA-bound kernels of another")
    ap.add_argument("--stream-offset", action="store_true", help="with --streams > 1: start the streams out of phase")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the all-gather even with one rank (rehearsal of the N>1 path)")
    return ap.parse_args()


def pmc_traffic(arch, batch):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read
    from inside the process); None unless a profile of this arch and forward batch exists."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        with open(path) as fh:
            j = json.load(fh)
        if j.get("forward_batch") == batch and arch == j.get("arch", "resnet101"):
            return j["traffic_bytes_per_launch"]
    return None


def cpu_baseline(arch, n_masks):
    """Reference-style loop (oracle/scorer.py: batch-1 fp32 forward per mask, mask built per
    superpixel) on this box's host cores, bounded sample of the same workload."""
    from network_interpretation_imagenet_amd import synth
    from oracle import scorer
    sd = synth.make_state_dict(arch)
    img = synth.make_images(1, kind="noise")[0]
    x = scorer.to_tensor_normalize(img)
    seg = synth.grid_segments()
    onoff = synth.random_onoff(n_masks, 196)
    scorer.score_masks_reference_loop(sd, arch, x, seg, onoff[:1], 0)    # warm the thread pool
    t0 = time.perf_counter()
    scorer.score_masks_reference_loop(sd, arch, x, seg, onoff, 0)
    dt = time.perf_counter() - t0
    return {"value": n_masks / dt, "unit": "masked-forward-passes/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "%s, 1 image x %d masks, batch-1 fp32 torch-CPU loop (%.1f s)" % (arch, n_masks, dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # "nccl" is RCCL on ROCm

    import __graft_entry__ as g
    if use_dist:            # one builder per node; the others load the finished library
        if local_rank == 0:
            g.build()
        dist.barrier()
    g.build()
    from network_interpretation_imagenet_amd import shard, synth
    from network_interpretation_imagenet_amd.engine import MaskedForwardEngine

    n_img, n_mask, ipf = args.images, args.masks, args.images_per_forward
    if n_img % ipf:
        raise SystemExit("--images must be a multiple of --images-per-forward")
    batch = ipf * n_mask
    sd = synth.make_state_dict(args.arch)
    engines = [MaskedForwardEngine(args.arch, max_batch=batch, device=local_rank).load_state_dict(sd)
               for _ in range(max(1, args.streams))]
    streams = [torch.cuda.Stream(device=dev) for _ in engines]
    eng = engines[0]
    # synthetic inputs, resident in HBM: this rank's images, the shared 14x14-block label map (S=196),
    # per-image Bernoulli(0.4) mask-vectors, labels = unmasked argmax (the reference's correctness gate)
    imgs = torch.from_numpy(synth.make_images(n_img, seed=1234 + rank, kind="noise")).to(dev)
    seg = torch.from_numpy(synth.grid_segments()).to(dev)
    onoff = torch.from_numpy(synth.random_onoff(n_img * n_mask, 196, seed=4321 + rank)).view(n_img, n_mask, 196).to(dev)
    ones = torch.ones(1, 196, dtype=torch.uint8, device=dev)
    labels = []
    for i0 in range(0, n_img, batch):          # unmasked forwards, one slot per image
        nb = min(batch, n_img - i0)
        for j in range(nb):
            eng.stage_masks(imgs[i0 + j], seg, ones, j)
        _s, p = eng.forward(nb, torch.zeros(nb, dtype=torch.int32, device=dev))
        labels.append(p)
    labels = torch.cat(labels)
    label_rows = labels.view(n_img, 1).expand(n_img, n_mask).contiguous()
    scores = torch.empty(n_img, n_mask, dtype=torch.float32, device=dev)
    total = world * n_img * n_mask

    def step(profile):
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        if len(engines) > 1 and args.stream_offset:
            # put stream k a fraction k/n of a forward behind stream 0, so that one stream's HBM-bound layers
            # meet another's MFMA-bound layers instead of running the same layer side by side
            for k in range(1, len(engines)):
                with torch.cuda.stream(streams[k]):
                    nb = batch * k // len(engines)
                    engines[k].forward(nb, label_rows[:ipf].view(-1)[:nb].contiguous())
        for f, i0 in enumerate(range(0, n_img, ipf)):
            e, st = engines[f % len(engines)], streams[f % len(engines)]
            prof = profile and (f % PROFILE_EVERY == 0) and e is eng
            with torch.cuda.stream(st):
                if prof:
                    e.profile(True)
                for j in range(ipf):
                    e.stage_masks(imgs[i0 + j], seg, onoff[i0 + j], j * n_mask)
                s, _p = e.forward(batch, label_rows[i0:i0 + ipf].view(-1))
                scores[i0:i0 + ipf] = s.view(ipf, n_mask)
                if prof:
                    e.profile(False)
        for st in streams:
            torch.cuda.current_stream(dev).wait_stream(st)
        if use_dist:
            return shard.all_gather_blocks(scores.view(-1), total)
        return scores.view(-1)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    prof = {"ms": {}, "launches": {}}
    for _ in range(args.steps):
        out = step(True)
        part = eng.collect_profile()        # waits for this step's last recorded event (the pool is bounded)
        for key in ("ms", "launches"):
            for k, v in part[key].items():
                prof[key][k] = prof[key].get(k, 0) + v
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.numel() == total and bool(torch.isfinite(out).all())

    if rank == 0:
        value = total * args.steps / dt
        conv_ms, conv_n = prof["ms"]["conv"], prof["launches"]["conv"]
        n_conv_layers = len(eng.layers)
        batches_profiled = conv_n / n_conv_layers if n_conv_layers else 0
        flops_per_batch = eng.flops_per_forward * batch
        roofline = None
        if conv_n:
            # dominant kernel = conv_f16x3_kernel (all conv/fc launches).  achieved = algorithmic FLOPs
            # of the launches / their summed HIP-event durations; MFMA-issued FLOPs are 3x algorithmic.
            achieved = flops_per_batch * batches_profiled / (conv_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_F16_MFMA_TFLOPS, "traffic": pmc_traffic(args.arch, batch),
                        "kernel": "conv_f16x3_kernel + conv3x3p_f16x3_kernel (all conv launches)", "launches": conv_n,
                        "avg_launch_us": conv_ms * 1e3 / conv_n,
                        "mfma_issued_frac": 3 * achieved / PEAK_F16_MFMA_TFLOPS,
                        "other_kernels_ms_per_batch": {k: v / max(batches_profiled, 1) for k, v in prof["ms"].items() if k != "conv"}}
        line = {
            "metric": "masked-forward-passes/sec (224x224, %s)" % args.arch,
            "value": value, "unit": "masked-forward-passes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16x3 (split-fp16 MFMA, fp32 accumulate)",
            "data": "synthetic (random-init torchvision-shaped weights, uniform-random u8 images, 14x14-block label map)",
            "config": {"workload": "%s, %d masks/image x %d images per GPU (BASELINE configs[2]; x%d GPUs)" % (args.arch, n_mask, n_img, world),
                       "images_per_gpu": n_img, "masks_per_image": n_mask, "forward_batch": batch, "streams": len(engines),
                       "parallelism": "mask-batch shard x%d + one all_gather of scores" % world},
            "tflops_algorithmic": value * eng.flops_per_forward / 1e12,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline(args.arch, args.cpu_masks) if (world == 1 and args.cpu_masks > 0) else None,
        }
        print(json.dumps(line))
    for e in engines:
        e.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
