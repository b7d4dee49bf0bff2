"""Mask-batch sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference has no distributed code (its --world-size/--dist-url flags are parsed and never
read: generate_gp_training_data_imagenet.py:72-77,572).  Every (image, mask) pair is independent
(:221-266), so the flattened work index w = img*M + m is cut into P contiguous blocks with no
data-path collective; the only exchange is ONE all-gather of the per-mask scores (RCCL over xGMI
when the backend is "nccl", gloo on CPU in the tests).  When the consumer wants the summed heat map of
an image instead of the scores (gp_superpixel_data_imagenet.py:322-323), every rank accumulates its
masks with K5 and ONE all-reduce of f32[224*224] (196 KiB) closes the image.
"""
import torch
import torch.distributed as dist


def block(total, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return total * rank // world, total * (rank + 1) // world


def image_ranges(lo, hi, masks_per_image):
    """Split the flat range [lo,hi) into per-image pieces: [(img, m_lo, m_hi)], m_* within the image."""
    out = []
    w = lo
    while w < hi:
        img = w // masks_per_image
        m_lo = w - img * masks_per_image
        m_hi = min(masks_per_image, m_lo + (hi - w))
        out.append((img, m_lo, m_hi))
        w += m_hi - m_lo
    return out


def all_gather_blocks(local, total, group=None):
    """local: 1-D tensor holding this rank's block (block(total, rank, world)) -> 1-D tensor of
    length `total` on every rank.  One collective; uneven blocks are padded to the largest."""
    if not dist.is_available() or not dist.is_initialized():
        if local.numel() != total:
            raise ValueError("single process must hold the whole range")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = block(total, rank, world)
    if local.numel() != hi - lo:
        raise ValueError("rank %d holds %d values, its block is %d" % (rank, local.numel(), hi - lo))
    width = -(-total // world)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[:hi - lo] = local
    gathered = torch.empty(world * width, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = []
    for r in range(world):
        rlo, rhi = block(total, r, world)
        parts.append(gathered[r * width:r * width + (rhi - rlo)])
    return torch.cat(parts)


def job_stem(engine, rows_per_image):
    """The staging of a JOB (engine.stem_for_rows): decided from the rows an image has over ALL ranks, never from the rows a rank, a
    block or a call happens to hold -- the stem table and K0 + the MFMA stem round differently (<= 2.5e-6 on a score), so a choice made
    per shard would give a 512-mask image the table on one GPU and K0 on eight.  None for engine stand-ins without the method."""
    fn = getattr(engine, "stem_for_rows", None)
    return fn(rows_per_image) if fn is not None else None


def score_sharded(score_image_fn, num_images, masks_per_image, device, group=None):
    """score_image_fn(img, m_lo, m_hi) -> 1-D f32 tensor of m_hi-m_lo scores on `device`.
    Runs this rank's block and returns all num_images*masks_per_image scores on every rank.  A block can cut an image's rows in
    two (its tail on rank r, its head on rank r + 1): for results bit-identical to the 1-GPU run `score_image_fn` must stage every piece
    the way the WHOLE image would be staged -- engine.score_masks(..., stem=job_stem(engine, masks_per_image)); with that the blocks
    run the same kernels on disjoint rows."""
    total = num_images * masks_per_image
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = block(total, rank, world)
    pieces = [score_image_fn(img, m_lo, m_hi) for img, m_lo, m_hi in image_ranges(lo, hi, masks_per_image)]
    local = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.float32, device=device)
    return all_gather_blocks(local, total, group)


def score_masks_sharded(engine, image, segments, onoff, label, group=None):
    """Single-image case (BASELINE config 5: one image, every candidate window of a BO round): shard the MASK
    axis, every rank holds the image.  Each rank scores its contiguous block of mask-vectors with its own engine
    and ONE all-gather returns the full (score f32[M], pred i32[M]) on every rank -- bit-identical to one engine
    scoring all M: the staging (stem table or K0 + the MFMA stem) is chosen from the GLOBAL row count M (job_stem) and handed to every
    rank's call, so the blocks run the kernels the unsplit call runs, on disjoint rows.  The scores travel as their bit patterns next to the predictions in one
    i32[2 * width] buffer per rank ("one RCCL all_gather"); nothing is converted, so NaN payloads and signed
    zeros arrive as they left."""
    import numpy as np
    m = int(onoff.shape[0])
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = block(m, rank, world)
    stem = job_stem(engine, m)
    kw = {} if stem is None else {"stem": stem}
    _o, score, pred = engine.score_masks(image, segments, onoff[lo:hi], label, **kw)
    score = np.ascontiguousarray(score, dtype=np.float32)
    pred = np.ascontiguousarray(pred, dtype=np.int32)
    if world == 1:
        return score, pred
    device = getattr(engine, "device", torch.device("cpu"))
    if dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    width = -(-m // world)
    mine = np.zeros((2, width), dtype=np.int32)
    mine[0, :hi - lo] = score.view(np.int32)
    mine[1, :hi - lo] = pred
    gathered = torch.empty(world * 2 * width, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(gathered, torch.from_numpy(mine.reshape(-1)).to(device), group=group)
    g = gathered.cpu().numpy().reshape(world, 2, width)
    s_all = np.empty(m, dtype=np.float32)
    p_all = np.empty(m, dtype=np.int32)
    for r in range(world):
        rlo, rhi = block(m, r, world)
        s_all[rlo:rhi] = g[r, 0, :rhi - rlo].view(np.float32)
        p_all[rlo:rhi] = g[r, 1, :rhi - rlo]
    return s_all, p_all


def all_reduce_heatmap(heat, group=None):
    """Sum the per-rank partial heat maps in place: ONE all_reduce(SUM) of f32[224,224] per image (RCCL when the
    tensor is on the GPU, gloo on the CPU).  The partial maps hold integer counts below 2^24, so the f32 sum is exact
    and independent of the reduction order.  No-op in a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(heat, op=dist.ReduceOp.SUM, group=group)
    return heat


HEAT_ELEMS = 224 * 224


def heatmap_sharded(engine, image, segments, onoff, label, group=None):
    """Heat map of ONE image with its M mask-vectors split over the ranks (contiguous blocks of the mask axis):
    each rank scores its block and accumulates sum_m [pred[m] == label] * onoff[m][seg[p]] straight into a device buffer
    f32[224*224 + 1] (K5 through engine.heatmap_device: no host round trip) whose LAST element carries the rank's number of
    correctly predicted masks, and ONE all_reduce(SUM) of that buffer closes the image ("one all_reduce of
    f32[224*224] per image"; the count rides along).  Equal to the single-engine map exactly: every rank stages its block the way the
    whole image's M rows would be staged (job_stem: the GLOBAL row count decides between the stem table and K0), so each mask's argmax is the
    single engine's, and the sums are integer counts below 2^24 -- the f32 result does not depend on the reduction order.  `segments` must be
    a rank map (engine.rank_segments).
    returns (heat f32[224,224] tensor on the engine's device, n_correct int over all ranks)."""
    import numpy as np
    m = int(onoff.shape[0])
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = block(m, rank, world)
    device = getattr(engine, "device", torch.device("cpu"))
    on_gloo = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "gloo"
    buf = torch.zeros(HEAT_ELEMS + 1, dtype=torch.float32, device=device)
    if hi > lo:
        if hasattr(engine, "heatmap_device"):
            engine.heatmap_device(image, segments, onoff[lo:hi], label, buf, stem=job_stem(engine, m))
        else:                           # engine stand-ins of the CPU tests: host K5
            stem = job_stem(engine, m)
            _o, _score, pred = engine.score_masks(image, segments, onoff[lo:hi], label, **({} if stem is None else {"stem": stem}))
            part = engine.heatmap(segments, onoff[lo:hi], pred, label)            # f64[224,224], exact
            buf[:HEAT_ELEMS] = torch.from_numpy(np.ascontiguousarray(part, dtype=np.float32).ravel())
            buf[HEAT_ELEMS] = float((np.asarray(pred) == label).sum())
    if on_gloo and buf.device.type != "cpu":
        buf = buf.cpu()
    all_reduce_heatmap(buf, group)
    return buf[:HEAT_ELEMS].view(224, 224), int(buf[HEAT_ELEMS].item())
