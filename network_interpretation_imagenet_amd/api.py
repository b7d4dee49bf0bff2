"""Reference-named entry points (drop-in for the GP / Bayesian-optimisation callers).

Same names, argument orders and return types as the reference's callables
(bayesian_active_learning_imagenet.py:116-298, generate_gp_training_data_imagenet.py:152-273,
gp_superpixel_data_imagenet.py:186-350); `model` is a MaskedForwardEngine instead of a
torchvision module, `val_loader` any iterable of (input f32[1,3,224,224], target[, gt_bbox]) and
`criterion` is accepted and ignored exactly as the reference ignores its result
(generate_gp_training_data_imagenet.py:194).  What changes is the cost: the reference pays one
dataset scan + one segmentation + two batch-1 forwards per score; here every candidate window of
an image is scored in ONE batched pass and later calls are table look-ups.

Quirks deliberately left behind: no dataset re-scan per call, no
`rm -rf ./masks` unless a mask_dir is passed, no float-valued PNG names.
"""
import os
import random

import numpy as np
import torch

from . import masks
from .engine import BasePredictionWrong, rank_segments, IMG

__all__ = ["SaliencySession", "sample_loss", "validate_nueral_network", "superpixel_mask",
           "validate", "validate_gp_superpixel", "validate_summed", "validate_summed_many", "validate_many", "fill_tables", "eval_superpixel",
           "score_masks", "default_segmenter",
           "jet_heatmap_u8",
           "img_show_u8", "load_images_from_folder", "prepare_training_data", "get_pixel_sorted_mask_label",
           "summed_heatmap_from_folder",
           "BasePredictionWrong", "configure"]

_CONFIG = {"eval_img_index": 1, "num_mask_samples": 100, "segmenter": None, "mask_dir": None, "seed": None}
_SESSIONS = []          # [(model, val_loader, eval_img_index, session)], most recent last; identity-checked, bounded
_MAX_SESSIONS = 8
_LAST = {"session": None}


def configure(**kw):
    """Stand-in for the reference's argparse globals (`args.eval_img_index`,
    `args.num_mask_samples`; generate_gp_training_data_imagenet.py:78-81) plus the segmenter
    (felzenszwalb, a CPU third-party step that stays outside the engine) and the optional PNG
    output directory."""
    for k, v in kw.items():
        if k not in _CONFIG:
            raise KeyError("unknown option %r (have %s)" % (k, sorted(_CONFIG)))
        _CONFIG[k] = v
    del _SESSIONS[:]


def default_segmenter(img_u8_hwc):
    """felzenszwalb(img_as_float(img_show), scale=100, sigma=0.5, min_size=50)
    (generate_gp_training_data_imagenet.py:183) by the native CPU front-end (segment.py / libmpxseg.so,
    bit-exact against scikit-image 0.18.3 on the committed vectors); configure(segmenter=callable)
    swaps in anything else that maps u8[224,224,3] -> int[224,224]."""
    from . import segment
    return segment.felzenszwalb(img_u8_hwc, scale=100, sigma=0.5, min_size=50)


def img_show_u8(input_chw):
    """Min-max rescale to u8 HWC, the picture the reference segments
    (generate_gp_training_data_imagenet.py:171-178; truncating astype)."""
    img = np.array(input_chw, dtype=np.float32, copy=True).transpose(1, 2, 0)
    img -= img.min()
    img /= img.max()
    img *= 255
    return img.astype(np.uint8)


def _as_int(target):
    """target arrives as an int, tensor[1] (ImageFolder loader) or tensor[1,1] (localization loader)."""
    return int(np.asarray(target).reshape(-1)[0])


class SaliencySession:
    """One (image, label, segmentation) with every window score cached."""

    def __init__(self, engine, input_chw, target, segments=None, segmenter=None, check_base=True):
        if not (hasattr(engine, "score_masks") and hasattr(engine, "predict")):
            raise TypeError("model must be a MaskedForwardEngine (score_masks/predict), got %r" % type(engine))
        x = torch.as_tensor(input_chw)
        if x.dim() == 4:
            x = x[0]
        if x.dtype != torch.float32 or tuple(x.shape) != (3, IMG, IMG):
            raise ValueError("input must be float32[1,3,224,224] (normalised), got %s%s" % (x.dtype, tuple(x.shape)))
        self.engine = engine
        self.input = x.contiguous()
        self.label = _as_int(target)
        if segments is None:
            seg_fn = segmenter or _CONFIG["segmenter"] or default_segmenter
            segments = seg_fn(img_show_u8(self.input.numpy()))
        self.seg_rank, self.num_segments = rank_segments(segments)
        self.window = masks.window_size(self.num_segments)
        self.upper_bound = masks.bo_upper_bound(self.num_segments)
        self._table = None
        self.base_pred = None
        # ONE staging for everything this session ever scores (the two stagings round differently, <= 2.5e-6 on a score): the kind an image
        # with S + 2 rows gets -- the unmasked row + every window start, what fill_tables packs -- whether the table is computed here with or
        # without the unmasked row, by fill_tables next to other images, or a single window is scored outside the table
        self.stem = engine.stem_for_rows(self.num_segments + 2) if hasattr(engine, "stem_for_rows") else None
        if check_base:
            # the unmasked row rides in the same forward batch as the S + 1 window starts every caller asks for next (as
            # fill_tables does for several images): one pass instead of a batch-1 forward -- ~4 ms of kernel latencies on
            # ResNet-101 -- followed by the table's.  An all-ones mask row IS the unmasked image (normalise, then mask), and a
            # score does not depend on the slot it is computed in.
            onoff = np.concatenate([np.ones((1, self.num_segments), dtype=np.uint8),
                                    masks.windows_onoff(self.num_segments, range(0, self.num_segments + 1))])
            _o, score, pred = self._score(onoff)
            self.base_pred = int(pred[0])
            if self.base_pred != self.label:
                raise BasePredictionWrong("unmasked prediction %d != label %d" % (self.base_pred, self.label))
            self._table = (score[1:], pred[1:])

    def _score(self, onoff):
        if self.stem is None:           # an engine without the two stagings (test doubles)
            return self.engine.score_masks(self.input, self.seg_rank, onoff, self.label)
        return self.engine.score_masks(self.input, self.seg_rank, onoff, self.label, stem=self.stem)

    def score_windows(self, first_indices):
        onoff = masks.windows_onoff(self.num_segments, first_indices)
        _o, score, pred = self._score(onoff)
        return onoff, score, pred

    def table(self):
        """Scores of every window start 0..S (one batched pass; the BO domain is [0, int(0.6*S)])."""
        if self._table is None:
            idx = list(range(0, self.num_segments + 1))
            _o, score, pred = self.score_windows(idx)
            self._table = (score, pred)
        return self._table

    def score(self, first_index):
        f = int(first_index)
        if 0 <= f <= self.num_segments:
            score, pred = self.table()
            return score[f], int(pred[f])
        _o, score, pred = self.score_windows([f])
        return score[0], int(pred[0])

    def summed_labels(self, firsts, correct):
        """f64[224,224] = sum over the correct masks of their pixel mask (gp_superpixel_data_imagenet.py:322-323):
        the engine's K5 heat-map kernel (mpx_heatmap_accumulate) on the mask-vectors of `firsts`."""
        onoff = masks.windows_onoff(self.num_segments, firsts)
        pred = np.where(np.asarray(correct, dtype=bool), self.label, -1).astype(np.int32)
        return self.engine.heatmap(self.seg_rank, onoff, pred, self.label)

    def mask_u8(self, first_index):
        """u8[224,224] in {0,255} (`mask*255`, bayesian_active_learning_imagenet.py:276)."""
        return masks.expand_pixel_mask(self.seg_rank, masks.window_onoff(self.num_segments, first_index)) * np.uint8(255)


def _pick(val_loader, eval_img_index):
    count = 0
    for item in val_loader:
        count += 1
        if count > eval_img_index:
            break
        if count == eval_img_index:
            return item
    return None


def _session(val_loader, model, eval_img_index):
    """The cached session of (model, val_loader, eval_img_index).  The cache holds strong references to the model and
    the loader and compares them with `is` (an id() key could be reused by a NEW object after garbage collection and
    hand back another dataset's table), and keeps at most _MAX_SESSIONS entries."""
    idx = int(eval_img_index)
    for k, (m, l, i, s) in enumerate(_SESSIONS):
        if m is model and l is val_loader and i == idx:
            _SESSIONS.append(_SESSIONS.pop(k))
            _LAST["session"] = s
            return s
    item = _pick(val_loader, idx)
    if item is None:
        return None
    s = SaliencySession(model, item[0], item[1])
    _SESSIONS.append((model, val_loader, idx, s))
    del _SESSIONS[:-_MAX_SESSIONS]
    _LAST["session"] = s
    return s


def _write_png(path, mask_u8):
    from PIL import Image
    Image.fromarray(mask_u8, mode="L").save(path)


def validate_nueral_network(val_loader, model, criterion, bo_iter, firstIndex):
    """-> np.float32 softmax probability of the true class for the window starting at firstIndex
    (bayesian_active_learning_imagenet.py:116-218).  Raises BasePredictionWrong where the
    reference raises a bare Exception (:219-221).  With configure(mask_dir=...) also writes
    mask_{bo_iter}_{0|1}.png (:210,215)."""
    s = _session(val_loader, model, _CONFIG["eval_img_index"])
    if s is None:
        return None
    score, pred = s.score(int(firstIndex))
    if _CONFIG["mask_dir"]:
        os.makedirs(_CONFIG["mask_dir"], exist_ok=True)
        _write_png(os.path.join(_CONFIG["mask_dir"], "mask_{}_{}.png".format(bo_iter, 1 if pred == s.label else 0)),
                   s.mask_u8(int(firstIndex)))
    return np.float32(score)


def sample_loss(params, val_loader, model, criterion):
    """The BO objective, bayesian_active_learning_imagenet.py:278-298:
    firstIndex = int(params[0]); returns the class probability score."""
    firstIndex = int(params[0])
    bo_iter = params[0]
    return validate_nueral_network(val_loader, model, criterion, bo_iter, firstIndex)


def superpixel_mask(firstIndex, session=None):
    """u8[224,224] in {0,255} for the window at firstIndex
    (bayesian_active_learning_imagenet.py:224-276; called by BayesianOptimization.py:204-205).
    Uses the most recent session instead of re-reading the dataset."""
    s = session or _LAST["session"]
    if s is None:
        raise RuntimeError("superpixel_mask needs a session: call sample_loss/validate_nueral_network first")
    return s.mask_u8(int(firstIndex))


def _generator_pass(val_loader, model, eval_img_index, num_mask_samples, rng):
    s = _session(val_loader, model, eval_img_index)
    if s is None:
        return None
    rng = rng or (random.Random(_CONFIG["seed"]) if _CONFIG["seed"] is not None else random)
    firsts = masks.draw_first_indices(s.num_segments, num_mask_samples, rng)
    table_score, table_pred = s.table()
    pred = np.array([table_pred[f] for f in firsts], dtype=np.int64)
    correct = pred == s.label
    return s, firsts, correct


def validate(val_loader, model, criterion, eval_img_index, num_mask_samples=None, rng=None):
    """-> correct_pred_count (int), generate_gp_training_data_imagenet.py:152-273: draws
    num_mask_samples random windows, labels each 1 if the masked prediction is still the target.
    With configure(mask_dir=...) writes mask_{i}_{label}.png (:260,265).  Returns 0 after printing
    "wrong prediction" when the unmasked prediction is wrong (:269-273)."""
    n = _CONFIG["num_mask_samples"] if num_mask_samples is None else num_mask_samples
    try:
        out = _generator_pass(val_loader, model, eval_img_index, n, rng)
    except BasePredictionWrong:
        print("wrong prediction")
        return 0
    if out is None:
        return 0
    s, firsts, correct = out
    if _CONFIG["mask_dir"]:
        os.makedirs(_CONFIG["mask_dir"], exist_ok=True)
        for i, (f, ok) in enumerate(zip(firsts, correct)):
            _write_png(os.path.join(_CONFIG["mask_dir"], "mask_{}_{}.png".format(i, int(ok))), s.mask_u8(f))
    return int(correct.sum())


_JET_BGR = None


def jet_heatmap_u8(summed):
    """summed f64[H,W] -> u8[H,W,3] (BGR), the picture gp_superpixel_data_imagenet.py:337-343 builds: min-max rescale
    to u8 (`-= min; /= max; *= 255; astype(np.uint8)`, truncating) then cv2.applyColorMap(..., cv2.COLORMAP_JET).
    cv2 is not a dependency: the colour map is a 256-entry LUT of the classic piecewise-linear jet,
    channel(x) = clamp(1.5 - |4x - c|, 0, 1) with c = 3, 2, 1 for R, G, B and x = i/255, rounded to u8.  (PARITY
    UNPINNED for the exact LUT bytes: OpenCV is absent from this image; the leading table entries 1.5/255, 5.5/255
    agree with OpenCV's published table.)  A constant map gives 0/0 upstream; here it maps to LUT[0]."""
    global _JET_BGR
    if _JET_BGR is None:
        x = np.arange(256, dtype=np.float64) / 255.0
        ch = lambda c: np.rint(np.clip(1.5 - np.abs(4.0 * x - c), 0.0, 1.0) * 255.0).astype(np.uint8)
        _JET_BGR = np.stack([ch(1.0), ch(2.0), ch(3.0)], axis=1)         # B, G, R
    show = np.array(summed, dtype=np.float64, copy=True)
    show -= show.min()
    mx = show.max()
    if mx > 0:
        show /= mx
    show *= 255
    return _JET_BGR[show.astype(np.uint8)]


def validate_gp_superpixel(val_loader, model, criterion, eval_img_index, num_mask_samples=None, rng=None):
    """-> (summed_superpixel_labels f64[224,224], summed_labels_heatmap u8[224,224,3]), the return of the
    gp_superpixel flavour of validate() (gp_superpixel_data_imagenet.py:186-350, unpacked by its caller at :617):
    the sum over the correctly predicted masks of their pixel mask (:322-323; equals gp_regression.py:82-94's
    y[p] = sum_i label_i*mask_i[p]) and its JET picture (:337-343).  When the unmasked prediction is wrong the
    reference prints "wrong prediction" and falls off the end of the function (returns None, :351-352): so does this."""
    n = _CONFIG["num_mask_samples"] if num_mask_samples is None else num_mask_samples
    try:
        out = _generator_pass(val_loader, model, eval_img_index, n, rng)
    except BasePredictionWrong:
        print("wrong prediction")
        return None
    if out is None:
        return None
    s, firsts, correct = out
    summed = s.summed_labels(firsts, correct)
    return summed, jet_heatmap_u8(summed)


validate_summed = validate_gp_superpixel      # earlier name of the same entry point


def fill_tables(engine, sessions):
    """Score the unmasked image and EVERY window start 0..S of several sessions in packed forward batches
    (MaskedForwardEngine.score_images: the rows of consecutive images share batches of up to max_batch slots) -- one image's
    table is only S+2 = 50 .. 350 rows, a fraction of the batch the engine is fast at.  Sets each session's base_pred and
    table; -> [bool]: the unmasked prediction equals the label (the reference's gate, generate_gp_training_data_imagenet.py:215).
    Every image is staged by ITS OWN S + 2 rows (SaliencySession.stem = engine.stem_for_rows(S + 2), the same rule score_images applies per
    image), so its scores are bit-identical to SaliencySession.table() of that image alone, whatever it is packed with: the binary labels
    (generate_gp_training_data_imagenet.py:248,257) and the heat-map sum (gp_superpixel_data_imagenet.py:322-323) do not depend on grouping
    or lookahead."""
    todo = [s for s in sessions if s._table is None or s.base_pred is None]
    if todo:
        rows = []
        for s in todo:
            onoff = np.empty((s.num_segments + 2, s.num_segments), dtype=np.uint8)
            onoff[0] = 1                                                    # row 0: the unmasked image
            onoff[1:] = masks.windows_onoff(s.num_segments, range(0, s.num_segments + 1))
            rows.append(onoff)
        res = engine.score_images([s.input for s in todo], [s.seg_rank for s in todo], rows, [s.label for s in todo])
        for s, (score, pred) in zip(todo, res):
            s.base_pred = int(pred[0])
            s._table = (score[1:], pred[1:])
    return [s.base_pred == s.label for s in sessions]


def _many(val_loader, model, eval_img_indices, num_mask_samples, rng, workers, lookahead, emit):
    """Common driver of validate_many / validate_summed_many: one pass over the loader; the CPU segmentation of the next
    images (segment.SegmenterPool) runs while the GPU scores the current group; the tables of consecutive images are packed
    into full forward batches (fill_tables), one group per kind of staging.  emit(index, session, firsts, correct) -> the per-image
    result, called in loader order."""
    from collections import deque
    from . import segment
    n = _CONFIG["num_mask_samples"] if num_mask_samples is None else num_mask_samples
    want = set(int(i) for i in eval_img_indices)
    if not want:
        return {}
    custom = _CONFIG["segmenter"]
    lookahead = int(lookahead or 2 * workers)
    rng = rng or (random.Random(_CONFIG["seed"]) if _CONFIG["seed"] is not None else random)
    out = {}
    pending = deque()       # (index, input, target, future of the label map), loader order
    # Sessions whose label map is in wait for a full batch PER KIND of staging (SaliencySession.stem: an image of 256+ rows takes the stem
    # table, a smaller one K0 + the MFMA stem, and one forward takes one kind): a loader whose pictures straddle S = 254 then still runs
    # full batches of each kind instead of two partial ones per group.  Results are EMITTED in loader order (the window draws of the
    # reference happen image after image, generate_gp_training_data_imagenet.py:221-230: the draws of image i must not depend on how the
    # images were grouped), so a scored session waits in `scored` until every earlier one has been scored too; MAX_WAITING bounds that
    # queue when one kind is rare.
    groups, rows = {}, {}   # kind -> sessions waiting for a full batch / their rows
    scored = deque()        # (index, session), loader order, waiting for emission
    MAX_WAITING = max(64, 4 * lookahead)

    def drain():
        while scored and scored[0][1].base_pred is not None:
            idx, s = scored.popleft()
            if s.base_pred != s.label:          # the reference's "wrong prediction" branch: no draws
                out[idx] = None
                continue
            firsts = masks.draw_first_indices(s.num_segments, n, rng)
            _score, table_pred = s.table()
            correct = np.array([table_pred[f] for f in firsts], dtype=np.int64) == s.label
            out[idx] = emit(idx, s, firsts, correct)

    def flush(kind=None):
        for k in ([kind] if kind is not None else list(groups)):
            if groups.get(k):
                fill_tables(model, groups[k])
                groups[k], rows[k] = [], 0
        drain()

    def segmented(idx, x, target, fut):
        s = SaliencySession(model, x, target, segments=fut.result(), check_base=False)
        groups.setdefault(s.stem, []).append(s)
        rows[s.stem] = rows.get(s.stem, 0) + s.num_segments + 2
        scored.append((idx, s))
        if rows[s.stem] >= model.max_batch:
            flush(s.stem)
        if len(scored) > MAX_WAITING:
            flush()

    with segment.SegmenterPool(workers=workers) as pool:
        count = 0
        for item in val_loader:
            count += 1
            if count in want:
                x = torch.as_tensor(item[0])
                x = (x[0] if x.dim() == 4 else x).contiguous()
                if custom is not None:
                    fut = pool._ex.submit(lambda a=x.numpy().copy(): custom(img_show_u8(a)))
                else:
                    fut = pool.submit(x.numpy())
                pending.append((count, x, item[1], fut))
                if len(pending) > lookahead:
                    segmented(*pending.popleft())
            if count >= max(want):
                break
        while pending:
            segmented(*pending.popleft())
        flush()
    return out


def validate_summed_many(val_loader, model, criterion, eval_img_indices, num_mask_samples=None, rng=None,
                         workers=4, lookahead=None):
    """The heat map of validate_gp_superpixel for several images of one pass over the loader: {index: f64[224,224] or None}
    (None where the unmasked prediction is wrong, the reference's "wrong prediction" branch).  The reference segments and
    scores strictly one image and one mask after the other (gp_superpixel_data_imagenet.py:206-232,276-334); here the
    segmentation of the next images overlaps the GPU and the window tables of consecutive images share full forward batches."""
    return _many(val_loader, model, eval_img_indices, num_mask_samples, rng, workers, lookahead,
                 lambda idx, s, firsts, correct: s.summed_labels(firsts, correct))


def validate_many(val_loader, model, criterion, eval_img_indices, num_mask_samples=None, rng=None, workers=4, lookahead=None):
    """validate() for several images of one pass over the loader: {index: correct_pred_count, or None where the unmasked
    prediction is wrong} (generate_gp_training_data_imagenet.py:152-273 run once per image, its random window draws included).
    With configure(mask_dir=...) the PNGs of image `index` go to <mask_dir>/img_<index>/mask_{i}_{label}.png (:260,265)."""
    def emit(idx, s, firsts, correct):
        if _CONFIG["mask_dir"]:
            d = os.path.join(_CONFIG["mask_dir"], "img_%d" % idx)
            os.makedirs(d, exist_ok=True)
            for i, (f, ok) in enumerate(zip(firsts, correct)):
                _write_png(os.path.join(d, "mask_{}_{}.png".format(i, int(ok))), s.mask_u8(f))
        return int(correct.sum())
    return _many(val_loader, model, eval_img_indices, num_mask_samples, rng, workers, lookahead, emit)


# the two small-network scripts' constants: (superpixels removed per mask, felzenszwalb min_size, image index, burn the unused randint)
_SMALLNET_DEFAULTS = {"cifar": (5, 10, 5, False),      # generate_gp_training_data_cifar.py:308,284,281
                      "mnist": (1, 5, 2, True)}        # generate_gp_training_data_mnist.py:206-215,181,177


def eval_superpixel(test_loader, model, eval_img_index=None, num_removed=None, num_mask_samples=1000, min_size=None,
                    rng=None, segments=None):
    """-> correct_pred_count (int): eval_superpixel() of the reference's CIFAR / MNIST scripts
    (generate_gp_training_data_cifar.py:236-342, generate_gp_training_data_mnist.py:153-269) for the image at
    `eval_img_index` of the loader (1-based; CIFAR script: 5, MNIST script: 2): min-max the picture to u8, felzenszwalb(scale=100,
    sigma=0.5, min_size = 10 / 5), then num_mask_samples (1000 upstream) masks, each switching OFF num_removed (5 / 1) superpixels
    drawn with random.sample(range(uniq[0], uniq[-1]), k); a mask counts as correct when the masked prediction is still the target.
    Unlike the ImageNet generators these scripts do NOT gate on the unmasked prediction (it is computed and unused, :289-290).
    `model` is a MaskedForwardEngine("cifar_resnet56" / "mnist_net", ...) -- all masks of the image are one batched pass
    (score_masks_removed: the scorers' double min-max convention on the device); the defaults follow the script the engine's
    architecture belongs to.  With configure(mask_dir=...) writes mask_{i}_{0|1}.png = the {0,255} mask exactly as the scripts do
    (:329-335).  `segments` overrides the segmentation (any integer label map)."""
    fam = "mnist" if getattr(model, "arch", "").startswith("mnist") else "cifar"
    d_removed, d_min, d_index, burn = _SMALLNET_DEFAULTS[fam]
    k = d_removed if num_removed is None else int(num_removed)
    idx = d_index if eval_img_index is None else int(eval_img_index)
    item = _pick(test_loader, idx)
    if item is None:
        return 0
    x = torch.as_tensor(item[0])
    x = (x[0] if x.dim() == 4 else x).contiguous().to(torch.float32)
    label = _as_int(item[1])
    if segments is None:
        from . import segment
        segments = segment.felzenszwalb(img_show_u8(x.numpy()), scale=100, sigma=0.5, min_size=d_min if min_size is None else int(min_size))
    segments = np.asarray(segments)
    uniq = np.unique(segments)
    rng = rng or (random.Random(_CONFIG["seed"]) if _CONFIG["seed"] is not None else random)
    sets = masks.draw_removed_sets(uniq, k, num_mask_samples, rng, burn_window_draw=burn)
    _r, _score, pred = model.score_masks_removed(x.numpy(), segments, masks.removed_onoff(uniq, sets), label)
    correct = np.asarray(pred) == label
    if _CONFIG["mask_dir"]:
        os.makedirs(_CONFIG["mask_dir"], exist_ok=True)
        for i, (vals, ok) in enumerate(zip(sets, correct)):
            _write_png(os.path.join(_CONFIG["mask_dir"], "mask_{}_{}.png".format(i, int(ok))), masks.removed_mask_u8(segments, vals))
    return int(correct.sum())


def load_images_from_folder(folder):
    """-> (img_filenames, labels): every file of `folder`, label = the text between the second '_' and the
    first '.' of `mask_{i}_{label}.png` (gp_regression.py:63-72; generate_gp_training_data_imagenet.py:474-483).
    Listing order is os.listdir's, as upstream."""
    img_filenames, labels = [], []
    for filename in os.listdir(folder):
        labels.append(filename.split('_')[2].split('.')[0])
        img_filenames.append(os.path.join(folder, filename))
    return img_filenames, labels


def _summed_from_folder(folder, n):
    from PIL import Image
    files, labels = load_images_from_folder(folder)
    summed = np.zeros((n, n), dtype=np.int64)
    seen = np.zeros((n, n), dtype=bool)
    correct = 0
    for path, lab in zip(files, labels):
        img = np.asarray(Image.open(path).convert("L"))        # cv2.imread(path, 0)
        if img.shape != (n, n):
            raise ValueError("%s is %s, expected %dx%d" % (path, img.shape, n, n))
        on = img == 255
        lab = int(lab)
        correct += lab == 1
        summed += on * lab
        seen |= on
    return summed, seen, len(files), correct


def get_pixel_sorted_mask_label(folder='./masks', n=IMG):
    """-> dict {(row, col): sum of the labels of the masks that keep this pixel}, only pixels some mask keeps
    (generate_gp_training_data_imagenet.py:490-516), read back from the PNG wire format validate() writes."""
    summed, seen, total, correct = _summed_from_folder(folder, n)
    print("%d samples, the corrrect prediction number: %d " % (total, correct))
    rows, cols = np.nonzero(seen)
    return {(int(r), int(c)): int(summed[r, c]) for r, c in zip(rows, cols)}


def summed_heatmap_from_folder(folder='./masks/', n=IMG):
    """-> result_gray_img f64[n,n] = sum_i label_i * [mask_i == 255] (gp_regression.py:82-101)."""
    return _summed_from_folder(folder, n)[0].astype(np.float64)


def prepare_training_data(folder='./masks/', n=IMG):
    """-> (train_x f32[P,2], train_y f32[P]) torch tensors: the GP's training set, gp_regression.py:74-153
    (pixels some mask keeps, raster order; y[p] = sum_i label_i * [mask_i[p] == 255]).  Upstream moves them to
    the GPU for gpytorch; here they stay on the host (the GP is the caller's).  One vectorised pass per PNG
    instead of upstream's n*n Python loop per PNG; the JET picture (:107-124) is plotting and not produced."""
    summed, seen, _total, _correct = _summed_from_folder(folder, n)
    rows, cols = np.nonzero(seen)
    train_x = torch.from_numpy(np.stack([rows, cols], axis=1).astype(np.float32))
    train_y = torch.from_numpy(summed[rows, cols].astype(np.float32))
    return train_x, train_y


def score_masks(model, image, segments, onoff, label, stem=None):
    """The batched surface: -> (onoff u8[M,S], score f32[M], pred i32[M]).  `stem`: None = the engine stages by this call's row
    count; a caller that scores PART of an image's rows (several calls, several ranks) passes model.stem_for_rows(all of them) so that every
    part carries the bits of the unsplit call (engine.MaskedForwardEngine.stem_for_rows, shard.job_stem)."""
    return model.score_masks(image, segments, onoff, label) if stem is None else model.score_masks(image, segments, onoff, label, stem=stem)
