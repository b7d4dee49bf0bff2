"""Host side of the masked-perturbation scorer: PyTorch-ROCm provides device memory and streams,
every computation is a hand-written HIP kernel behind the C-ABI of include/mpx.h.

Replaces the inner loop of the reference's three hot-path scripts
(generate_gp_training_data_imagenet.py:221-266, gp_superpixel_data_imagenet.py:276-334,
bayesian_active_learning_imagenet.py:170-218): mask build, `input * mask`, per-mask H2D copy,
batch-1 forward, per-mask D2H sync.  Here M masks of one image are staged and scored in one
batched launch sequence and only the M scores come back.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import MpxError, IMG, NUM_CLASSES

# transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
# generate_gp_training_data_imagenet.py:590-591
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5   # torchvision BatchNorm2d default

ARCH_IDS = {"resnet18": 18, "resnet34": 34, "resnet50": 50, "resnet101": 101, "resnet152": 152,
            # the reference's two small networks (include/mpx.h MPX_ARCH_*)
            "mnist_net": 1, "cifar_resnet20": 2020, "cifar_resnet56": 2056, "cifar_resnet110": 2110}


COMPUTE_UNITS = 256        # MI355X; only what whole_round_batch falls back to when no GPU is visible (CPU tests, documentation)
ROUND_PIXELS_14 = 256      # pixels per tile of the kernels that run ONE workgroup per CU on the 14x14 maps (mpx_conv3p.h, mpx_conv256.h)


def device_compute_units(device=None):
    """Compute units of the GPU the engine will run on -- the number mpx_create reads (hipDeviceAttributeMultiprocessorCount; the
    engine reports it as MaskedForwardEngine.num_cus / mpx_num_cus) and the persistent kernels size their grids from.  256 on a
    whole MI355X; a partitioned or CU-masked device reports fewer.  Falls back to 256 without a GPU."""
    if torch.cuda.is_available():
        return int(torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device).multi_processor_count)
    return COMPUTE_UNITS


def whole_round_batch(limit, num_cus=None):
    """Largest forward batch <= `limit` whose 14x14 maps cut into whole rounds of 256-pixel tiles over the device's `num_cus`
    compute units (None = device_compute_units(); 256 CUs: 334.37 images per round).  The 3x3 patch kernel and the 256x256 tile
    keep one workgroup per CU, so a batch just ABOVE a whole number of rounds pays a whole extra round on the stage that holds
    2/3 of an ImageNet ResNet's time: measured on ResNet-101 (tools/batch_sweep.sh, profiles/r03_batch_sweep.txt) 44.8 us of
    conv time per masked image at 2006 / 2340 / 2674 / 3009 against 45.6-45.9 at 2010 and 2048.  Returns `limit` itself below
    one round."""
    per_round = (device_compute_units() if num_cus is None else int(num_cus)) * ROUND_PIXELS_14
    rounds = int(limit) * 196 // per_round
    if (rounds + 1) * per_round // 196 <= int(limit):       # the last tile of a batch may be partial: 2340 images are 1791.6 -> 1792 tiles
        rounds += 1
    return rounds * per_round // 196 if rounds >= 1 else int(limit)


class BasePredictionWrong(Exception):
    """The unmasked prediction differs from the label; the reference only defines the scorer when
    it is correct (generate_gp_training_data_imagenet.py:215,269-273;
    bayesian_active_learning_imagenet.py:167,219-221)."""


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def rank_segments(segments):
    """Arbitrary integer label map -> (i32[224,224] ranks in np.unique(segments) order, S).
    The reference indexes superpixels through np.unique(segments)
    (generate_gp_training_data_imagenet.py:223,230)."""
    seg = np.asarray(segments)
    if seg.shape != (IMG, IMG):
        raise ValueError("segments must be [%d,%d], got %s" % (IMG, IMG, seg.shape))
    if not np.issubdtype(seg.dtype, np.integer):
        raise ValueError("segments must be an integer label map, got %s" % seg.dtype)
    # ranks in ascending label order = np.unique's order.  Label maps are small non-negative integers (felzenszwalb, grids, an already
    # ranked map): a histogram ranks them in one pass -- np.unique sorts 50,176 values (1 ms, twice per score_masks call, next to
    # ~7 ms of GPU time for a BO round's window table)
    flat = seg.reshape(-1)
    lo, hi = int(flat.min()), int(flat.max())
    if lo >= 0 and hi < 4 * flat.size:
        present = np.bincount(flat, minlength=hi + 1) > 0
        s = int(present.sum())
        if s == hi + 1:                                     # already 0 .. S-1 with no gap
            return np.ascontiguousarray(seg, dtype=np.int32), s
        rank = (np.cumsum(present) - 1).astype(np.int32)
        return rank[flat].reshape(IMG, IMG), s
    uniq, inv = np.unique(seg, return_inverse=True)
    return inv.reshape(IMG, IMG).astype(np.int32), int(len(uniq))


class MaskedForwardEngine:
    """One engine per process per GPU (one RCCL rank).  `arch` is the reference's `-a/--arch`."""

    def __init__(self, arch="resnet101", max_batch=512, device=None, stem=None):
        """stem: how score_packed / score_masks / score_images stage the masks of an image on the ImageNet ResNets --
        "table" (default): the stem by superposition (mpx_stem_table_build once per image, mpx_stem_table_apply per block of mask rows:
        K0, the stem conv and its max pool for all masks of an image without materialising a masked image) for every IMAGE that brings at
        least `stem_table_min_rows` (256) rows -- decided per image, however the rows are packed into calls and batches --, K0 + the MFMA
        stem otherwise (a BO round's 28 .. 118 windows); "conv": always
        K0 into the input staging, then the MFMA stem + max pool inside the forward (rounds 1-3).  stage_masks() is always K0."""
        if arch not in ARCH_IDS:
            raise ValueError("unsupported arch %r (torchvision ResNets and the reference's small networks: %s)" % (arch, sorted(ARCH_IDS)))
        if not torch.cuda.is_available():
            raise MpxError("no MI355X visible: the scorer has no CPU path")
        self._lib = _lib.load()
        self.arch = arch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.max_batch = int(max_batch)
        h = C.c_void_p()
        rc = self._lib.mpx_create(ARCH_IDS[arch], self.max_batch, self.device.index, C.byref(h))
        if rc != 0:
            raise MpxError("mpx_create(%s, max_batch=%d) failed rc=%d" % (arch, max_batch, rc))
        self._h = h
        self.layers = []
        for i in range(self._lib.mpx_num_convs(h)):
            d = _lib.ConvDesc()
            _lib.check(h, self._lib.mpx_conv_info(h, i, C.byref(d)), "mpx_conv_info")
            self.layers.append(d)
        self.flops_per_forward = float(self._lib.mpx_flops_per_forward(h))
        self.num_cus = int(self._lib.mpx_num_cus(h))        # what the persistent kernels' grids are sized from (whole_round_batch)
        self._mean, self._std = _f3(MEAN), _f3(STD)
        g = [C.c_int() for _ in range(4)]
        _lib.check(h, self._lib.mpx_geometry(h, *[C.byref(v) for v in g]), "mpx_geometry")
        self.image_size, self.in_channels, self.num_classes, self.logit_pitch = (int(v.value) for v in g)
        self.small = self.image_size != IMG
        # below this many mask rows per image the table does not pay for every label map: building it costs what K0 + the MFMA stem cost
        # for ~40 masks and a short apply launch leaves the chip half empty.  tools/stem_bench.py, ms per 2340 rows, table / K0 + MFMA
        # stem: 16-pixel grid 96 rows per image 3.2 / 3.6, 256 rows 1.9 / 3.5, 512 rows 1.5 / 3.5; the (fragmented) felzenszwalb fixture
        # 96 rows 5.5 / 3.7, 256 rows 3.1 / 3.5, 512 rows 2.3 / 3.6 -- at 256 no map loses
        self.stem_table_min_rows = 256
        if stem not in (None, "table", "conv"):
            raise ValueError("stem must be 'table' or 'conv', got %r" % (stem,))
        if self.small and stem == "table":
            raise ValueError("%s has no 7x7 stem: stem='table' is the ImageNet ResNets' path" % arch)
        self.stem = "conv" if self.small else (stem or "table")

    def stem_for_rows(self, rows_per_image):
        """The staging an IMAGE that brings `rows_per_image` mask rows to a job gets on this engine: "table" (the stem by superposition) from
        `stem_table_min_rows` rows on, "conv" (K0 + the MFMA stem) below or when the engine was created with stem="conv".  The two stagings
        round differently (<= 2.5e-6 on a score), so whoever SPLITS a job -- shard.score_masks_sharded / heatmap_sharded over ranks, a caller
        cutting an image's rows into several calls -- decides ONCE from the job's row count with this function and hands the answer to
        every call as `stem=`: the shards then carry the bits of the unsplit call."""
        return "table" if (self.stem == "table" and int(rows_per_image) >= self.stem_table_min_rows) else "conv"

    def _staging(self, stem, rows):
        """Resolve the `stem=` argument for ONE image that brings `rows` mask rows: None = stem_for_rows(rows), else the caller's choice."""
        if stem is None:
            return self.stem_for_rows(rows)
        if stem not in ("table", "conv"):
            raise ValueError("stem must be None, 'table' or 'conv', got %r" % (stem,))
        if stem == "table" and self.small:
            raise ValueError("%s has no 7x7 stem: stem='table' is the ImageNet ResNets' path" % self.arch)
        return stem

    # ---- life cycle ----
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mpx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self):
        """nn.Module.eval() of the reference (generate_gp_training_data_imagenet.py:159): the engine
        only has the inference mode (BatchNorm running statistics)."""
        return self

    def cuda(self):
        return self

    @property
    def workspace_bytes(self):
        return int(self._lib.mpx_workspace_bytes(self._h))

    # ---- weights ----
    def load_state_dict(self, sd, eps=BN_EPS, only=None):
        """`sd`: torchvision ResNet state_dict (key names as `models.<arch>().state_dict()`), e.g.
        torch.load(local_path, weights_only=True).  `module.` prefixes (DataParallel) are accepted.  `only`: conv names
        ("layer1.1.conv3", "fc") to (re)load instead of every layer -- the engine rebuilds whatever it derived from a reloaded
        layer (the K-concatenated conv3 | downsample planes, a block tail's permuted copy)."""
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        if only is not None:
            only = set(only)
            unknown = only - {d.name.decode() for d in self.layers}
            if unknown:
                raise KeyError("no such conv layers: %s" % sorted(unknown))

        def get(key, shape):
            if key not in sd:
                raise KeyError("state_dict lacks %r" % key)
            t = sd[key].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError("%s has shape %s, expected %s" % (key, tuple(t.shape), shape))
            return t

        for i, d in enumerate(self.layers):
            name, bn = d.name.decode(), d.bn_name.decode()
            if only is not None and name not in only:
                continue
            wshape = (d.cout, d.cin) if (d.ksize == 1 and sd[name + ".weight"].dim() == 2) else (d.cout, d.cin, d.ksize, d.ksize)
            w = get(name + ".weight", wshape)
            if not bn:          # no BatchNorm: fc, or the MNIST net's conv6 -- the layer's own bias goes in as `beta`
                b = get(name + ".bias", (d.cout,))
                args = (_ptr(w), None, None, _ptr(b), None, None)
            else:
                cb = get(name + ".bias", (d.cout,)) if (name + ".bias") in sd else None      # nn.Conv2d(bias=True) + BN
                g = get(bn + ".weight", (d.cout,))
                b = get(bn + ".bias", (d.cout,))
                m = get(bn + ".running_mean", (d.cout,))
                v = get(bn + ".running_var", (d.cout,))
                args = (_ptr(w), _ptr(cb), _ptr(g), _ptr(b), _ptr(m), _ptr(v))
            _lib.check(self._h, self._lib.mpx_set_conv_weights(self._h, i, *args, float(eps)),
                       "mpx_set_conv_weights(%s)" % name)
        return self

    # ---- kernels ----
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def stage_masks(self, image, seg, onoff, slot0=0, out_f32=None):
        """K0.  image: device u8[224,224,3] (raw) or f32[3,224,224] (normalised); seg: device
        i32[224,224] ranks; onoff: device u8[M,S].  Fills input slots [slot0, slot0+M)."""
        for name, t in (("image", image), ("seg", seg), ("onoff", onoff)):
            if t.device != self.device or not t.is_contiguous():
                raise ValueError("%s must be a contiguous tensor on %s" % (name, self.device))
        if seg.dtype != torch.int32 or tuple(seg.shape) != (IMG, IMG):
            raise ValueError("seg must be int32[%d,%d]" % (IMG, IMG))
        if onoff.dtype != torch.uint8 or onoff.dim() != 2:
            raise ValueError("onoff must be uint8[M,S]")
        m, s = onoff.shape
        if image.dtype == torch.uint8 and tuple(image.shape) == (IMG, IMG, 3):
            u8, f32 = _ptr(image), None
        elif image.dtype == torch.float32 and tuple(image.shape) == (3, IMG, IMG):
            u8, f32 = None, _ptr(image)
        else:
            raise ValueError("image must be uint8[224,224,3] or float32[3,224,224], got %s%s"
                             % (image.dtype, tuple(image.shape)))
        if out_f32 is not None and (out_f32.dtype != torch.float32 or tuple(out_f32.shape) != (m, 3, IMG, IMG)
                                    or not out_f32.is_contiguous()):
            raise ValueError("out_f32 must be contiguous float32[M,3,224,224]")
        _lib.check(self._h, self._lib.mpx_mask_apply_normalize(
            self._h, u8, f32, _ptr(seg), _ptr(onoff), int(m), int(s), self._mean, self._std,
            int(slot0), _ptr(out_f32), self._stream()), "mpx_mask_apply_normalize")

    def build_stem_table(self, image, seg, num_segments):
        """mpx_stem_table_build: the mask-independent terms of the stem for ONE image (device u8[224,224,3] or f32[3,224,224]) and its
        rank map (device i32[224,224], labels in [0, num_segments)); the engine holds one table at a time."""
        for name, t in (("image", image), ("seg", seg)):
            if t.device != self.device or not t.is_contiguous():
                raise ValueError("%s must be a contiguous tensor on %s" % (name, self.device))
        if seg.dtype != torch.int32 or tuple(seg.shape) != (IMG, IMG):
            raise ValueError("seg must be int32[%d,%d]" % (IMG, IMG))
        if image.dtype == torch.uint8 and tuple(image.shape) == (IMG, IMG, 3):
            u8, f32 = _ptr(image), None
        elif image.dtype == torch.float32 and tuple(image.shape) == (3, IMG, IMG):
            u8, f32 = None, _ptr(image)
        else:
            raise ValueError("image must be uint8[224,224,3] or float32[3,224,224], got %s%s" % (image.dtype, tuple(image.shape)))
        _lib.check(self._h, self._lib.mpx_stem_table_build(self._h, u8, f32, _ptr(seg), int(num_segments), self._mean, self._std,
                                                           self._stream()), "mpx_stem_table_build")

    def apply_stem_table(self, onoff, slot0=0):
        """mpx_stem_table_apply: the pooled stem output of M mask rows (device u8[M,S]) of the image whose table is in place, into slots
        [slot0, slot0+M); the next forward over those slots starts behind the max pool."""
        if onoff.device != self.device or not onoff.is_contiguous() or onoff.dtype != torch.uint8 or onoff.dim() != 2:
            raise ValueError("onoff must be a contiguous uint8[M,S] tensor on %s" % self.device)
        m, s = onoff.shape
        _lib.check(self._h, self._lib.mpx_stem_table_apply(self._h, _ptr(onoff), int(m), int(s), int(slot0), self._stream()),
                   "mpx_stem_table_apply")

    def forward(self, batch, labels, want_logits=False, score_out=None, pred_out=None):
        """Whole network over the staged slots [0,batch).  labels: device i32[batch].
        returns (score f32[batch], pred i32[batch][, logits f32[batch,1000]]) on the device; score_out / pred_out
        (contiguous device tensors of that shape and dtype) receive the results without any allocation."""
        if labels.dtype != torch.int32 or labels.device != self.device or labels.numel() != batch or not labels.is_contiguous():
            raise ValueError("labels must be contiguous int32[%d] on %s" % (batch, self.device))
        for name, t, dt in (("score_out", score_out, torch.float32), ("pred_out", pred_out, torch.int32)):
            if t is not None and (t.dtype != dt or t.device != self.device or t.numel() != batch or not t.is_contiguous()):
                raise ValueError("%s must be contiguous %s[%d] on %s" % (name, dt, batch, self.device))
        score = score_out if score_out is not None else torch.empty(batch, dtype=torch.float32, device=self.device)
        pred = pred_out if pred_out is not None else torch.empty(batch, dtype=torch.int32, device=self.device)
        logits = torch.empty(batch, self.logit_pitch, dtype=torch.float32, device=self.device) if want_logits else None
        _lib.check(self._h, self._lib.mpx_forward(self._h, _ptr(labels), _ptr(score), _ptr(pred), _ptr(logits),
                                                  int(batch), self._stream()), "mpx_forward")
        return (score, pred, logits[:, :self.num_classes]) if want_logits else (score, pred)

    def input_plane_shape(self, n):
        """Shape of n slots of the engine's input staging: [n,230,230,4] (padded NHWC4, the 7x7 stem's operand) for the ImageNet
        ResNets, [n,H,W,32] (channels padded to 32) for the small networks (csrc/mpx_api.hip mpx_create)."""
        if self.small:
            return (int(n), self.image_size, self.image_size, 32)
        return (int(n), _lib.IMG_PAD, _lib.IMG_PAD, 4)

    def input_planes(self, n=None):
        """Zero-copy fp16 views of the engine-owned input staging planes (hi, lo), shaped input_plane_shape(n).
        For tests and diagnostics; reading them changes nothing.  Whoever WRITES slots by hand calls mark_input_staged() before forward()."""
        n = self.max_batch if n is None else int(n)
        if not 0 < n <= self.max_batch:
            raise ValueError("input_planes: n must be in [1, max_batch=%d]" % self.max_batch)
        hi, lo = C.c_void_p(), C.c_void_p()
        _lib.check(self._h, self._lib.mpx_input_planes(self._h, C.byref(hi), C.byref(lo)), "mpx_input_planes")
        shape = self.input_plane_shape(n)

        class _View:
            def __init__(self, ptr):
                self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": "<f2", "version": 2}

        return (torch.as_tensor(_View(hi.value), device=self.device),
                torch.as_tensor(_View(lo.value), device=self.device))

    def mark_input_staged(self, slot0, m):
        """mpx_mark_input_staged: slots [slot0, slot0+m) of the input planes were written by hand -- the next forward runs the stem on them."""
        _lib.check(self._h, self._lib.mpx_mark_input_staged(self._h, int(slot0), int(m)), "mpx_mark_input_staged")

    def stem_planes(self, n=None):
        """Zero-copy fp16 views (hi, lo) [n,56,56,64] of the engine-owned pooled stem output planes (ImageNet ResNets).  For tests."""
        n = self.max_batch if n is None else int(n)
        if self.small or not 0 < n <= self.max_batch:
            raise ValueError("stem_planes: an ImageNet ResNet engine and n in [1, max_batch=%d]" % self.max_batch)
        hi, lo = C.c_void_p(), C.c_void_p()
        _lib.check(self._h, self._lib.mpx_stem_planes(self._h, C.byref(hi), C.byref(lo)), "mpx_stem_planes")
        shape = (n, 56, 56, 64)

        class _View:
            def __init__(self, ptr):
                self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": "<f2", "version": 2}

        return (torch.as_tensor(_View(hi.value), device=self.device), torch.as_tensor(_View(lo.value), device=self.device))

    # ---- the batched surface ----
    def _image_to_device(self, image):
        t = torch.as_tensor(image)
        if t.dim() == 4 and t.shape[0] == 1:
            t = t[0]
        if t.dtype == torch.uint8:
            if tuple(t.shape) != (IMG, IMG, 3):
                raise ValueError("uint8 image must be [224,224,3] (HWC), got %s" % (tuple(t.shape),))
        elif t.dtype == torch.float32:
            if tuple(t.shape) != (3, IMG, IMG):
                raise ValueError("float32 image must be [3,224,224] (normalised CHW), got %s" % (tuple(t.shape),))
        else:
            raise ValueError("image must be uint8 HWC or float32 CHW, got %s" % t.dtype)
        return t.contiguous().to(self.device)

    def score_masks_removed(self, image_chw, segments, removed, label, return_logits=False, return_inputs=False):
        """The small networks' scorer: (image f32[C,H,W] as the loader yields it, segments int[H,W],
        removed u8[M,S] with 1 = superpixel switched OFF, label) -> (removed, score f32[M], pred i32[M][, logits][, inputs]).
        The mask convention is generate_gp_training_data_cifar.py:274-321 / generate_gp_training_data_mnist.py:167-242:
        min-max the picture to [0,255], multiply by the {0,255} mask, min-max again, * f32(1/255); pred[m] is the reference's
        `pred_mask`, score[m] the softmax probability of `label`."""
        if not self.small:
            raise ValueError("score_masks_removed is the small networks' convention; %s uses score_masks" % self.arch)
        x = torch.as_tensor(image_chw)
        if x.dim() == 4 and x.shape[0] == 1:
            x = x[0]
        n = self.image_size
        if x.dtype != torch.float32 or tuple(x.shape) != (self.in_channels, n, n):
            raise ValueError("image must be float32[%d,%d,%d], got %s%s" % (self.in_channels, n, n, x.dtype, tuple(x.shape)))
        seg = np.asarray(segments)
        if seg.shape != (n, n) or not np.issubdtype(seg.dtype, np.integer):
            raise ValueError("segments must be an integer [%d,%d] label map" % (n, n))
        uniq, inv = np.unique(seg, return_inverse=True)
        s = int(len(uniq))
        removed = np.ascontiguousarray(removed)
        if removed.dtype != np.uint8 or removed.ndim != 2 or removed.shape[1] != s:
            raise ValueError("removed must be uint8[M,%d], got %s%s" % (s, removed.dtype, removed.shape))
        if not 0 <= int(label) < self.num_classes:
            raise ValueError("label %r outside [0,%d)" % (label, self.num_classes))
        m = removed.shape[0]
        score = np.empty(m, dtype=np.float32)
        pred = np.empty(m, dtype=np.int32)
        logits = np.empty((m, self.num_classes), dtype=np.float32) if return_logits else None
        inputs = np.empty((m, self.in_channels, n, n), dtype=np.float32) if return_inputs else None
        x_d = x.contiguous().to(self.device)
        seg_d = torch.from_numpy(inv.reshape(n, n).astype(np.int32)).to(self.device)
        rem_d = torch.from_numpy(removed).to(self.device)
        for s0 in range(0, m, self.max_batch):
            b = min(self.max_batch, m - s0)
            labels = torch.full((b,), int(label), dtype=torch.int32, device=self.device)
            out_f32 = torch.empty(b, self.in_channels, n, n, dtype=torch.float32, device=self.device) if return_inputs else None
            _lib.check(self._h, self._lib.mpx_mask_apply_minmax(self._h, _ptr(x_d), _ptr(seg_d), _ptr(rem_d[s0:s0 + b].contiguous()),
                                                                int(b), int(s), 0, _ptr(out_f32), self._stream()),
                       "mpx_mask_apply_minmax")
            out = self.forward(b, labels, want_logits=return_logits)
            score[s0:s0 + b] = out[0].cpu().numpy()
            pred[s0:s0 + b] = out[1].cpu().numpy()
            if return_logits:
                logits[s0:s0 + b] = out[2].cpu().numpy()
            if return_inputs:
                inputs[s0:s0 + b] = out_f32.cpu().numpy()
        res = (removed, score, pred)
        if return_logits:
            res += (logits,)
        if return_inputs:
            res += (inputs,)
        return res

    def score_masks(self, image, segments, onoff, label, return_logits=False, stem=None):
        """(image, segments, onoff u8[M,S], label) -> (onoff u8[M,S], score f32[M], pred i32[M]).
        score[m] = softmax(model(normalised_image * mask_m))[label]
        (bayesian_active_learning_imagenet.py:187-198); pred[m] == label is the generators' binary
        label (generate_gp_training_data_imagenet.py:248,257).  stem: None = staged by this call's row count (stem_for_rows(M)); "table" /
        "conv" = the caller's choice -- a caller that scores PART of an image's rows passes stem_for_rows(all of them)."""
        if self.small:
            raise ValueError("%s scores with score_masks_removed (the small networks' mask convention)" % self.arch)
        seg_rank, s = rank_segments(segments)
        onoff = np.ascontiguousarray(onoff)
        if onoff.dtype != np.uint8 or onoff.ndim != 2 or onoff.shape[1] != s:
            raise ValueError("onoff must be uint8[M,%d] (S = number of distinct segment labels), got %s%s"
                             % (s, onoff.dtype, onoff.shape))
        if not 0 <= int(label) < NUM_CLASSES:
            raise ValueError("label %r outside [0,1000)" % (label,))
        m = onoff.shape[0]
        if not return_logits:
            # one upload, forwards of max_batch slots back to back, ONE download at the end (no per-chunk synchronisation)
            score, pred = self.score_images([image], [seg_rank], [onoff], [label], stem=stem)[0]
            return onoff, score, pred
        score = np.empty(m, dtype=np.float32)
        pred = np.empty(m, dtype=np.int32)
        logits = np.empty((m, NUM_CLASSES), dtype=np.float32)
        if m == 0:
            return onoff, score, pred, logits
        img_d = self._image_to_device(image)
        seg_d = torch.from_numpy(seg_rank).to(self.device)
        onoff_d = torch.from_numpy(onoff).to(self.device)
        table = self._staging(stem, m) == "table"
        for s0 in range(0, m, self.max_batch):
            b = min(self.max_batch, m - s0)
            labels = torch.full((b,), int(label), dtype=torch.int32, device=self.device)
            if table:
                if s0 == 0:
                    self.build_stem_table(img_d, seg_d, s)
                self.apply_stem_table(onoff_d[s0:s0 + b], 0)
            else:
                self.stage_masks(img_d, seg_d, onoff_d[s0:s0 + b], 0)
            out = self.forward(b, labels, want_logits=return_logits)
            score[s0:s0 + b] = out[0].cpu().numpy()
            pred[s0:s0 + b] = out[1].cpu().numpy()
            if return_logits:
                logits[s0:s0 + b] = out[2].cpu().numpy()
        return (onoff, score, pred, logits) if return_logits else (onoff, score, pred)

    def score_packed(self, images, segs, onoffs, label_rows, score_out, pred_out, stem=None):
        """Device-resident packed scoring: the mask rows of SEVERAL images share forward batches of up to max_batch slots
        (an image's rows may straddle two batches), so the network always runs at the batch size it is fast at -- the
        reference scores one mask per forward (generate_gp_training_data_imagenet.py:240-248,
        gp_superpixel_data_imagenet.py:299-307).  images: sequence of device tensors (u8[224,224,3] or f32[3,224,224]);
        segs: ONE device i32[224,224] rank map shared by all images, or a sequence with one per image; onoffs: sequence of
        device u8[M_i, S_i]; label_rows: device i32[sum M_i] (image i's label repeated M_i times); score_out f32 / pred_out
        i32 [sum M_i] receive the results.  Allocates nothing, synchronises nothing.  Staging is decided PER IMAGE, from that image's own
        row count: `stem` = None stages image i by stem_for_rows(M_i) (the table from stem_table_min_rows rows on, K0 + the MFMA stem below),
        "table" / "conv" force one kind for every image of the call (a caller that hands over PART of an image's rows passes
        stem_for_rows(all of them)).  A forward takes its slots from one kind of staging (mpx_forward refuses a mixed batch), so the pending
        forward is flushed where the kind changes between consecutive images -- callers that can reorder (score_images) group the images by
        kind first.  A row's bits depend neither on where it sits in a batch nor on what else the call scores (kernel choice does not depend
        on the position of a mask, staging only on the row's own image): results are bit-identical to scoring every image on its own."""
        n = len(images)
        shared = isinstance(segs, torch.Tensor)
        if len(onoffs) != n or (not shared and len(segs) != n):
            raise ValueError("images, segs and onoffs must have one entry per image")
        total = sum(int(o.shape[0]) for o in onoffs)
        for name, t, dt in (("label_rows", label_rows, torch.int32), ("score_out", score_out, torch.float32), ("pred_out", pred_out, torch.int32)):
            if t.dtype != dt or t.device != self.device or t.numel() != total or not t.is_contiguous():
                raise ValueError("%s must be contiguous %s[%d] on %s" % (name, dt, total, self.device))
        label_rows, score_out, pred_out = label_rows.view(-1), score_out.view(-1), pred_out.view(-1)
        done = 0            # rows already handed to a forward
        used = 0            # slots staged for the next forward
        used_kind = None    # how those slots were staged
        if stem is not None:
            self._staging(stem, 0)

        def flush():
            nonlocal done, used
            self.forward(used, label_rows[done:done + used], score_out=score_out[done:done + used], pred_out=pred_out[done:done + used])
            done += used
            used = 0

        for i in range(n):
            m, r = int(onoffs[i].shape[0]), 0
            if not m:
                continue
            seg = segs if shared else segs[i]
            kind = self.stem_for_rows(m) if stem is None else stem
            if used and kind != used_kind:
                flush()                 # one forward, one kind of staging
            used_kind = kind
            if kind == "table":
                self.build_stem_table(images[i], seg, int(onoffs[i].shape[1]))      # once per image; its rows follow in one or two forwards
            while r < m:
                take = min(m - r, self.max_batch - used)
                if kind == "table":
                    self.apply_stem_table(onoffs[i][r:r + take], used)
                else:
                    self.stage_masks(images[i], seg, onoffs[i][r:r + take], used)
                used += take
                r += take
                if used == self.max_batch:
                    flush()
        if used:
            flush()

    def score_images(self, images, segments, onoffs, labels, stem=None):
        """Host convenience over score_packed: [(image, segments, onoff u8[M_i,S_i], label)] for several images ->
        [(score f32[M_i], pred i32[M_i])], with ONE upload of the inputs and ONE download of all scores.  `segments` are
        arbitrary integer label maps (ranked here as score_masks does); `stem` as in score_packed: None stages every image by its own row
        count, so an image gets the same bits here as when it is scored alone, whatever it is packed with."""
        if self.small:
            raise ValueError("%s scores with score_masks_removed (the small networks' mask convention)" % self.arch)
        n = len(images)
        if not (len(segments) == len(onoffs) == len(labels) == n):
            raise ValueError("images, segments, onoffs and labels must have the same length")
        self._staging(stem, 0)           # a bad `stem` fails here, whatever the rows
        img_d, seg_d, onoff_d, lab, sizes = [], [], [], [], []
        for i in range(n):
            seg_rank, s = rank_segments(segments[i])
            o = np.ascontiguousarray(onoffs[i])
            if o.dtype != np.uint8 or o.ndim != 2 or o.shape[1] != s:
                raise ValueError("onoffs[%d] must be uint8[M,%d] (S = number of distinct segment labels), got %s%s" % (i, s, o.dtype, o.shape))
            if not 0 <= int(labels[i]) < NUM_CLASSES:
                raise ValueError("label %r outside [0,1000)" % (labels[i],))
            sizes.append(o.shape[0])
            if o.shape[0] == 0:
                continue
            img_d.append(self._image_to_device(images[i]))
            seg_d.append(torch.from_numpy(seg_rank).to(self.device))
            onoff_d.append(torch.from_numpy(o).to(self.device))
            lab.append(np.full(o.shape[0], int(labels[i]), dtype=np.int32))
        total = int(sum(sizes))
        if total == 0:
            return [(np.empty(0, np.float32), np.empty(0, np.int32)) for _ in range(n)]
        # every image is staged by ITS OWN row count (score_packed); the images of one kind go first so that the forward is flushed at most
        # once for the change of kind (stable: the order within a kind is the caller's)
        order = list(range(len(img_d)))
        if stem is None:
            order.sort(key=lambda k: self.stem_for_rows(int(onoff_d[k].shape[0])) != "table")
        label_rows = torch.from_numpy(np.concatenate([lab[k] for k in order])).to(self.device)
        score = torch.empty(total, dtype=torch.float32, device=self.device)
        pred = torch.empty(total, dtype=torch.int32, device=self.device)
        self.score_packed([img_d[k] for k in order], [seg_d[k] for k in order], [onoff_d[k] for k in order], label_rows, score, pred, stem=stem)
        score, pred = score.cpu().numpy(), pred.cpu().numpy()
        packed, at = {}, 0
        for k in order:
            m = int(onoff_d[k].shape[0])
            packed[k] = (score[at:at + m].copy(), pred[at:at + m].copy())
            at += m
        out, k = [], 0
        for m in sizes:
            if m == 0:
                out.append((np.empty(0, np.float32), np.empty(0, np.int32)))
            else:
                out.append(packed[k])
                k += 1
        return out

    def predict(self, image):
        """Unmasked forward: (argmax class, softmax f32[1000])
        (generate_gp_training_data_imagenet.py:193,202)."""
        seg = np.zeros((IMG, IMG), dtype=np.int32)
        _o, _s, pred, logits = self.score_masks(image, seg, np.ones((1, 1), dtype=np.uint8), 0, return_logits=True)
        z = logits[0].astype(np.float64)
        p = np.exp(z - z.max())
        return int(pred[0]), (p / p.sum()).astype(np.float32)

    def heatmap_accumulate(self, seg, onoff, pred, labels, heat):
        """K5: heat[p] += sum_m [pred[m] == labels[m]] * onoff[m][seg[p]] on the device
        (gp_superpixel_data_imagenet.py:322-323).  seg i32[224,224], onoff u8[M,S], pred/labels i32[M],
        heat f32[224,224] -- all device tensors; heat is updated in place."""
        m, s = onoff.shape
        for name, t, dt in (("seg", seg, torch.int32), ("onoff", onoff, torch.uint8), ("pred", pred, torch.int32),
                            ("labels", labels, torch.int32), ("heat", heat, torch.float32)):
            if t.device != self.device or t.dtype != dt or not t.is_contiguous():
                raise ValueError("%s must be a contiguous %s tensor on %s" % (name, dt, self.device))
        if tuple(seg.shape) != (IMG, IMG) or tuple(heat.shape) != (IMG, IMG) or pred.numel() != m or labels.numel() != m:
            raise ValueError("shapes: seg/heat [224,224], pred/labels [M], onoff [M,S]")
        _lib.check(self._h, self._lib.mpx_heatmap_accumulate(self._h, _ptr(seg), _ptr(onoff), _ptr(pred), _ptr(labels),
                                                             int(m), int(s), _ptr(heat), self._stream()),
                   "mpx_heatmap_accumulate")
        return heat

    def heatmap_device(self, image, seg_rank, onoff, label, buf, stem=None):
        """Score the M mask-vectors of one image and accumulate their heat map WITHOUT a host round trip:
        buf (device f32[224*224 + 1]) gets heat[p] += sum_m [pred[m] == label] * onoff[m][seg[p]] in its first 224*224 elements
        (K5, gp_superpixel_data_imagenet.py:322-323) and the number of correctly predicted masks added to its last element --
        the layout shard.heatmap_sharded closes with ONE all_reduce (which passes stem_for_rows(the image's rows over ALL ranks) as `stem`).
        -> (score f32[M], pred i32[M]) device tensors."""
        if buf.dtype != torch.float32 or buf.device != self.device or buf.numel() != IMG * IMG + 1 or not buf.is_contiguous():
            raise ValueError("buf must be a contiguous float32[%d] tensor on %s" % (IMG * IMG + 1, self.device))
        seg_rank, s = rank_segments(seg_rank)               # a rank map passes through on the histogram's fast path
        onoff = np.ascontiguousarray(onoff)
        if onoff.dtype != np.uint8 or onoff.ndim != 2 or onoff.shape[1] != s:
            raise ValueError("onoff must be uint8[M,%d] (S = number of distinct segment labels), got %s%s" % (s, onoff.dtype, onoff.shape))
        if not 0 <= int(label) < NUM_CLASSES:
            raise ValueError("label %r outside [0,1000)" % (label,))
        m = int(onoff.shape[0])
        self._staging(stem, m)
        score = torch.empty(m, dtype=torch.float32, device=self.device)
        pred = torch.empty(m, dtype=torch.int32, device=self.device)
        if m == 0:
            return score, pred
        seg_d = torch.from_numpy(seg_rank).to(self.device)
        onoff_d = torch.from_numpy(onoff).to(self.device)
        labels = torch.full((m,), int(label), dtype=torch.int32, device=self.device)
        self.score_packed([self._image_to_device(image)], seg_d, [onoff_d], labels, score, pred, stem=stem)
        self.heatmap_accumulate(seg_d, onoff_d, pred, labels, buf[:IMG * IMG].view(IMG, IMG))
        buf[IMG * IMG] += (pred == labels).sum()
        return score, pred

    def heatmap(self, seg_rank, onoff, pred, label):
        """Host-array convenience over K5: -> f64[224,224] = sum_m [pred[m] == label] * onoff[m][seg[p]]
        (counts are integers < 2^24, so the f32 accumulation on the device is exact)."""
        m = int(np.asarray(onoff).shape[0])
        heat = torch.zeros(IMG, IMG, dtype=torch.float32, device=self.device)
        if m:
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
            self.heatmap_accumulate(t(seg_rank, np.int32), t(onoff, np.uint8), t(pred, np.int32),
                                    torch.full((m,), int(label), dtype=torch.int32, device=self.device), heat)
        return heat.cpu().numpy().astype(np.float64)

    # ---- kernel variants (tuning / tests) ----
    def set_conv_tile(self, layer, tile):
        """Select the conv kernel variant of one layer (index or torchvision name); tile < 0 = default."""
        i = layer if isinstance(layer, int) else [d.name.decode() for d in self.layers].index(layer)
        _lib.check(self._h, self._lib.mpx_set_conv_tile(self._h, int(i), int(tile)), "mpx_set_conv_tile")

    def set_fusion(self, on=True):
        """mpx_set_fusion.  True (default state) = every fusion: a stage's first conv3 with its downsample conv as one
        K-concatenated launch, the ImageNet stem with its max pool, and layer1's block tails (conv2 -> conv3 + identity -> the
        next block's conv1, mpx_bottleneck_tail).  False = one launch per layer.  An int is passed through as the mask
        (1 = round 2's fusions without the block tails).  The stem fusion is bit-identical to the separate launches; the other
        two change the fp32 summation order (same tolerance)."""
        mask = (3 if on else 0) if isinstance(on, bool) else int(on)
        _lib.check(self._h, self._lib.mpx_set_fusion(self._h, mask), "mpx_set_fusion")

    def bottleneck_tails(self):
        """[(conv2, conv3, downsample or -1, next conv1)] layer indices of the blocks whose tail runs as one launch."""
        out = []
        for k in range(self._lib.mpx_num_bottleneck_tails(self._h)):
            v = [C.c_int() for _ in range(4)]
            _lib.check(self._h, self._lib.mpx_bottleneck_tail_info(self._h, k, *[C.byref(x) for x in v]), "mpx_bottleneck_tail_info")
            out.append(tuple(int(x.value) for x in v))
        return out

    def conv_tile(self, layer):
        i = layer if isinstance(layer, int) else [d.name.decode() for d in self.layers].index(layer)
        return int(self._lib.mpx_get_conv_tile(self._h, int(i)))

    # ---- profiling ----
    def profile(self, on=True):
        _lib.check(self._h, self._lib.mpx_profile_enable(self._h, 1 if on else 0), "mpx_profile_enable")

    def collect_profile(self):
        """{'ms': {kind: ms}, 'launches': {kind: n}, 'per_conv_ms': [...]} accumulated since the last call."""
        ms = (C.c_double * 4)()
        n = (C.c_longlong * 4)()
        per = (C.c_double * len(self.layers))()
        _lib.check(self._h, self._lib.mpx_profile_collect(self._h, ms, n, per), "mpx_profile_collect")
        kinds = ("conv", "mask_apply_normalize", "pool", "head")
        return {"ms": dict(zip(kinds, list(ms))), "launches": dict(zip(kinds, list(n))),
                "per_conv_ms": list(per)}
