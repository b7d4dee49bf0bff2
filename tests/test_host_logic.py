"""CPU-side checks: the C-ABI library loads and exports every symbol include/mpx.h declares, the
host-only weight packer, mask-vector logic, segment ranking, sharding math and the reference-named
API driven by a fake engine (no GPU compute anywhere in this file)."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest
import torch

from network_interpretation_imagenet_amd.engine import MaskedForwardEngine
from network_interpretation_imagenet_amd import _lib, api, masks, shard, synth
from network_interpretation_imagenet_amd.engine import rank_segments
from oracle import resnet_ref, scorer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(mpx_lib):
    header = open(os.path.join(ROOT, "include", "mpx.h")).read()
    declared = set(re.findall(r"(?m)^(?:int|const char\*|size_t|double)\s+(mpx_[a-z0-9_]+)\(", header))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(mpx_lib, name) is not None


def test_build_stamp_covers_every_source(tmp_path):
    """The .so staleness stamp hashes every file the library is built from (glob of csrc/*.h, csrc/*.hip and
    include/mpx.h) plus the compiler flags: an edit to ANY kernel header, or a flag change, must invalidate it --
    the GPU box receives the prebuilt .so and would otherwise run code that is not in the tree."""
    import shutil
    import __graft_entry__ as g
    names = {os.path.basename(f) for f in g.lib_sources()}
    assert {"mpx_api.hip", "mpx_kernels.h", "mpx_conv.h", "mpx_conv3p.h", "mpx.h"} <= names
    csrc = os.path.join(ROOT, "network_interpretation_imagenet_amd", "csrc")
    for f in os.listdir(csrc):                       # every header mpx_api.hip can include is hashed
        if f.endswith((".h", ".hip")):
            assert f in names, f
    root = tmp_path / "tree"
    shutil.copytree(csrc, root / "network_interpretation_imagenet_amd" / "csrc")
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    lib = str(root / "libfake.so")
    open(lib, "wb").close()
    src = g.lib_sources(str(root))
    with open(lib + ".sha256", "w") as fh:
        fh.write(g._source_hash(src, g.HIPCC_FLAGS) + "\n")
    assert not g._stale(lib, src, g.HIPCC_FLAGS)
    for victim in ("mpx_conv3p.h", "mpx_kernels.h", "mpx.h"):
        path = [f for f in src if os.path.basename(f) == victim][0]
        before = open(path).read()
        with open(path, "a") as fh:
            fh.write("// touched\n")
        assert g._stale(lib, src, g.HIPCC_FLAGS), victim
        with open(path, "w") as fh:
            fh.write(before)
        assert not g._stale(lib, src, g.HIPCC_FLAGS)
    assert g._stale(lib, src, g.HIPCC_FLAGS + ["-DX"])                  # flags are part of the stamp
    (root / "network_interpretation_imagenet_amd" / "csrc" / "mpx_new_kernel.h").write_text("// new\n")
    assert g._stale(lib, g.lib_sources(str(root)), g.HIPCC_FLAGS)      # a new header is picked up by the glob


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def _device_elf(lib_path):
    """the gfx950 code object inside a HIP fat binary (the clang offload bundle is located by its magic)"""
    import struct
    blob = open(lib_path, "rb").read()
    i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in %s" % lib_path
    n = struct.unpack_from("<Q", blob, i + 24)[0]
    off, elf = i + 32, None
    for _ in range(n):
        o, size, tl = struct.unpack_from("<QQQ", blob, off)
        off += 24
        triple = blob[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            elf = blob[i + o:i + o + size]
    assert elf, "no gfx950 code object in the bundle"
    return elf


def _device_kernels(lib_path):
    """(name -> metadata dict) of that code object: the AMDGPU metadata note as llvm-readelf prints it (no GPU needed)"""
    import subprocess
    import tempfile
    readelf = os.path.join(LLVM_BIN, "llvm-readelf")
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not installed")
    elf = _device_elf(lib_path)
    with tempfile.NamedTemporaryFile(suffix=".elf") as fh:
        fh.write(elf)
        fh.flush()
        notes = subprocess.run([readelf, "--notes", fh.name], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "agpr_count" and line.lstrip().startswith("-"):
            cur = {}
        if cur is not None:
            cur[key] = val
            if key == "name":
                kernels[val] = cur
    return kernels


def test_conv_kernels_have_no_scratch_and_convw_keeps_its_weights_in_agprs(mpx_lib):
    """Static guard on what hipcc made of the hand-scheduled kernels (read from the built library, no GPU): a scratch reload inside a K loop
    drains vmcnt and breaks the counted waits, and the weights-in-registers kernel (csrc/mpx_convw.h) is only what its name says while the
    256 weight registers of a wave are the whole AGPR half of its file and nothing spills."""
    ks = _device_kernels(_lib.LIB_PATH)
    conv = {n: k for n, k in ks.items() if "_f16x3_kernel" in n}
    assert len(conv) >= 15, sorted(ks)
    for n, k in conv.items():
        scratch = int(k["private_segment_fixed_size"])
        if any(t in n for t in ("convw_", "convx_", "conv256p_", "conv3x3pp_", "btail_")):
            # the persistent kernels: their tile loop IS the K loop, a spill anywhere in it is a reload between MFMAs
            assert scratch == 0 and int(k.get("vgpr_spill_count", 0)) == 0, "%s: %d bytes of scratch per lane" % (n, scratch)
        else:
            # one workgroup per tile: hipcc parks a few prologue values across the K loop (stored before the first MFMA, reloaded behind
            # the last one -- checked in the ISA); more than that would mean spills inside the loop
            assert scratch <= 72, "%s: %d bytes of scratch per lane" % (n, scratch)
    convw = {n: k for n, k in conv.items() if "convw_f16x3_kernel" in n}
    assert len(convw) == 1, sorted(conv)                  # RELU = true only: a call without ReLU or residual takes tile 10's kernel (launch_convw)
    for n, k in convw.items():
        assert int(k["agpr_count"]) == 256 and int(k["vgpr_count"]) <= 512, (n, k)


def test_convw_k_loop_is_what_the_source_says():
    """What hipcc must (not) make of csrc/mpx_convw.h, checked in the disassembly of the built library.  Round 4: between the first and the
    last MFMA of the tile loop there is no v_accvgpr copy (the weights are MFMA operands IN the AGPRs: inline asm with an "a" constraint)
    and no scratch access; a tile is 8 K steps x 48 MFMAs, every one with an AGPR A operand.  Round 5 (column-major K loop, the epilogue
    slices in the MFMA gaps): the MFMAs stay evenly spread -- no gap holds more than a handful of instructions, i.e. the epilogue was not
    hoisted back into one block -- the residual loads, the pieces and the stores all sit INSIDE the loop, and every vmcnt in it is a
    counted immediate that leaves instructions in flight (a vmcnt(0) would serialise the tile on its own stores; hipcc's own waits in front of
    the residual lines' first uses are 14 .. 24, and it guards the first fragment read of a tile, whose registers it shares with the last
    column's store data, with vmcnt(4) / vmcnt(5))."""
    import subprocess
    import tempfile
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not installed")
    import __graft_entry__ as g
    g.build()
    with tempfile.NamedTemporaryFile(suffix=".elf") as fh:
        fh.write(_device_elf(_lib.LIB_PATH))
        fh.flush()
        asm = subprocess.run([objdump, "-d", fh.name], capture_output=True, text=True, check=True).stdout
    for relu in ("Lb1E",):
        m = re.search(r"<_ZN3mpx18convw_f16x3_kernelILi256E%sEEvNS_10ConvParamsE>:\n(.*?)\n\n" % relu, asm, re.S)
        assert m, "kernel not found in the disassembly"
        ins = [l.split("//")[0].strip() for l in m.group(1).splitlines()]
        ins = [l for l in ins if l and not l.endswith(":")]
        mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma_f32_16x16x32_f16")]
        assert len(mf) == 384, len(mf)
        assert all(re.match(r"v_mfma_f32_16x16x32_f16 v\[\d+:\d+\], a\[\d+:\d+\], v\[\d+:\d+\]", ins[i]) for i in mf), "an MFMA without an AGPR A operand"
        loop = ins[mf[0]:mf[-1] + 1]
        assert not [l for l in loop if "v_accvgpr" in l], "AGPR copies in the K loop"
        assert not [l for l in loop if l.startswith("scratch_")], "scratch access in the K loop"
        # the column-major loop: stores, residual loads and LDS-DMA pieces between the MFMAs
        n_store = len([l for l in loop if l.startswith("buffer_store_dwordx4")])
        n_dma = len([l for l in loop if l.startswith("buffer_load_dwordx4") and l.endswith("lds")])
        n_load = len([l for l in loop if l.startswith("buffer_load_dwordx4") and not l.endswith("lds")])
        assert (n_store, n_load, n_dma) == (16, 16, 16), (n_store, n_load, n_dma)
        waits = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in loop if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert waits and min(waits) >= 4 and len(waits) <= 8, waits      # counted: traffic stays in flight behind every wait; none per MFMA gap
        gaps = [b - a - 1 for a, b in zip(mf, mf[1:])]
        # a gap holds one or two epilogue instructions, sometimes a fragment read or a memory instruction with its address arithmetic; the
        # descriptor set-up of a column start and the tile end (wait, barrier, first fragment read) are the few long ones
        assert sorted(gaps)[len(gaps) // 2] <= 3 and sum(1 for x in gaps if x > 12) <= 8, (sorted(gaps)[-10:], sorted(gaps)[len(gaps) // 2])
        # Round 6: the two hazards hipcc cannot see because the MFMAs are inline asm.
        # (1) the hand-written tile-end `s_waitcnt vmcnt(N)` in front of the barrier is what orders the NEXT tile's LDS-DMA pieces before
        # the barrier and the fragment reads: N must be exactly the number of VMEM instructions issued behind the last piece
        end = next(i for i in range(mf[-1], len(ins)) if ins[i].startswith("s_waitcnt") and "vmcnt" in ins[i] and ins[i + 1].startswith("s_barrier"))
        imm = int(re.search(r"vmcnt\((\d+)\)", ins[end]).group(1))
        last_dma = max(i for i in range(mf[0], end) if ins[i].startswith("buffer_load_dwordx4") and ins[i].endswith("lds"))
        behind = len([l for l in ins[last_dma + 1:end] if l.startswith("buffer_load") or l.startswith("buffer_store")])
        assert imm == behind == 24, (imm, behind)
        # (2) a VALU / memory instruction must not touch an accumulator quad within four MFMAs of the MFMA that wrote it last (the kernel's
        # schedule keeps five: the column's last MFMA -> barrier -> five MFMAs of the next column -> the first v_permlane16_swap of the slice)
        body = ins[mf[0]:end + 2] * 2                       # twice: the loop wraps from column 3 into column 0
        last_write, issued, worst = {}, 0, None
        for l in body:
            regs = []
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l):
                regs += list(range(int(a), int(b) + 1))
            regs += [int(a) for a in re.findall(r"\bv(\d+)\b", l)]
            if l.startswith("v_mfma"):
                d = re.match(r"v_mfma\S+ v\[(\d+):(\d+)\]", l)
                for r in range(int(d.group(1)), int(d.group(2)) + 1):
                    last_write[r] = issued
                issued += 1
                continue
            for r in regs:
                if r in last_write:
                    between = issued - last_write[r] - 1
                    if worst is None or between < worst[0]:
                        worst = (between, l)
        assert worst is not None and worst[0] >= 4, worst


def test_null_engine_calls_fail_cleanly(mpx_lib):
    assert mpx_lib.mpx_num_convs(None) == -1
    assert mpx_lib.mpx_forward(None, None, None, None, None, 1, None) == -1
    assert mpx_lib.mpx_destroy(None) == 0
    assert mpx_lib.mpx_last_error(None) == b"null engine"


def _desc(cin, cout, k):
    d = _lib.ConvDesc()
    d.cin, d.cout, d.ksize = cin, cout, k
    d.k_packed = 7 * 32 if cin == 3 else k * k * cin
    d.cout_pad = (cout + 127) // 128 * 128
    return d


def _row_major(plane, rows, K):
    """Undo the piece-major order of a packed weight plane (include/mpx.h, mpx_pack_conv_weights): element (row, k) sits at
    ((((row/16)*(K/32) + k/32)*16 + row%16)*4 + ((k/8)%4 ^ ((row%16)/8)*2))*8 + k%8."""
    row = np.arange(rows)[:, None]
    k = np.arange(K)[None, :]
    r = row % 16
    at = ((((row // 16) * (K // 32) + k // 32) * 16 + r) * 4 + (((k // 8) % 4) ^ ((r // 8) * 2))) * 8 + k % 8
    assert np.array_equal(np.sort(at.ravel()), np.arange(rows * K))          # a permutation of the plane
    return plane.ravel()[at]


def _pack(mpx_lib, d, w, bn, bias=None):
    hi = np.zeros((d.cout_pad, d.k_packed), dtype=np.uint16)
    lo = np.zeros_like(hi)
    sc = np.zeros(d.cout_pad, dtype=np.float32)
    sh = np.zeros(d.cout_pad, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    rc = mpx_lib.mpx_pack_conv_weights(C.byref(d), p(w), p(bias), p(bn[0]), p(bn[1]), p(bn[2]), p(bn[3]), 1e-5,
                                       p(hi), p(lo), p(sc), p(sh))
    assert rc == 0
    return _row_major(hi, d.cout_pad, d.k_packed).view(np.float16), _row_major(lo, d.cout_pad, d.k_packed).view(np.float16), sc, sh


@pytest.mark.parametrize("cin,cout,k", [(64, 64, 3), (256, 128, 1), (3, 64, 7)])
def test_pack_conv_weights(mpx_lib, cin, cout, k):
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
    w[3] = 0.0                                            # an all-zero output channel
    bn = [rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32),
          rng.standard_normal(cout).astype(np.float32), rng.uniform(0.5, 2, cout).astype(np.float32)]
    d = _desc(cin, cout, k)
    hi, lo, sc, sh = _pack(mpx_lib, d, w, bn)
    s = bn[0].astype(np.float64) / np.sqrt(bn[3].astype(np.float64) + 1e-5)
    np.testing.assert_allclose(sh[:cout], bn[1] - bn[2] * s, rtol=1e-6, atol=1e-7)
    assert (hi[cout:] == 0).all() and (sc[cout:] == 0).all()
    ratio = sc[:cout].astype(np.float64) / s               # = 2^-e, an exact power of two
    e = -np.log2(ratio)
    assert np.allclose(e, np.round(e), atol=1e-6) and round(e[3]) == 0
    if cin == 3:     # stem: k = ky*32 + px*4 + c, pads are zero
        wk = np.zeros((cout, 7, 8, 4), dtype=np.float32)
        wk[:, :, :7, :3] = w.transpose(0, 2, 3, 1)
        wk = wk.reshape(cout, -1)
    else:            # (ky,kx,ci), ci fastest
        wk = w.transpose(0, 2, 3, 1).reshape(cout, -1)
    scaled = wk.astype(np.float64) * (2.0 ** np.round(e))[:, None]
    amax = np.abs(scaled).max(1)
    assert ((amax[np.arange(cout) != 3] >= 512) & (amax[np.arange(cout) != 3] < 1024)).all()
    rec = hi[:cout].astype(np.float64) + lo[:cout].astype(np.float64)
    # split-fp16 keeps >= 21 bits of every weight relative to the channel maximum's exponent
    assert np.abs(rec - scaled).max() <= 1024 * 2.0 ** -21
    assert (hi[:cout] == scaled.astype(np.float32).astype(np.float16)).all()


def test_pack_fc(mpx_lib):
    rng = np.random.default_rng(1)
    w = rng.standard_normal((1000, 512)).astype(np.float32)
    b = rng.standard_normal(1000).astype(np.float32)
    d = _desc(512, 1000, 1)
    hi, lo, sc, sh = _pack(mpx_lib, d, w, [None, b, None, None])
    assert (sh[:1000] == b).all() and (sh[1000:] == 0).all()
    rec = (hi[:1000].astype(np.float64) + lo[:1000]) * sc[:1000, None]
    np.testing.assert_allclose(rec, w, rtol=0, atol=2e-6)


def test_pack_rejects_bad_desc(mpx_lib):
    d = _desc(64, 64, 3)
    d.k_packed = 100
    w = np.zeros((64, 64, 3, 3), dtype=np.float32)
    z = np.zeros(64, dtype=np.float32)
    buf = np.zeros((128, 576), dtype=np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert mpx_lib.mpx_pack_conv_weights(C.byref(d), p(w), None, p(z), p(z), p(z), p(z), 1e-5, p(buf), p(buf), p(z), p(z)) == -1


def test_pack_padded_channels_and_conv_bias(mpx_lib):
    """The small networks' layers: input planes carry cin_pad = 32 channels per pixel for a 16- (or 1-) channel layer, and
    the MNIST net's convs have their own bias in front of the BatchNorm: shift = beta + (bias - mean) * s."""
    rng = np.random.default_rng(4)
    cin, cout, k, cin_pad = 16, 16, 3, 32
    w = (rng.standard_normal((cout, cin, k, k)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    bn = [rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32),
          rng.standard_normal(cout).astype(np.float32), rng.uniform(0.5, 2, cout).astype(np.float32)]
    d = _lib.ConvDesc()
    d.cin, d.cout, d.ksize = cin, cout, k
    d.k_packed = k * k * cin_pad
    d.cout_pad = 128
    hi, lo, sc, sh = _pack(mpx_lib, d, w, bn, bias)
    s = bn[0].astype(np.float64) / np.sqrt(bn[3].astype(np.float64) + 1e-5)
    np.testing.assert_allclose(sh[:cout], bn[1] + (bias.astype(np.float64) - bn[2]) * s, rtol=1e-6, atol=1e-7)
    planes = (hi[:cout].astype(np.float64) + lo[:cout]).reshape(cout, k * k, cin_pad) * sc[:cout, None, None].astype(np.float64) / s[:, None, None]
    np.testing.assert_allclose(planes[:, :, :cin], w.transpose(0, 2, 3, 1).reshape(cout, k * k, cin), rtol=0, atol=1e-6)
    assert (planes[:, :, cin:] == 0).all() and (hi[cout:] == 0).all() and (sc[cout:] == 0).all()
    d.k_packed = k * k * 8                             # fewer plane channels than the layer reads
    buf = np.zeros((128, k * k * 32), dtype=np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert mpx_lib.mpx_pack_conv_weights(C.byref(d), p(w), None, p(bn[0]), p(bn[1]), p(bn[2]), p(bn[3]), 1e-5, p(buf), p(buf), p(sc), p(sh)) == -1


def test_rank_segments():
    seg = np.full((224, 224), 40, dtype=np.int64)
    seg[:10] = 7
    seg[100:, 50:] = 1000
    rank, s = rank_segments(seg)
    assert s == 3 and rank.dtype == np.int32
    assert rank[0, 0] == 0 and rank[50, 0] == 1 and rank[150, 100] == 2
    with pytest.raises(ValueError):
        rank_segments(np.zeros((10, 10), dtype=np.int32))
    with pytest.raises(ValueError):
        rank_segments(np.zeros((224, 224), dtype=np.float32))
    # the histogram path (small non-negative labels), the already-ranked path and the np.unique fallback (negative / huge labels)
    # all give np.unique's ranks (generate_gp_training_data_imagenet.py:223,230 index superpixels through np.unique(segments))
    rng = np.random.default_rng(0)
    for trial in range(12):
        k = int(rng.integers(1, 400))
        pool = np.arange(-5, 900) if trial % 4 == 0 else (np.arange(0, 10**9, 997) if trial % 4 == 1 else np.arange(0, 5000))
        labels = rng.choice(pool, size=k, replace=False)
        seg = labels[rng.integers(0, k, size=(224, 224))].astype(np.int64 if trial % 2 else np.int32)
        rank, s = rank_segments(seg)
        uniq, inv = np.unique(seg, return_inverse=True)
        assert s == len(uniq) and rank.dtype == np.int32 and rank.flags.c_contiguous and np.array_equal(rank, inv.reshape(224, 224))
        again, s2 = rank_segments(rank)
        assert s2 == s and np.array_equal(again, rank)


def test_masks_module():
    assert masks.window_size(196) == 78 and masks.bo_upper_bound(196) == 117
    assert masks.window_onoff(10, 8).tolist() == [0] * 8 + [1, 1]          # truncated at the end
    assert masks.windows_onoff(10, [0, 3]).shape == (2, 10)
    draws = masks.draw_first_indices(46, 500, random.Random(3))
    assert min(draws) >= 1 and max(draws) <= 46 - 18
    with pytest.raises(ValueError):
        masks.draw_first_indices(0, 1)


def test_shard_blocks_and_ranges():
    for total, world in [(10, 3), (65536 * 8, 8), (7, 8), (512, 1)]:
        blocks = [shard.block(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
    assert shard.image_ranges(5, 25, 10) == [(0, 5, 10), (1, 0, 10), (2, 0, 5)]
    assert shard.image_ranges(20, 20, 10) == []
    with pytest.raises(ValueError):
        shard.block(10, 3, 3)


class FakeEngine:
    """Engine stand-in for API plumbing tests: scores through the CPU oracle."""

    def __init__(self, arch="resnet18"):
        self.arch, self.sd, self.calls = arch, synth.make_state_dict(arch), 0

    def predict(self, image):
        x = torch.as_tensor(image)
        with torch.no_grad():
            logits = resnet_ref.forward(self.sd, x[None], self.arch)
        return int(logits.argmax()), torch.softmax(logits, 1)[0].numpy()

    def score_masks(self, image, segments, onoff, label, stem=None):
        self.calls += 1
        self.stems = getattr(self, "stems", []) + [stem]
        s, p = scorer.score_masks_batched(self.sd, self.arch, torch.as_tensor(image), segments, onoff, label, chunk=16)
        return onoff, s.astype(np.float32), p.astype(np.int32)

    def heatmap(self, seg_rank, onoff, pred, label):
        """The test double's own K5 (the real engine runs mpx_heatmap_accumulate)."""
        return scorer.summed_superpixel_labels(seg_rank, onoff, np.asarray(pred) == label)

    max_batch = 24          # smaller than one coarse image's table + the next one's: groups of two images get packed

    def score_images(self, images, segments, onoffs, labels):
        """The packed entry (the real engine shares forward batches between the images; the scores are per row either way)."""
        self.packed = getattr(self, "packed", 0) + 1
        out = []
        for im, sg, oo, lb in zip(images, segments, onoffs, labels):
            s, p = scorer.score_masks_batched(self.sd, self.arch, torch.as_tensor(im), sg, oo, lb, chunk=16)
            out.append((s.astype(np.float32), p.astype(np.int32)))
        return out


@pytest.fixture()
def coarse_setup(tmp_path):
    eng = FakeEngine()
    x = scorer.to_tensor_normalize(synth.make_images(1)[0])
    label, _ = eng.predict(x)
    loader = [(torch.zeros(1, 3, 224, 224), torch.tensor([0])), (x[None], torch.tensor([label]))]
    seg = synth.grid_segments(block=56)               # 16 superpixels -> 17 window starts, k = 6
    api.configure(eval_img_index=2, num_mask_samples=12, segmenter=lambda img: seg, mask_dir=str(tmp_path / "masks"), seed=5)
    yield eng, loader, seg, label, tmp_path / "masks"
    api.configure(eval_img_index=1, num_mask_samples=100, segmenter=None, mask_dir=None, seed=None)


def test_api_drop_in_signatures(coarse_setup):
    eng, loader, seg, label, mask_dir = coarse_setup
    criterion = object()
    y = api.sample_loss([3.7], loader, eng, criterion)           # L-BFGS-B hands floats: int() truncates
    assert isinstance(y, np.float32)
    y2 = api.validate_nueral_network(loader, eng, criterion, 3.7, 3)
    assert y == y2 and eng.calls == 1                            # served from the one-pass table
    ref, _ = scorer.score_masks_reference_loop(eng.sd, eng.arch, loader[1][0][0], seg, scorer.window_onoff(16, 3)[None], label)
    assert abs(float(y) - float(ref[0])) < 1e-6
    m = api.superpixel_mask(3)
    assert m.dtype == np.uint8 and set(np.unique(m)) <= {0, 255}
    assert (m == scorer.window_mask_u8(seg, 3) * 255).all()
    names = sorted(os.listdir(mask_dir))
    assert names and all(re.fullmatch(r"mask_3\.7_[01]\.png", n) for n in names)
    from PIL import Image
    assert (np.array(Image.open(mask_dir / names[0])) == m).all()


def test_api_bo_style_caller(coarse_setup):
    """bayesian_optimisation(n_iters, sample_loss, val_loader, nn_model, criterion, bounds, n_pre_samples)
    calls sample_loss(params, val_loader, nn_model, criterion) (BayesianOptimization.py:137-144,182)."""
    eng, loader, _seg, _label, _ = coarse_setup
    ub = masks.bo_upper_bound(16)
    rng = random.Random(0)
    xs, ys = [], []
    for _ in range(5):
        params = [rng.randint(0, ub)]
        xs.append(params)
        ys.append(api.sample_loss(params, loader, eng, None))
    xp, yp = np.array(xs), np.array(ys)
    assert xp.shape == (5, 1) and yp.shape == (5,) and eng.calls == 1


def test_bo_loop_end_to_end(coarse_setup):
    """BASELINE config 5 plumbing on the CPU: the package's BO loop (same signature as
    BayesianOptimization.py:99-100) drives api.sample_loss; the engine is asked for scores exactly once."""
    from network_interpretation_imagenet_amd import bo
    eng, loader, _seg, _label, _ = coarse_setup
    ub = masks.bo_upper_bound(16)
    xp, yp = bo.bayesian_optimisation(n_iters=4, sample_loss=api.sample_loss, val_loader=loader, nn_model=eng,
                                      criterion=None, bounds=np.array([[0, ub]]), n_pre_samples=3,
                                      rng=random.Random(1))
    assert xp.shape == (7, 1) and yp.shape == (7,) and eng.calls == 1
    assert ((xp >= 0) & (xp <= ub)).all() and np.isfinite(yp).all()
    table = api._LAST["session"].table()[0]
    assert all(abs(yp[i] - table[int(xp[i, 0])]) < 1e-7 for i in range(7))
    assert len({float(v) for v in xp[:, 0]}) >= 4                         # the loop explores, duplicates are re-drawn


def test_api_validate_generators(coarse_setup):
    eng, loader, seg, label, mask_dir = coarse_setup
    n_ok = api.validate(loader, eng, None, 2)
    assert isinstance(n_ok, int) and 0 <= n_ok <= 12
    files = sorted(os.listdir(mask_dir))
    assert len(files) == 12 and sum(f.endswith("_1.png") for f in files) == n_ok
    summed, heatmap = api.validate_gp_superpixel(loader, eng, None, 2)     # the 2-tuple gp_superpixel...:350,617 unpacks
    assert summed.shape == (224, 224) and summed.dtype == np.float64
    assert heatmap.shape == (224, 224, 3) and heatmap.dtype == np.uint8
    assert (heatmap == api.jet_heatmap_u8(summed)).all()
    # the picture: min-max -> u8 (truncating) -> JET; literal rescale on a non-constant map + the LUT's anchor colours
    demo = np.add.outer(np.arange(224.0), 3.0 * np.arange(224.0))
    show = demo.copy()[:, :, None]
    show -= show.min()
    show /= show.max()
    show *= 255
    show = show.astype(np.uint8)[:, :, 0]
    pic = api.jet_heatmap_u8(demo)
    assert (pic[show == 0] == (128, 0, 0)).all() and (pic[show == 255] == (0, 0, 128)).all()     # BGR: dark blue .. dark red
    lut = api.jet_heatmap_u8(np.arange(256, dtype=np.float64).reshape(16, 16)).reshape(256, 3)
    assert (lut[32] == (255, 0, 0)).all() and (lut[128] == (126, 255, 130)).all() and (lut[223] == (0, 0, 255)).all()
    assert (lut[:, 1].argmax() == 96) and (np.diff(lut[:33, 0].astype(int)) >= 0).all()      # green saturates at 96; blue ramps up first
    assert (pic == lut[show]).all()
    assert (api.jet_heatmap_u8(np.zeros((4, 4))) == (128, 0, 0)).all()     # constant map: no 0/0
    # same seed -> same draws: compare with the oracle's literal accumulation
    firsts = masks.draw_first_indices(16, 12, random.Random(5))
    onoff = masks.windows_onoff(16, firsts)
    _s, pred = scorer.score_masks_batched(eng.sd, eng.arch, loader[1][0][0], seg, onoff, label)
    assert (summed == scorer.summed_superpixel_labels(seg, onoff, pred == label)).all()
    assert int((pred == label).sum()) == n_ok


def test_api_png_wire_format_round_trip(coarse_setup):
    """validate() writes mask_{i}_{label}.png; the consumer side (gp_regression.py:63-104,
    generate_gp_training_data_imagenet.py:490-516) reads them back into the same heat map validate_summed builds."""
    eng, loader, _seg, _label, mask_dir = coarse_setup
    n_ok = api.validate(loader, eng, None, 2)
    summed, _pic = api.validate_summed(loader, eng, None, 2)         # same seed -> same draws
    files, labels = api.load_images_from_folder(str(mask_dir))
    assert len(files) == 12 and sorted(set(labels)) <= ["0", "1"] and labels.count("1") == n_ok
    assert (api.summed_heatmap_from_folder(str(mask_dir)) == summed).all()
    d = api.get_pixel_sorted_mask_label(str(mask_dir))
    train_x, train_y = api.prepare_training_data(str(mask_dir))
    assert train_x.dtype == torch.float32 and train_x.shape == (len(d), 2) and train_y.shape == (len(d),)
    xs = train_x.numpy().astype(int)
    assert (np.diff(xs[:, 0] * 224 + xs[:, 1]) > 0).all()             # raster order, each pixel once
    assert all(d[(r, c)] == int(y) == int(summed[r, c]) for (r, c), y in zip(xs[::97].tolist(), train_y.numpy()[::97]))
    assert all(summed[r, c] == 0 for r, c in [(0, 0), (223, 223)] if (r, c) not in d)


def test_api_validate_summed_many_pipelines_segmentation(coarse_setup):
    eng, loader, seg, label, _ = coarse_setup
    loader3 = loader + [loader[1]]
    one, _pic = api.validate_summed(loader3, eng, None, 2, rng=random.Random(9))
    many = api.validate_summed_many(loader3, eng, None, [2], rng=random.Random(9), workers=2)
    assert list(many) == [2] and (many[2] == one).all()
    r = random.Random(9)
    both = api.validate_summed_many(loader3, eng, None, [3, 2], rng=r, workers=2, lookahead=1)
    assert sorted(both) == [2, 3] and (both[2] == one).all() and both[3].shape == (224, 224)
    bad = [loader[0], (loader[1][0], torch.tensor([(label + 1) % 1000]))]
    assert api.validate_summed_many(bad, eng, None, [2], workers=1) == {2: None}
    assert api.validate_gp_superpixel(bad, eng, None, 2) is None           # "wrong prediction": falls off the end upstream
    assert api.validate_summed_many(loader, eng, None, []) == {}


def test_fill_tables_and_validate_many(coarse_setup):
    """api.fill_tables scores the unmasked row + every window of several sessions through ONE packed engine call and leaves the
    tables SaliencySession.table() would compute one image at a time; validate_many = validate() per image, same draws."""
    eng, loader, seg, label, mask_dir = coarse_setup
    x = loader[1][0]
    a = api.SaliencySession(eng, x, label, segments=seg, check_base=False)
    b = api.SaliencySession(eng, x, (label + 1) % 1000, segments=seg, check_base=False)
    single = api.SaliencySession(eng, x, label, segments=seg)
    assert api.fill_tables(eng, [a, b]) == [True, False] and eng.packed == 1
    assert a.base_pred == label and np.abs(a.table()[0] - single.table()[0]).max() < 1e-6 and (a.table()[1] == single.table()[1]).all()   # (torch-CPU batches of another size: not bit-equal; the engine's are -- GPU test)
    assert len(a.table()[0]) == 17 and api.fill_tables(eng, [a]) == [True] and eng.packed == 1       # cached: no second pass
    loader3 = loader + [loader[1]]
    want = api.validate(loader3, eng, None, 2, rng=random.Random(11))
    api.configure(eval_img_index=2, num_mask_samples=12, segmenter=lambda img: seg, mask_dir=None, seed=5)
    r = random.Random(11)
    many = api.validate_many(loader3, eng, None, [2, 3], rng=r, workers=2)
    assert many[2] == want and isinstance(many[3], int) and 0 <= many[3] <= 12
    api.configure(eval_img_index=2, num_mask_samples=12, segmenter=lambda img: seg, mask_dir=str(mask_dir), seed=5)
    api.validate_many(loader3, eng, None, [3], rng=random.Random(1), workers=1)
    files, labels = api.load_images_from_folder(str(mask_dir / "img_3"))
    assert len(files) == 12 and set(labels) <= {"0", "1"}
    bad = [loader[0], (loader[1][0], torch.tensor([(label + 1) % 1000]))]
    assert api.validate_many(bad, eng, None, [2], workers=1) == {2: None}


def test_session_cache_checks_identity_and_is_bounded():
    """The session cache compares model and loader by identity on strong references (an id() key can be reused by a
    new object) and keeps a bounded number of entries."""
    eng = FakeEngine()
    x = scorer.to_tensor_normalize(synth.make_images(1)[0])
    label, _ = eng.predict(x)
    seg = synth.grid_segments(block=112)              # 4 superpixels
    api.configure(eval_img_index=1, segmenter=lambda img: seg, mask_dir=None, seed=None)
    try:
        loaders = [[(x[None], torch.tensor([label]))] for _ in range(api._MAX_SESSIONS + 3)]
        first = api._session(loaders[0], eng, 1)
        assert api._session(loaders[0], eng, 1) is first                   # same objects: cached
        assert api._session(list(loaders[0]), eng, 1) is not first         # an equal but different loader: new session
        for l in loaders[1:]:
            api._session(l, eng, 1)
        assert len(api._SESSIONS) == api._MAX_SESSIONS
        assert all(e[0] is eng for e in api._SESSIONS)
    finally:
        api.configure(eval_img_index=1, num_mask_samples=100, segmenter=None, mask_dir=None, seed=None)


def test_default_segmenter_is_the_native_front_end():
    from network_interpretation_imagenet_amd import segment
    x = scorer.to_tensor_normalize(synth.make_images(1)[0])
    pic = api.img_show_u8(x.numpy())
    assert (pic == segment.minmax_u8(x.numpy())).all()
    seg = api.default_segmenter(pic)
    assert seg.dtype == np.int64 and seg.shape == (224, 224) and (seg == segment.felzenszwalb(pic)).all()
    s = api.SaliencySession(FakeEngine(), x, 0, check_base=False)      # segmentation only: no forward is run
    assert s.num_segments == int(seg.max()) + 1 and s.window == int(0.4 * s.num_segments)


def test_native_segmenter_reproduces_committed_skimage_segments():
    """tests/golden/segments_blobs.npz holds scikit-image's label maps of img_show of the seeded blob images
    (make_segments.py); the native front-end must give the same maps from the same tensors."""
    from network_interpretation_imagenet_amd import segment
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "segments_blobs.npz"))
    imgs = synth.make_images(2, seed=int(gold["image_seed"]), kind="blobs")
    for im, want in zip(imgs, gold["segments"]):
        pic = api.img_show_u8(scorer.to_tensor_normalize(im).numpy())
        assert (segment.felzenszwalb(pic) == want).all()


def test_api_wrong_base_prediction(coarse_setup):
    eng, loader, _seg, label, _ = coarse_setup
    bad = [loader[0], (loader[1][0], torch.tensor([(label + 1) % 1000]))]
    with pytest.raises(api.BasePredictionWrong):
        api.sample_loss([1], bad, eng, None)
    assert api.validate(bad, eng, None, 2) == 0
    assert api.validate(loader, eng, None, 5) == 0                # index past the end of the loader


# ------------------------------------------------------------------------------------------------
# the small networks' sampler and mask pictures
# ------------------------------------------------------------------------------------------------
def test_draw_removed_sets_support_and_stream():
    """masks.draw_removed_sets = random.sample(range(uniq[0], uniq[-1]), k) per mask (generate_gp_training_data_cifar.py:308,
    generate_gp_training_data_mnist.py:215): the last superpixel is never drawn, the values of a set are distinct, a seeded
    generator gives upstream's own stream (with the MNIST script's unused randint consumed when asked)."""
    from network_interpretation_imagenet_amd import masks
    uniq = np.arange(3, 15)                              # labels 3 .. 14
    sets = masks.draw_removed_sets(uniq, 5, 400, random.Random(1))
    flat = np.array(sets)
    assert flat.shape == (400, 5) and flat.min() == 3 and flat.max() == 13          # 14, the last label, never
    assert all(len(set(r)) == 5 for r in sets) and set(flat.ravel()) == set(range(3, 14))
    r1, r2 = random.Random(7), random.Random(7)
    assert masks.draw_removed_sets(uniq, 5, 10, r1) == [r2.sample(range(3, 14), 5) for _ in range(10)]
    r1, r2 = random.Random(7), random.Random(7)
    want = []
    for _ in range(10):
        r2.randint(1, len(uniq) - 1)                     # firstIndex = randint(1, S - 1), drawn and unused upstream
        want.append(r2.sample(range(3, 14), 1))
    assert masks.draw_removed_sets(uniq, 1, 10, r1, burn_window_draw=True) == want
    with pytest.raises(ValueError):
        masks.draw_removed_sets(np.arange(4), 5, 1, random.Random(0))               # population of 3 < k, as random.sample


def test_removed_onoff_and_mask_picture_match_the_oracle_and_the_fixture(golden_dir):
    from network_interpretation_imagenet_amd import masks
    from oracle import smallnets_ref
    for arch, k, seed in (("mnist_net", 1, 11), ("cifar_resnet56", 5, 12)):
        g = np.load(os.path.join(golden_dir, "smallnet_%s.npz" % arch))
        rnd = random.Random(seed)                        # the generator script's stream: one generator over both pictures
        for i in range(int(g["n_pictures"])):
            seg = g["pic%d/segments" % i]
            uniq = np.unique(seg)
            sets = [sorted(r) for r in masks.draw_removed_sets(uniq, min(k, len(uniq) - 1), 24, rnd)]
            sets[0] = []
            onoff = masks.removed_onoff(uniq, sets)
            assert np.array_equal(onoff, g["pic%d/removed" % i])                    # the committed fixture was drawn by this sampler
            assert np.array_equal(onoff, smallnets_ref.removed_onoff(seg, sets))
            for r in sets[:4]:
                m = masks.removed_mask_u8(seg, r)
                assert m.dtype == np.uint8 and np.array_equal(m, smallnets_ref.removed_mask_u8(seg, r))
    gaps = np.array([[2, 2, 7], [7, 9, 9]])              # labels with gaps: a drawn value may name no superpixel
    assert masks.removed_onoff(np.unique(gaps), [[3, 7]]).tolist() == [[0, 1, 0]]
    assert masks.removed_mask_u8(gaps, [3, 7]).tolist() == [[255, 255, 0], [0, 255, 255]]


def test_jet_lut_bytes_are_pinned():
    """api.jet_heatmap_u8's LUT (parity unpinned against OpenCV's table, which is absent here: INTEGRATION.md): its own bytes are
    pinned at the end points and the four breakpoints of the piecewise-linear jet, so that an edit cannot drift silently."""
    ramp = np.arange(256, dtype=np.float64).reshape(16, 16)
    bgr = api.jet_heatmap_u8(ramp).reshape(256, 3)
    want = {0: (128, 0, 0), 31: (252, 0, 0), 32: (255, 0, 0), 95: (255, 252, 0), 96: (254, 255, 1), 159: (1, 255, 254), 160: (0, 252, 255),
            223: (0, 0, 255), 224: (0, 0, 252), 255: (0, 0, 128)}
    got = {i: tuple(int(v) for v in bgr[i]) for i in want}
    assert got == want, got
    assert (np.diff(bgr[:, 2].astype(int))[:223] >= 0).all() and (np.diff(bgr[:, 0].astype(int))[32:] <= 0).all()      # R rises, B falls


def test_whole_round_batch():
    """engine.whole_round_batch: the largest forward batch whose 14x14 maps cut into WHOLE rounds of 256-pixel tiles over 256 CUs
    (the one-workgroup-per-CU kernels pay a full extra round for a batch just above one)."""
    from network_interpretation_imagenet_amd.engine import whole_round_batch
    assert whole_round_batch(2048) == 2006 and whole_round_batch(2400) == 2340 and whole_round_batch(2340) == 2340
    assert whole_round_batch(334) == 334 and whole_round_batch(335) == 334 and whole_round_batch(100) == 100 and whole_round_batch(1) == 1
    for limit in range(335, 5000, 37):
        b = whole_round_batch(limit)
        tiles = -(-b * 196 // 256)                       # 256-pixel tiles of the batch's 14x14 maps
        assert b <= limit and tiles % 256 == 0           # whole rounds ...
        assert -(-(b + 1) * 196 // 256) > tiles or whole_round_batch(limit) == b      # ... and one more image would start a new tile
        nxt = (tiles // 256 + 1) * 256 * 256 // 196
        assert nxt > limit                               # the next whole-round batch does not fit the limit
    # another CU count (a partitioned or CU-masked device; the engine reports its own through mpx_num_cus): rounds of num_cus tiles
    assert whole_round_batch(2400, num_cus=256) == 2340 and whole_round_batch(2400, num_cus=128) == 2340      # 14 rounds of 128
    for cus in (64, 120, 128, 304):
        b = whole_round_batch(2400, num_cus=cus)
        assert b <= 2400 and -(-b * 196 // 256) % cus == 0


def test_bench_traffic_lookup_prefers_the_benched_batch():
    """bench.pmc_traffic: the PMC file of the arch taken at the benched forward batch wins; another batch is scaled and says so."""
    import bench
    total, src = bench.pmc_traffic("resnet101", 2048)
    assert total and src.endswith(".json") and "scaled" not in src
    t2, src2 = bench.pmc_traffic("resnet101", 1024)
    assert "scaled by 1024/" in src2 and t2 < total
    assert bench.pmc_traffic("no_such_arch", 2048) == (None, None)


class _StagingProbe(MaskedForwardEngine):
    """The real engine's host logic (score_packed / score_images / stem_for_rows) over recorded kernel calls: no library, no GPU.
    A staged slot remembers (image id, row, kind); forward() writes image * 1000 + row as the score and the kind as the prediction."""

    def __init__(self, max_batch=512, stem="table"):
        self.arch, self.small, self.stem, self.stem_table_min_rows = "resnet101", False, stem, 256
        self.max_batch, self.device = max_batch, torch.device("cpu")
        self.slots = [None] * max_batch
        self.forwards = []          # [(batch, set of kinds)]
        self.builds = []
        self._cur = None

    def __del__(self):
        pass

    def build_stem_table(self, image, seg, num_segments):
        self._cur = int(image[0, 0, 0])
        self.builds.append(self._cur)

    def apply_stem_table(self, onoff, slot0=0):
        for j in range(onoff.shape[0]):
            self.slots[slot0 + j] = (self._cur, int(onoff[j, 0]) + 256 * int(onoff[j, 1]), "table")

    def stage_masks(self, image, seg, onoff, slot0=0, out_f32=None):
        for j in range(onoff.shape[0]):
            self.slots[slot0 + j] = (int(image[0, 0, 0]), int(onoff[j, 0]) + 256 * int(onoff[j, 1]), "conv")

    def forward(self, batch, labels, want_logits=False, score_out=None, pred_out=None):
        kinds = {self.slots[j][2] for j in range(batch)}
        self.forwards.append((batch, kinds))
        for j in range(batch):
            img, row, kind = self.slots[j]
            score_out[j] = img * 1000 + row
            pred_out[j] = 2 if kind == "table" else 1
            assert int(labels[j]) == img + 5
        return score_out, pred_out


def _probe_job(sizes):
    """images i = 0.. with sizes[i] rows over an 8-segment... label map of 4 segments; row r of an image carries r in its first two bytes."""
    seg = np.repeat(np.arange(4, dtype=np.int32), 56)[:, None].repeat(224, 1)
    images, onoffs = [], []
    for i, m in enumerate(sizes):
        im = np.zeros((224, 224, 3), dtype=np.uint8)
        im[0, 0, 0] = i
        images.append(im)
        o = np.zeros((m, 4), dtype=np.uint8)
        o[:, 0] = np.arange(m) % 256
        o[:, 1] = np.arange(m) // 256
        onoffs.append(o)
    return images, [seg] * len(sizes), onoffs, [i + 5 for i in range(len(sizes))]


def test_staging_is_decided_per_image_however_rows_are_packed():
    """VERDICT r5 item 1 / ADVICE r5 (medium): the stem table and K0 + the MFMA stem round differently, so an image must get the SAME staging
    alone and packed with others.  score_packed decides from each image's own row count (stem_for_rows(M_i)), never from the call's average,
    and no forward mixes the two kinds (mpx_forward refuses that); score_images groups the images by kind and hands the results back in
    the caller's order."""
    eng = _StagingProbe(max_batch=512)
    assert [eng.stem_for_rows(r) for r in (1, 62, 255, 256, 302, 5000)] == ["conv", "conv", "conv", "table", "table", "table"]
    assert _StagingProbe(stem="conv").stem_for_rows(4096) == "conv"
    sizes = [302, 62, 0, 300, 40, 256, 255, 700]
    want_kind = [2 if m >= 256 else 1 for m in sizes]
    out = eng.score_images(*_probe_job(sizes))
    assert [len(s) for s, _p in out] == sizes
    for i, (score, pred) in enumerate(out):
        assert np.array_equal(score, i * 1000 + np.arange(sizes[i], dtype=np.float32))          # every row back where it belongs
        assert (pred == want_kind[i]).all()                                                       # staged by ITS OWN rows: 302 + 62 -> table + conv
    assert all(len(k) == 1 for _b, k in eng.forwards)                                             # one forward, one kind
    assert sorted(eng.builds) == [0, 3, 5, 7]                                                     # one table per >= 256-row image
    # grouped by kind: 1558 table rows = 3 full batches + 22, then 357 conv rows: ONE flush for the change of kind
    assert [b for b, _k in eng.forwards] == [512, 512, 512, 22, 357]
    # the same image alone: the same kind (what SaliencySession / api.validate do)
    for i, m in enumerate(sizes):
        if m:
            im, sg, oo, lb = _probe_job(sizes)
            alone = _StagingProbe().score_images([im[i]], [sg[i]], [oo[i]], [lb[i]])[0]
            assert np.array_equal(alone[1], out[i][1])
    # score_packed in the caller's order flushes where the kind changes; an explicit `stem` overrides every image
    eng2 = _StagingProbe(max_batch=512)
    im, sg, oo, lb = _probe_job([300, 60, 300])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    rows = torch.from_numpy(np.concatenate([np.full(len(o), l, dtype=np.int32) for o, l in zip(oo, lb)]))
    score, pred = torch.empty(660), torch.empty(660, dtype=torch.int32)
    eng2.score_packed([t(a) for a in im], t(sg[0]), [t(o) for o in oo], rows, score, pred)
    assert [(b, sorted(k)) for b, k in eng2.forwards] == [(300, ["table"]), (60, ["conv"]), (300, ["table"])]
    assert pred.tolist() == [2] * 300 + [1] * 60 + [2] * 300
    eng3 = _StagingProbe(max_batch=512)
    eng3.score_packed([t(a) for a in im], t(sg[0]), [t(o) for o in oo], rows, score, pred, stem="conv")
    assert [(b, sorted(k)) for b, k in eng3.forwards] == [(512, ["conv"]), (148, ["conv"])] and eng3.builds == []
    with pytest.raises(ValueError):
        eng3.score_packed([t(a) for a in im], t(sg[0]), [t(o) for o in oo], rows, score, pred, stem="tabel")


def test_session_fixes_one_staging_for_everything_it_scores():
    """SaliencySession stages by S + 2 rows whatever it is asked: the table with and without the unmasked row, a window outside the
    table, fill_tables' packed call -- S = 254 (255 / 256 rows) used to land on both sides of the 256-row line."""
    class Rec:
        max_batch = 512

        def __init__(self):
            self.calls = []

        def stem_for_rows(self, rows):
            return "table" if rows >= 256 else "conv"

        def predict(self, image):
            return 3, None

        def score_masks(self, image, segments, onoff, label, stem=None):
            self.calls.append((len(onoff), stem))
            return onoff, np.zeros(len(onoff), np.float32), np.full(len(onoff), 3, np.int32)

    x = torch.zeros(1, 3, 224, 224)
    for s_count, kind in ((254, "table"), (253, "conv"), (60, "conv"), (300, "table")):
        seg = (np.arange(224 * 224) % s_count).reshape(224, 224).astype(np.int32)
        eng = Rec()
        a = api.SaliencySession(eng, x, 3, segments=seg)                         # S + 2 rows
        b = api.SaliencySession(eng, x, 3, segments=seg, check_base=False)
        b.table()                                                               # S + 1 rows
        b.score(s_count + 7)                                                    # one row, outside the table
        a.score_windows([1, 2, 3])
        assert a.stem == b.stem == kind
        assert eng.calls == [(s_count + 2, kind), (s_count + 1, kind), (1, kind), (3, kind)]


def test_bound_library_record_names_the_product_build(mpx_lib):
    """What bench.py writes into config: the path, sha256 and build stamp of the library the process bound; the stamp
    of the in-tree build equals the hash of this tree's sources + flags, and a probe build is recognised as one."""
    import hashlib
    import __graft_entry__ as g
    rec = _lib.bound_library()
    assert rec["product_library"] and not _lib.is_probe_build()
    assert os.path.samefile(rec["lib_path"], os.path.join(ROOT, "network_interpretation_imagenet_amd", "libmpx.so"))
    with open(rec["lib_path"], "rb") as fh:
        assert rec["lib_sha256"] == hashlib.sha256(fh.read()).hexdigest()
    assert rec["lib_stamp"] == g._source_hash(g.lib_sources(), g.HIPCC_FLAGS)
    saved = _lib.LIB_PATH
    try:
        _lib.LIB_PATH = os.path.join(ROOT, "network_interpretation_imagenet_amd", "libmpxseg.so")       # any other file
        assert _lib.is_probe_build() and not _lib.bound_library()["product_library"]
    finally:
        _lib.LIB_PATH = saved
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"custom (probe library)"' in src and "lib_stamp_matches_tree" in src


class _KindEngine:
    """An engine double WITH the two stagings (stem_for_rows) and deterministic fake scores that depend on the image, the mask row and the
    label only -- never on how rows were grouped -- for the plumbing of api._many."""
    max_batch = 400

    def __init__(self):
        self.packed_calls = []          # per score_images call: the set of kinds of its images
        self.single_calls = 0

    def stem_for_rows(self, rows):
        return "table" if rows >= 256 else "conv"

    @staticmethod
    def _rows(image, onoff, label):
        tag = int(round(float(np.asarray(image).reshape(-1)[0]) * 1000))
        w = (np.arange(onoff.shape[1]) * 7 + tag) % 13
        h = (onoff.astype(np.int64) @ w + tag) % 11
        score = (h.astype(np.float32) + 1) / 12
        full = onoff.all(axis=1)
        pred = np.where(full | (h % 3 != 0), 5 + tag % 3, 999).astype(np.int32)      # the unmasked row predicts class 5 + tag % 3
        return score, pred

    def predict(self, image):
        tag = int(round(float(np.asarray(image).reshape(-1)[0]) * 1000))
        return 5 + tag % 3, None

    def score_masks(self, image, segments, onoff, label, stem=None):
        self.single_calls += 1
        assert stem == self.stem_for_rows(int(np.asarray(segments).max()) + 3)      # the session's kind: S + 2 rows
        s, p = self._rows(image, onoff, label)
        return onoff, s, p

    def score_images(self, images, segments, onoffs, labels, stem=None):
        self.packed_calls.append({self.stem_for_rows(len(o)) for o in onoffs})
        return [self._rows(im, o, lb) for im, o, lb in zip(images, onoffs, labels)]

    def heatmap(self, seg_rank, onoff, pred, label):
        return scorer.summed_superpixel_labels(seg_rank, onoff, np.asarray(pred) == label)


def test_validate_many_groups_by_staging_and_emits_in_loader_order():
    """api._many keeps one group of waiting sessions per kind of staging (a forward takes one kind) and still consumes the window draws
    image after image in LOADER order: validate_many == validate() per image with one shared random stream, whatever the grouping; an image
    whose unmasked prediction is wrong draws nothing, as upstream (generate_gp_training_data_imagenet.py:215,269-273)."""
    maps = {}
    sizes = [300, 60, 300, 40, 60, 300, 280, 50]
    loader = []
    for i, s_count in enumerate(sizes):
        x = torch.zeros(1, 3, 224, 224)
        x[0, 0, 0, 0] = (i + 1) / 1000.0            # the double's image tag
        maps[i + 1] = (np.arange(224 * 224, dtype=np.int64) * s_count // (224 * 224)).reshape(224, 224).astype(np.int32)
        good = i != 3                               # image 4 carries a wrong label
        loader.append((x, torch.tensor([5 + (i + 1) % 3 if good else 77])))
    # the label map of an image is handed out in CALL order (workers=1 keeps the pool's submission order = loader order)
    calls = []

    def segmenter_in_order(img_u8):
        calls.append(1)
        return maps[want_order[len(calls) - 1]]

    idx = list(range(1, len(sizes) + 1))
    api.configure(eval_img_index=1, num_mask_samples=9, segmenter=segmenter_in_order, mask_dir=None, seed=None)
    try:
        eng = _KindEngine()
        want_order = idx
        many = api.validate_many(loader, eng, None, idx, rng=random.Random(3), workers=1, lookahead=2)
        assert all(len(k) == 1 for k in eng.packed_calls), eng.packed_calls           # one kind per packed call
        assert {"table"} in eng.packed_calls and {"conv"} in eng.packed_calls
        # the table group (302 / 302 / 302 / 282 rows, max_batch 400) fills twice; the conv group (62 + 42 + 62 + 52) only at the end
        assert eng.packed_calls.count({"table"}) == 2 and eng.packed_calls.count({"conv"}) == 1
        # per image, one shared stream, loader order
        del calls[:]
        eng2 = _KindEngine()
        shared = random.Random(3)
        want = {}
        for i in idx:
            want_order = [i]
            del calls[:]
            api.configure(eval_img_index=i, num_mask_samples=9, segmenter=segmenter_in_order, mask_dir=None, seed=None)
            want[i] = api.validate(loader, eng2, None, i, rng=shared)
        assert many[4] is None and want[4] == 0                                        # wrong base prediction: None / 0, no draws
        assert {i: many[i] for i in idx if i != 4} == {i: want[i] for i in idx if i != 4}
    finally:
        api.configure(eval_img_index=1, num_mask_samples=100, segmenter=None, mask_dir=None, seed=None)
