"""ctypes binding of libmpx.so (include/mpx.h).  There is no CPU fallback: if the HIP library is
missing or a call fails, this module raises."""
import ctypes as C
import os

# torch first: PyTorch-ROCm ships its own libamdhip64; libmpx.so must bind to that one runtime (the
# one that owns the tensors and streams it is handed).  Loading libmpx.so before torch would pull
# /opt/rocm's copy into the process and the two runtimes do not share devices or streams.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPX_LIB_PATH: another build of the library for this process AND the processes it starts (tools/with_lib.py sets it for a probe
# build, so that `bench.py --gpus N`'s child ranks bind what the parent was told to bind); unset = the product library in the tree
PRODUCT_LIB_PATH = os.path.join(_HERE, "libmpx.so")
LIB_PATH = os.environ.get("MPX_LIB_PATH") or PRODUCT_LIB_PATH

IMG = 224
IMG_PAD = 230
NUM_CLASSES = 1000


class MpxError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("bn_name", C.c_char * 48),
                ("cin", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32),
                ("stride", C.c_int32), ("pad", C.c_int32), ("hin", C.c_int32),
                ("hout", C.c_int32), ("relu", C.c_int32), ("residual", C.c_int32),
                ("k_packed", C.c_int32), ("cout_pad", C.c_int32)]


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/mpx.h declares
SIGNATURES = {
    "mpx_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "mpx_destroy": (_i, [_vp]),
    "mpx_last_error": (C.c_char_p, [_vp]),
    "mpx_max_batch": (_i, [_vp]),
    "mpx_num_cus": (_i, [_vp]),
    "mpx_workspace_bytes": (C.c_size_t, [_vp]),
    "mpx_num_convs": (_i, [_vp]),
    "mpx_conv_info": (_i, [_vp, _i, C.POINTER(ConvDesc)]),
    "mpx_set_conv_weights": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f]),
    "mpx_geometry": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "mpx_mask_apply_minmax": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mpx_avgpool2_pad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mpx_weights_complete": (_i, [_vp]),
    "mpx_set_conv_tile": (_i, [_vp, _i, _i]),
    "mpx_get_conv_tile": (_i, [_vp, _i]),
    "mpx_last_conv_kernels": (_i, [_vp]),
    "mpx_pack_conv_weights": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "mpx_mask_apply_normalize": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _fp, _fp, _i, _vp, _vp]),
    "mpx_stem_table_build": (_i, [_vp, _vp, _vp, _vp, _i, _fp, _fp, _vp]),
    "mpx_stem_table_apply": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mpx_conv_bn_act": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mpx_conv_dual_bn_act": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mpx_set_fusion": (_i, [_vp, _i]),
    "mpx_bottleneck_tail": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mpx_num_bottleneck_tails": (_i, [_vp]),
    "mpx_bottleneck_tail_info": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "mpx_stem_conv_maxpool": (_i, [_vp, _vp, _vp, _i, _vp]),
    "mpx_maxpool3x3s2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mpx_global_avgpool": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mpx_head_softmax_gather": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mpx_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mpx_heatmap_accumulate": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mpx_input_planes": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "mpx_mark_input_staged": (_i, [_vp, _i, _i]),
    "mpx_stem_planes": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "mpx_profile_enable": (_i, [_vp, _i]),
    "mpx_profile_collect": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "mpx_flops_per_forward": (C.c_double, [_vp]),
}

_lib = None


def load():
    """Load libmpx.so (built by __graft_entry__.build() / csrc/build.sh)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpxError("HIP extension %s not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        if is_probe_build():
            import sys
            sys.stderr.write("mpx: binding %s (MPX_LIB_PATH / with_lib.py) -- NOT the product library %s; probe builds may be timing-only\n"
                             % (LIB_PATH, PRODUCT_LIB_PATH))
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def is_probe_build():
    """True when this process binds anything but the in-tree product library (MPX_LIB_PATH, tools/with_lib.py)."""
    return os.path.realpath(LIB_PATH) != os.path.realpath(PRODUCT_LIB_PATH)


def bound_library():
    """What a record of a run must carry to say WHICH library it ran (bench.py's config): the path this process binds, the
    sha256 of that file, the build stamp next to it (`<lib>.sha256` = __graft_entry__._source_hash of the sources and flags it was built
    from; None when there is none) and whether it is the in-tree product library."""
    import hashlib
    h = hashlib.sha256()
    with open(LIB_PATH, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    stamp = None
    if os.path.exists(LIB_PATH + ".sha256"):
        with open(LIB_PATH + ".sha256") as fh:
            stamp = fh.read().strip()
    return {"lib_path": LIB_PATH, "lib_sha256": h.hexdigest(), "lib_stamp": stamp, "product_library": not is_probe_build()}


def check(handle, rc, what):
    if rc != 0:
        msg = load().mpx_last_error(handle).decode() if handle else ""
        raise MpxError("%s failed (rc=%d): %s" % (what, rc, msg))
