#!/bin/bash
# AddressSanitizer + UBSan over the code that runs on the HOST (GPU sanitizers are not available on this pool): libmpxseg.so whole, and
# the CPU paths of libmpx.so's C-ABI (weight packing, argument checks, null-engine calls; device code is built without instrumentation).
# The sanitized builds live in /tmp and are bound per process; the product libraries are not touched.   usage: tools/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++17 -shared -fPIC -pthread -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I include \
    -o /tmp/libmpxseg_asan.so network_interpretation_imagenet_amd/csrc/mpx_seg.cpp
( cd network_interpretation_imagenet_amd/csrc && ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC \
    -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -o /tmp/libmpx_asan.so mpx_api.hip )
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python - <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np
from network_interpretation_imagenet_amd import segment, synth
segment.LIB_PATH = "/tmp/libmpxseg_asan.so"
segment.load()
n = 0
for kind in ("blobs", "noise"):
    for im in synth.make_images(3, seed=11, kind=kind):
        assert segment.felzenszwalb(im, scale=100, sigma=0.5, min_size=50).shape == (224, 224)
        n += 1
for shape in ((1, 1, 3), (2, 3, 3), (17, 5, 3), (28, 28, 1), (32, 32, 3)):       # degenerate and small pictures (the CIFAR / MNIST scripts' sizes)
    im = (np.random.default_rng(0).random(shape) * 255).astype(np.uint8)
    segment.felzenszwalb(im if shape[2] == 3 else im[:, :, 0], scale=100, sigma=0.5, min_size=5)
    n += 1
with segment.SegmenterPool(workers=4) as pool:
    for f in [pool.submit(np.random.default_rng(i).standard_normal((3, 224, 224)).astype(np.float32)) for i in range(8)]:
        assert f.result().shape == (224, 224)
print("libmpxseg under ASan + UBSan: %d segmentations + a 4-thread pool, clean" % n)
PY
ASAN_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python - <<'PY' 2>&1 | grep -v "^mpx: binding"
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from network_interpretation_imagenet_amd import _lib
_lib.LIB_PATH = "/tmp/libmpx_asan.so"
lib = _lib.load()
import test_host_logic as T
for cin, cout, k in [(64, 64, 3), (256, 128, 1), (3, 64, 7)]:
    T.test_pack_conv_weights(lib, cin, cout, k)
T.test_pack_fc(lib)
T.test_pack_rejects_bad_desc(lib)
T.test_pack_padded_channels_and_conv_bias(lib)
T.test_null_engine_calls_fail_cleanly(lib)
print("libmpx C-ABI host paths under ASan + UBSan (pack_conv_weights x3, pack_fc, bad descriptors, padded channels, null-engine calls): clean")
PY
