#!/bin/bash
# Timing-only ablation of the persistent expanding-1x1 kernel (csrc/mpx_convx.h, tile 10): probe builds in which one class of memory
# instructions carries an out-of-range offset (still issued and counted by vmcnt, no memory access, WRONG results) -- which of the pixel
# fetch, the weight fetch, the residual loads and the output stores does the layer's time consist of?  The builds are made HERE (hipcc
# cross-compiles) into tools/probes/ so that they travel with the tree; run with `tools/ablate_convx.sh run` on the GPU box.
set -e
cd "$(dirname "$0")/.."
MASKS="0 1 2 3 4 8 12 15"
if [ "$1" = "run" ]; then
  B=${2:-2340}
  for L in layer3.5.conv3 layer2.1.conv3; do
    for M in $MASKS; do
      printf "CX_ABL=%-2s " $M; python tools/with_lib.py tools/probes/libmpx_cxabl$M.so tools/conv_bench.py resnet101 $L $B 40 10 2>/dev/null | grep TFLOP | cut -c1-100
    done
  done
else
  for M in $MASKS; do
    ( cd network_interpretation_imagenet_amd/csrc && ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCX_ABL=$M -o ../../tools/probes/libmpx_cxabl$M.so mpx_api.hip ) &
    if [ $(jobs -r | wc -l) -ge 4 ]; then wait -n; fi
  done
  wait
  ls -la tools/probes/libmpx_cxabl*.so
fi
