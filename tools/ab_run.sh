#!/bin/bash
# parity subset + in-network layer profile (one gpurun call); usage: tools/ab_run.sh <tag>
set -o pipefail
T=${1:-x}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "conv_ or fused or cfg3 or cfg1 or trained" > gpurun_out/ab_${T}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/ab_${T}_tests.log; [ $rc -ne 0 ] && exit $rc
python tools/layer_profile.py resnet101 2048 3 > gpurun_out/lp_${T}.txt 2>&1 || exit 1
python tools/layer_profile.py resnet101 2048 3 > gpurun_out/lp_${T}_2.txt 2>&1 || exit 1
tail -n 1 gpurun_out/lp_${T}.txt gpurun_out/lp_${T}_2.txt
