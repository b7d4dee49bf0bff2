#!/bin/bash
# parity subset + in-network layer profile (one gpurun call); usage: tools/ab_run.sh <tag> [pytest -k expression]
set -o pipefail
T=${1:-x}
K=${2:-"stem or fusion or cfg1 or cfg3 or golden or live_oracle"}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -q -x -k "$K" > gpurun_out/ab_${T}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/ab_${T}_tests.log; [ $rc -ne 0 ] && exit $rc
python tools/layer_profile.py resnet101 2048 3 > gpurun_out/lp_${T}.txt 2>&1 || exit 1
python tools/layer_profile.py resnet18 2048 3 > gpurun_out/lp_${T}_r18.txt 2>&1 || exit 1
tail -n 1 gpurun_out/lp_${T}.txt gpurun_out/lp_${T}_r18.txt; grep "3->64" gpurun_out/lp_${T}.txt
