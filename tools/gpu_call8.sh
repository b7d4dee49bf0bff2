#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "round_split or conv256 or cfg3 or patch_kernel" > gpurun_out/r2_pytest_gpu_8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r2_pytest_gpu_8.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_8.log 2>&1; tail -1 gpurun_out/r2_layers_8.log
timeout -k 10 200 python tools/layer_profile.py resnet18 2048 3 > gpurun_out/r2_layers18_8.log 2>&1; tail -1 gpurun_out/r2_layers18_8.log
grep -E "k3 s1|k3 s2" gpurun_out/r2_layers_8.log
