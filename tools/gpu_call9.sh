#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "convx" > gpurun_out/r2_pytest_gpu_9.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r2_pytest_gpu_9.log
[ $rc -ne 0 ] && exit $rc
{
for L in layer3.5.conv3 layer2.1.conv3 layer4.1.conv3 layer3.5.conv1; do
  timeout -k 10 120 python tools/conv_bench.py resnet101 $L 2048 20 -1,10,-1,10 || exit 1
done
} > gpurun_out/r2_convbench_9.log 2>&1
echo "convbench rc=$?"; grep -E "ms " gpurun_out/r2_convbench_9.log | awk '{print $2, $4, $5, $6, $11, $12, $13, $14}'
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_base_9.log 2>&1; tail -1 gpurun_out/r2_layers_base_9.log
MPX_TILE_RULES=k1exp:10 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1exp10_9.log 2>&1; tail -1 gpurun_out/r2_layers_k1exp10_9.log
