#!/bin/bash
# round-2 GPU call 5: the 256x256-tile 1x1 kernel: parity, isolated and in-network timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv256" > gpurun_out/r2_pytest_gpu_5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r2_pytest_gpu_5.log
[ $rc -ne 0 ] && exit $rc
{
for L in layer3.5.conv1 layer3.5.conv3 layer4.1.conv1 layer4.1.conv3 layer2.1.conv3 layer1.1.conv3 layer3.0.conv1 layer4.0.conv1; do
  timeout -k 10 120 python tools/conv_bench.py resnet101 $L 2048 20 -1,9,-1,9 || exit 1
done
} > gpurun_out/r2_convbench_5.log 2>&1
echo "convbench rc=$?"; grep -E "ms " gpurun_out/r2_convbench_5.log | awk '{print $2, $4, $5, $6, $11, $12, $13, $14}'
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_base_5.log 2>&1; tail -1 gpurun_out/r2_layers_base_5.log
MPX_TILE_RULES=k1red:9 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1red9_5.log 2>&1; tail -1 gpurun_out/r2_layers_k1red9_5.log
MPX_TILE_RULES=k1red:9,k1exp:9 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1all9_5.log 2>&1; tail -1 gpurun_out/r2_layers_k1all9_5.log
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv1 2048 9,2 > gpurun_out/r2_timeline_5.log 2>&1; grep -E "==|median|share|K steps" gpurun_out/r2_timeline_5.log
