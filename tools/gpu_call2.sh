#!/bin/bash
# round-2 GPU call 2: parity of the fused-downsample and persistent kernels, then their timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or persistent or every_tile or forward_fused or cfg3 or resnet101_vs_golden" > gpurun_out/r2_pytest_gpu_2.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r2_pytest_gpu_2.log
[ $rc -ne 0 ] && exit $rc
{
for L in layer3.5.conv3 layer3.5.conv1 layer2.1.conv3 layer1.1.conv3 layer4.1.conv3 layer2.1.conv1 layer3.0.conv2; do
  timeout -k 10 120 python tools/conv_bench.py resnet101 $L 2048 20 7,9,2,8,7,9,2,8 || exit 1
done
} > gpurun_out/r2_convbench_2.log 2>&1
echo "convbench rc=$?"; grep "ms" gpurun_out/r2_convbench_2.log | tail -60
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_default_2.log 2>&1; tail -3 gpurun_out/r2_layers_default_2.log
MPX_TILE_RULES=k1exp:9 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1exp9_2.log 2>&1; tail -3 gpurun_out/r2_layers_k1exp9_2.log
MPX_TILE_RULES=k1exp:9,k1red:8 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1exp9_k1red8_2.log 2>&1; tail -3 gpurun_out/r2_layers_k1exp9_k1red8_2.log
MPX_TILE_RULES=k1exp:8,k1red:8 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1_8_2.log 2>&1; tail -3 gpurun_out/r2_layers_k1_8_2.log
