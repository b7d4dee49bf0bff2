#!/bin/bash
# round-2 GPU call 3: small-network parity; nt on/off in the register epilogue; downsample fusion in the network
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small or trained_small" > gpurun_out/r2_pytest_gpu_3.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r2_pytest_gpu_3.log
{
for D in 0 1 2 3; do
  echo "== MPX_DBG=$D (bit0: nt residual loads, bit1: nt stores)"
  for L in layer3.5.conv3 layer1.1.conv3 layer2.1.conv3; do
    MPX_DBG=$D timeout -k 10 120 python tools/conv_bench.py resnet101 $L 2048 20 7,9,8,9,8 || exit 1
  done
done
} > gpurun_out/r2_convbench_3.log 2>&1
echo "convbench rc=$?"; grep -E "ms |MPX_DBG" gpurun_out/r2_convbench_3.log | awk '{print $1, $2, $3, $4, $5, $6, $12, $13, $14}'
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_fused_3.log 2>&1; tail -2 gpurun_out/r2_layers_fused_3.log
MPX_NO_FUSION=1 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_unfused_3.log 2>&1; tail -2 gpurun_out/r2_layers_unfused_3.log
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_fused_3b.log 2>&1; tail -2 gpurun_out/r2_layers_fused_3b.log
