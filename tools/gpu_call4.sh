#!/bin/bash
# round-2 GPU call 4: persistent kernel with the full-line register epilogue: parity, isolated and in-network timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or persistent or every_tile or small_network_ops" > gpurun_out/r2_pytest_gpu_4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r2_pytest_gpu_4.log
[ $rc -ne 0 ] && exit $rc
{
for D in 0 2; do
  echo "== MPX_DBG=$D (bit1: nt stores)"
  for L in layer3.5.conv3 layer1.1.conv3 layer2.1.conv3 layer3.5.conv1 layer4.1.conv3; do
    MPX_DBG=$D timeout -k 10 120 python tools/conv_bench.py resnet101 $L 2048 20 7,8,2,8 || exit 1
  done
done
} > gpurun_out/r2_convbench_4.log 2>&1
echo "convbench rc=$?"; grep -E "ms |MPX_DBG" gpurun_out/r2_convbench_4.log | awk '{print $1, $2, $3, $4, $5, $6, $12, $13, $14}'
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_base_4.log 2>&1; tail -1 gpurun_out/r2_layers_base_4.log
MPX_TILE_RULES=k1exp:8 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1exp8_4.log 2>&1; tail -1 gpurun_out/r2_layers_k1exp8_4.log
MPX_TILE_RULES=k1exp:8,k1red:8 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1all8_4.log 2>&1; tail -1 gpurun_out/r2_layers_k1all8_4.log
MPX_DBG=2 MPX_TILE_RULES=k1exp:8 timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_k1exp8_nt_4.log 2>&1; tail -1 gpurun_out/r2_layers_k1exp8_nt_4.log
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_base_4b.log 2>&1; tail -1 gpurun_out/r2_layers_base_4b.log
