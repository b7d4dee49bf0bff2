#!/bin/bash
# round-2 GPU call 1: full GPU suite (new cfg-2 / cfg-3 / cfg-5 tests) + workgroup timelines of the 1x1 layers
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_gpu_1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2_pytest_gpu_1.log
tail -5 gpurun_out/r2_pytest_gpu_1.log
{
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv3 2048 7,2,0
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv1 2048 2,0,7
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv2 2048 6,2
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer1.1.conv3 2048 7,2
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer2.1.conv3 2048 7
} > gpurun_out/r2_timeline_1.log 2>&1
echo "timeline rc=$?"
tail -30 gpurun_out/r2_timeline_1.log
