#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv3 2048 7,10,3
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv1 2048 2,3
} > gpurun_out/r2_timeline_7.log 2>&1
grep -E "==|k-loop|K steps|resident" gpurun_out/r2_timeline_7.log
timeout -k 10 120 python tools/conv_bench.py resnet101 layer3.5.conv3 2048 20 7,10,7,10 2>&1 | grep ms
timeout -k 10 120 python tools/conv_bench.py resnet101 layer2.1.conv3 2048 20 7,10,7,10 2>&1 | grep ms
