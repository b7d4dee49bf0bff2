#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "convs_pixel" > gpurun_out/convs_tests.log 2>&1
rc=$?; tail -5 gpurun_out/convs_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python tools/conv_bench.py resnet101 layer3.5.conv3 2048 10 7,10,11 2>&1 | grep "ms "
timeout -k 10 120 python tools/conv_bench.py resnet101 layer2.1.conv3 2048 10 7,10,11 2>&1 | grep "ms "
