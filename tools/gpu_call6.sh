#!/bin/bash
# round-2 GPU call 6: full GPU suite with the new defaults (fused downsample, 256x256 tile on reducing 1x1), layer profile, bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_gpu_6.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r2_pytest_gpu_6.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/layer_profile.py resnet101 2048 3 > gpurun_out/r2_layers_6.log 2>&1; tail -1 gpurun_out/r2_layers_6.log
timeout -k 10 300 python bench.py > gpurun_out/r2_bench_6.json 2> gpurun_out/r2_bench_6.err; cat gpurun_out/r2_bench_6.json | cut -c1-1500
