#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_gpu_final.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r2_pytest_gpu_final.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; cut -c1-400 gpurun_out/r3_bench_final.json
