#!/bin/bash
# The expanding 1x1 layer 256 -> 1024 on tile 10 (weights through the LDS) and tile 14 (weights in registers), batch 2340, in ONE gpurun
# call: time, HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes, units and the x2 of MI355X_MICROARCH.md) and SQ / GRBM counters.
# usage: tools/pmc_convw.sh <round-tag>
set -e -o pipefail
R=${1:-r04}
cd "$(dirname "$0")/.."
O=gpurun_out/pmc_convw_$R
mkdir -p $O
export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
{
echo "# python3 tools/conv_bench.py resnet101 layer3.5.conv3 2340 20 10,14   (no profiler)"
python3 tools/conv_bench.py resnet101 layer3.5.conv3 2340 20 10,14 2>&1 | grep "ms " | cut -c1-160
for tile in 10 14; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "$SQ"; do
    rm -rf $O/run
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/run -- python3 tools/conv_bench.py resnet101 layer3.5.conv3 2340 5 $tile > $O/run.log 2>&1
    echo "## tile $tile, --pmc $pass: $(grep 'ms ' $O/run.log | tail -1 | cut -c1-120)"
    python tools/pmc_summary.py $O/run _f16x3_kernel
  done
done
} > $O/${R}_pmc_convw.txt 2>&1
cat $O/${R}_pmc_convw.txt
