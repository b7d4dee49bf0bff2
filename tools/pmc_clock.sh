#!/bin/bash
# Effective clock and MFMA duty of the persistent kernels against the kernels they replace (isolated layers, batch 2340).
# usage: tools/pmc_clock.sh <round-tag>
set -e -o pipefail
R=${1:-r03}
O=gpurun_out/pmc_clock_$R
mkdir -p $O
export TMPDIR=/tmp
{
for spec in "layer3.5.conv2 6" "layer3.5.conv2 12" "layer3.5.conv1 9" "layer3.5.conv1 13" "layer3.5.conv3 10" "layer2.1.conv2 6" "layer2.1.conv2 12"; do
  set -- $spec
  rm -rf $O/run
  if rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/run -- python3 tools/conv_bench.py resnet101 $1 2340 40 $2 > $O/run.log 2>&1; then
    echo "## $1 tile $2: $(grep 'ms ' $O/run.log | tail -1 | cut -c1-120)"
    python tools/pmc_clock.py $O/run _f16x3_kernel
  else
    echo "## $1 tile $2: pass failed"; tail -3 $O/run.log
  fi
done
} > $O/${R}_pmc_clock.txt 2>&1
cat $O/${R}_pmc_clock.txt
rm -rf $O/run
