#!/bin/bash
# SQ / GRBM counters of the three dominant conv kernels, one layer each at batch 2048 (one gpurun call).
# usage: tools/pmc_kernels.sh <round-tag>
set -e -o pipefail
R=${1:-r02}
O=gpurun_out/pmc_$R
mkdir -p $O
export TMPDIR=/tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
{
echo "# rocprofv3 --pmc $CNT --kernel-trace -- python3 tools/conv_bench.py resnet101 <layer> 2048 5 <tile>   (MI355X, batch 2048)"
for spec in "layer3.5.conv3 7" "layer3.5.conv1 9" "layer3.5.conv1 2" "layer3.5.conv2 6" "layer3.5.conv3 10"; do
  set -- $spec
  rm -rf $O/run
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/run -- python3 tools/conv_bench.py resnet101 $1 2048 5 $2 > $O/run.log 2>&1
  echo "## $1 tile $2: $(grep 'ms ' $O/run.log | tail -1 | cut -c1-160)"
  python tools/pmc_summary.py $O/run _f16x3_kernel
done
} > $O/${R}_pmc_conv_kernels.txt 2>&1
cat $O/${R}_pmc_conv_kernels.txt
{
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv3 2048 7,10
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv1 2048 2,9
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer3.5.conv2 2048 6
timeout -k 10 120 python tools/probes/conv_timeline.py resnet101 layer1.1.conv3 2048 7
} > $O/${R}_timeline.txt 2>&1
grep -E "==|median|K steps|resident|tile 10" $O/${R}_timeline.txt
