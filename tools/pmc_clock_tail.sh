#!/bin/bash
# effective clock and MFMA duty of the three block-tail launches (one rocprofv3 --pmc pass per tail)
set -e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
O=gpurun_out/pmc_clock_tail; mkdir -p $O
for K in 0 1 2; do
  rm -rf $O/run
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/run -- python3 tools/tail_bench.py $K 2340 30 > $O/run.log 2>&1
  echo "## tail $K: $(grep 'ms per launch' $O/run.log | tail -1)"
  python tools/pmc_clock.py $O/run btail_f16x3
done
rm -rf $O/run
