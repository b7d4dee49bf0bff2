#!/bin/bash
# SQ counters of the stem-table apply launches (one gpurun call): where do the waves of stem_apply_kernel spend their cycles?
# usage: tools/pmc_stem.sh [grid|felz|grid8]
set -e -o pipefail
K=${1:-grid}
cd "$(dirname "$0")/.."
O=gpurun_out/pmc_stem
mkdir -p $O
export TMPDIR=/tmp
for CNT in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  rm -rf $O/run
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/run -- python3 tools/stem_bench.py 2340 512 $K 2 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
  python tools/pmc_summary.py $O/run stem_apply
done
grep "masks" $O/run.log | tail -1
