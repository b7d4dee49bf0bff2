#!/bin/bash
# SQ / LDS / TA counters of the block-tail kernel (csrc/mpx_btail.h), the identity tail at batch 2048 (one gpurun call).
# usage: tools/pmc_btail.sh <round-tag>
set -e -o pipefail
R=${1:-r03}
O=gpurun_out/pmc_btail_$R
mkdir -p $O
export TMPDIR=/tmp
P0="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P1="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
{
echo "# rocprofv3 --pmc <set> --kernel-trace -- python3 tools/tail_bench.py <k> 2048 5   (MI355X)"
for K in 1 0; do
  for CNT in "$P0" "$P1" "$P2" "$P3"; do
    rm -rf $O/run
    if rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/run -- python3 tools/tail_bench.py $K 2048 5 > $O/run.log 2>&1; then
      echo "## $(grep 'ms per launch' $O/run.log | tail -1)"
      python tools/pmc_summary.py $O/run btail_f16x3
    else
      echo "## tail $K: pass failed: $CNT"; tail -3 $O/run.log
    fi
  done
done
} > $O/${R}_pmc_btail.txt 2>&1
cat $O/${R}_pmc_btail.txt
