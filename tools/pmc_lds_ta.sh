#!/bin/bash
# Which unit of a CU is busy in the conv K loops: LDS array, texture addresser (TA, also carries LDS-DMA), L1 (TCP)?  Two PMC passes
# per kernel (the counters do not all fit one pass).  usage: tools/pmc_lds_ta.sh <round-tag>
set -e -o pipefail
R=${1:-r02}
O=gpurun_out/pmc2_$R
mkdir -p $O
export TMPDIR=/tmp
P1="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
{
for spec in "layer3.5.conv2 6" "layer3.5.conv1 9" "layer3.5.conv3 7"; do
  set -- $spec
  for CNT in "$P1" "$P2" "$P3"; do
    rm -rf $O/run
    if rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/run -- python3 tools/conv_bench.py resnet101 $1 2048 5 $2 > $O/run.log 2>&1; then
      echo "## $1 tile $2: $(grep 'ms ' $O/run.log | tail -1 | cut -c1-120)"
      python tools/pmc_summary.py $O/run _f16x3_kernel
    else
      echo "## $1 tile $2: pass failed: $CNT"; tail -3 $O/run.log
    fi
  done
done
} > $O/${R}_pmc_lds_ta.txt 2>&1
cat $O/${R}_pmc_lds_ta.txt
