#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per counter group over tools/layer_bench.py (the fused propagation
# layers of a Rec workload), summarised per kernel into gpurun_out/<tag>/diag_pmc.txt.
#   tools/diag_pmc.sh TAG [workload] [kernel-name pattern] [tool]     (tool: layer_bench.py or set_bench.py)
set -o pipefail
TAG=${1:-diag}; WL=${2:-tmall}; PAT=${3:-gcn_layer}; TOOL=${4:-layer_bench.py}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/diag_pmc.txt
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $c --output-format csv -d $O/diag_$i -o p -- python $R/tools/$TOOL $WL 0,0 > $O/diag_$i.log 2>&1 || { echo "pass $i ($c) failed" >> $O/diag_pmc.txt; tail -3 $O/diag_$i.log >> $O/diag_pmc.txt; continue; }
  echo "## --pmc $c" >> $O/diag_pmc.txt
  python $R/tools/pmc_summary.py $O/diag_$i $PAT >> $O/diag_pmc.txt
  rm -rf $O/diag_$i
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GRBM_TC_BUSY GRBM_EA_BUSY
LIST
cat $O/diag_pmc.txt
