#!/bin/bash
# Runs on the GPU box: rocprofv3 --pmc passes over the Tmall training step, summarised for the two single-step
# denoiser kernels (loss_bwd_kernel, denoise_kernel<1>).   tools/pmc_loss.sh TAG
set -o pipefail
TAG=${1:-pmc_loss}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_loss.txt
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/p_$i -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --other-workloads "" --one-gpu-forward "" > $O/p_$i.log 2>&1 || { echo "pass $i ($c) failed" >> $O/pmc_loss.txt; tail -3 $O/p_$i.log >> $O/pmc_loss.txt; continue; }
  echo "## --pmc $c" >> $O/pmc_loss.txt
  python $R/tools/pmc_summary.py $O/p_$i loss_bwd >> $O/pmc_loss.txt
  python $R/tools/pmc_summary.py $O/p_$i "denoise_kernel<1>" >> $O/pmc_loss.txt
  rm -rf $O/p_$i
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES
GRBM_GUI_ACTIVE MeanOccupancyPerCU
LIST
cat $O/pmc_loss.txt
