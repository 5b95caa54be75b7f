#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines, rocprofv3 kernel statistics and the PMC traffic
# passes whose summaries are committed under profiles/.  Usage: tools/collect_profiles.sh TAG   (e.g. r01)
set -e -o pipefail
TAG=${1:-r01}
LEGS=${2:-all}     # "all"; "tmall": the Tmall training-step legs only (traffic passes, bench line, kernel stats, step timeline); "rest": the others
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# traffic passes first: the bench lines below read the JSON they produce
if [ "$LEGS" != rest ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_tmall_$c -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_tmall_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_industry_$c -o p -- python $R/bench.py --workload industry --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_industry_$c.log 2>&1
done
python $R/tools/make_traffic_json.py $O/traffic.json tmall $O/pmc_tmall_FETCH_SIZE $O/pmc_tmall_WRITE_SIZE industry $O/pmc_industry_FETCH_SIZE $O/pmc_industry_WRITE_SIZE > /dev/null
cp $O/traffic.json $R/profiles/${TAG}_traffic.json
timeout -k 10 300 python $R/bench.py --steps 50 --warmup 10 > $O/tmall_bench.json 2> $O/tmall_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_tmall -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/ks_tmall.log 2>&1
cp $O/ks_tmall/p_kernel_stats.csv $O/tmall_train_step_kernel_stats.csv
# one replayed step kernel by kernel, with the hardware queue each ran on (what runs beside the propagation)
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt_tmall -o step -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_tmall.log 2>&1
python $R/tools/step_timeline.py $O/kt_tmall/step_results.db 15 > $O/tmall_train_step_timeline.txt
fi
if [ "$LEGS" = tmall ]; then echo done; exit 0; fi
timeout -k 10 300 python $R/bench.py --workload industry --steps 20 --warmup 5 > $O/industry_bench.json 2> $O/industry_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_industry -o p -- python $R/bench.py --workload industry --steps 20 --warmup 5 --no-cpu-baseline > $O/ks_industry.log 2>&1
# SQ / TA / L2 counters of the layer kernel (forward propagation only, tools/layer_bench.py)
i=0
: > $O/pmc_gcn_layer.txt
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "GRBM_GUI_ACTIVE MeanOccupancyPerCU MemUnitStalled" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_layer_$i -o p -- python $R/tools/layer_bench.py tmall 0,0 > $O/pmc_layer_$i.log 2>&1
  echo "## rocprofv3 --pmc $c -- python tools/layer_bench.py tmall 0,0   (per-dispatch means; FETCH/WRITE_SIZE in KiB)" >> $O/pmc_gcn_layer.txt
  python $R/tools/pmc_summary.py $O/pmc_layer_$i set_layer >> $O/pmc_gcn_layer.txt
  python $R/tools/pmc_summary.py $O/pmc_layer_$i set_hub >> $O/pmc_gcn_layer.txt
done
for w in eval_tmall eval_ijcai negsample_tmall metapath_dblp predict_ijcai nc_dblp nc_train_dblp epoch_tmall; do
  timeout -k 10 200 python $R/bench.py --workload $w --steps 10 --warmup 3 > $O/${w}_bench.json 2> $O/${w}_bench.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_eval -o p -- python $R/bench.py --workload eval_tmall --steps 10 --warmup 3 --no-cpu-baseline > $O/ks_eval.log 2>&1
cp $O/ks_eval/p_kernel_stats.csv $O/eval_tmall_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_epoch -o p -- python $R/bench.py --workload epoch_tmall --steps 5 --warmup 2 --no-cpu-baseline > $O/ks_epoch.log 2>&1
cp $O/ks_epoch/p_kernel_stats.csv $O/epoch_tmall_kernel_stats.csv
cp $O/ks_industry/p_kernel_stats.csv $O/industry_1gpu_kernel_stats.csv
echo done
