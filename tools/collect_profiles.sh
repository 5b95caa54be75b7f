#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines, rocprofv3 kernel statistics and the PMC traffic
# passes whose summaries are committed under profiles/.  Usage: tools/collect_profiles.sh TAG   (e.g. r01)
set -e -o pipefail
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/bench.py --steps 50 --warmup 10 > $O/tmall_bench.json 2> $O/tmall_bench.err
timeout -k 10 300 python $R/bench.py --workload industry --steps 20 --warmup 5 > $O/industry_bench.json 2> $O/industry_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_tmall -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/ks_tmall.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_industry -o p -- python $R/bench.py --workload industry --steps 20 --warmup 5 --no-cpu-baseline > $O/ks_industry.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_tmall_$c -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_tmall_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_industry_$c -o p -- python $R/bench.py --workload industry --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_industry_$c.log 2>&1
done
python $R/tools/make_traffic_json.py $O/traffic.json tmall $O/pmc_tmall_FETCH_SIZE $O/pmc_tmall_WRITE_SIZE industry $O/pmc_industry_FETCH_SIZE $O/pmc_industry_WRITE_SIZE > /dev/null
cp $O/ks_tmall/p_kernel_stats.csv $O/tmall_train_step_kernel_stats.csv
cp $O/ks_industry/p_kernel_stats.csv $O/industry_1gpu_kernel_stats.csv
echo done
