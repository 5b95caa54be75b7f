#!/bin/bash
# Copies the summaries tools/collect_profiles.sh left under gpurun_out/TAG (merged back from the GPU box) into profiles/
# under TAG_ names: the bench lines, kernel statistics, the step timeline, the traffic file, the counter summary.
# Usage: tools/publish_profiles.sh TAG
set -e
TAG=${1:?tag}
S=gpurun_out/$TAG
for f in $S/*_bench.json $S/*_kernel_stats.csv $S/tmall_train_step_timeline.txt $S/traffic.json $S/pmc_gcn_layer.txt; do
  [ -s "$f" ] && cp "$f" profiles/${TAG}_$(basename $f)
done
ls profiles | grep "^${TAG}_"
