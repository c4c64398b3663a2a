#!/bin/bash
# GPU occupancy over time of a bench step (tools/timeline.py on a rocprofv3 kernel trace):
#   bash tools/timeline.sh <tag> <marker kernel substring> [bench args]     e.g.  timeline.sh tl_train gaussian_target --workload hrnet_w32_train --amp O2 --batch 128 --leg --no-roofline
set -e
tag=$1; marker=$2; shift 2
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
export MINDPOSE_TUNE_CACHE=$out/${tag}_tune.json
python3 bench.py "$@" --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_tune.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 $root/bench.py "$@" --steps 6 --warmup 3 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
f=$(find $out/${tag}_prof -name "*kernel_trace.csv" | head -1)
python3 $root/tools/timeline.py "$f" --marker "$marker" --steps 3 --json $out/${tag}_timeline.json --dump $out/${tag}_last_step.csv ${TIMELINE_SPLIT:+--split "$TIMELINE_SPLIT"}
rm -rf $out/${tag}_prof
