#!/bin/bash
# rocprofv3 kernel stats of the training step:  bash tools/profile_train.sh <tag> [bench args]
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
# fill the tuner cache first so that the profiled run replays the choices instead of timing trial launches
export MINDPOSE_TUNE_CACHE=$out/${tag}_tune.json
python3 bench.py --workload hrnet_w32_train "$@" --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_tune.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 $root/bench.py --workload hrnet_w32_train "$@" > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
f=$(find $out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
rm -rf $out/${tag}_prof
head -30 $out/${tag}_kernel_stats.csv | cut -c1-150
