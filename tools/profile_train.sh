#!/bin/bash
# rocprofv3 kernel stats of the training step:  bash tools/profile_train.sh <tag> [bench args]
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 $root/bench.py --workload hrnet_w32_train "$@" > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
f=$(find $out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
rm -rf $out/${tag}_prof
head -30 $out/${tag}_kernel_stats.csv | cut -c1-150
