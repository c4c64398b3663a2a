#!/bin/bash
# kernel timeline of the graph-replayed amp-O2 training step: rocprofv3 --kernel-trace, then tools/trace_gaps.py on the csv
set -e
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
export MINDPOSE_TUNE_CACHE=$out/trace_tune.json
python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 2 --warmup 1 --leg --no-roofline > /dev/null 2> $out/trace_tune.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace_prof -o trace -- python3 $root/bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 6 --warmup 3 --leg --no-roofline > $out/trace_bench.json 2> $out/trace_rocprof.err
f=$(find $out/trace_prof -name "*kernel_trace.csv" | head -1)
cd $root
head -2 "$f" > $out/trace_head.txt
python3 tools/trace_gaps.py "$f" $out/trace_timeline.csv > $out/trace_gaps.txt
rm -rf $out/trace_prof
tail -60 $out/trace_gaps.txt
