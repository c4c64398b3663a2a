#!/bin/bash
# kernel timeline of the inference plan replay:  bash tools/trace_infer.sh [bench args, e.g. --amp O2]
set -e
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
export MINDPOSE_TUNE_CACHE=$out/trace_inf_tune.json
python3 bench.py "$@" --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-roofline > /dev/null 2> $out/trace_inf_tune.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace_inf_prof -o trace -- python3 $root/bench.py "$@" --steps 6 --warmup 3 --no-extra --no-cpu-baseline --no-roofline > $out/trace_inf_bench.json 2> $out/trace_inf_rocprof.err
f=$(find $out/trace_inf_prof -name "*kernel_trace.csv" | head -1)
cd $root
python3 tools/trace_gaps.py "$f" $out/trace_inf_timeline.csv decode_kernel > $out/trace_inf_gaps.txt
rm -rf $out/trace_inf_prof
grep -v "^gap" $out/trace_inf_gaps.txt | head -40
