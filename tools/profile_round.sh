#!/bin/bash
# Evidence for one bench configuration on the GPU box: bench line, rocprofv3 kernel stats, PMC HBM traffic.
#   bash tools/profile_round.sh <tag> [bench args...]      e.g.  bash tools/profile_round.sh r01_e
# Writes gpurun_out/<tag>_* ; copy what should be judged into profiles/.
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
export MINDPOSE_TUNE_CACHE=$out/${tag}_tune.json
# 1. the bench line itself (also fills the tuner cache that the profiled runs replay)
python3 bench.py "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench.json; echo
# 2. per-kernel times.  bench.py's roofline section times every launch alone on one stream (HIP events around
#    mp_plan_run_range); the matching rocprof summary therefore comes from a single-lane replay (MINDPOSE_PLAN_LANES=0).
#    A second summary of the default multi-lane run (kernels of the four HRNet branches overlap, so each one's duration
#    stretches) is kept next to it.
cd /tmp
export MINDPOSE_PLAN_LANES=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 $root/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
f=$(find $out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
head -8 $out/${tag}_kernel_stats.csv
unset MINDPOSE_PLAN_LANES
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_lanes -o ${tag} -- python3 $root/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_rocprof_lanes.err
f=$(find $out/${tag}_prof_lanes -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats_lanes.csv
export MINDPOSE_PLAN_LANES=0
# 3. HBM traffic counters, one pass each
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o ${tag} -- python3 $root/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o ${tag} -- python3 $root/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_mfma -o ${tag} -- python3 $root/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_pmc_mfma.err || true
cd $root
python3 tools/pmc_summary.py ${tag} $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_mfma
cp profiles/${tag}_pmc_traffic.json $out/
# keep the merge-back small
rm -rf $out/${tag}_prof $out/${tag}_prof_lanes $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_mfma
