#!/bin/bash
# round-4 evidence (gpurun_out/r04_*; copy what should be judged into profiles/)
#   bash tools/round4_evidence.sh c5 <tag>       config 5 (HRNet-W48 384x288 UDP / DARK flip, fp16, N = 64): bench line, rocprofv3 kernel
#                                                stats, PMC traffic + MFMA-busy, SQ wait counters, per-launch table
#   bash tools/round4_evidence.sh o2 <tag>       HRNet-W32 amp-O2 inference: the same set
#   bash tools/round4_evidence.sh head <tag>     the driver's bench line + the fp32 headline's rocprofv3 / PMC set
#   bash tools/round4_evidence.sh train <tag>    amp-O2 training step: rocprofv3 kernel stats + bench line with per-entry tables
set -e
part=${1:-c5}
tag=${2:-r04_a_$part}
out=gpurun_out
mkdir -p $out
if [ "$part" = "c5" ]; then
    args="--workload hrnet_w48_384_udp_flip --amp O2 --batch 64 --no-extra"
    bash tools/profile_round.sh $tag $args 2>&1 | tail -12
    bash tools/pmc_waits.sh $tag $args 2>&1 | tail -4
    MINDPOSE_TUNE_CACHE=$out/${tag}_tune.json python3 bench.py $args --no-cpu-baseline --layers $out/${tag}_layers.csv > $out/${tag}_layers_bench.json 2> $out/${tag}_layers.err
elif [ "$part" = "o2" ]; then
    args="--amp O2 --no-extra"
    bash tools/profile_round.sh $tag $args 2>&1 | tail -12
    bash tools/pmc_waits.sh $tag $args 2>&1 | tail -4
    MINDPOSE_TUNE_CACHE=$out/${tag}_tune.json python3 bench.py $args --no-cpu-baseline --layers $out/${tag}_layers.csv > $out/${tag}_layers_bench.json 2> $out/${tag}_layers.err
elif [ "$part" = "head" ]; then
    python3 bench.py > $out/${tag}_bench_full.json 2> $out/${tag}_bench_full.err
    tail -c 300 $out/${tag}_bench_full.json; echo
    bash tools/profile_round.sh $tag --no-extra 2>&1 | tail -12
elif [ "$part" = "train" ]; then
    bash tools/profile_train.sh $tag --amp O2 --batch 128 --steps 10 --warmup 3 --leg --no-roofline 2>&1 | tail -8
    MINDPOSE_BENCH_TRAIN_SHAPES=$out/${tag}_shapes.csv MINDPOSE_TUNE_CACHE=$out/${tag}_tune.json python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > $out/${tag}_bench.json 2> $out/${tag}.err
    tail -c 300 $out/${tag}_bench.json; echo
fi
