#!/bin/bash
# all round-2 evidence (final kernels), two GPU calls of < 20 min each (gpurun_out/r02_*; copy what should be judged into profiles/)
#   bash tools/round_evidence.sh 1     headline as the driver runs it + its rocprofv3 / PMC set, the same for --amp O2
#   bash tools/round_evidence.sh 2     SimpleBaseline-R50 (configs[1]) set, the two training steps
set -e
part=${1:-1}
out=gpurun_out
mkdir -p $out
if [ "$part" = "1" ]; then
    # the bench line as the driver runs it (headline + cpu_baseline + extra_workloads)
    python3 bench.py > $out/r02_h_bench_full.json 2> $out/r02_h_bench_full.err
    tail -c 300 $out/r02_h_bench_full.json; echo
    # rocprofv3 / PMC evidence; --no-extra: a profiled process must not start child processes (the profiler's preload has already
    # initialised the GPU when the program starts)
    bash tools/profile_round.sh r02_h --no-extra 2>&1 | tail -12
    bash tools/profile_round.sh r02_i_o2 --amp O2 --no-extra 2>&1 | tail -12
else
    bash tools/profile_round.sh r02_l_sb --workload simplebaseline_r50 --leg 2>&1 | tail -12
    bash tools/profile_train.sh r02_j_train_o2 --amp O2 --batch 128 --steps 5 --warmup 2 --leg 2>&1 | tail -14
    MINDPOSE_TUNE_CACHE=$out/r02_j_train_o2_tune.json python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 10 --warmup 3 --leg > $out/r02_j_train_o2_bench.json 2> $out/r02_j.err
    tail -c 300 $out/r02_j_train_o2_bench.json; echo
    bash tools/profile_train.sh r02_k_train_f32 --batch 128 --steps 5 --warmup 2 --leg 2>&1 | tail -14
    MINDPOSE_TUNE_CACHE=$out/r02_k_train_f32_tune.json python3 bench.py --workload hrnet_w32_train --batch 128 --steps 10 --warmup 3 --leg > $out/r02_k_train_f32_bench.json 2> $out/r02_k.err
    tail -c 300 $out/r02_k_train_f32_bench.json; echo
fi
