#!/bin/bash
# all round-2 evidence in one GPU call (gpurun_out/r02_*; copy what should be judged into profiles/)
#   bash tools/round_evidence.sh
set -e
out=gpurun_out
mkdir -p $out
# the bench line as the driver runs it (headline + cpu_baseline + extra_workloads)
python3 bench.py > $out/r02_d_bench_full.json 2> $out/r02_d_bench_full.err
tail -c 300 $out/r02_d_bench_full.json; echo
# rocprofv3 / PMC evidence; --no-extra: a profiled process must not start child processes (the profiler's preload has already
# initialised the GPU when the program starts)
bash tools/profile_round.sh r02_d --no-extra 2>&1 | tail -12
bash tools/profile_round.sh r02_e_o2 --amp O2 --no-extra 2>&1 | tail -12
bash tools/profile_train.sh r02_f_train_o2 --amp O2 --batch 128 --steps 5 --warmup 2 --leg 2>&1 | tail -14
export MINDPOSE_TUNE_CACHE=$out/r02_f_train_o2_tune.json
python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 10 --warmup 3 --leg > $out/r02_f_train_o2_bench.json 2> $out/r02_f.err
tail -c 300 $out/r02_f_train_o2_bench.json; echo
