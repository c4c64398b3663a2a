#!/bin/bash
# all round-end evidence in one GPU call
set -e
bash tools/profile_round.sh r01_e 2>&1 | tail -12
bash tools/profile_round.sh r01_f_o2 --amp O2 2>&1 | tail -12
export MINDPOSE_TUNE_CACHE=gpurun_out/r01_g_tune.json
python3 bench.py --workload hrnet_w48_384_udp_flip --batch 64 --amp O2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r01_g_config5_o2_bench.json 2> gpurun_out/r01_g.err
python3 bench.py --workload hrnet_w32_train --batch 128 --steps 3 --warmup 2 > gpurun_out/r01_h_train_bench.json 2> gpurun_out/r01_h.err
python3 bench.py --workload hrnet_w32_train --batch 128 --amp O2 --steps 10 --warmup 3 > gpurun_out/r01_i_train_o2_bench.json 2> gpurun_out/r01_i.err
python3 bench.py --workload simplebaseline_r50_train --batch 128 --steps 3 --warmup 2 > gpurun_out/r01_j_sb_train_bench.json 2> gpurun_out/r01_j0.err
python3 bench.py --workload simplebaseline_r50_train --batch 128 --amp O2 --steps 10 --warmup 3 > gpurun_out/r01_j_sb_train_o2_bench.json 2> gpurun_out/r01_j.err
python3 bench.py --workload simplebaseline_r50 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r01_k_simplebaseline_bench.json 2> gpurun_out/r01_k.err
python3 bench.py --workload simplebaseline_r50 --batch 64 --amp O2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r01_k_simplebaseline_o2_bench.json 2> gpurun_out/r01_k2.err
tail -c 400 gpurun_out/r01_g_config5_o2_bench.json; echo; tail -c 300 gpurun_out/r01_h_train_bench.json
