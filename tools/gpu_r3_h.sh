#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3h_train_shapes.csv timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3h_train.json 2>gpurun_out/r3h_train.err || exit 1
grep -E "bn_train|sum_tensors|fuse" gpurun_out/r3h_train_shapes.csv
