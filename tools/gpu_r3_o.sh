#!/bin/bash
# per-candidate tuner timings of the amp-O2 training step's convolutions (plain / statistics mode 1 / mode 2)
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/r3o_tune_log.tsv
MINDPOSE_TUNE_LOG=gpurun_out/r3o_tune_log.tsv timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 5 --warmup 2 --leg --no-roofline > gpurun_out/r3o.json 2>gpurun_out/r3o.err || exit 1
wc -l gpurun_out/r3o_tune_log.tsv
