#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3h_train_shapes.csv timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3h_train.json 2>gpurun_out/r3h_train.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3h_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k, v in d["roofline"]["per_entry"].items(): print(k, v)
print(d["roofline"]["sum_kernel_ms"])
PY
grep -E "^mp_f16_conv2d_fwd,|bn_train_bwd,|bn_train_fwd,|upsample|sum_tensors|fuse_sum" gpurun_out/r3h_train_shapes.csv | head -50
