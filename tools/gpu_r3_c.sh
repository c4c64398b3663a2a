#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3c_train_shapes.csv timeout -k 10 300 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3c_train_o2.json 2> gpurun_out/r3c_train_o2.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c_train_o2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["sum_kernel_ms"])
for k, v in d["roofline"]["per_entry"].items():
    print(k, v)
PY
