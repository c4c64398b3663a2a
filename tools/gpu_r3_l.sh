#!/bin/bash
# fp16 weight gradient: workgroups per launch (slab traffic against parallelism), per-shape table per setting
set -o pipefail
mkdir -p gpurun_out
export MINDPOSE_EXPERIMENT_KNOBS=1
for w in 256 384 768; do
MP_WGRAD16_WGS=$w MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3l_shapes_$w.csv timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3l_$w.json 2>gpurun_out/r3l_$w.err || exit 1
python - <<PY
import json
d = json.loads(open("gpurun_out/r3l_$w.json").read().strip().splitlines()[-1])
print("wgs $w", d.get("value"), d.get("ms_per_step"), d["roofline"]["per_entry"]["mp_f16_conv_wgrad_grouped"])
PY
grep -E "wgrad" gpurun_out/r3l_shapes_$w.csv | head -8
done
