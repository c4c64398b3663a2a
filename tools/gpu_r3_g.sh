#!/bin/bash
# training leg at HEAD (twice: with and without the experiment-knob gate open), then the whole -m gpu suite
set -o pipefail
mkdir -p gpurun_out
for k in 0 1; do
MINDPOSE_EXPERIMENT_KNOBS=$k timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3g_train_k$k.json 2>gpurun_out/r3g_train_k$k.err || exit 1
python - <<PY
import json
d = json.loads(open("gpurun_out/r3g_train_k$k.json").read().strip().splitlines()[-1])
print("knobs $k", d.get("value"), d.get("ms_per_step"))
PY
done
bash tools/gpu_full_tests.sh
