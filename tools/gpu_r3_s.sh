#!/bin/bash
# how much do the branch streams inside the captured step give today?
set -o pipefail
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3s_$tag.json 2>gpurun_out/r3s_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3s_$tag.json").read().strip().splitlines()[-1])
print("$tag", d.get("value"), d.get("ms_per_step"))
PY
}
run streams A=1
run serial MINDPOSE_TRAIN_BRANCH_STREAMS=0
run lanes MINDPOSE_TRAIN_WGRAD_LANES=1
run streams2 A=1
run serial2 MINDPOSE_TRAIN_BRANCH_STREAMS=0
