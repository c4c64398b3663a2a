#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3j_$tag.json 2>gpurun_out/r3j_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3j_$tag.json").read().strip().splitlines()[-1])
print("$tag", d.get("value"), d.get("ms_per_step"))
PY
}
run base A=1
run nosync MP_EXPERIMENT_NO_SYNC=1
run base2 A=1
run nosync2 MP_EXPERIMENT_NO_SYNC=1
