#!/bin/bash
# lockstep branch blocks with grouped BatchNorm apply passes: tests, then A/B of the training leg on one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bn_fuse.py tests/test_gpu_train_f16.py tests/test_gpu_train_full.py -x -q > gpurun_out/r3t_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3t_tests.log
tail -12 gpurun_out/r3t_tests.log
grep -q "tests rc=0" gpurun_out/r3t_tests.log || exit 1
run() {
  tag=$1; shift
  env "$@" timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3t_$tag.json 2>gpurun_out/r3t_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3t_$tag.json").read().strip().splitlines()[-1])
print("$tag", d.get("value"), d.get("ms_per_step"))
PY
}
run group MINDPOSE_BN_GROUP=1
run nogroup A=1
run group2 MINDPOSE_BN_GROUP=1
run nogroup2 A=1
