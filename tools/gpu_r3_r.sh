#!/bin/bash
# stride-2 data gradient: four phases in one launch - parity tests, then the training leg with and without it (same box)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train_f16.py tests/test_gpu_bn_fuse.py tests/test_gpu_train_full.py -x -q > gpurun_out/r3r_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3r_tests.log
tail -5 gpurun_out/r3r_tests.log
grep -q "tests rc=0" gpurun_out/r3r_tests.log || exit 1
run() {
  tag=$1; shift
  env "$@" timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3r_$tag.json 2>gpurun_out/r3r_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3r_$tag.json").read().strip().splitlines()[-1])
print("$tag", d.get("value"), d.get("ms_per_step"))
PY
}
run merged A=1
run nostats MINDPOSE_BN_FUSE_PARTS=15
run merged2 A=1
run nostats2 MINDPOSE_BN_FUSE_PARTS=15
