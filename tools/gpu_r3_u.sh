#!/bin/bash
# exchange-unit rows on side streams: training tests, then A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bn_fuse.py tests/test_gpu_train_f16.py tests/test_gpu_train_full.py tests/test_gpu_dp.py -x -q > gpurun_out/r3u_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3u_tests.log
tail -5 gpurun_out/r3u_tests.log
grep -q "tests rc=0" gpurun_out/r3u_tests.log || exit 1
run() {
  tag=$1; shift
  env "$@" timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3u_$tag.json 2>gpurun_out/r3u_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3u_$tag.json").read().strip().splitlines()[-1])
print("$tag", d.get("value"), d.get("ms_per_step"))
PY
}
run rows A=1
run base MINDPOSE_TRAIN_FUSE_STREAMS=0
run rows2 A=1
run base2 MINDPOSE_TRAIN_FUSE_STREAMS=0
