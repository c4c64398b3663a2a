#!/bin/bash
# fp16 weight gradient, narrow (64 x 16) tile for the stem: parity tests, then the training leg with and without it (same box) and the per-shape table
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train_f16.py tests/test_gpu_bn_fuse.py tests/test_gpu_train_full.py -x -q > gpurun_out/r3k_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3k_tests.log
tail -5 gpurun_out/r3k_tests.log
grep -q "tests rc=0" gpurun_out/r3k_tests.log || exit 1
run() {
  tag=$1; shift
  env "$@" timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3k_$tag.json 2>gpurun_out/r3k_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3k_$tag.json").read().strip().splitlines()[-1])
print("$tag", d.get("value"), d.get("ms_per_step"))
PY
}
run narrow A=1
run old MINDPOSE_EXPERIMENT_KNOBS=1 MP_WGRAD16_NARROW=0
run narrow2 A=1
run old2 MINDPOSE_EXPERIMENT_KNOBS=1 MP_WGRAD16_NARROW=0
MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3k_train_shapes.csv timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3k_train.json 2>gpurun_out/r3k_train.err || exit 1
grep -E "wgrad" gpurun_out/r3k_train_shapes.csv | grep -E " s2 |3->64"
