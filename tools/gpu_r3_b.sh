#!/bin/bash
# round 3, call B: BatchNorm-fusion tests + O2 training bench (fused vs per-cell)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bn_fuse.py tests/test_gpu_train_full.py tests/test_gpu_train_f16.py -x -q > gpurun_out/r3b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b_tests.log
tail -15 gpurun_out/r3b_tests.log
grep -q "tests rc=0" gpurun_out/r3b_tests.log || exit 1
MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3b_train_shapes.csv timeout -k 10 300 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3b_train_o2.json 2> gpurun_out/r3b_train_o2.err
echo "bench rc=$?"


python - <<'PY'
import json
for f in ("gpurun_out/r3b_train_o2.json", "gpurun_out/r3b_train_o2_nofuse.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("final_loss"))
    except Exception as e:
        print(f, "unreadable", e)
PY
