#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_train_f16.py tests/test_gpu_bn_fuse.py tests/test_gpu_train_full.py -x -q > gpurun_out/r3n_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3n_tests.log
tail -3 gpurun_out/r3n_tests.log
grep -q "tests rc=0" gpurun_out/r3n_tests.log || exit 1
for rep in 1 2; do
MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3n_shapes_$rep.csv timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3n_train_$rep.json 2>gpurun_out/r3n_train_$rep.err || exit 1
python - <<PY
import json
d = json.loads(open("gpurun_out/r3n_train_$rep.json").read().strip().splitlines()[-1])
pe = d["roofline"]["per_entry"]
print("rep $rep", d.get("value"), d.get("ms_per_step"), "fwd", pe["mp_f16_bn_train_fwd_stats"], "bwd", pe["mp_f16_bn_train_bwd_stats"])
PY
done
grep -E "bn_train_(fwd|bwd)_stats" gpurun_out/r3n_shapes_2.csv | head -14
