#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_train_f16.py tests/test_gpu_bn_fuse.py tests/test_gpu_train_full.py -x -q > gpurun_out/r3n_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3n_tests.log
tail -3 gpurun_out/r3n_tests.log
grep -q "tests rc=0" gpurun_out/r3n_tests.log || exit 1
for rep in 1 2; do
timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3n_train_$rep.json 2>gpurun_out/r3n_train_$rep.err || exit 1
python - <<PY
import json
d = json.loads(open("gpurun_out/r3n_train_$rep.json").read().strip().splitlines()[-1])
print("rep $rep", d.get("value"), d.get("ms_per_step"))
PY
done
