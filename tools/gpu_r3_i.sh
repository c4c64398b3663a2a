#!/bin/bash
# BatchNorm apply kernels: in-block fold of the conv's partial slots (default) against a fold launch + thin apply blocks
set -o pipefail
mkdir -p gpurun_out
export MINDPOSE_EXPERIMENT_KNOBS=1
run() {
  tag=$1; shift
  env "$@" timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > gpurun_out/r3i_$tag.json 2>gpurun_out/r3i_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3i_$tag.json").read().strip().splitlines()[-1])
print("$tag", d.get("value"), d.get("ms_per_step"))
PY
}
run base A=1
run fold16 MP_BN_PREFOLD_ABOVE=16
run fold16_b1024 MP_BN_PREFOLD_ABOVE=16 MP_BN_PRE_BLOCKS=1024 MP_BN_PRE_MIN=1024
run fold16_b2048 MP_BN_PREFOLD_ABOVE=16 MP_BN_PRE_BLOCKS=2048 MP_BN_PRE_MIN=512
run b1024 MP_BN_PRE_BLOCKS=1024 MP_BN_PRE_MIN=1024
run fold64_b1024 MP_BN_PREFOLD_ABOVE=64 MP_BN_PRE_BLOCKS=1024 MP_BN_PRE_MIN=1024
run base2 A=1
