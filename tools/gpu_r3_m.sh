#!/bin/bash
# BatchNorm apply kernels: per-entry device time for block-count settings
set -o pipefail
mkdir -p gpurun_out
export MINDPOSE_EXPERIMENT_KNOBS=1
run() {
  tag=$1; shift
  env "$@" MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3m_shapes_$tag.csv timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3m_$tag.json 2>gpurun_out/r3m_$tag.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3m_$tag.json").read().strip().splitlines()[-1])
pe = d["roofline"]["per_entry"]
print("$tag", d.get("value"), d.get("ms_per_step"), "fwd", pe["mp_f16_bn_train_fwd_stats"]["ms"], "bwd", pe["mp_f16_bn_train_bwd_stats"]["ms"])
PY
}
run b1024 MP_BN_PRE_BLOCKS=1024 MP_BN_PRE_MIN=1024
run b1024m512 MP_BN_PRE_BLOCKS=1024 MP_BN_PRE_MIN=512
run b1536m512 MP_BN_PRE_BLOCKS=1536 MP_BN_PRE_MIN=512
run b2048m512 MP_BN_PRE_BLOCKS=2048 MP_BN_PRE_MIN=512
run b2048m256 MP_BN_PRE_BLOCKS=2048 MP_BN_PRE_MIN=256
run b3072m512 MP_BN_PRE_BLOCKS=3072 MP_BN_PRE_MIN=512
run b1024b MP_BN_PRE_BLOCKS=1024 MP_BN_PRE_MIN=1024
