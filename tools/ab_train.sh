#!/bin/bash
# A/B of the amp-O2 training leg on ONE GPU box (boxes differ by several per cent: only same-box numbers compare).
#   bash tools/ab_train.sh "tagA:VAR=val VAR2=val" "tagB:" ...     each setting is run twice, interleaved
# e.g.  bash tools/ab_train.sh "rows:" "norows:MINDPOSE_TRAIN_FUSE_STREAMS=0"
#       bash tools/ab_train.sh "b1024:" "b2048:MINDPOSE_EXPERIMENT_KNOBS=1 MP_BN_PRE_BLOCKS=2048 MP_BN_PRE_MIN=512"
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    env $envs timeout -k 10 420 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline \
        > gpurun_out/ab_${tag}_$rep.json 2> gpurun_out/ab_${tag}_$rep.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${tag}_$rep.json").read().strip().splitlines()[-1])
print("$tag rep $rep:", d.get("value"), "img/s", d.get("ms_per_step"), "ms")
PY
  done
done
