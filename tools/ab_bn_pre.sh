#!/bin/bash
# amp-O2 training leg with the BatchNorm apply left to the consumer conv's operand staging (MINDPOSE_BN_PRE=1, default) against
# the apply-pass form (=0), interleaved, each with its own tuner cache:   bash tools/ab_bn_pre.sh [rounds]
out=gpurun_out
mkdir -p $out
for r in $(seq 1 ${1:-2}); do
    for m in 0 1; do
        MINDPOSE_BN_PRE=$m MINDPOSE_TUNE_CACHE=$out/ab_bn_pre_tune_$m.json python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline \
            > $out/ab_bn_pre_${m}_$r.json 2> $out/ab_bn_pre_${m}_$r.err || { tail -5 $out/ab_bn_pre_${m}_$r.err; exit 1; }
        python3 -c "import json,sys; d=json.loads(open('$out/ab_bn_pre_${m}_$r.json').read().strip().splitlines()[-1]); print('BN_PRE=$m round $r:', d.get('value'), d.get('unit'), d.get('ms_per_step'))"
    done
done
