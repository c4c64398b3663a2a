#!/bin/bash
# Same-box A/B of environment switches on one bench leg:  bash tools/ab_env.sh "<VAR=a VAR=b ...>" [bench args]   ("-" = no variable)
sets=$1; shift
out=gpurun_out; mkdir -p $out
export MINDPOSE_TUNE_CACHE=$out/ab_env_tune.json
args=${@:-"--workload hrnet_w32_train --amp O2 --batch 128 --leg --no-roofline --steps 20 --warmup 5"}
python3 bench.py $args > /dev/null 2> $out/ab_env_tune.err   # fills the tuner cache
for rep in 1 2; do
  for v in $sets; do
    if [ "$v" = "-" ]; then e=""; else e=$(echo $v | tr ',' ' '); fi
    r=$(env $e python3 bench.py $args 2> $out/ab_env_last.err | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])")
    echo "$v: $r"
  done
done
