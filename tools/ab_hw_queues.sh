#!/bin/bash
# Same-box A/B of the number of hardware queues the HIP runtime multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4): the
# captured training step forks the HRModule branches / exchange-unit rows onto side streams, and chains that share a hardware queue
# run one after the other.   bash tools/ab_hw_queues.sh "<values>" [bench args]
vals=${1:-"4 8"}; shift
out=gpurun_out; mkdir -p $out
export MINDPOSE_TUNE_CACHE=$out/ab_hwq_tune.json
args=${@:-"--workload hrnet_w32_train --amp O2 --batch 128 --leg --no-roofline --steps 20 --warmup 5"}
python3 bench.py $args > /dev/null 2> $out/ab_hwq_tune.err   # fills the tuner cache
for rep in 1 2; do
  for v in $vals; do
    r=$(GPU_MAX_HW_QUEUES=$v python3 bench.py $args 2> $out/ab_hwq_$v.err | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])")
    echo "GPU_MAX_HW_QUEUES=$v: $r"
  done
done
