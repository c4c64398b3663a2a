#!/bin/bash
mkdir -p gpurun_out
for amp in O0 O2; do for lanes in 0 1; do
  MINDPOSE_PLAN_LANES=$lanes MINDPOSE_TUNE_CACHE=gpurun_out/tune_ab.json timeout -k 10 300 python bench.py --amp $amp --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('amp $amp lanes $lanes', r['ms_per_step'], 'ms', r['value'], 'img/s')"
done; done
MINDPOSE_TUNE_CACHE=gpurun_out/tune_ab.json timeout -k 10 300 python bench.py --workload hrnet_w48_384_udp_flip --batch 64 --amp O2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('config5 O2 lanes 1', r['ms_per_step'], 'ms', r['value'], 'img/s')"
