#!/bin/bash
# small-batch latency sweep (serving): hipGraph on/off, fp32 and fp16
mkdir -p gpurun_out
for amp in O0 O2; do for g in 1 0; do for n in 1 8 32; do
  MINDPOSE_HIP_GRAPH=$g timeout -k 10 200 python bench.py --batch $n --amp $amp --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('amp $amp graph $g N $n', r['ms_per_step'], 'ms', r['value'], 'img/s')"
done; done; done
