#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r3p_bench_full.json 2> gpurun_out/r3p_bench_full.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3p_bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k, v in (d.get("extra_workloads") or {}).items():
    if isinstance(v, dict): print(k, v.get("value"), v.get("ms_per_step"), v.get("unit"))
print(d.get("cpu_baseline"))
PY
