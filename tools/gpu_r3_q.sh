#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload hrnet_w48_384_udp_flip --amp O2 --batch 64 --steps 20 --warmup 5 --leg --layers gpurun_out/r3q_c5_layers.csv > gpurun_out/r3q_c5.json 2> gpurun_out/r3q_c5.err || exit 1
python - <<'PY'
import json, csv
d = json.loads(open("gpurun_out/r3q_c5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"))
rows = list(csv.DictReader(open("gpurun_out/r3q_c5_layers.csv")))
agg = {}
for r in rows:
    k = (r["kernel"][:44], r["cin"], r["cout"], r["h"] + "x" + r["w"], r["k"], r["stride"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["us"])
tot = sum(a[1] for a in agg.values())
print("sum us", tot)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(k, a[0], round(a[1] / a[0], 1), "us", round(a[1] / 1e3, 2), "ms")
PY
