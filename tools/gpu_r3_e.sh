#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_winograd.py -x -q > gpurun_out/r3e_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3e_tests.log
tail -4 gpurun_out/r3e_tests.log
grep -q "tests rc=0" gpurun_out/r3e_tests.log || exit 1
for nb in 128 32; do
python bench.py --batch $nb --no-extra --no-cpu-baseline --layers gpurun_out/r3e_layers_n$nb.csv > gpurun_out/r3e_headline_n$nb.json 2>/dev/null
MINDPOSE_EXPERIMENT_KNOBS=1 MP_WINO_KSPLIT=0 python bench.py --batch $nb --no-extra --no-cpu-baseline --no-roofline > gpurun_out/r3e_headline_n${nb}_nok.json 2>/dev/null
python - <<PY
import json, csv
for tag in ("", "_nok"):
    d = json.loads(open("gpurun_out/r3e_headline_n$nb%s.json" % tag).read().strip().splitlines()[-1])
    print("N=$nb", tag or "ksplit-auto", d["value"], d["ms_per_step"])
rows = list(csv.DictReader(open("gpurun_out/r3e_layers_n$nb.csv")))
agg = {}
for r in rows:
    if "wino" in r["kernel"]:
        k = (r["kernel"], r["cin"], r["cout"], r["h"] + "x" + r["w"]); a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["us"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("   ", k, a[0], round(a[1] / a[0], 1), "us avg")
PY
done
