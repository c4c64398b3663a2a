#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bn_fuse.py tests/test_gpu_train_full.py -x -q > gpurun_out/r3d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3d_tests.log
tail -4 gpurun_out/r3d_tests.log
grep -q "tests rc=0" gpurun_out/r3d_tests.log || exit 1
timeout -k 10 300 python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 30 --warmup 5 --leg --no-roofline > gpurun_out/r3d_train_o2.json 2> gpurun_out/r3d_train_o2.err
python -c "
import json
d = json.loads(open('gpurun_out/r3d_train_o2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
bash tools/profile_train.sh r03_b_train_o2 --amp O2 --batch 128 --steps 10 --warmup 3 --leg --no-roofline > /dev/null 2>&1
python - <<'PY'
import csv, re
rows = list(csv.DictReader(open("gpurun_out/r03_b_train_o2_kernel_stats.csv")))
steps = 16.0  # 3 eager warm-ups + 13 replays
def cat(n):
    if "wgrad" in n: return "wgrad"
    if "bn16_fold" in n: return "bn fold"
    if "bn16_apply_pre" in n: return "bn fwd apply(pre)"
    if "bn16_bwd_apply_pre" in n: return "bn bwd apply(pre)"
    if "bn16" in n or "bn_" in n: return "bn old"
    m = re.search(r"conv_f16\w*kernel<([^>]*)>", n)
    if m: return "conv stats" + m.group(1).split(",")[-1].strip()
    if "fuse_sum" in n or "sum_tensors" in n: return "fuse/sum"
    if "copyBuffer" in n or "fill" in n.lower(): return "copy/fill"
    return "other"
c = {}
for r in rows:
    k = cat(r["Name"]); c.setdefault(k, [0, 0]); c[k][0] += float(r["TotalDurationNs"]) / 1e6 / steps; c[k][1] += int(r["Calls"]) / steps
for k, (t, n) in sorted(c.items(), key=lambda kv: -kv[1][0]): print(f"{k:22s} {t:7.2f} ms {n:7.1f}/step")
print(sum(v[0] for v in c.values()))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("bn16", "fuse_sum", "sum_tensors")):
        print(n[:50].ljust(50), r["Calls"], "avg", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1))
PY
