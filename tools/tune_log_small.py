"""Read a MINDPOSE_TUNE_LOG file and print, per shape that the small-problem fp32 kernel (tuner ids 11 / 12) was timed on, its
time beside the best other candidate (ms per 5 launches):  python tools/tune_log_small.py gpurun_out/tune.log"""
import ast
import collections
import sys

rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    parts = line.rstrip("\n").split("\t")
    if len(parts) != 3:
        continue
    rows.setdefault(ast.literal_eval(parts[0]), {})[int(parts[1])] = float(parts[2])
for key, v in rows.items():
    if 11 not in v and 12 not in v:
        continue
    others = {a: b for a, b in v.items() if a not in (11, 12)}
    bo = min(others, key=others.get) if others else None
    print("N=%d cin=%d %dx%d cout=%d k=%d s=%d | best %d | small %s wide %s | other %s %.4f" % (
        key[0], key[1], key[2], key[3], key[4], key[5], key[7], min(v, key=v.get), v.get(11), v.get(12), bo, others.get(bo, float("nan"))))
