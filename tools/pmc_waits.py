#!/usr/bin/env python3
"""Per-kernel SQ counter averages from the passes of tools/pmc_waits.sh -> gpurun_out/<tag>_sq.json."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import key_of  # noqa: E402


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    per = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = key_of(row["Kernel_Name"])
                    e = per.setdefault(k, {}).setdefault(row["Counter_Name"], {})
                    did = row.get("Dispatch_Id", row.get("Correlation_Id"))
                    e[did] = e.get(did, 0.0) + float(row["Counter_Value"])
    out = {k: {c: round(sum(v.values()) / len(v), 1) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())}
           for k, cs in per.items()}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", f"{tag}_sq.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1]["dispatches"])[:10]:
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        print(k, v["dispatches"], {c: round(x / wc, 3) for c, x in v.items() if c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_VALU_MFMA"))})


if __name__ == "__main__":
    main()
