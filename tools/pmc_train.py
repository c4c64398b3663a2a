#!/usr/bin/env python3
"""Per-kernel HBM traffic, MFMA-busy share and SQ wait counters of the training step (tools/pmc_train.sh) ->
gpurun_out/<tag>_pmc_traffic.json and <tag>_sq_counters.json.  Kernels are keyed by their FULL template argument list (the statistics
mode and the residual flag are template arguments of the conv builds and matter here).  FETCH_SIZE / WRITE_SIZE are KiB per dispatch,
FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md); durations come from the separate --stats pass (counters perturb them)."""
import csv
import glob
import json
import os
import re
import sys


def key_of(name):
    s = name.replace("void ", "").replace("mp::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^>]*>)?", s)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else s


def collect(dirs):
    per = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    e = per.setdefault(key_of(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], {})
                    did = row.get("Dispatch_Id", row.get("Correlation_Id"))
                    e[did] = e.get(did, 0.0) + float(row["Counter_Value"])
    return {k: {c: (sum(v.values()) / len(v), len(v)) for c, v in cs.items()} for k, cs in per.items()}


def step_totals(traffic, dur):
    """HBM bytes and summed kernel time of ONE training step: every forward + backward pass of the stats run (the eager warm-ups of
    the capture and every replay) launches the loss gradient `mse_bwd_kernel` exactly once, so calls / that count = launches per
    step (the tuner's trial launches were done by an earlier process: the profiled run replays its cache)."""
    steps = dur.get("mse_bwd_kernel", (0, 0.0))[0]
    if not steps:
        return None
    tot_b = sum(traffic[k]["hbm_bytes_per_launch"] * dur[k][0] / steps for k in traffic if k in dur)
    tot_ns = sum(t for _, t in dur.values()) / steps
    cats = {}
    for k, (calls, t) in dur.items():
        c = ("batchnorm apply forward" if k.startswith("bn16_apply") else "batchnorm apply backward" if k.startswith("bn16_bwd") else
             "weight gradient" if "wgrad" in k else "convolution (forward / data gradient)" if k.startswith(("conv_f16", "stem_", "expand_")) else
             "fan-in / exchange unit" if ("fuse" in k or "sum_tensors" in k) else "other")
        e = cats.setdefault(c, [0.0, 0.0, 0.0])
        e[0] += t / steps / 1e6
        e[1] += (traffic[k]["hbm_bytes_per_launch"] if k in traffic else 0) * calls / steps / 1e9
        e[2] += calls / steps
    return {"steps_in_stats_pass": steps, "hbm_GB_per_step": round(tot_b / 1e9, 2), "sum_kernel_ms_per_step": round(tot_ns / 1e6, 2),
            "by_kind": {c: {"ms": round(v[0], 2), "GB": round(v[1], 2), "launches": round(v[2], 1), "TB_per_s": round(v[1] / v[0], 2) if v[0] else None}
                        for c, v in sorted(cats.items(), key=lambda kv: -kv[1][0])}}


def main():
    if sys.argv[1] == "--resummarize":  # <tag>_pmc_traffic.json + the stats CSV -> the same JSON with the per-step totals added
        path, stats_csv = sys.argv[2], sys.argv[3]
        with open(path) as f:
            doc = json.load(f)
        dur = {}
        with open(stats_csv) as f:
            for r in csv.DictReader(f):
                k = key_of(r["Name"])
                calls, tot = dur.get(k, (0, 0.0))
                dur[k] = (calls + int(r["Calls"]), tot + float(r["TotalDurationNs"]))
        doc = {"units": doc["units"], "step": step_totals(doc["kernels"], dur), "kernels": doc["kernels"]}
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)
        print(json.dumps(doc["step"], indent=1))
        return
    tag, stats_csv, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    dur = {}
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            k = key_of(r["Name"])
            calls, tot = dur.get(k, (0, 0.0))
            dur[k] = (calls + int(r["Calls"]), tot + float(r["TotalDurationNs"]))
    total_ns = sum(t for _, t in dur.values())
    per = collect(dirs)
    traffic, sq = {}, {}
    for k, cs in per.items():
        calls, tot = dur.get(k, (0, 0.0))
        avg_us = tot / calls / 1e3 if calls else None
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            fb = cs.get("FETCH_SIZE", (0.0, 0))[0] * 1024 * 2
            wb = cs.get("WRITE_SIZE", (0.0, 0))[0] * 1024
            e = dict(fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb), hbm_bytes_per_launch=round(fb + wb),
                     dispatches_counted=cs.get("FETCH_SIZE", cs.get("WRITE_SIZE"))[1], avg_launch_us=None if avg_us is None else round(avg_us, 2),
                     share_of_kernel_time=round(tot / total_ns, 4) if total_ns else None)
            if avg_us:
                e["hbm_tb_per_s"] = round((fb + wb) / avg_us / 1e6, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE", (0, 0))[0]:
                e["mfma_busy_fraction_of_active_cycles"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024 * cs["GRBM_GUI_ACTIVE"][0] / 8), 4)
            traffic[k] = e
        s = {c: round(v[0], 1) for c, v in cs.items() if c.startswith("SQ_") and c != "SQ_VALU_MFMA_BUSY_CYCLES"}
        if s:
            wc = s.get("SQ_WAVE_CYCLES") or 0
            if wc:
                s["fractions_of_wave_cycles"] = {c: round(x / wc, 3) for c, x in s.items() if c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_BUSY"))}
            sq[k] = s
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    order = sorted(traffic, key=lambda k: -(traffic[k]["share_of_kernel_time"] or 0))
    doc = {"units": "bytes per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE); avg_launch_us from the --stats pass of the same command",
           "step": step_totals(traffic, dur), "kernels": {k: traffic[k] for k in order}}
    with open(os.path.join(root, "gpurun_out", f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    with open(os.path.join(root, "gpurun_out", f"{tag}_sq_counters.json"), "w") as f:
        json.dump({k: sq[k] for k in order if k in sq}, f, indent=1)
    for k in order[:14]:
        e = traffic[k]
        print(f"{k[:70]:70s} {e['avg_launch_us']} us  {e['hbm_bytes_per_launch'] / 1e6:8.1f} MB  {e.get('hbm_tb_per_s')} TB/s  mfma {e.get('mfma_busy_fraction_of_active_cycles')}  share {e['share_of_kernel_time']}")


if __name__ == "__main__":
    main()
