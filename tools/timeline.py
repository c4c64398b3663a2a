#!/usr/bin/env python3
"""How full is the GPU during a step?  Reads a rocprofv3 --kernel-trace CSV, cuts it into steps at the first launch of a marker
kernel (once per step: e.g. `gaussian_target` for the training step, `to_c8_kernel` / `decode_kernel` for inference) and prints for
the last steps: wall time, sum of kernel durations, the time at least one kernel runs (union), the time exactly one / two / three+
run, the idle time between kernels and the largest gaps with the kernels around them.
   python tools/timeline.py <kernel_trace.csv> --marker gaussian_target [--steps 3]"""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", required=True)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--dump", default=None, help="write the LAST step's launches (start us, duration us, queue, grid, short name) to this CSV")
    ap.add_argument("--split", default=None, help="kernel substring: report the part of a step before / from its first launch separately")
    a = ap.parse_args()
    rows, extra = [], {}
    with open(a.trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
            extra[(int(r["Start_Timestamp"]), r["Kernel_Name"])] = (r.get("Queue_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    rows.sort()
    marks = []
    for s, e, name in rows:
        if a.marker in name and (not marks or s - marks[-1] > 2_000_000):  # one mark per step: launches of the marker 2 ms apart
            marks.append(s)
    if len(marks) < 2:
        raise SystemExit(f"marker {a.marker!r}: {len(marks)} occurrence(s)")
    if a.dump:
        t0, t1 = marks[-2], marks[-1]
        with open(a.dump, "w") as fh:
            fh.write("start_us,dur_us,queue,grid,wg,name\n")
            for s_, e_, n_ in rows:
                if t0 <= s_ < t1:
                    q, g, wg = extra.get((s_, n_), ("", "", ""))
                    short = n_.replace("void ", "").replace("mp::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
                    fh.write(f"{(s_ - t0) / 1e3:.2f},{(e_ - s_) / 1e3:.2f},{q},{g},{wg},\"{short}\"\n")
    out = []
    for k in range(max(0, len(marks) - 1 - a.steps), len(marks) - 1):
        t0, t1 = marks[k], marks[k + 1]
        ks = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
        ev = sorted([(s, 1) for s, e, n in ks] + [(min(e, t1), -1) for s, e, n in ks])
        depth, last, hist = 0, t0, {}
        for t, d in ev:
            hist[depth] = hist.get(depth, 0) + (t - last)
            depth += d
            last = t
        hist[depth] = hist.get(depth, 0) + (t1 - last)
        wall = t1 - t0
        busy = wall - hist.get(0, 0)
        # idle gaps: intervals with no kernel running
        gaps, depth, last_end, last_name = [], 0, None, None
        running = 0
        cur_end, cur_name = t0, "(step start)"
        for s, e, n in ks:
            if s > cur_end:
                gaps.append((s - cur_end, cur_name, n))
            if e > cur_end:
                cur_end, cur_name = e, n
        gaps.sort(reverse=True)
        rec = dict(step=k, kernels=len(ks), wall_ms=wall / 1e6, sum_ms=sum(e - s for s, e, n in ks) / 1e6, busy_ms=busy / 1e6,
                   idle_ms=hist.get(0, 0) / 1e6, one_ms=hist.get(1, 0) / 1e6, two_ms=hist.get(2, 0) / 1e6,
                   three_plus_ms=sum(v for d, v in hist.items() if d >= 3) / 1e6, gaps=len(gaps),
                   gaps_over_5us=sum(1 for g in gaps if g[0] > 5000), median_gap_us=(sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0),
                   top_gaps=[(round(g[0] / 1e3, 1), g[1][:60], g[2][:60]) for g in gaps[:6]])
        # who runs ALONE (the serial part of the step), by kernel family; and the concurrency before / from the split kernel
        alone = {}
        depth, last, names = 0, t0, []
        ev2 = sorted([(s, 1, n) for s, e, n in ks] + [(min(e, t1), -1, n) for s, e, n in ks], key=lambda v: (v[0], v[1]))
        split_t = next((s for s, e, n in ks if a.split and a.split in n), None)
        phase = {"before": {}, "from": {}}
        for t, d, n in ev2:
            if depth == 1 and names:
                fam = names[0].replace("void ", "").replace("mp::(anonymous namespace)::", "").replace("mp::", "").split("(")[0].split("<")[0][:48]
                alone[fam] = alone.get(fam, 0) + (t - last)
            if split_t is not None:
                ph = phase["before" if last < split_t else "from"]
                ph[min(depth, 3)] = ph.get(min(depth, 3), 0) + (t - last)
            if d == 1:
                names.append(n)
            else:
                names.remove(n)
            depth += d
            last = t
        rec["alone_top"] = [(k2, round(v / 1e6, 3)) for k2, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]]
        fams = {}
        for s_, e_, n_ in ks:
            f_ = n_.replace("void ", "").replace("mp::(anonymous namespace)::", "").replace("mp::", "").split("(")[0].split("<")[0][:48]
            c_ = fams.setdefault(f_, [0, 0])
            c_[0] += 1
            c_[1] += e_ - s_
        rec["families"] = [(k2, v[0], round(v[1] / 1e6, 3)) for k2, v in sorted(fams.items(), key=lambda kv: -kv[1][1])[:16]]
        if split_t is not None:
            rec["phases"] = {k2: {str(d2): round(v / 1e6, 3) for d2, v in sorted(ph.items())} for k2, ph in phase.items()}
        out.append(rec)
        print(f"step {k}: {rec['kernels']} kernels, wall {rec['wall_ms']:.2f} ms, sum of durations {rec['sum_ms']:.2f}, >= 1 running "
              f"{rec['busy_ms']:.2f} (idle {rec['idle_ms']:.2f} in {rec['gaps']} gaps, {rec['gaps_over_5us']} over 5 us, median "
              f"{rec['median_gap_us']:.1f} us), exactly 1: {rec['one_ms']:.2f}, 2: {rec['two_ms']:.2f}, 3+: {rec['three_plus_ms']:.2f}")
    for g in out[-1]["top_gaps"]:
        print("   gap", g)
    print("   running alone (ms):", out[-1]["alone_top"])
    print("   kernel families (launches, ms of duration):", out[-1]["families"])
    if "phases" in out[-1]:
        print("   ms with 0 / 1 / 2 / 3+ kernels running, before and from the first", a.split, ":", out[-1]["phases"])
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
