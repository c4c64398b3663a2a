#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/<tag>_pmc_traffic.json (the file bench.py's roofline.traffic reads).

    python tools/pmc_summary.py <tag> <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass>

Each directory is a `rocprofv3 --kernel-trace --pmc <COUNTER> -d <dir> -- python3 bench.py ...` output tree (separate
passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Counters are KiB per dispatch; FETCH_SIZE is doubled
(gfx950 reports half of a wide coalesced read), WRITE_SIZE is exact for >= 8-byte stores.  Keys are the kernel names
bench.py uses: the template head <KS,S,PS,CS,WAVES_P,WAVES_C> plus "/occN" for the light / multi-tile builds.
"""
import csv
import glob
import json
import os
import re
import sys


def key_of(name):
    m = re.search(r"(conv_f16_ws_kernel|conv_f16_wreg_kernel)<([^>]*)>", name)
    if m:  # ws: <k-steps, cout tiles per wave, pixel-splitting waves, pixel tiles, occupancy | statistics mode, residual>; wreg: <KS, S,
        # pixel tiles, cout tiles per wave, pixel-splitting waves, occupancy | statistics mode>: the head is bench.py's kernel_name
        keep = 5 if m.group(1) == "conv_f16_ws_kernel" else 6
        return f"{m.group(1)}<{','.join(a.strip() for a in m.group(2).split(',')[:keep])}>"
    m = re.search(r"(expand_reduce_f32_w8_kernel|expand_reduce_f32_kernel|expand_reduce_f16_kernel|basicblock_f16_c64_kernel|basicblock_f16_v2_kernel|basicblock_f16_kernel|conv_wino_f32_kernel|conv1x1_f32_stream_kernel|conv1x1_f32_gemm_kernel)<([^>]*)>", name)
    if m:  # these names are used with their full template argument list
        return f"{m.group(1)}<{','.join(a.strip() for a in m.group(2).split(','))}>"
    m = re.search(r"(conv_mfma_kernel|conv_f16_mt_kernel|conv_f16_kernel)<([^>]*)>", name)
    if not m:
        m2 = re.search(r"([A-Za-z_0-9]+)(<|\()", name.replace("void ", "").replace("mp::", "").replace("(anonymous namespace)::", ""))
        return m2.group(1) if m2 else name
    kind, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    head = ",".join(args[:6])
    occ = args[-1]
    if kind == "conv_mfma_kernel":
        return f"{kind}<{head}>" + ("/occ3" if occ == "3" else "")
    if kind == "conv_f16_kernel":
        return f"{kind}<{head}>" + ("/occ3" if occ == "3" else "")
    return f"{kind}<{head}>/occ{occ}"


def collect(d, counter):
    per = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for path in files:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = key_of(row["Kernel_Name"])
                # one row per (dispatch, counter[, dimension]): sum the dimensions of a dispatch
                e = per.setdefault(k, {})
                did = row.get("Dispatch_Id", row.get("Correlation_Id"))
                e[did] = e.get(did, 0.0) + float(row["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}, {k: len(v) for k, v in per.items()}


def main():
    tag, d_fetch, d_write = sys.argv[1:4]
    d_mfma = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, nf = collect(d_fetch, "FETCH_SIZE")
    write, _ = collect(d_write, "WRITE_SIZE")
    mfma, gui = {}, {}
    if d_mfma:
        # SQ_VALU_MFMA_BUSY_CYCLES: MFMA-pipe busy cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE: active cycles
        # summed over the 8 XCDs (MI355X_MICROARCH.md).  busy fraction = busy / (1024 * active / 8)
        mfma, _ = collect(d_mfma, "SQ_VALU_MFMA_BUSY_CYCLES")
        gui, _ = collect(d_mfma, "GRBM_GUI_ACTIVE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fb = fetch.get(k, 0.0) * 1024 * 2
        wb = write.get(k, 0.0) * 1024
        kernels[k] = {"fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                      "hbm_bytes_per_launch": round(fb + wb), "dispatches": nf.get(k, 0)}
        if k in mfma and gui.get(k):
            kernels[k]["mfma_busy_cycles_per_launch"] = round(mfma[k])
            kernels[k]["gui_active_cycles_per_launch_per_xcd"] = round(gui[k] / 8)
            kernels[k]["mfma_busy_fraction_of_active_cycles"] = round(mfma[k] / (1024 * gui[k] / 8), 4)
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py (autotuner choices replayed "
                    "from MINDPOSE_TUNE_CACHE, so no trial launches are mixed in). KiB per dispatch; FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md, WRITE_SIZE as reported. Average HBM-side bytes per launch of each instantiation.",
           "kernels": kernels}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", f"{tag}_pmc_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(kernels), "kernels")


if __name__ == "__main__":
    main()
