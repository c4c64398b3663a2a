#!/usr/bin/env python3
"""Record which (launch shape, tile variant) pairs the tuner picks for the fp16 bench plans -> tests/golden/bench_plan_picks.json.

    python tools/dump_plan_picks.py [--out tests/golden/bench_plan_picks.json]        (on an MI355X)

Runs the three fp16 bench legs of bench.py (amp-O2 HRNet-W32 inference N = 128, config 5 = HRNet-W48 384x288 with the batched flip
test 2N = 128, the amp-O2 training step N = 128) as child processes with a fresh MINDPOSE_TUNE_CACHE each and turns the persisted
choice tables into {desc, variant, n_res, stats} records.  tests/test_f16_matrix_cpu.py::test_bench_plan_picks_are_still_served
asserts - host-only, through mp_f16_conv_supported - that every recorded pair is still accepted by the forced-variant entry, so a
kernel that drops out of the tuner's candidate set fails the suite instead of only slowing the bench.
"""
import argparse
import ast
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LEGS = {
    "hrnet_w32_infer_ampO2_n128": ["--workload", "hrnet_w32", "--amp", "O2", "--batch", "128"],
    "config5_hrnet_w48_384x288_flip_ampO2_n64": ["--workload", "hrnet_w48_384_udp_flip", "--amp", "O2", "--batch", "64"],
    "config3_hrnet_w32_train_ampO2_n128": ["--workload", "hrnet_w32_train", "--amp", "O2", "--batch", "128"],
}


def parse_key(key, fields):
    """repr(tuple) written by models/layers.py::tune_conv_variant -> dict, or None for keys of other tuners."""
    try:
        t = ast.literal_eval(key)
    except (ValueError, SyntaxError):
        return None
    if not isinstance(t, tuple) or len(t) < len(fields) + 4:
        return None
    desc = dict(zip(fields, t[:len(fields)]))
    res1, res2, _dev, half = t[len(fields):len(fields) + 4]
    rest = t[len(fields) + 4:]
    if not all(isinstance(v, int) for v in desc.values()) or not isinstance(half, bool):
        return None
    stats = 0
    if "stats" in rest:
        stats = int(rest[rest.index("stats") + 1])
    return dict(desc=desc, n_res=int(bool(res1)) + int(bool(res2)), half=half, stats=stats, wino="wino" in rest, pre="pre" in rest)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "bench_plan_picks.json"))
    args = ap.parse_args()
    from mindpose_amd import _lib
    fields = [f for f, _ in _lib.ConvDesc._fields_]
    picks, seen = [], set()
    for name, leg in LEGS.items():
        with tempfile.TemporaryDirectory() as tmp:
            cache = os.path.join(tmp, "tune.json")
            env = dict(os.environ, MINDPOSE_TUNE_CACHE=cache)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--leg", "--no-roofline", "--steps", "3", "--warmup", "2", *leg]
            print("running", name, file=sys.stderr, flush=True)
            subprocess.run(cmd, env=env, check=True, stdout=subprocess.DEVNULL)
            with open(cache) as fh:
                doc = json.load(fh)
        for key, variant in doc["choices"].items():
            rec = parse_key(key, fields)
            if rec is None or not rec["half"] or rec["pre"] or int(variant) < 0:
                continue
            ident = (tuple(rec["desc"].items()), int(variant), rec["n_res"], rec["stats"])
            if ident in seen:
                continue
            seen.add(ident)
            picks.append(dict(leg=name, desc=rec["desc"], variant=int(variant), n_res=rec["n_res"], stats=rec["stats"]))
    lib = _lib.load()
    doc = {"source": "tools/dump_plan_picks.py on one MI355X", "library": lib.mp_version().decode(), "f16": picks}
    with open(args.out, "w") as fh:
        json.dump(doc, fh, indent=0, sort_keys=True)
        fh.write("\n")
    print(f"{len(picks)} fp16 picks -> {args.out}", file=sys.stderr)


if __name__ == "__main__":
    main()
