#!/usr/bin/env python3
"""Package power and shader clock (rocm-smi, 4 samples per second) while a bench leg runs:
   python tools/power_sample.py [bench args]        e.g.  --workload hrnet_w32 --leg --no-roofline --steps 400
Prints the median / maximum power and the median clock over the samples taken while the GPU was busy (> 600 W)."""
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, stop = [], False


def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)["card0"]
            samples.append((float(d["Current Socket Graphics Package Power (W)"]), int(d["sclk clock speed:"].strip("()Mhz"))))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.25)


t = threading.Thread(target=sample)
t.start()
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
stop = True
t.join()
line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
if line:
    d = json.loads(line[-1])
    print("bench:", d.get("value"), d.get("unit"), d.get("ms_per_step"), "ms/step")
busy = [s for s in samples if s[0] > 600]
if busy:
    print(f"{len(busy)} busy samples of {len(samples)}: power median {statistics.median(p for p, _ in busy):.0f} W, max {max(p for p, _ in busy):.0f} W; "
          f"sclk median {statistics.median(c for _, c in busy)} MHz, min {min(c for _, c in busy)} MHz")
else:
    print("no busy samples", samples[:5])
