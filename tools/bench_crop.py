#!/usr/bin/env python3
"""Micro-benchmark of the fused crop kernel (warpAffine + Normalize + HWC2CHW): N boxes of one 480x640 frame -> the
network's [N,3,256,192] fp32 input.  Algorithmic bytes per crop = 3*256*192*4 written (589 824 B) + the <= 4 source
pixels per destination pixel that miss L2 (the source box itself, read once)."""
import json
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindpose_amd as mp  # noqa: E402
from tests.golden import recipes  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = dict(image_size=[192, 256], heatmap_size=[48, 64], flip_pairs=recipes.FLIP_PAIRS, upper_body_ids=list(range(11)),
           pixel_std=200.0, scale_padding=1.25)
rng = np.random.RandomState(0)
img = torch.from_numpy(rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)).to(dev)
boxes = np.stack([rng.uniform(0, 400, n), rng.uniform(0, 250, n), rng.uniform(60, 220, n), rng.uniform(120, 400, n)], 1).astype(np.float32)
c, s = mp.TopDownBoxToCenterScale(False, cfg).transform_batch(boxes)
aff = mp.TopDownAffine(False, cfg)
out = torch.empty(n, 3, 256, 192, device=dev)
mats = np.stack([aff.get_matrix(c[i], s[i], 0.0) for i in range(n)])
aff._launch([img], [0] * n, mats, True, out, aff.NORMALIZE_MEAN, aff.NORMALIZE_STD)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    aff._launch([img], [0] * n, mats, True, out, aff.NORMALIZE_MEAN, aff.NORMALIZE_STD)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / reps  # includes the three small host->device descriptor copies of _launch
import time
t0 = time.perf_counter()
for _ in range(20):
    aff.crop_batch(img, c, s, out=out)
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps({"crops": n, "launch_ms": round(ms, 4), "crops_per_s_launch": round(n / ms * 1e3), "write_GBps": round(n * 589824 / ms / 1e6, 1),
                  "crop_batch_ms_incl_host_matrices": round(full, 3), "crops_per_s_end_to_end": round(n / full * 1e3)}))
