#!/usr/bin/env python3
"""Overfit one fixed synthetic batch: loss must fall steadily in fp32 and under amp O2 (gradients, optimizer, loss scaling and
the graph-captured step all in the loop).  python tools/train_smoke.py [O0|O2] [steps] [backbone head]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindpose_amd as mp  # noqa: E402
from mindpose_amd.utils import AdamWeightDecay, DynamicLossScaleManager, GraphedTrainStep  # noqa: E402

amp = sys.argv[1] if len(sys.argv) > 1 else "O2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bb, hd = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("hrnet_w32", "hrnet_head")
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = mp.init_synthetic(mp.create_network(bb, hd), seed=0).to(dev).train()
mgr = None
if amp != "O0":
    mp.models.auto_mixed_precision(net, amp)
    mgr = DynamicLossScaleManager()
nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
g = torch.Generator().manual_seed(1)
n = 32
x = torch.randn(n, 3, 256, 192, generator=g).to(dev)
kp = torch.empty(n, 17, 3)
kp[..., 0] = torch.rand(n, 17, generator=g) * 180 + 6
kp[..., 1] = torch.rand(n, 17, generator=g) * 240 + 8
kp[..., 2] = 1.0
tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
target, weight = tgt(kp.to(dev))
step = GraphedTrainStep(nwl, opt, (x, target, weight), loss_scale_manager=mgr)
hist = []
for it in range(steps):
    loss = float(step(x, target, weight).detach())
    hist.append(loss)
    if it % 10 == 0 or it == steps - 1:
        print(f"step {it:4d} loss {loss:.6f} loss_scale {mgr.loss_scale if mgr else None} skipped {mgr.skipped_steps if mgr else 0}", flush=True)
assert all(map(lambda v: v == v, hist)), "NaN loss"
assert hist[-1] < 0.5 * hist[0], (hist[0], hist[-1])
print("ok: loss", hist[0], "->", hist[-1])
