#!/usr/bin/env python3
"""How long does the HOST spend inside one replay of the captured training step, next to the step's device time?

    python tools/graph_host_time.py [--batch 128] [--segments 1]

If hipGraphLaunch returns in well under the step's device time the launch is pre-baked; if it takes a large share of it, the runtime
enqueues the graph's kernel nodes one by one at launch time and chains that come late in its enqueue order start late on the GPU
whatever the graph's dependencies allow (tools/timeline.py shows the HRModule branches running one after the other)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--segments", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import mindpose_amd as mp
    from mindpose_amd.utils import AdamWeightDecay, DynamicLossScaleManager, GraphedTrainStep
    dev = torch.device("cuda:0")
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(dev).train()
    mp.models.auto_mixed_precision(net, "O2")
    scaler = DynamicLossScaleManager()
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    gen = torch.Generator(device="cpu").manual_seed(1000)
    n = a.batch
    image = torch.randn(n, 3, 256, 192, generator=gen).to(dev)
    kp = torch.empty(n, 17, 3)
    kp[..., 0] = torch.rand(n, 17, generator=gen) * 232 - 20
    kp[..., 1] = torch.rand(n, 17, generator=gen) * 296 - 20
    kp[..., 2] = (torch.rand(n, 17, generator=gen) < 0.7).float()
    t0_, w0_ = tgt(kp.to(dev))
    step = GraphedTrainStep(nwl, opt, (image, t0_, w0_), loss_scale_manager=scaler, segments=a.segments)
    for _ in range(3):
        step.replay(exchange=False)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step.replay(exchange=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    host.sort(); total.sort()
    print(f"replay(): host {host[len(host) // 2]:.2f} ms inside the call, {total[len(total) // 2]:.2f} ms until the device is done "
          f"(median of {a.reps}; batch {n}, {len(step.graphs)} graph(s))")


if __name__ == "__main__":
    main()
