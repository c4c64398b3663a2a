"""Host time of one replay of the inference plan (the native call that enqueues every launch) against its GPU time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindpose_amd as mp  # noqa: E402

amp = sys.argv[1] if len(sys.argv) > 1 else "O0"
dev = torch.device("cuda:0")
net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(dev).eval()
if amp != "O0":
    mp.models.auto_mixed_precision(net, amp)
x = torch.randn(128, 3, 256, 192, device=dev)
with torch.no_grad():
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(30):
        h0 = time.perf_counter()
        net(x)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 30
print(f"amp {amp}: wall {wall * 1e3:.3f} ms per step, host enqueue {sum(host) / len(host) * 1e3:.3f} ms per step (min {min(host) * 1e3:.3f})")
