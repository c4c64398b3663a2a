"""Does a captured HRModule really run its branches concurrently?  Times hipGraph replays of ONE module's training forward
(stage 3, three branches) with the branches on side streams and on one stream, and each branch alone."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindpose_amd as mp  # noqa: E402
from mindpose_amd.models import train_ops as T  # noqa: E402
from mindpose_amd.models.backbones import hrnet as H  # noqa: E402

dev = torch.device("cuda:0")
net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(dev).train()
mp.models.auto_mixed_precision(net, "O2")
bb = net.backbone
N = 128
xs0 = [T.to_c8(torch.randn(N, c, h, w, device=dev)) for c, h, w in ((32, 64, 48), (64, 32, 24), (128, 16, 12))]


def timed(fn, tag):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per replay", flush=True)


with torch.no_grad():
    for mi in (0, 1):
        mod = bb.stage3[mi]
        for streams in (True, False):
            H.set_branch_streams(streams)
            timed(lambda: mod.train_forward(list(xs0)), f"stage3[{mi}] forward, branch streams {streams}")
    H.set_branch_streams(True)

    def two():
        ys = bb.stage3[0].train_forward(list(xs0))
        bb.stage3[1].train_forward(ys)
    timed(two, "stage3[0] + stage3[1], streams")
    H.set_branch_streams(False)
    mod = bb.stage3[1]
    for i in range(3):
        def one(i=i):
            x = xs0[i]
            for blk in mod.branches[i]:
                x = blk.train_forward(x)
        timed(one, f"branch {i} alone")
