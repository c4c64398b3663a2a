"""Per-tensor gradient comparison of the fused-BatchNorm step against the per-cell step (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindpose_amd as mp
from mindpose_amd.utils import AdamWeightDecay

DEV = torch.device("cuda:0")


def step(fuse, parts):
    os.environ["MINDPOSE_BN_FUSE"] = "1" if fuse else "0"
    os.environ["MINDPOSE_BN_FUSE_PARTS"] = str(parts)
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
    mp.models.auto_mixed_precision(net, "O2")
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
    g = torch.Generator().manual_seed(3)
    n, h, w = 3, 128, 96
    x = torch.randn(n, 3, h, w, generator=g).to(DEV)
    kp = (torch.rand(n, 17, 3, generator=g) * torch.tensor([float(w), float(h), 2.0])).to(DEV)
    target, weight = mp.TopDownGenerateTarget(config=dict(image_size=[w, h], heatmap_size=[w // 4, h // 4]), sigma=2.0)(kp)
    opt.zero_grad()
    loss = nwl(x, target, weight)
    (loss * 1024.0).backward()
    return float(loss.detach()), {k: p.grad.detach().clone().double().flatten() for k, p in net.named_parameters()}


l0, g0 = step(False, 7)
for parts in (1, 2, 4, 3, 7):
    l1, g1 = step(True, parts)
    cos = {k: float((g1[k] @ g0[k]) / (g1[k].norm() * g0[k].norm()).clamp_min(1e-300)) for k in g0}
    a0, a1 = torch.cat(list(g0.values())), torch.cat(list(g1.values()))
    worst = sorted(cos, key=cos.get)[:6]
    print(f"parts={parts}: loss {l1:.6f} vs {l0:.6f}; global cos {float((a0 @ a1) / (a0.norm() * a1.norm())):.6f}")
    for k in worst:
        print(f"    {k} {cos[k]:.5f} numel {g0[k].numel()}")
