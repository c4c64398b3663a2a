"""GPU parity of the training-side kernels (BatchNorm train fwd/bwd, conv data/weight gradients, exchange-unit sum
backward, AdamWeightDecay) and of one full HRNet-W32 training step, against torch-CPU autograd over the oracle graph.
fp32 everywhere; tolerances are relative to each tensor's scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import mindpose_amd as mp  # noqa: E402
from mindpose_amd.models import train_ops as T  # noqa: E402
from mindpose_amd.models.layers import BatchNorm2d, Conv2d  # noqa: E402
from oracle import nets as onets  # noqa: E402

DEV = torch.device("cuda:0")


def _rel(got, ref):
    return float((got.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-20))


@pytest.mark.parametrize("shape,relu,res", [((4, 8, 16, 12), True, True), ((3, 32, 8, 6), True, False),
                                            ((5, 5, 7, 3), False, True), ((64, 17, 4, 4), False, False)])
def test_bn_train_fwd_bwd_vs_torch(shape, relu, res):
    g = torch.Generator().manual_seed(sum(shape))
    n, c, h, w = shape
    z = torch.randn(shape, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    r = torch.randn(shape, generator=g) if res else None
    dy = torch.randn(shape, generator=g)
    mm, mv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    # torch reference
    zt, gt, bt = z.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rt = r.clone().requires_grad_() if res else None
    rm, rv = mm.clone(), mv.clone()
    y_ref = F.batch_norm(zt, rm, rv, gt, bt, training=True, momentum=0.1, eps=1e-5)
    if res:
        y_ref = y_ref + rt
    if relu:
        y_ref = F.relu(y_ref)
    y_ref.backward(dy)
    # HIP
    zd, gd, bd = z.to(DEV).requires_grad_(), gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    rd = r.to(DEV).requires_grad_() if res else None
    mmd, mvd = mm.to(DEV), mv.to(DEV)
    y = T.BatchNormActFn.apply(zd, gd, bd, rd, mmd, mvd, relu)
    y.backward(dy.to(DEV))
    assert _rel(y.detach(), y_ref.detach()) < 1e-5
    assert _rel(zd.grad, zt.grad) < 1e-4
    assert _rel(gd.grad, gt.grad) < 1e-4 and _rel(bd.grad, bt.grad) < 1e-4
    if res:
        assert _rel(rd.grad, rt.grad) < 1e-6
    assert _rel(mmd, rm) < 1e-5 and _rel(mvd, rv) < 1e-5  # moving statistics (unbiased variance convention)
    # determinism
    zd2 = z.to(DEV).requires_grad_()
    y2 = T.BatchNormActFn.apply(zd2, gd.detach(), bd.detach(), rd.detach() if res else None, mm.to(DEV), mv.to(DEV), relu)
    assert torch.equal(y2, y)


CONV_TRAIN_CASES = [
    # n, cin, cout, k, s, h, w, bias
    (2, 32, 32, 3, 1, 16, 12, False),
    (3, 64, 32, 1, 1, 8, 8, False),
    (2, 32, 64, 3, 2, 16, 12, False),
    (2, 16, 40, 1, 2, 8, 6, False),
    (2, 3, 64, 3, 2, 32, 24, False),
    (4, 32, 17, 1, 1, 16, 12, True),
    (1, 5, 7, 3, 1, 9, 7, False),
    (2, 256, 256, 3, 1, 8, 6, False),
    (8, 32, 32, 3, 1, 64, 48, False),
    (2, 64, 64, 3, 2, 128, 96, False),   # stem conv2: the wide stride-2 layer (pipelined wgrad with one workgroup per CU)
]


@pytest.mark.parametrize("case", CONV_TRAIN_CASES, ids=[f"{c[1]}to{c[2]}_k{c[3]}s{c[4]}_{c[5]}x{c[6]}" for c in CONV_TRAIN_CASES])
def test_conv_fwd_dgrad_wgrad_vs_torch(case):
    n, cin, cout, k, s, h, w, bias = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(), wt.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    ref = F.conv2d(xr, wr, br, stride=s, padding=k // 2)
    dz = torch.randn(ref.shape, generator=g)
    ref.backward(dz)
    xd, wd = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if bias else None
    z = T.Conv2dFn.apply(xd, wd, bd, s, k // 2)
    z.backward(dz.to(DEV))
    assert _rel(z.detach(), ref.detach()) < 2e-5
    assert _rel(xd.grad, xr.grad) < 2e-5
    assert _rel(wd.grad, wr.grad) < 5e-5
    if bias:
        assert _rel(bd.grad, br.grad) < 1e-5
    # weight-gradient determinism (slab reduction, no atomics)
    xd2, wd2 = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_()
    T.Conv2dFn.apply(xd2, wd2, None, s, k // 2).backward(dz.to(DEV))
    assert torch.equal(wd2.grad, wd.grad)


@pytest.mark.parametrize("scales", [[1], [1, 2], [2, 4, 8], [1, 1, 2]])
def test_fuse_sum_fwd_bwd_vs_torch(scales):
    g = torch.Generator().manual_seed(len(scales) * 7)
    n, c, h, w = 2, 6, 16, 24
    base = torch.randn(n, c, h, w, generator=g)
    ts = [torch.randn(n, c, h // s, w // s, generator=g) for s in scales]
    dy = torch.randn(n, c, h, w, generator=g)
    br = base.clone().requires_grad_()
    tr = [t.clone().requires_grad_() for t in ts]
    acc = br
    for t, s in zip(tr, scales):
        acc = acc + (F.interpolate(t, size=(h, w), mode="nearest") if s > 1 else t)
    ref = F.relu(acc)
    ref.backward(dy)
    bd = base.to(DEV).requires_grad_()
    td = [t.to(DEV).requires_grad_() for t in ts]
    out = T.fuse_sum(bd, list(zip(td, scales)))
    out.backward(dy.to(DEV))
    assert torch.equal(out.detach().cpu(), ref.detach())
    assert _rel(bd.grad, br.grad) < 1e-6
    for a, b_ in zip(td, tr):
        assert _rel(a.grad, b_.grad) < 1e-5


def test_adamw_no_bias_correction():
    lib = mp._lib.load()
    g = torch.Generator().manual_seed(1)
    n = 100003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v = torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.01
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-6, 0.05
    # MindSpore keeps beta1/beta2/eps/lr as fp32 tensors, so (1 - beta) is formed in fp32
    tb1, tb2, one = torch.tensor(b1), torch.tensor(b2), torch.tensor(1.0)
    m_ref = tb1 * m + (one - tb1) * gr
    v_ref = tb2 * v + (one - tb2) * gr * gr
    p_ref = p - torch.tensor(lr) * (m_ref / (v_ref.sqrt() + torch.tensor(eps)) + torch.tensor(wd) * p)
    pd, gd, md, vd = (t.to(DEV).contiguous() for t in (p, gr, m, v))
    mp._lib.check(lib.mp_adamw_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, lr, b1, b2, eps, wd, None),
                  "adamw")
    assert _rel(pd, p_ref) < 1e-6 and _rel(md, m_ref) < 1e-6 and _rel(vd, v_ref) < 1e-6


def test_hrnet_w32_training_step_vs_oracle_autograd():
    """NetWithLoss in training mode: loss and EVERY parameter gradient of HRNet-W32 + head vs torch-CPU autograd of the
    oracle graph (batch-statistics BatchNorm), plus the moving statistics after the step.

    ~300 layers of ReLU + batch-statistics BatchNorm make a few deep gradients ill-conditioned (one ReLU mask flip
    changes an element's gradient by O(1)): torch-CPU fp32 itself is 5e-2 away from an fp64 run on the worst tensor.
    The yardstick is therefore an fp64 oracle, and the HIP path must be as close to it as fp32 torch-CPU is."""
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0)

    def leaf_params(dtype):
        d = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in net.state_dict().items()}
        for k, v in d.items():
            if v.dtype.is_floating_point and not k.endswith(("moving_mean", "moving_variance")):
                v.requires_grad_()
        return d

    p32, p64 = leaf_params(torch.float32), leaf_params(torch.float64)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 3, 128, 96, generator=g)
    kp = torch.rand(4, 17, 3, generator=g) * torch.tensor([96.0, 128.0, 2.0])
    net = net.to(DEV).train()
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[96, 128], heatmap_size=[24, 32]), sigma=2.0)
    target, weight = tgt(kp.to(DEV))
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    loss = nwl(x.to(DEV), target, weight)
    loss.backward()

    def oracle_step(params, dtype):
        out = onets.net_forward_train(params, x.to(dtype), "hrnet_w32", "hrnet_head")
        ref_loss = (((out - target.cpu().to(dtype)) ** 2) * weight.cpu().to(dtype)[..., None, None]).mean()
        ref_loss.backward()
        return float(ref_loss.detach())

    l32, l64 = oracle_step(p32, torch.float32), oracle_step(p64, torch.float64)
    assert abs(float(loss.detach()) - l64) <= 1e-6 * abs(l64)

    def rel64(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))

    e_hip = {n: rel64(p.grad, p64[n].grad) for n, p in net.named_parameters()}
    e_cpu = {n: rel64(p32[n].grad, p64[n].grad) for n in e_hip}
    assert all(p.grad is not None for p in net.parameters())
    med_hip, med_cpu = float(np.median(list(e_hip.values()))), float(np.median(list(e_cpu.values())))
    worst_hip, worst_cpu = max(e_hip.values()), max(e_cpu.values())
    print(f"gradient error vs fp64 oracle: HIP median {med_hip:.2e} worst {worst_hip:.2e}; "
          f"torch-CPU fp32 median {med_cpu:.2e} worst {worst_cpu:.2e}; loss {float(loss.detach())} / {l32} / {l64}")
    assert med_hip < max(2 * med_cpu, 1e-4)
    assert worst_hip < max(2 * worst_cpu, 1e-3)
    # well-conditioned tensors (last layers of the backward) are tight
    for name in ("head.head.weight", "head.head.bias", "backbone.stage4.2.fuse_layers.0.3.0.weight"):
        assert e_hip[name] < 5e-5, (name, e_hip[name])
    sd = net.state_dict()
    for name in ("backbone.bn1.moving_mean", "backbone.stage4.2.branches.3.3.bn2.moving_variance"):
        assert rel64(sd[name], p64[name].detach()) < 1e-4, name


def test_optimizer_step_reduces_loss_and_matches_reference_update():
    """AdamWeightDecay wrapper (flat arenas, decay / no-decay groups) + gradient arena on one GPU: first update equals
    the reference formula on every parameter, and a few steps reduce the loss."""
    from mindpose_amd.utils import AdamWeightDecay
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True)
    for k, v in net.named_parameters():  # flattening must not change values or names
        assert torch.equal(v.detach(), before[k])
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    kp = (torch.rand(4, 17, 3, generator=g) * torch.tensor([64.0, 64.0, 2.0])).to(DEV)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)
    losses = []
    for it in range(4):
        opt.zero_grad()
        target, weight = tgt(kp)
        loss = nwl(x, target, weight)
        loss.backward()
        if it == 0:
            grads = {k: v.grad.detach().clone() for k, v in net.named_parameters()}
        opt.step()
        if it == 0:
            one, b1, b2 = torch.tensor(1.0), torch.tensor(0.9), torch.tensor(0.999)
            for k, v in net.named_parameters():
                gk, pk = grads[k].cpu(), before[k].cpu()
                m, vv = (one - b1) * gk, (one - b2) * gk * gk
                wd = 0.0 if k.endswith(("beta", "gamma", "bias")) else 0.05
                ref = pk - torch.tensor(1e-3) * (m / (vv.sqrt() + torch.tensor(1e-6)) + torch.tensor(wd) * pk)
                assert _rel(v.detach(), ref) < 1e-5, k
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    assert opt.global_step == 4


def test_create_optimizer_and_scheduler_drive_the_native_adamw():
    from mindpose_amd.optim import create_optimizer
    from mindpose_amd.scheduler import create_lr_scheduler
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
    sched = create_lr_scheduler("warmup_multi_step_decay", lr=1e-3, total_epochs=2, steps_per_epoch=4, warmup=2, milestones=[2])
    opt = create_optimizer(net, name="adamw", learning_rate=sched, weight_decay=0.05, filter_bias_and_bn=True)
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    x = torch.randn(2, 3, 64, 64, device=DEV)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)
    target, weight = tgt(torch.rand(2, 17, 3, device=DEV) * 60)
    lrs = []
    for _ in range(6):
        opt.zero_grad()
        nwl(x, target, weight).backward()
        opt.step()
        lrs.append(opt.lr)
    assert lrs == [sched(i) for i in range(6)] and lrs[0] == 0.0 and lrs[2] == 1e-3 and lrs[4] == 1e-3 * 0.1
    # weight decay only through the decay / no-decay grouping, as the reference wires it
    assert create_optimizer(net, "adamw", 1e-3, weight_decay=0.05, filter_bias_and_bn=False).weight_decay == 0.0
    with pytest.raises(ValueError, match="Unkown components"):
        create_optimizer(net, name="lamb")


class _TwoParam(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.weight = torch.nn.Parameter(torch.randn(37, 5, generator=g))
        self.bias = torch.nn.Parameter(torch.randn(19, generator=g))


@pytest.mark.parametrize("name,kwargs", [("adam", {}), ("sgd", dict(momentum=0.9, dampening=0.1)), ("sgd", dict(momentum=0.9, nesterov=True)),
                                         ("sgd", {}), ("momentum", dict(momentum=0.9)), ("momentum", dict(momentum=0.8, use_nesterov=True)),
                                         ("adagrad", {})])
def test_other_registered_optimizers_vs_their_documented_update_rules(name, kwargs):
    """create_optimizer resolves every name the reference registers (optim_factory.py:9-14); four steps of each on a tiny
    module against the documented update rule evaluated in fp64 (static loss scale 8, L2 decay on the weight only)."""
    from mindpose_amd.optim import create_optimizer
    net = _TwoParam().to(DEV)
    lr, wd, ls = 0.05, 0.01, 8.0
    opt = create_optimizer(net, name=name, learning_rate=lr, weight_decay=wd, filter_bias_and_bn=True, loss_scale=ls, overlap=False, **kwargs)
    ref = {k: v.detach().double().cpu() for k, v in net.named_parameters()}
    st = {k: [torch.full_like(v, 0.1 if name == "adagrad" else 0.0), torch.zeros_like(v)] for k, v in ref.items()}
    g = torch.Generator().manual_seed(9)
    for t in range(1, 5):
        grads = {k: torch.randn(v.shape, generator=g) for k, v in ref.items()}
        opt.zero_grad()
        for k, prm in net.named_parameters():
            prm.grad.copy_(grads[k].to(DEV))  # p.grad are views of the optimizer's gradient arena
        opt.step()
        for k in ref:
            gi = grads[k].double() / ls + (wd if k == "weight" else 0.0) * ref[k]
            s1, s2 = st[k]
            if name == "adam":
                s1.mul_(0.9).add_(0.1 * gi)
                s2.mul_(0.999).add_(0.001 * gi * gi)
                ref[k] = ref[k] - lr * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t) * s1 / (s2.sqrt() + 1e-8)
            elif name == "sgd":
                mom, damp, nest = kwargs.get("momentum", 0.0), kwargs.get("dampening", 0.0), kwargs.get("nesterov", False)
                d = gi
                if mom:
                    s1.copy_(gi if t == 1 else mom * s1 + (1 - damp) * gi)
                    d = gi + mom * s1 if nest else s1
                ref[k] = ref[k] - lr * d
            elif name == "momentum":
                mom = kwargs["momentum"]
                s1.mul_(mom).add_(gi)
                ref[k] = ref[k] - lr * (gi + mom * s1 if kwargs.get("use_nesterov") else s1)
            else:
                s1.add_(gi * gi)
                ref[k] = ref[k] - lr * gi / s1.sqrt()
    for k, prm in net.named_parameters():
        assert torch.allclose(prm.detach().cpu().double(), ref[k], rtol=2e-5, atol=2e-6), (name, k)


def test_deconv_fp32_autograd_vs_torch():
    from mindpose_amd.models import train_ops as T
    g = torch.Generator().manual_seed(9)
    for n, cin, cout, h, w in ((3, 64, 32, 12, 10), (4, 2048, 256, 4, 3), (4, 256, 256, 8, 6)):  # + the SimpleBaseline head's shapes
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
        dy = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
        xt, wtt = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        ref = F.conv_transpose2d(xt, wtt, None, stride=2, padding=1)
        ref.backward(dy)
        xd, wd = x.to(DEV).requires_grad_(True), wt.to(DEV).requires_grad_(True)
        y = T.DeconvFn.apply(xd, wd)
        y.backward(dy.to(DEV))
        assert _rel(y.detach(), ref.detach()) < 2e-5 and _rel(xd.grad, xt.grad) < 2e-5 and _rel(wd.grad, wtt.grad) < 2e-5


def test_simplebaseline_r50_fp32_training_step_vs_oracle():
    """SimpleBaseline-ResNet50 in fp32: loss and every parameter gradient against an fp64 oracle run; the HIP path must be as
    close to it as torch-CPU fp32 is (same yardstick as the HRNet step: the deep ReLU + batch-statistics graph is
    ill-conditioned, here with only 4 x 12 positions per channel in the last stage)."""
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("resnet50", "simple_baseline_head"), seed=0)

    def leaf_params(dtype):
        d = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in net.state_dict().items()}
        for k, v in d.items():
            if v.dtype.is_floating_point and not k.endswith(("moving_mean", "moving_variance")):
                v.requires_grad_()
        return d

    p32, p64 = leaf_params(torch.float32), leaf_params(torch.float64)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(4, 3, 128, 96, generator=g)
    target = torch.rand(4, 17, 32, 24, generator=g)
    weight = (torch.rand(4, 17, generator=g) > 0.3).float()

    def oracle_step(params, dtype):
        out = onets.net_forward_train(params, x.to(dtype), "resnet50", "simple_baseline_head")
        ref_loss = (((out - target.to(dtype)) ** 2) * weight.to(dtype)[..., None, None]).mean()
        ref_loss.backward()
        return float(ref_loss.detach())

    l32, l64 = oracle_step(p32, torch.float32), oracle_step(p64, torch.float64)
    net = net.to(DEV).train()
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    loss = nwl(x.to(DEV), target.to(DEV), weight.to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - l64) <= 1e-5 * abs(l64)

    def rel64(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))

    e_hip = {n: rel64(p.grad, p64[n].grad) for n, p in net.named_parameters()}
    e_cpu = {n: rel64(p32[n].grad, p64[n].grad) for n in e_hip}
    med_hip, med_cpu = float(np.median(list(e_hip.values()))), float(np.median(list(e_cpu.values())))
    worst_hip, worst_cpu = max(e_hip.values()), max(e_cpu.values())
    print(f"SimpleBaseline fp32 gradient error vs fp64 oracle: HIP median {med_hip:.2e} worst {worst_hip:.2e}; torch-CPU fp32 median "
          f"{med_cpu:.2e} worst {worst_cpu:.2e}; deconv6 HIP {e_hip['head.deconv_layer.6.weight']:.2e} CPU "
          f"{e_cpu['head.deconv_layer.6.weight']:.2e}; loss {float(loss.detach())} / {l32} / {l64}")
    top = sorted(e_hip, key=e_hip.get, reverse=True)[:6]
    print({n: (f"{e_hip[n]:.2e}", f"{e_cpu[n]:.2e}") for n in top})
    # The bulk of the tensors (median, 90th percentile) must be as close to fp64 as torch-CPU fp32 is.  The WORST tensor is not a
    # stable quantity: the last stage has 4 x 12 positions per channel behind 50 fp32 layers, and where exactly one ReLU mask flips
    # differs between any two fp32 implementations - a single flip moves a BatchNorm beta gradient of that stage by ~10 % of its
    # largest entry (seen: 0.06 - 0.11 across summation orders of the same kernels, all of which match fp64 to 1e-6 layer by layer:
    # test_bn_train_fwd_bwd_vs_torch, test_deconv_fp32_autograd_vs_torch, test_conv_fwd_dgrad_wgrad_vs_torch) - so it only gets a loose bound.
    p90_hip, p90_cpu = float(np.percentile(list(e_hip.values()), 90)), float(np.percentile(list(e_cpu.values()), 90))
    assert med_hip < max(2 * med_cpu, 1e-4) and p90_hip < max(2 * p90_cpu, 2e-2) and worst_hip < max(2 * worst_cpu, 0.3)
    for name in ("head.final_layer.weight", "head.final_layer.bias"):
        assert e_hip[name] < 5e-5, (name, e_hip[name])


def test_batched_fp32_weight_pack_equals_individual_packs():
    """mp_conv_pack_weight_batch == one mp_conv_pack_weight per job, bit for bit, over every packing mode."""
    import numpy as np
    from mindpose_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(22)
    specs = [((40, 24, 3, 3), 40, 24, 3, 0, 0, 0), ((17, 32, 1, 1), 17, 32, 1, 0, 0, 0), ((40, 24, 3, 3), 24, 40, 3, 2, 0, 0),
             ((40, 24, 3, 3), 24, 40, 2, 3, 1, 0), ((24, 40, 4, 4), 40, 24, 2, 1, 0, 1), ((24, 40, 4, 4), 24, 40, 2, 4, 1, 1),
             ((64, 3, 7, 7), 64, 3, 7, 0, 0, 0)]
    ws = [torch.randn(sh, generator=g).to(DEV) for sh, *_ in specs]
    single, batch = [], []
    arr = (T._PackJob * len(specs))()
    first = np.zeros(len(specs) + 1, dtype=np.uint32)
    for i, (w, (_, cout, cin, k, mode, py, px)) in enumerate(zip(ws, specs)):
        nb = lib.mp_conv_packed_weight_bytes(cout, cin, k, k)
        a, b = torch.full((nb // 4,), 7.0, device=DEV), torch.full((nb // 4,), 9.0, device=DEV)
        _lib.check(lib.mp_conv_pack_weight(_lib.ptr(w), _lib.ptr(a), cout, cin, k, k, mode, py, px, _lib.stream()), "pack")
        arr[i] = T._PackJob(w.data_ptr(), b.data_ptr(), cout, cin, k, k, mode, py, px, 0)
        first[i + 1] = first[i] + ((cin + 3) // 4 * 4 * k * k * ((cout + 15) // 16 * 4) + 255) // 256
        single.append(a)
        batch.append(b)
    jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    first_dev = torch.from_numpy(first.view(np.int32)).to(DEV)
    _lib.check(lib.mp_conv_pack_weight_batch(_lib.ptr(jobs_dev), _lib.ptr(first_dev), len(specs), int(first[-1]), _lib.stream()), "batch")
    torch.cuda.synchronize()
    for a, b, sp in zip(single, batch, specs):
        assert torch.equal(a, b), sp


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("k", [2, 3, 4, 6])
def test_fan_out_sums_consumer_gradients_in_one_launch(half, k):
    """T.fan_out(x, k): k handles on x; the backward pass returns the sum of the k consumers' gradients (mp_sum_tensors: fp32
    arithmetic, for fp16 one rounding per running sum of up to four operands)."""
    from mindpose_amd.models import train_ops as T
    g = torch.Generator().manual_seed(k)
    shape = (3, 4, 6, 10, 8) if half else (3, 32, 6, 10)
    x = torch.randn(shape, generator=g).to(DEV)
    x = (x.half() if half else x).requires_grad_(True)
    ws = [torch.randn(shape, generator=g).to(DEV).to(x.dtype) for _ in range(k)]
    hs = T.fan_out(x, k)
    assert len(hs) == k and all(h.data_ptr() == x.data_ptr() for h in hs)
    sum((h * w).sum() for h, w in zip(hs, ws)).backward()
    ref = torch.stack([w.double() for w in ws]).sum(0)
    tol = 2e-3 if half else 1e-6
    assert float((x.grad.double() - ref).abs().max()) <= tol * float(ref.abs().max())
    # a consumer that produces no gradient is skipped
    x2 = x.detach().clone().requires_grad_(True)
    a, b, c = T.fan_out(x2, 3)
    ((a * ws[0]).sum() + (c * ws[1]).sum() + b.detach().sum()).backward()
    ref2 = ws[0].double() + ws[1].double()
    assert float((x2.grad.double() - ref2).abs().max()) <= tol * float(ref2.abs().max())
