"""fp16 (amp O2) training kernels against torch-CPU autograd on fp16-rounded operands (fp32 accumulation on both sides),
through the C ABI.  Tolerances: fp16 outputs one ulp (2^-9 relative + 1e-4 of scale); fp32 weight gradients 1e-4 of scale
(accumulation order only)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from mindpose_amd import _lib  # noqa: E402
from mindpose_amd.models.layers import ActC8  # noqa: E402
from tests import f16_matrix as fm  # noqa: E402

DEV = torch.device("cuda:0")
LIB = _lib.load()


def _h(x):
    return x.half().float()


def _to_c8(x):
    n, c, h, w = x.shape
    a = ActC8(n, c, h, w, DEV)
    _lib.check(LIB.mp_f16_to_c8(_lib.ptr(x.to(DEV).contiguous()), _lib.ptr(a), n, c, h, w, _lib.stream()), "to_c8")
    return a


def _from_c8(a):
    n, c, h, w = a.shape
    out = torch.empty(n, c, h, w, device=DEV)
    _lib.check(LIB.mp_f16_from_c8(_lib.ptr(a), _lib.ptr(out), n, c, h, w, _lib.stream()), "from_c8")
    return out.cpu()


def _close16(got, ref, what=""):
    tol = ref.abs() * 2.0 ** -9 + 2e-4 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} beyond tolerance, max diff {float((got - ref).abs().max())}"


def _desc(n, cin, h, w, cout, k, s, pad, ho, wo, oh=None, ow=None, mul=1, oy=0, ox=0, pt=None, pl=None):
    return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=s, pad_top=pad if pt is None else pt,
                         pad_left=pad if pl is None else pl, conv_h=ho, conv_w=wo, out_h=oh or ho, out_w=ow or wo, out_mul=mul,
                         out_rep=1, out_off_y=oy, out_off_x=ox, relu=0, flags=0)


WGRAD_CASES = [
    # n, cin, cout, k, s, h, w
    (3, 32, 32, 3, 1, 64, 48),
    (4, 64, 64, 3, 1, 32, 24),
    (5, 128, 128, 3, 1, 16, 12),
    (6, 256, 256, 3, 1, 8, 6),
    (2, 3, 64, 3, 2, 64, 48),
    (2, 3, 64, 3, 2, 256, 192),   # full-size stem: single-buffered wide stride-2 tile
    (2, 64, 64, 3, 2, 128, 96),
    (3, 64, 128, 3, 2, 32, 24),
    (3, 48, 96, 3, 2, 24, 18),
    (2, 64, 256, 1, 1, 32, 24),
    (2, 256, 64, 1, 1, 16, 12),
    (3, 32, 17, 1, 1, 64, 48),
    (2, 40, 24, 3, 1, 9, 7),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_f16_vs_torch(case):
    n, cin, cout, k, s, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    x = torch.randn(n, cin, h, w, generator=g)
    dz = torch.randn(n, cout, ho, wo, generator=g)
    wt = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(_h(x), wt, None, stride=s, padding=pad).backward(_h(dz))
    ref = wt.grad * 0.5
    d = _desc(n, cin, h, w, cout, k, s, pad, ho, wo)
    nb = LIB.mp_f16_conv_wgrad_workspace_bytes(ctypes.byref(d))
    assert nb > 0
    ws = torch.empty(nb // 4, device=DEV)
    dw = torch.full((cout, cin, k, k), float("nan"), device=DEV)
    xa, dza = _to_c8(x), _to_c8(dz)  # keep the device tensors alive: the ABI only sees raw pointers
    _lib.check(LIB.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(xa), _lib.ptr(dza), _lib.ptr(dw), 0.5, 0, _lib.ptr(ws), nb,
                                     _lib.stream()), "wgrad")
    got = dw.cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err
    # deterministic: bit-identical on a second launch
    dw2 = torch.empty_like(dw)
    _lib.check(LIB.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(xa), _lib.ptr(dza), _lib.ptr(dw2), 0.5, 0, _lib.ptr(ws), nb,
                                     _lib.stream()), "wgrad")
    assert torch.equal(dw, dw2)
    # accumulate mode adds into the destination (gradient arena slot)
    _lib.check(LIB.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(xa), _lib.ptr(dza), _lib.ptr(dw2), 0.5, 1, _lib.ptr(ws), nb,
                                     _lib.stream()), "wgrad")
    assert torch.equal(dw2, dw + dw)


DGRAD_CASES = [(2, 32, 48, 3, 1, 32, 24), (3, 64, 64, 3, 1, 16, 12), (2, 64, 128, 3, 2, 32, 24), (2, 48, 96, 3, 2, 24, 16),
               (2, 64, 256, 1, 1, 16, 12)]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_dgrad_f16_vs_torch(case):
    n, cin, cout, k, s, h, w = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    dz = torch.randn(n, cout, ho, wo, generator=g)
    x = torch.zeros(n, cin, h, w, requires_grad=True)
    F.conv2d(x, _h(wt), None, stride=s, padding=pad).backward(_h(dz))
    ref = _h(x.grad)
    dza = _to_c8(dz)
    dx = ActC8(n, cin, h, w, DEV)
    ones = torch.ones((cin + 15) // 16 * 16, device=DEV)
    zeros = torch.zeros_like(ones)
    wdev = wt.to(DEV)

    def run(d, mode, kk, py=0, px=0):
        nb = LIB.mp_f16_packed_weight_bytes(cin, cout, kk, kk)
        packed = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
        _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wdev), _lib.ptr(packed), cin, cout, kk, kk, mode, py, px, _lib.stream()), "pack")
        _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(d), -1, _lib.ptr(dza), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                         None, _lib.ptr(dx), _lib.stream()), "dgrad")
    if s == 1:
        run(_desc(n, cout, ho, wo, cin, k, 1, k - 1 - pad, h, w), 2, k)
    else:
        for py in (0, 1):
            for px in (0, 1):
                run(_desc(n, cout, ho, wo, cin, 2, 1, 0, ho, wo, oh=h, ow=w, mul=2, oy=py, ox=px), 3, 2, py, px)
    _close16(_from_c8(dx), ref, "dx")


@pytest.mark.parametrize("case,variant", [pytest.param(c, -1, id=f"case{i}-heuristic") for i, c in enumerate(fm.PHASES4_CASES)]
                         + fm.served_pairs(fm.PHASES4_CASES, fm.PHASES4_VARIANTS, fm.phases4_case_desc))
def test_stride2_dgrad_four_phases_in_one_launch_equal_four_launches(case, variant):
    """MP_CONV_PHASES4: phase = second grid dimension, weight slice and output offset from it - bit-identical to the four launches
    (one-tile and persistent multi-tile variants; only the pairs the library serves are collected - tests/f16_matrix.py)."""
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    ho, wo = h // 2, w // 2
    wdev = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(DEV)
    dza = _to_c8(torch.randn(n, cout, ho, wo, generator=g))
    ones = torch.ones((cin + 15) // 16 * 16, device=DEV)
    zeros = torch.zeros_like(ones)
    nb = LIB.mp_f16_packed_weight_bytes(cin, cout, 2, 2)
    packed = torch.empty(4 * (nb // 2), device=DEV, dtype=torch.float16)
    for py in (0, 1):
        for px in (0, 1):
            sl = packed[(2 * py + px) * (nb // 2):]
            _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wdev), sl.data_ptr(), cin, cout, 2, 2, 3, py, px, _lib.stream()), "pack")
    one, four = ActC8(n, cin, h, w, DEV), ActC8(n, cin, h, w, DEV)
    d = _desc(n, cout, ho, wo, cin, 2, 1, 0, ho, wo, oh=h, ow=w, mul=2)
    d.flags = _lib.MP_CONV_PHASES4
    rc = LIB.mp_f16_conv2d_fwd(ctypes.byref(d), variant, _lib.ptr(dza), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None, None,
                               _lib.ptr(one), _lib.stream())
    _lib.check(rc, f"phases4, variant {variant}")
    for py in (0, 1):
        for px in (0, 1):
            dd = _desc(n, cout, ho, wo, cin, 2, 1, 0, ho, wo, oh=h, ow=w, mul=2, oy=py, ox=px)
            sl = packed[(2 * py + px) * (nb // 2):]
            _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(dd), variant, _lib.ptr(dza), sl.data_ptr(), _lib.ptr(ones), _lib.ptr(zeros), None,
                                             None, _lib.ptr(four), _lib.stream()), "phase")
    assert torch.equal(_from_c8(one), _from_c8(four))
    # weights-in-registers variants do not take the flag
    assert LIB.mp_f16_conv2d_fwd(ctypes.byref(d), 26, _lib.ptr(dza), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None, None,
                                 _lib.ptr(one), _lib.stream()) != 0


@pytest.mark.parametrize("c,h,w,relu,with_res", [(32, 64, 48, True, True), (64, 16, 12, True, False), (17, 8, 6, False, False),
                                                 (48, 24, 18, True, True)])
def test_bn_train_f16_fwd_bwd_vs_torch(c, h, w, relu, with_res):
    g = torch.Generator().manual_seed(c + h)
    n = 6
    z = (torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3)
    res = torch.randn(n, c, h, w, generator=g) if with_res else None
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    dy = torch.randn(n, c, h, w, generator=g)
    mm, mv = torch.zeros(c), torch.ones(c)
    zt = _h(z).requires_grad_(True)
    rt = _h(res).requires_grad_(True) if with_res else None
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmm, rmv = mm.clone(), mv.clone()
    y = F.batch_norm(zt, rmm, rmv, gt, bt, training=True, momentum=0.1, eps=1e-5)
    if with_res:
        y = y + rt
    if relu:
        y = F.relu(y)
    yh = _h(y.detach())
    # the kernel masks with the STORED (fp16) output: make the reference use the same mask
    y.backward(_h(dy) * ((yh > 0).float() if relu else 1.0))

    za, ya = _to_c8(z), ActC8(n, c, h, w, DEV)
    ra = _to_c8(res) if with_res else None
    nb = LIB.mp_bn_workspace_bytes(c)
    ws = torch.empty(nb // 4 + 1, device=DEV)
    mean, invstd = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    dmm, dmv = mm.to(DEV), mv.to(DEV)
    dgam, dbet = gamma.to(DEV), beta.to(DEV)
    _lib.check(LIB.mp_f16_bn_train_fwd(_lib.ptr(za), _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(ra), _lib.ptr(ya),
                                       _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(dmm), _lib.ptr(dmv), n, c, h * w, 1e-5, 0.9,
                                       int(relu), _lib.ptr(ws), nb, _lib.stream()), "bn fwd")
    _close16(_from_c8(ya), yh, "y")
    assert torch.allclose(dmm.cpu(), rmm, atol=1e-5) and torch.allclose(dmv.cpu(), rmv, atol=1e-5)
    dza, dra = ActC8(n, c, h, w, DEV), (ActC8(n, c, h, w, DEV) if with_res else None)
    dgm, dbt = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    acc_g, acc_b = torch.full((c,), 1.0, device=DEV), torch.full((c,), 2.0, device=DEV)
    dya = _to_c8(dy)
    _lib.check(LIB.mp_f16_bn_train_bwd(_lib.ptr(dya), _lib.ptr(za), _lib.ptr(ya), _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(mean),
                                       _lib.ptr(invstd), _lib.ptr(dza), _lib.ptr(dra), _lib.ptr(dgm), _lib.ptr(dbt), _lib.ptr(acc_g),
                                       _lib.ptr(acc_b), n, c, h * w, int(relu), _lib.ptr(ws), nb, _lib.stream()), "bn bwd")
    assert torch.equal(acc_g, dgm + 1.0) and torch.equal(acc_b, dbt + 2.0)  # also added into the caller's buffers
    if relu and not with_res:
        # the same pass with the ReLU mask re-derived from z (y not passed): bit-identical outputs
        dz2, dg2, db2 = ActC8(n, c, h, w, DEV), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        _lib.check(LIB.mp_f16_bn_train_bwd(_lib.ptr(dya), _lib.ptr(za), None, _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(mean),
                                           _lib.ptr(invstd), _lib.ptr(dz2), None, _lib.ptr(dg2), _lib.ptr(db2), None, None, n, c, h * w, 1,
                                           _lib.ptr(ws), nb, _lib.stream()), "bn bwd, mask from z")
        assert torch.equal(dz2.c8_tensor, dza.c8_tensor) and torch.equal(dg2, dgm) and torch.equal(db2, dbt)
    if relu and with_res:  # a residual layer cannot re-derive its mask: rejected, not silently wrong
        assert LIB.mp_f16_bn_train_bwd(_lib.ptr(dya), _lib.ptr(za), None, _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(mean), _lib.ptr(invstd),
                                       _lib.ptr(dza), _lib.ptr(dra), _lib.ptr(dgm), _lib.ptr(dbt), None, None, n, c, h * w, 1, _lib.ptr(ws),
                                       nb, _lib.stream()) == -1  # MP_ERR_NULL
    # where the kernel's fp16 y and torch's fp32 y disagree about the ReLU mask (y within an ulp of 0) nothing is compared
    _close16(_from_c8(dza), _h(zt.grad), "dz")
    assert torch.allclose(dgm.cpu(), gt.grad, rtol=2e-3, atol=2e-3 * float(gt.grad.abs().max()))
    assert torch.allclose(dbt.cpu(), bt.grad, rtol=2e-3, atol=2e-3 * float(bt.grad.abs().max()))
    if with_res:
        _close16(_from_c8(dra), _h(rt.grad), "dres")


def test_fuse_sum_f16_bwd_vs_torch():
    g = torch.Generator().manual_seed(3)
    n, c, h, w = 3, 32, 32, 24
    scales = (2, 4, 8)
    base = torch.randn(n, c, h, w, generator=g)
    terms = [torch.randn(n, c, h // s, w // s, generator=g) for s in scales]
    dy = torch.randn(n, c, h, w, generator=g)
    leaves = [_h(base).requires_grad_(True)] + [_h(t).requires_grad_(True) for t in terms]
    y = leaves[0]
    for t, s in zip(leaves[1:], scales):
        y = y + F.interpolate(t, scale_factor=s, mode="nearest")
    out = _h(F.relu(y).detach())
    y.backward(_h(dy) * (out > 0).float())
    outa = _to_c8(out)
    dbase = ActC8(n, c, h, w, DEV)
    dts = [ActC8(n, c, h // s, w // s, DEV) for s in scales]
    args = []
    for t, s in zip(dts, scales):
        args += [_lib.ptr(t), s]
    dya = _to_c8(dy)
    _lib.check(LIB.mp_f16_fuse_upsample_sum_bwd(_lib.ptr(dya), _lib.ptr(outa), _lib.ptr(dbase), *args, n, c, h, w, 1,
                                                _lib.stream()), "fuse bwd")
    assert torch.equal(_from_c8(dbase), _h(leaves[0].grad))
    for t, leaf in zip(dts, leaves[1:]):
        _close16(_from_c8(t), _h(leaf.grad), "dt")



def test_o2_optimizer_steps_with_dynamic_loss_scale():
    import mindpose_amd as mp
    from mindpose_amd.utils import AdamWeightDecay, DynamicLossScaleManager
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
    mp.models.auto_mixed_precision(net, "O2")
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True)
    mgr = DynamicLossScaleManager()  # 2**24, factor 2, window 2000 as tools/train.py:170-173
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    kp = (torch.rand(4, 17, 3, generator=g) * torch.tensor([64.0, 64.0, 2.0])).to(DEV)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)
    losses, updated = [], 0
    for it in range(16):
        opt.zero_grad()
        target, weight = tgt(kp)
        loss = nwl(x, target, weight)
        mgr.scale(loss).backward()
        updated += int(opt.step(loss_scale_manager=mgr))
        losses.append(float(loss.detach()))
    # the initial 2**24 overflows fp16 activation gradients: a few skipped steps halve it, then every step updates
    assert updated >= 8 and mgr.skipped_steps == 16 - updated and mgr.loss_scale == 2.0 ** 24 / 2 ** mgr.skipped_steps
    assert losses[-1] < losses[0], losses
    assert all(torch.isfinite(p).all() for p in net.parameters())


def test_graphed_o2_step_equals_eager():
    """forward + loss + backward replayed from one hipGraph: same loss and bit-identical gradients as the eager step
    (same kernels, same order), then the optimizer update and the loss-scale bookkeeping run outside the graph."""
    import mindpose_amd as mp
    from mindpose_amd.utils import AdamWeightDecay, DynamicLossScaleManager, GraphedTrainStep

    def build():
        torch.manual_seed(0)
        net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
        mp.models.auto_mixed_precision(net, "O2")
        nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
        opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
        return net, nwl, opt

    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    kp = (torch.rand(4, 17, 3, generator=g) * torch.tensor([64.0, 64.0, 2.0])).to(DEV)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)
    target, weight = tgt(kp)
    net_e, nwl_e, opt_e = build()
    opt_e.zero_grad()
    loss_e = nwl_e(x, target, weight)
    (loss_e * 4096.0).backward()
    grads_e = opt_e.grads.arena.clone()
    net_g, nwl_g, opt_g = build()
    mgr = DynamicLossScaleManager(init_loss_scale=4096.0)
    step = GraphedTrainStep(nwl_g, opt_g, (x, target, weight), loss_scale_manager=mgr, warmup=2)
    before = opt_g.flat.clone()
    loss_g = step(x, target, weight)
    assert float(loss_g.detach()) == float(loss_e.detach())
    assert step.updated and mgr.skipped_steps == 0
    # the arena keeps the loss-scaled sums (the 1 / loss_scale is folded into the update kernel's read); the parameters moved
    assert torch.equal(opt_g.grads.arena, grads_e) and opt_g.grad_scale == 1.0 / 4096.0
    assert not torch.equal(opt_g.flat, before)
    losses = [float(step(x, target, weight).detach()) for _ in range(6)]
    assert losses[-1] < float(loss_e.detach())


def test_batched_weight_pack_equals_individual_packs():
    """mp_f16_pack_weight_batch == one mp_f16_pack_weight per job, bit for bit, over every packing mode; then three optimizer
    steps of a network whose packings are refreshed by the batched launch equal three steps with per-layer packing."""
    import ctypes
    import numpy as np
    from mindpose_amd.models import train_ops as T

    g = torch.Generator().manual_seed(21)
    specs = [  # (weight shape, cout, cin, k, mode, py, px)
        ((40, 24, 3, 3), 40, 24, 3, 0, 0, 0), ((17, 32, 1, 1), 17, 32, 1, 0, 0, 0), ((40, 24, 3, 3), 24, 40, 3, 2, 0, 0),
        ((40, 24, 3, 3), 24, 40, 2, 3, 1, 0), ((24, 40, 4, 4), 40, 24, 2, 1, 0, 1), ((24, 40, 4, 4), 24, 40, 2, 4, 1, 1),
        ((64, 64, 1, 1), 64, 64, 1, 2, 0, 0)]
    ws = [torch.randn(sh, generator=g).to(DEV) for sh, *_ in specs]
    single, batch = [], []
    arr = (T._PackJob * len(specs))()
    first = np.zeros(len(specs) + 1, dtype=np.uint32)
    for i, (w, (_, cout, cin, k, mode, py, px)) in enumerate(zip(ws, specs)):
        nb = LIB.mp_f16_packed_weight_bytes(cout, cin, k, k)
        a, b = torch.full((nb // 2,), 7.0, device=DEV, dtype=torch.float16), torch.full((nb // 2,), 9.0, device=DEV, dtype=torch.float16)
        _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(w), _lib.ptr(a), cout, cin, k, k, mode, py, px, _lib.stream()), "pack")
        arr[i] = T._PackJob(w.data_ptr(), b.data_ptr(), cout, cin, k, k, mode, py, px, 0)
        # one thread per (8 input channels, cout) pair; every other job with the x k * k block count of round 4 (surplus blocks leave)
        first[i + 1] = first[i] + ((cin + 31) // 32 * (k * k if i % 2 else 1) * 4 * ((cout + 15) // 16 * 16) + 255) // 256
        single.append(a)
        batch.append(b)
    jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    first_dev = torch.from_numpy(first.view(np.int32)).to(DEV)
    _lib.check(LIB.mp_f16_pack_weight_batch(_lib.ptr(jobs_dev), _lib.ptr(first_dev), len(specs), int(first[-1]), _lib.stream()), "batch")
    torch.cuda.synchronize()
    for a, b, sp in zip(single, batch, specs):
        assert torch.equal(a, b), sp

    import mindpose_amd as mp
    from mindpose_amd.utils import AdamWeightDecay

    def run(batched):
        torch.manual_seed(0)
        net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
        mp.models.auto_mixed_precision(net, "O2")
        nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
        opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
        gg = torch.Generator().manual_seed(3)
        x = torch.randn(2, 3, 64, 64, generator=gg).to(DEV)
        kp = (torch.rand(2, 17, 3, generator=gg) * torch.tensor([64.0, 64.0, 2.0])).to(DEV)
        target, weight = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)(kp)
        losses, refreshed = [], []
        real = T.repack_weights
        if not batched:
            T.repack_weights = lambda module: 0
        try:
            for _ in range(3):
                refreshed.append(T.repack_weights(net) if batched else 0)  # what Net.train_forward is about to do again
                opt.zero_grad()
                loss = nwl(x, target, weight)
                (loss * 1024.0).backward()
                opt.grads.arena.mul_(1.0 / 1024.0)
                opt.step()
                losses.append(float(loss.detach()))
        finally:
            T.repack_weights = real
        return losses, opt.flat.clone(), refreshed

    la, pa, ra = run(True)
    lb, pb, _ = run(False)
    assert ra[0] == 0 and ra[1] > 500 and ra[2] == ra[1]  # step one packs per layer, later steps refresh all at once
    assert la == lb and torch.equal(pa, pb)


def test_residual_block_as_one_autograd_node_matches_per_cell_functions(monkeypatch):
    """ResidualBlock16Fn (identity gradient added in the first conv's data-gradient epilogue, one rounding) against the per-cell
    composition (autograd's separate fp16 add): same loss bit for bit, gradients equal up to that one extra fp16 rounding."""
    import mindpose_amd as mp
    from mindpose_amd.utils import AdamWeightDecay

    monkeypatch.setenv("MINDPOSE_BN_FUSE", "0")  # the round-3 fused chains replace both forms (tests/test_gpu_bn_fuse.py)

    def run(fused):
        monkeypatch.setenv("MINDPOSE_FUSE_RESIDUAL", "1" if fused else "0")
        torch.manual_seed(0)
        net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
        mp.models.auto_mixed_precision(net, "O2")
        nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
        opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
        kp = (torch.rand(2, 17, 3, generator=g) * torch.tensor([64.0, 64.0, 2.0])).to(DEV)
        target, weight = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)(kp)
        opt.zero_grad()
        loss = nwl(x, target, weight)
        (loss * 1024.0).backward()
        return float(loss.detach()), opt.grads.arena.clone()

    l1, g1 = run(True)
    l0, g0 = run(False)
    assert l1 == l0
    cos = torch.nn.functional.cosine_similarity(g1.double(), g0.double(), dim=0)
    assert float(cos) > 0.9995, float(cos)
    assert float((g1 - g0).norm() / g0.norm()) < 3e-2


def test_maxpool_bwd_and_stem_wgrad_vs_torch():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 5, 18, 14, generator=g)
    dy = torch.randn(3, 5, 9, 7, generator=g)
    xt = x.clone().requires_grad_(True)
    from oracle.nets import maxpool3x3s2_same as oracle_pool
    oracle_pool(xt).backward(dy)
    dx = torch.empty(3, 5, 18, 14, device=DEV)
    xd, dyd = x.to(DEV), dy.to(DEV)
    _lib.check(LIB.mp_maxpool3x3s2_same_bwd(_lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dx), 3, 5, 18, 14, _lib.stream()), "pool bwd")
    assert torch.equal(dx.cpu(), xt.grad)
    # 7x7 stride-2 stem weight gradient (3 input channels)
    x = torch.randn(4, 3, 64, 48, generator=g)
    dz = torch.randn(4, 64, 32, 24, generator=g)
    w = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(x, w, None, stride=2, padding=3).backward(dz)
    dw = torch.empty(64, 3, 7, 7, device=DEV)
    xd, dzd = x.to(DEV), dz.to(DEV)
    _lib.check(LIB.mp_stem_conv_wgrad(_lib.ptr(xd), _lib.ptr(dzd), _lib.ptr(dw), 4, 3, 64, 48, 64, 7, _lib.stream()), "stem wgrad")
    assert float((dw.cpu() - w.grad).abs().max() / w.grad.abs().max()) < 1e-5


def test_deconv_f16_autograd_vs_torch():
    from mindpose_amd.models import train_ops as T
    g = torch.Generator().manual_seed(9)
    n, cin, cout, h, w = 3, 64, 32, 12, 10
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5)
    dy = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
    xt, wtt = _h(x).requires_grad_(True), _h(wt).requires_grad_(True)
    F.conv_transpose2d(xt, wtt, None, stride=2, padding=1).backward(_h(dy))
    xa = _to_c8(x).c8_tensor.requires_grad_(True)
    wd = wt.to(DEV).requires_grad_(True)
    y = T.Deconv16Fn.apply(xa, wd)
    y.backward(_to_c8(dy).c8_tensor)
    ya = ActC8(n, cout, 2 * h, 2 * w, DEV); ya.c8_tensor = y.detach()
    _close16(_from_c8(ya), _h(F.conv_transpose2d(_h(x), _h(wt), None, stride=2, padding=1)), "deconv y")
    dxa = ActC8(n, cin, h, w, DEV); dxa.c8_tensor = xa.grad
    _close16(_from_c8(dxa), _h(xt.grad), "deconv dx")
    err = float((wd.grad.cpu() - wtt.grad).abs().max() / wtt.grad.abs().max())
    assert err < 2e-4, err


def test_simplebaseline_r50_o2_training_step_vs_oracle():
    """SimpleBaseline-ResNet50 under amp O2 (fp32 stem + max-pool, fp16 bottlenecks, transposed-conv head): loss and
    parameter gradients against torch-CPU fp32 autograd of the oracle graph (same bars as the HRNet O2 step)."""
    import numpy as np
    import mindpose_amd as mp
    from oracle import nets as onets
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("resnet50", "simple_baseline_head"), seed=0)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    for k, v in params.items():
        if v.dtype.is_floating_point and not k.endswith(("moving_mean", "moving_variance")):
            v.requires_grad_()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(4, 3, 128, 96, generator=g)
    target = torch.rand(4, 17, 32, 24, generator=g)
    weight = (torch.rand(4, 17, generator=g) > 0.3).float()
    out = onets.net_forward_train(params, x, "resnet50", "simple_baseline_head")
    ref_loss = (((out - target) ** 2) * weight[..., None, None]).mean()
    ref_loss.backward()
    net = net.to(DEV).train()
    mp.models.auto_mixed_precision(net, "O2")
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    scale = 256.0
    loss = nwl(x.to(DEV), target.to(DEV), weight.to(DEV))
    (loss * scale).backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-2 * abs(float(ref_loss.detach())), (float(loss.detach()), float(ref_loss.detach()))
    cos, rel = {}, {}
    for k, v in net.named_parameters():
        assert v.grad is not None and torch.isfinite(v.grad).all(), k
        a, b = (v.grad / scale).double().cpu().flatten(), params[k].grad.double().flatten()
        cos[k] = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))
        rel[k] = float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
    med = float(np.median(list(cos.values())))
    print(f"SimpleBaseline O2 vs fp32 oracle: per-tensor cosine median {med:.5f} min {min(cos.values()):.4f}; "
          f"final layer rel {rel['head.final_layer.weight']:.2e}, deconv0 rel {rel['head.deconv_layer.0.weight']:.2e}, "
          f"stem rel {rel['backbone.conv1.weight']:.2e}")
    # 4 crops of 128x96 leave 48 samples per channel for the last stage's batch statistics: fp16 noise is amplified more than
    # in the HRNet step; every tensor still points the same way and the layer next to the loss is tight
    assert med > 0.97 and min(cos.values()) > 0.9
    assert rel["head.final_layer.weight"] < 5e-3 and rel["head.final_layer.bias"] < 5e-3
