"""GPU parity of the direct fp32-MFMA convolution family and of the whole networks, through the C ABI
(launch plan), against independent torch-CPU formulations / the CPU oracle.

Tolerance: BASELINE.json asks heat-maps within 1e-3 (fp32).  v_mfma_f32_16x16x4_f32 is an exact fp32 FMA
chain, so single layers are held to 2e-5 of the output scale and whole networks to 1e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import mindpose_amd as mp  # noqa: E402
from mindpose_amd.models.layers import BatchNorm2d, Conv2d, Conv2dTranspose, Plan  # noqa: E402
from oracle import nets as onets  # noqa: E402

DEV = torch.device("cuda:0")


def _rand_bn(c, g):
    bn = BatchNorm2d(c)
    with torch.no_grad():
        bn.gamma.copy_(torch.rand(c, generator=g) + 0.5)
        bn.beta.copy_(torch.randn(c, generator=g) * 0.1)
        bn.moving_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.moving_variance.copy_(torch.rand(c, generator=g) + 0.5)
    return bn


def _nerr(got, ref):
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20))


CONV_CASES = [
    # n, cin, cout, k, s, pad, h, w, relu, res, bias
    (2, 32, 32, 3, 1, 1, 64, 48, True, True, False),    # W32 branch 0
    (3, 64, 64, 3, 1, 1, 32, 24, True, True, False),    # branch 1
    (3, 128, 128, 3, 1, 1, 16, 12, True, False, False),  # branch 2
    (5, 256, 256, 3, 1, 1, 8, 6, True, True, False),    # branch 3 (multi-image tiles, W % 4 != 0)
    (2, 3, 64, 3, 2, 1, 256, 192, True, False, False),  # stem conv1 (cin padded to 4)
    (2, 64, 64, 3, 2, 1, 128, 96, True, False, False),  # stem conv2
    (2, 256, 32, 3, 1, 1, 64, 48, True, False, False),  # transition1.0
    (2, 256, 64, 3, 2, 1, 64, 48, True, False, False),  # transition1.1
    (2, 64, 256, 1, 1, 0, 64, 48, False, True, False),  # bottleneck conv3 + identity
    (2, 256, 64, 1, 1, 0, 64, 48, True, False, False),  # bottleneck conv1
    (2, 32, 17, 1, 1, 0, 64, 48, False, False, True),   # HRNet head (bias, cout=17)
    (2, 32, 64, 3, 2, 1, 64, 48, False, True, False),   # fuse down
    (2, 48, 48, 3, 1, 1, 64, 48, True, True, False),    # W48 branch 0
    (2, 96, 192, 3, 2, 1, 18, 14, True, False, False),  # odd-ish sizes
    (1, 5, 7, 3, 1, 1, 9, 7, True, True, False),        # tiny ragged
    (3, 8, 40, 1, 2, 0, 10, 6, False, False, False),    # 1x1 stride 2 (ResNet down_sample), cout=40
    (2, 3, 64, 7, 2, 3, 64, 48, True, False, False),    # ResNet stem 7x7
    (1, 3, 64, 7, 2, 3, 256, 192, True, False, False),  # ResNet stem full size
    (2, 512, 2048, 1, 1, 0, 8, 6, False, True, False),  # ResNet layer4 conv3
    (7, 16, 16, 3, 1, 1, 4, 4, True, True, False),      # N not a multiple of the image group
    (2, 32, 32, 3, 1, 1, 96, 72, True, True, False),    # 384x288 config, W=72
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_k{c[3]}s{c[4]}_{c[6]}x{c[7]}" for c in CONV_CASES])
def test_conv_bn_act_vs_torch(case):
    n, cin, cout, k, s, pad, h, w, relu, res, bias = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    conv = Conv2d(cin, cout, k, stride=s, padding=pad, has_bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g))
    bn = None if bias else _rand_bn(cout, g)
    x = torch.randn(n, cin, h, w, generator=g)
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    r = torch.randn(n, cout, ho, wo, generator=g) if res else None
    ref = F.conv2d(x, conv.weight, conv.bias, stride=s, padding=pad)
    if bn is not None:
        ref = F.batch_norm(ref, bn.moving_mean, bn.moving_variance, bn.gamma, bn.beta, False, 0.0, 1e-5)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    plan = Plan(DEV)
    xd = x.to(DEV)
    rd = r.to(DEV) if res else None
    out = plan.conv(xd, conv, bn, relu=relu, res1=rd)
    plan.run()
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert _nerr(out.cpu(), ref.detach()) < 2e-5


@pytest.mark.parametrize("up", [2, 4, 8])
def test_fuse_upsample_add_two_residuals(up):
    # HRModule fuse term j > i: relu(res1 + up_nearest(bn(conv1x1(x))) + res2), in place on res1
    g = torch.Generator().manual_seed(up)
    cin, cout, h, w, n = 32 * up, 32, 64 // up, 48 // up, 2
    conv = Conv2d(cin, cout, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.0 / cin) ** 0.5)
    bn = _rand_bn(cout, g)
    x = torch.randn(n, cin, h, w, generator=g)
    acc = torch.randn(n, cout, 64, 48, generator=g)
    idt = torch.randn(n, cout, 64, 48, generator=g)
    t = F.batch_norm(F.conv2d(x, conv.weight), bn.moving_mean, bn.moving_variance, bn.gamma, bn.beta, False, 0.0, 1e-5)
    ref = F.relu(acc + F.interpolate(t, size=(64, 48), mode="nearest") + idt)
    plan = Plan(DEV)
    accd = acc.to(DEV)
    out = plan.conv(x.to(DEV), conv, bn, relu=True, res1=accd, res2=idt.to(DEV), out=accd, upsample=up)
    plan.run()
    torch.cuda.synchronize()
    assert out.data_ptr() == accd.data_ptr()
    assert _nerr(out.cpu(), ref.detach()) < 2e-5


@pytest.mark.parametrize("shape", [(2, 64, 32, 8, 6), (1, 2048, 256, 8, 6), (3, 20, 24, 5, 7)])
def test_deconv4x4s2_bn_relu_vs_torch(shape):
    n, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(cin)
    dc = Conv2dTranspose(cin, cout, 4)
    with torch.no_grad():
        dc.weight.copy_(torch.randn(dc.weight.shape, generator=g) * (0.5 / cin) ** 0.5)
    bn = _rand_bn(cout, g)
    x = torch.randn(n, cin, h, w, generator=g)
    ref = F.relu(F.batch_norm(F.conv_transpose2d(x, dc.weight, None, stride=2, padding=1), bn.moving_mean,
                              bn.moving_variance, bn.gamma, bn.beta, False, 0.0, 1e-5))
    plan = Plan(DEV)
    out = plan.deconv4x4s2(x.to(DEV), dc, bn, relu=True)
    plan.run()
    torch.cuda.synchronize()
    assert out.shape == (n, cout, 2 * h, 2 * w)
    assert _nerr(out.cpu(), ref.detach()) < 2e-5


def _net(backbone, head):
    return mp.init_synthetic(mp.create_network(backbone, head), seed=0).to(DEV).eval()


def test_reference_shape_tests():
    # the reference's own (shape-only) tests: tests/models/backbones/test_hrnet.py, test_resnet.py, heads/*
    x = torch.rand(4, 3, 32, 32, device=DEV)
    for name, ch in (("hrnet_w32", 32), ("hrnet_w48", 48)):
        bb = mp.init_synthetic(mp.create_backbone(name), 0).to(DEV)
        assert bb(x).shape == (4, ch, 8, 8) and bb.out_channels == ch
    bb = mp.init_synthetic(mp.create_backbone("resnet50"), 0).to(DEV)
    assert bb(x).shape == (4, 2048, 1, 1) and bb.out_channels == 2048
    head = mp.init_synthetic(mp.create_head("hrnet_head", in_channels=32), 0).to(DEV)
    assert head(torch.rand(4, 32, 8, 8, device=DEV)).shape == (4, 17, 8, 8)
    head = mp.init_synthetic(mp.create_head("simple_baseline_head", in_channels=32), 0).to(DEV)
    assert head(torch.rand(4, 32, 8, 8, device=DEV)).shape == (4, 17, 64, 64)


@pytest.mark.parametrize("backbone,head,shape", [
    ("hrnet_w32", "hrnet_head", (2, 3, 96, 64)),
    ("hrnet_w32", "hrnet_head", (3, 3, 256, 192)),
    ("hrnet_w48", "hrnet_head", (1, 3, 128, 96)),
    ("resnet50", "simple_baseline_head", (2, 3, 256, 192)),
    ("resnet50", "simple_baseline_head", (1, 3, 64, 64)),
    ("resnet101", "simple_baseline_head", (2, 3, 64, 64)),   # resnet.py:298-340: layers [3, 4, 23, 3] / [3, 8, 36, 3]
    ("resnet152", "simple_baseline_head", (2, 3, 64, 64)),
])
def test_network_heatmaps_vs_oracle(backbone, head, shape):
    net = _net(backbone, head)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(0))
    got = net(x.to(DEV)).cpu()
    ref = onets.net_forward({k: v.cpu() for k, v in net.state_dict().items()}, x, backbone, head)
    assert got.shape == ref.shape
    err = _nerr(got, ref)
    assert err < 1e-3, f"normalised max error {err}"
    # arg-max agreement wherever the oracle's top-1/top-2 margin exceeds the heat-map tolerance
    n, k = got.shape[:2]
    rf = ref.reshape(n, k, -1)
    top2 = rf.topk(2, dim=2).values
    safe = (top2[..., 0] - top2[..., 1]) > 2e-3 * ref.abs().max()
    assert torch.equal(got.reshape(n, k, -1).argmax(2)[safe], rf.argmax(2)[safe])


def test_eval_net_and_flip_tta_end_to_end():
    from oracle import decoder as od
    from tests.golden import recipes
    net = _net("hrnet_w32", "hrnet_head")
    dec = mp.create_decoder("topdown_heatmap", shift_coordinate=True).to(DEV)
    ev = mp.create_eval_network(net, dec, output_raw=True)
    x = torch.randn(2, 3, 256, 192, generator=torch.Generator().manual_seed(3))
    center, scale, score = (torch.from_numpy(a) for a in recipes.boxes(2, 4))
    (preds, boxes), hm = ev(x.to(DEV), center.to(DEV), scale.to(DEV), score.to(DEV))
    rp, rb, ri = od.decode(hm.cpu().numpy(), center.numpy(), scale.numpy(), score.numpy(), shift_coord=True)
    assert np.array_equal(dec.last_argmax.cpu().numpy(), ri.astype(np.int32))
    assert np.array_equal(preds.cpu().numpy(), rp) and np.array_equal(boxes.cpu().numpy(), rb)
    hm0 = hm.clone()
    inf = mp.TopDownHeatMapInferencer(ev, config=dict(has_heatmap_output=True, hflip_tta=True, shift_heatmap=True,
                                                      flip_pairs=recipes.FLIP_PAIRS), decoder=dec)
    recs = inf([dict(image=x.to(DEV), center=center.to(DEV), scale=scale.to(DEV), bbox_scores=score.to(DEV))])
    hf = net(torch.flip(x, dims=[3]).to(DEV)).cpu().numpy()
    avg = od.flip_aggregate(hm0.cpu().numpy(), hf, recipes.FLIP_INDEX, shift_heatmap=True)
    rp, rb, _ = od.decode(avg, center.numpy(), scale.numpy(), score.numpy(), shift_coord=True)
    assert len(recs) == 2 and set(recs[0]) == {"pred", "box", "image_path", "bbox_id"}
    assert np.array_equal(np.array([r["pred"] for r in recs], dtype=np.float32), rp)
    assert np.array_equal(np.array([r["box"] for r in recs], dtype=np.float32), rb)


def test_network_with_loss_and_cpu_input_fails_loudly():
    net = _net("hrnet_w32", "hrnet_head")
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    x = torch.randn(2, 3, 64, 64, device=DEV)
    t = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)
    kp = torch.tensor([[[10.0, 20.0, 1.0]] * 17, [[30.0, 40.0, 1.0]] * 17], device=DEV)
    target, w = t(kp)
    loss = nwl(x, target, w)
    assert loss.numel() == 1 and torch.isfinite(loss)
    with pytest.raises(mp._lib.MindposeHipError):
        net(torch.randn(1, 3, 64, 64))  # CPU input: no fallback


def test_config5_w48_384x288_udp_dark_flip_end_to_end():
    # BASELINE.json configs[4] shape (HRNet-W48 384x288, UDP + DARK k=17, flip test) at N=2, fp32
    from oracle import decoder as od
    from tests.golden import recipes
    from mindpose_amd.engine.inferencer.topdown_inferencer import _MultiRunNet
    net = _net("hrnet_w48", "hrnet_head")
    dec = mp.create_decoder("topdown_heatmap", use_udp=True, dark_udp_refine=True, kernel_size=17).to(DEV)
    ev = mp.create_eval_network(net, dec, output_raw=True)
    mr = _MultiRunNet(ev, dec, np.array(recipes.FLIP_INDEX), shift_heatmap=False).to(DEV)
    x = torch.randn(2, 3, 384, 288, generator=torch.Generator().manual_seed(5))
    center, scale, score = (torch.from_numpy(a) for a in recipes.boxes(2, 6))
    preds, boxes = mr(x.to(DEV), center.to(DEV), scale.to(DEV), score.to(DEV))
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    h = onets.net_forward(sd, x, "hrnet_w48", "hrnet_head").numpy()
    hf = onets.net_forward(sd, torch.flip(x, dims=[3]), "hrnet_w48", "hrnet_head").numpy()
    assert h.shape == (2, 17, 96, 72)
    # heat-maps of the HIP path vs oracle (both runs)
    got_h = net(x.to(DEV)).cpu().numpy()
    assert np.abs(got_h - h).max() / np.abs(h).max() < 1e-3
    # decode on the oracle-aggregated map: identical arg-max wherever the top-2 margin is safe, boxes exact
    avg = od.flip_aggregate(h, hf, recipes.FLIP_INDEX, shift_heatmap=False)
    rp, rb, ri = od.decode(avg, center.numpy(), scale.numpy(), score.numpy(), use_udp=True, dark_udp_refine=True, kernel_size=17)
    flat = avg.reshape(2, 17, -1)
    top2 = np.sort(flat, axis=2)[..., -2:]
    safe = (top2[..., 1] - top2[..., 0]) > 2e-3 * np.abs(avg).max()
    assert np.array_equal(dec.last_argmax.cpu().numpy()[safe], ri.astype(np.int32)[safe])
    assert np.array_equal(boxes.cpu().numpy(), rb)


@pytest.mark.parametrize("terms", [[2], [2, 4], [2, 4, 8]])
def test_fuse_upsample_sum_vs_torch(terms):
    # out = relu(((base + up(t1)) + up(t2)) + up(t3)), same order as hrnet.py:327-339 -> bit-exact vs torch
    g = torch.Generator().manual_seed(len(terms))
    n, c, h, w = 3, 8, 32, 24
    base = torch.randn(n, c, h, w, generator=g)
    ts = [torch.randn(n, c, h // s, w // s, generator=g) for s in terms]
    ref = base.clone()
    for t, s in zip(ts, terms):
        ref = ref + F.interpolate(t, size=(h, w), mode="nearest")
    ref = F.relu(ref)
    plan = Plan(DEV)
    out = plan.alloc(n, c, h, w)
    plan.fuse_sum(base.to(DEV), [(t.to(DEV), s) for t, s in zip(ts, terms)], out, relu=True)
    plan.run()
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("amp", ["O0", "O2"])
def test_execution_lanes_are_bit_identical_to_single_stream(amp, monkeypatch):
    # the four-lane replay (branches / exchange-unit rows on side streams, event fork / barriers / join) must produce exactly
    # the single-stream result, run after run: a missing ordering edge would show up as a difference at a filling batch size
    # Runs 1 - 2 of a plan go through the native multi-stream replay, the second one captures, runs 3 - 4 replay the hipGraph driven
    # over torch streams with star-shaped barriers (Plan._replay_lanes); "native": MINDPOSE_PLAN_GRAPH_LANES=0, never captured.
    x = torch.randn(48, 3, 256, 192, generator=torch.Generator().manual_seed(7)).to(DEV)
    outs = {}
    for tag, lanes, graph in (("graph", "1", "1"), ("native", "1", "0"), ("single", "0", "1")):
        monkeypatch.setenv("MINDPOSE_PLAN_LANES", lanes)
        monkeypatch.setenv("MINDPOSE_PLAN_GRAPH_LANES", graph)
        net = _net("hrnet_w32", "hrnet_head")
        mp.models.auto_mixed_precision(net, amp)
        outs[tag] = [net(x).clone() for _ in range(4)]
        torch.cuda.synchronize()
    for tag in outs:
        assert all(torch.equal(o, outs["single"][0]) for o in outs[tag]), tag


PW_CASES = [
    # n, cin, cout, h, w, relu, res
    (3, 64, 256, 64, 48, False, True),    # stage-1 Bottleneck conv3 + identity (hrnet.py:86-146)
    (2, 256, 64, 64, 48, True, False),    # Bottleneck conv1: four 64-channel chunks per tile
    (2, 64, 64, 64, 48, True, False),     # first Bottleneck conv1
    (5, 128, 128, 16, 12, True, True),    # two chunks, two cout blocks per wave, N x HW / 64 not a multiple of the workgroup run
    (2, 64, 200, 8, 8, False, True),      # cout not a multiple of 16 / 64: padding channels masked
]


@pytest.mark.parametrize("case", PW_CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}" for c in PW_CASES])
def test_streaming_1x1_kernel_vs_torch_and_direct(case):
    """Forced variant 8 of mp_conv2d_fwd_variant: the persistent streaming 1x1 kernel (weights in registers, all couts of a
    64-pixel tile in one workgroup) - same packed weights, same result as the direct kernel to fp32 rounding."""
    import ctypes
    from mindpose_amd import _lib
    n, cin, cout, h, w, relu, res = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    r1 = torch.randn(n, cout, h, w, generator=g) if res else None
    ref = F.conv2d(x.double(), wt.double()) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    if res:
        ref = ref + r1.double()
    if relu:
        ref = F.relu(ref)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=h, conv_w=w, out_h=h, out_w=w,
                      out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=0)
    st = _lib.stream()
    xd, wd, sc, sh = x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV)
    rd = r1.to(DEV) if res else None
    pk = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, 1, 1) // 4, device=DEV)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(pk), cout, cin, 1, 1, 0, 0, 0, st), "pack")
    out = torch.full((n, cout, h, w), float("nan"), device=DEV)
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 8, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd), None,
                                         _lib.ptr(out), st), "streaming 1x1")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert _nerr(out.double().cpu(), ref) <= 2e-5
    # outside its form: refused, the tuner then never sees it
    d2 = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=h, conv_w=w, out_h=h, out_w=w,
                       out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
    assert lib.mp_conv2d_fwd_variant(ctypes.byref(d2), 8, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd), _lib.ptr(out),
                                     _lib.ptr(out), st) == -3  # a second residual tensor
    d3 = _lib.ConvDesc(n=1, cin=48, h=8, w=8, cout=64, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=8, conv_w=8, out_h=8, out_w=8,
                       out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
    assert lib.mp_conv2d_fwd_variant(ctypes.byref(d3), 8, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), None, None, _lib.ptr(out),
                                     st) == -3  # Cin = 48 is not a built shape


@pytest.mark.parametrize("form", ["identity", "down_sample", "expand_only"])
@pytest.mark.parametrize("shape", [(3, 64, 48), (2, 16, 12), (5, 8, 8), (2, 96, 72), (1, 4, 16), (7, 32, 24)])
def test_expand_reduce_chain_f32(shape, form, monkeypatch):
    """mp_expand_reduce_fwd - expand conv of Bottleneck i + reduce conv of Bottleneck i + 1 (hrnet.py:107-146) as ONE persistent,
    weight-stationary fp32 launch (csrc/pwchain_f32.hip) - against fp64 torch at the direct kernel's bar, and against the
    mp_conv2d_fwd launches it replaces (same values up to the association of the k sums).  Forms: the residual is a tensor
    (identity), the residual is the first block's down-sample conv computed in the launch (hrnet.py:74-81), the expand conv alone
    (last block).  HRNet's stage-1 map, W48's, small maps with one / few tiles per image, more tiles than workgroups
    (MP_PWCHAIN32_WGS: a workgroup walks several tiles) and fewer."""
    import ctypes
    from mindpose_amd import _lib
    n, h, w = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(n * h * w)
    mid, res, x0 = torch.randn(n, 64, h, w, generator=g), torch.randn(n, 256, h, w, generator=g), torch.randn(n, 64, h, w, generator=g)
    w3 = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    wd = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    w1 = torch.randn(64, 256, 1, 1, generator=g) * (2.0 / 256) ** 0.5
    s3, b3 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    sd, bd = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    s1, b1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    bc = lambda v: v.double()[None, :, None, None]  # noqa: E731
    r_ref = F.conv2d(x0.double(), wd.double()) * bc(sd) + bc(bd) if form == "down_sample" else res.double()
    y_ref = F.relu(F.conv2d(mid.double(), w3.double()) * bc(s3) + bc(b3) + r_ref)
    z_ref = F.relu(F.conv2d(y_ref, w1.double()) * bc(s1) + bc(b1))
    st = _lib.stream()
    dev = lambda t: t.to(DEV)  # noqa: E731
    midd, resd, x0d, s3d, b3d, sdd, bdd, s1d, b1d = map(dev, (mid, res, x0, s3, b3, sd, bd, s1, b1))

    def pack(wt, co, ci):
        pk = torch.empty(lib.mp_conv_packed_weight_bytes(co, ci, 1, 1) // 4, device=DEV)
        _lib.check(lib.mp_conv_pack_weight(_lib.ptr(dev(wt)), _lib.ptr(pk), co, ci, 1, 1, 0, 0, 0, st), "pack")
        return pk
    pk3, pkd, pk1 = pack(w3, 256, 64), pack(wd, 256, 64), pack(w1, 64, 256)
    red = form != "expand_only"
    for wgs in (None, "3"):  # default grid; three persistent workgroups (every one walks several tiles, uneven counts)
        if wgs is not None:
            monkeypatch.setenv("MP_PWCHAIN32_WGS", wgs)
        y = torch.full((n, 256, h, w), float("nan"), device=DEV)
        z = torch.full((n, 64, h, w), float("nan"), device=DEV) if red else None
        ds = form == "down_sample"
        _lib.check(lib.mp_expand_reduce_fwd(_lib.ptr(midd), None if ds else _lib.ptr(resd), _lib.ptr(x0d) if ds else None,
                                            _lib.ptr(pkd) if ds else None, _lib.ptr(sdd) if ds else None, _lib.ptr(bdd) if ds else None,
                                            _lib.ptr(pk3), _lib.ptr(s3d), _lib.ptr(b3d), _lib.ptr(pk1) if red else None,
                                            _lib.ptr(s1d) if red else None, _lib.ptr(b1d) if red else None, _lib.ptr(y), _lib.ptr(z), n, 64, 256, 64,
                                            h, w, st), "mp_expand_reduce_fwd")
        torch.cuda.synchronize()
        assert torch.isfinite(y).all() and _nerr(y.double().cpu(), y_ref) <= 2e-5
        if red:
            assert torch.isfinite(z).all() and _nerr(z.double().cpu(), z_ref) <= 2e-5
    monkeypatch.delenv("MP_PWCHAIN32_WGS")
    if form != "down_sample":
        # the default form (four waves on 32-pixel tiles, two workgroups per CU) against the one-workgroup forms - four waves / eight
        # waves on 64-pixel tiles: same bits
        for waves in ("4", "8"):
            monkeypatch.setenv("MP_PWCHAIN32_WAVES", waves)
            y4 = torch.full((n, 256, h, w), float("nan"), device=DEV)
            z4 = torch.full((n, 64, h, w), float("nan"), device=DEV) if red else None
            _lib.check(lib.mp_expand_reduce_fwd(_lib.ptr(midd), _lib.ptr(resd), None, None, None, None, _lib.ptr(pk3), _lib.ptr(s3d), _lib.ptr(b3d),
                                                _lib.ptr(pk1) if red else None, _lib.ptr(s1d) if red else None, _lib.ptr(b1d) if red else None,
                                                _lib.ptr(y4), _lib.ptr(z4), n, 64, 256, 64, h, w, st), f"mp_expand_reduce_fwd ({waves} waves)")
            torch.cuda.synchronize()
            assert torch.equal(y4, y) and (not red or torch.equal(z4, z))
        monkeypatch.delenv("MP_PWCHAIN32_WAVES")

    # the launches it replaces
    def conv(x, pk, sc, sh, r, cin, cout, relu):
        d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=h, conv_w=w, out_h=h, out_w=w,
                          out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=relu, flags=0)
        out = torch.empty(n, cout, h, w, device=DEV)
        _lib.check(lib.mp_conv2d_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(out), st), "conv")
        return out
    r2 = conv(x0d, pkd, sdd, bdd, None, 64, 256, 0) if form == "down_sample" else resd
    y2 = conv(midd, pk3, s3d, b3d, r2, 64, 256, 1)
    torch.cuda.synchronize()
    assert _nerr(y.double().cpu(), y2.double().cpu()) <= 2e-6
    if red:
        z2 = conv(y2, pk1, s1d, b1d, None, 256, 64, 1)
        torch.cuda.synchronize()
        assert _nerr(z.double().cpu(), z2.double().cpu()) <= 4e-6


def test_expand_reduce_chain_f32_rejects_what_it_is_not_built_for():
    from mindpose_amd import _lib
    lib = _lib.load()
    f = torch.zeros(256 * 64 * 64, device=DEV)
    p = _lib.ptr(f)
    args = lambda cm, ce, cr, h, w: (p, p, None, None, None, None, p, p, p, p, p, p, p, p, 1, cm, ce, cr, h, w, _lib.stream())  # noqa: E731
    assert lib.mp_expand_reduce_fwd(*args(32, 128, 32, 8, 8)) == -3   # other widths: MP_ERR_UNSUPPORTED
    assert lib.mp_expand_reduce_fwd(*args(64, 256, 64, 8, 6)) == -3   # 48 pixels: a tile would straddle images
    assert lib.mp_expand_reduce_fwd(None, *args(64, 256, 64, 8, 8)[1:]) == -1  # MP_ERR_NULL
    both = list(args(64, 256, 64, 8, 8)); both[2:6] = [p, p, p, p]
    assert lib.mp_expand_reduce_fwd(*both) == -1                       # a residual tensor AND a down-sample conv
    ds_only = list(args(64, 256, 64, 8, 8)); ds_only[1] = None; ds_only[2:6] = [p, p, p, p]; ds_only[9:12] = [None, None, None]; ds_only[13] = None
    assert lib.mp_expand_reduce_fwd(*ds_only) == -3                    # down-sample form without a reduce conv: not built


def test_fp32_network_with_the_stage1_chain_launches_vs_one_launch_per_conv(monkeypatch):
    """The fp32 plan with the chain launches of stage 1 (MINDPOSE_FUSE_PWCHAIN32, default: three expand + reduce launches - the first
    with the down-sample conv inside - and the last block's expand conv) against the plan with one launch per conv: heat-maps equal
    to fp32 rounding, and the entries are really in the plan."""
    x = torch.randn(3, 3, 256, 192, generator=torch.Generator().manual_seed(3)).to(DEV)
    outs, kinds = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MINDPOSE_FUSE_PWCHAIN32", flag)
        net = _net("hrnet_w32", "hrnet_head")
        outs[flag] = net(x).clone()
        plan = next(iter(net._plans.values()))
        kinds[flag] = [e["kind"] for e in plan.layer_info]
    assert kinds["1"].count("pwchain_f32") == 4 and kinds["0"].count("pwchain_f32") == 0
    assert len(kinds["1"]) == len(kinds["0"]) - 4  # three reduce convs and the down-sample conv have no launch of their own
    assert _nerr(outs["1"].double().cpu(), outs["0"].double().cpu()) <= 1e-5


GEMM_CASES = [
    # n, cin, cout, h, w, stride, relu, res
    (3, 256, 1024, 16, 12, 1, False, True),   # ResNet layer3 conv3 + identity (resnet.py:74-138): 576 columns = 4.5 column tiles
    (2, 512, 128, 32, 24, 1, True, False),    # layer2 conv1: one cout tile, 32 chunks
    (5, 512, 2048, 8, 6, 1, False, True),     # layer4 conv3: 48-pixel planes, a column tile spans 2.67 images
    (2, 64, 256, 64, 48, 1, False, True),     # stage-1 conv3 (the streaming kernel's shape: both must agree)
    (3, 256, 512, 64, 48, 2, False, False),   # layer2 down_sample: stride-2 column gather, even rows / columns only
    (3, 1024, 2048, 16, 12, 2, False, False),  # layer4 down_sample: 8x6 output rows of 6 (a staging unit straddles rows)
    (2, 32, 200, 10, 6, 1, True, True),       # cout not a multiple of 16 / 128, two chunks, 120 columns in one tile
    (1, 16, 96, 7, 4, 2, False, True),        # odd height under stride 2 (conv_h = 4), one chunk
]


@pytest.mark.parametrize("ni", ["1", "2", "11"])
@pytest.mark.parametrize("case", GEMM_CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_s{c[5]}" for c in GEMM_CASES])
def test_gemm_1x1_kernel_vs_torch_and_direct(case, ni, monkeypatch):
    """Forced variant 10 of mp_conv2d_fwd_variant: the blocked-GEMM 1x1 kernel (128 x 128 / 128 x 64 output tiles over the pixel
    columns of the whole batch, 32x32x2 MFMA; the tile shape - 128 x 64, 128 x 128, 64 x 64 - is forced here, the library picks it by the
    workgroup count) on the direct kernel's packed weights - against fp64 torch at the direct kernel's own bar, and against the direct
    kernel's result."""
    import ctypes
    from mindpose_amd import _lib
    monkeypatch.setenv("MP_GEMM_NI", ni)
    n, cin, cout, h, w, stride, relu, res = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + cout + h + stride)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    r1 = torch.randn(n, cout, ho, wo, generator=g) if res else None
    ref = F.conv2d(x.double(), wt.double(), stride=stride) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    if res:
        ref = ref + r1.double()
    if relu:
        ref = F.relu(ref)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=1, kw=1, stride=stride, pad_top=0, pad_left=0, conv_h=ho, conv_w=wo, out_h=ho,
                      out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=0)
    st = _lib.stream()
    xd, wd, sc, sh = x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV)
    rd = r1.to(DEV) if res else None
    pk = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, 1, 1) // 4, device=DEV)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(pk), cout, cin, 1, 1, 0, 0, 0, st), "pack")
    out = torch.full((n, cout, ho, wo), float("nan"), device=DEV)
    out_d = torch.empty(n, cout, ho, wo, device=DEV)
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 10, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd), None,
                                         _lib.ptr(out), st), "gemm 1x1")
    _lib.check(lib.mp_conv2d_fwd(ctypes.byref(d), _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd), None,
                                 _lib.ptr(out_d), st), "direct 1x1")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()  # every element written
    assert _nerr(out.double().cpu(), ref) <= 2e-5
    assert _nerr(out.double().cpu(), out_d.double().cpu()) <= 2e-5
    # in-place on the residual (the training path's accumulate-into form) gives the same bits
    if res:
        acc = rd.clone()
        _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 10, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(acc), None,
                                             _lib.ptr(acc), st), "gemm 1x1 in place")
        torch.cuda.synchronize()
        assert torch.equal(acc, out)


def test_gemm_1x1_kernel_rejects_what_it_does_not_cover():
    import ctypes
    from mindpose_amd import _lib
    lib = _lib.load()
    st = _lib.stream()
    buf = torch.zeros(1 << 20, device=DEV)
    base = dict(n=2, cin=64, h=8, w=8, cout=128, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=8, conv_w=8, out_h=8, out_w=8,
                out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)

    def rc(res2=None, **kw):
        f = dict(base)
        f.update(kw)
        d = _lib.ConvDesc(**f)
        return lib.mp_conv2d_fwd_variant(ctypes.byref(d), 10, _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf), None, res2,
                                         _lib.ptr(buf), st)

    assert rc() == 0
    assert rc(res2=_lib.ptr(buf)) == 0                                   # a second residual tensor (round 4)
    assert rc(kh=3, kw=3, pad_top=1, pad_left=1, stride=2, conv_h=4, conv_w=4, out_h=4, out_w=4, cout=32) == -3  # 3x3 s2: < 48 couts
    assert rc(cin=40) == -3                                              # not whole 16-channel chunks
    assert rc(cout=64) == -3                                             # less than most of one 128-channel tile
    assert rc(kh=3, kw=3, pad_top=1, pad_left=1) == -3                   # neither pointwise nor a 2x2 phase
    assert rc(h=5, w=5, conv_h=5, conv_w=5, out_h=5, out_w=5) == -3      # 25-pixel planes: columns are staged in fours
    assert rc(out_mul=2, out_rep=2, out_h=16, out_w=16) == -3            # fused up-sampling stays with the direct kernel
    torch.cuda.synchronize()


GEMM_S2_CASES = [
    # n, cin, cout, h, w, relu, residual tensors - the 3x3 stride-2 pad-1 convolutions of the HRNet transitions and exchange units
    (3, 32, 64, 64, 48, True, 0),     # transition / exchange 32 -> 64 @64x48: 64-cout tile
    (5, 64, 128, 32, 24, False, 2),   # exchange unit's last down conv: running sum + identity
    (6, 128, 256, 16, 12, False, 1),  # 8x6 output planes: a column tile spans images
    (2, 256, 64, 64, 48, True, 0),    # transition 256 -> 64: 144 k-loop steps
    (3, 32, 128, 32, 24, True, 1),
    (2, 64, 64, 15, 10, False, 1),    # odd height (conv_h = 8), 40 output pixels per plane
    (2, 48, 96, 12, 8, True, 0),      # W48 widths, 24-pixel planes
]


@pytest.mark.parametrize("ni", ["0", "1", "2", "11"])
@pytest.mark.parametrize("case", GEMM_S2_CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_r{c[6]}" for c in GEMM_S2_CASES])
def test_gemm_kernel_3x3_stride2_vs_torch_and_direct(case, ni, monkeypatch):
    """Forced variant 10 on a 3x3 stride-2 pad-1 convolution (hrnet.py:280-313, 440-496): the blocked-GEMM kernel's nine-tap gather
    form - k loop over (cin chunk, tap) pairs, taps outside the image from the buffer range check - on the direct kernel's packed
    weights, against fp64 torch at the direct kernel's own bar and against the direct kernel, with up to two residual tensors."""
    import ctypes
    from mindpose_amd import _lib
    if ni != "0":
        monkeypatch.setenv("MP_GEMM_NI", ni)  # "0": the library's own tile choice
    n, cin, cout, h, w, relu, n_res = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    res = [torch.randn(n, cout, ho, wo, generator=g) for _ in range(n_res)]
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    for r in res:
        ref = ref + r.double()
    if relu:
        ref = F.relu(ref)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=3, kw=3, stride=2, pad_top=1, pad_left=1, conv_h=ho, conv_w=wo, out_h=ho,
                      out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=0)
    st = _lib.stream()
    xd, wd, sc, sh = x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV)
    rd = [r.to(DEV) for r in res] + [None, None]
    pk = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, 3, 3) // 4, device=DEV)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(pk), cout, cin, 3, 3, 0, 0, 0, st), "pack")
    out = torch.full((n, cout, ho, wo), float("nan"), device=DEV)
    out_d = torch.empty(n, cout, ho, wo, device=DEV)
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 10, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd[0]),
                                         _lib.ptr(rd[1]), _lib.ptr(out), st), "gemm 3x3 s2")
    _lib.check(lib.mp_conv2d_fwd(ctypes.byref(d), _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd[0]), _lib.ptr(rd[1]),
                                 _lib.ptr(out_d), st), "direct 3x3 s2")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()  # every element written
    assert _nerr(out.double().cpu(), ref) <= 2e-5
    assert _nerr(out.double().cpu(), out_d.double().cpu()) <= 2e-5
    if n_res:  # in-place on the running sum (how the exchange unit accumulates): same bits
        acc = rd[0].clone()
        _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 10, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(acc),
                                             _lib.ptr(rd[1]), _lib.ptr(acc), st), "gemm 3x3 s2 in place")
        torch.cuda.synchronize()
        assert torch.equal(acc, out)


@pytest.mark.parametrize("ni", ["1", "2", "11"])
@pytest.mark.parametrize("case", [(3, 64, 256, 8, 6), (2, 32, 128, 16, 12), (5, 16, 100, 4, 4)], ids=lambda c: f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}")
def test_gemm_kernel_deconv_phases_vs_conv_transpose(case, ni, monkeypatch):
    """Conv2dTranspose(k=4, s=2, p=1) + scale / shift + ReLU (simple_baseline_head.py:80-90) as four 2x2 sub-pixel phase launches of
    the blocked-GEMM kernel (forced variant 10: k loop over (cin chunk, tap), zero padding through the buffer range check, output on
    every second pixel) against torch's conv_transpose2d in fp64; every output pixel is written by exactly one phase."""
    import ctypes
    from mindpose_amd import _lib
    monkeypatch.setenv("MP_GEMM_NI", ni)
    n, cin, cout, h, w = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) * (2.0 / (cin * 4)) ** 0.5  # (Cin, Cout, 4, 4): the transposed conv's layout
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1) * scale.double()[None, :, None, None]
                 + shift.double()[None, :, None, None])
    st = _lib.stream()
    xd, wd, sc, sh = x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV)
    out = torch.full((n, cout, 2 * h, 2 * w), float("nan"), device=DEV)
    keep = []
    for py in (0, 1):
        for px in (0, 1):
            pk = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, 2, 2) // 4, device=DEV)
            keep.append(pk)
            _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(pk), cout, cin, 2, 2, 1, py, px, st), "pack phase")
            d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=2, kw=2, stride=1, pad_top=1 - py, pad_left=1 - px, conv_h=h, conv_w=w,
                              out_h=2 * h, out_w=2 * w, out_mul=2, out_rep=1, out_off_y=py, out_off_x=px, relu=1, flags=0)
            _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 10, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), None, None,
                                                 _lib.ptr(out), st), "gemm phase")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert _nerr(out.double().cpu(), ref) <= 2e-5


def test_simplebaseline_whole_network_tuned_forms_vs_library_heuristic(monkeypatch):
    """SimpleBaseline-R50 at the recipe's resolution, N large enough for the tuner to time candidates: the tuned plan - blocked-GEMM
    kernel on the pointwise / stride-2 / transposed-conv-phase launches, Winograd on the 3x3 ones - against the plan recorded with
    MINDPOSE_AUTOTUNE=0 (direct kernel everywhere, the configuration the oracle tests pin): heat-maps within 2e-5 of the output
    scale, same arg-max wherever the top-1 / top-2 margin exceeds that."""
    import mindpose_amd as mp
    x = torch.randn(32, 3, 256, 192, generator=torch.Generator().manual_seed(7)).to(DEV)
    outs, variants = [], []
    for env in ("1", "0"):
        monkeypatch.setenv("MINDPOSE_AUTOTUNE", env)
        net = mp.init_synthetic(mp.create_network("resnet50", "simple_baseline_head"), seed=0).to(DEV).eval()
        outs.append(net(x).clone())
        plan = next(iter(net._plans.values()))
        infos = [plan.entry_info(i) for i in range(len(plan))]
        variants.append([(i["kind_id"], i["variant"]) for i in infos])
    tuned, plain = variants
    assert sum(1 for k, v in tuned if k == 0 and v == 10) >= 10 and sum(1 for k, v in tuned if k == 9) >= 5
    assert all(k == 0 and v < 8 for k, v in plain if k in (0, 9))
    a, b = outs
    span = float(b.abs().max())
    assert float((a - b).abs().max()) / span <= 2e-5
    n, k = a.shape[:2]
    top2 = b.reshape(n, k, -1).topk(2, dim=2).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-4 * span
    assert torch.equal(a.reshape(n, k, -1).argmax(2)[safe], b.reshape(n, k, -1).argmax(2)[safe])


@pytest.mark.parametrize("ni", ["1", "2", "11"])
@pytest.mark.parametrize("case", [(3, 64, 256, 8, 6), (2, 32, 128, 16, 12), (5, 16, 100, 4, 4)], ids=lambda c: f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}")
def test_deconv_all_phases_in_one_gemm_launch(case, ni, monkeypatch):
    """mp_deconv4x4s2_gemm_fwd: the four sub-pixel phases of Conv2dTranspose(k=4, s=2, p=1) + scale / shift + ReLU as ONE launch (phase =
    grid dimension, the four phase packings back to back) - against conv_transpose2d in fp64, and bit-identical to the four
    per-phase launches of the same kernel."""
    import ctypes
    from mindpose_amd import _lib
    monkeypatch.setenv("MP_GEMM_NI", ni)
    n, cin, cout, h, w = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + cout + h + 1)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) * (2.0 / (cin * 4)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1) * scale.double()[None, :, None, None]
                 + shift.double()[None, :, None, None])
    st = _lib.stream()
    xd, wd, sc, sh = x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV)
    per = lib.mp_conv_packed_weight_bytes(cout, cin, 2, 2) // 4
    buf = torch.empty(4 * per, device=DEV)
    out1 = torch.full((n, cout, 2 * h, 2 * w), float("nan"), device=DEV)
    out4 = torch.full((n, cout, 2 * h, 2 * w), float("nan"), device=DEV)
    descs = []
    for i, (py, px) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(buf[i * per:(i + 1) * per]), cout, cin, 2, 2, 1, py, px, st), "pack phase")
        descs.append(_lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=2, kw=2, stride=1, pad_top=1 - py, pad_left=1 - px, conv_h=h, conv_w=w,
                                   out_h=2 * h, out_w=2 * w, out_mul=2, out_rep=1, out_off_y=py, out_off_x=px, relu=1, flags=0))
    assert lib.mp_deconv4x4s2_gemm_supported(ctypes.byref(descs[0])) == 0
    _lib.check(lib.mp_deconv4x4s2_gemm_fwd(ctypes.byref(descs[0]), _lib.ptr(xd), _lib.ptr(buf), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(out1), st),
               "one launch")
    for i, d in enumerate(descs):
        _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 10, _lib.ptr(xd), _lib.ptr(buf[i * per:(i + 1) * per]), _lib.ptr(sc), _lib.ptr(sh),
                                             None, None, _lib.ptr(out4), st), "phase launch")
    torch.cuda.synchronize()
    assert torch.isfinite(out1).all()
    assert _nerr(out1.double().cpu(), ref) <= 2e-5
    assert torch.equal(out1, out4)
    # only the phase (0, 0) descriptor of the k=4 / s=2 / p=1 recipe is taken
    assert lib.mp_deconv4x4s2_gemm_supported(ctypes.byref(descs[1])) == -3


@pytest.mark.parametrize("shape", [(3, 256, 192), (2, 384, 288), (2, 64, 64), (5, 8, 32), (1, 6, 96)])
@pytest.mark.parametrize("relu", [1, 0])
def test_stem_conv_streaming_kernel_fp32(shape, relu):
    """mp_stem_conv_fwd (the network's first conv, hrnet.py:377-385, with (tap, channel) as the k axis of the fp32 matrix cores)
    against an fp64 convolution: 1e-5 of the output scale, like the direct kernel - odd tile counts, maps smaller than a workgroup's
    8 rows, W48's 384x288 input."""
    import ctypes
    from mindpose_amd import _lib
    lib = _lib.load()
    n, h, w = shape
    g = torch.Generator().manual_seed(n * h + w + relu)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    out = torch.full((n, 64, h // 2, w // 2), 7.0, device=DEV)
    xd, wd, sd, bd = x.to(DEV), wt.to(DEV).contiguous(), scale.to(DEV), shift.to(DEV)
    _lib.check(lib.mp_stem_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(sd), _lib.ptr(bd), relu, _lib.ptr(out), n, h, w, _lib.stream()), "stem")
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if relu:
        ref = ref.clamp_min(0)
    assert _nerr(out.cpu().double(), ref) < 1e-5
    assert lib.mp_stem_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(sd), _lib.ptr(bd), relu, _lib.ptr(out), n, h + 1, w, _lib.stream()) == -3
    assert lib.mp_stem_conv_fwd(None, _lib.ptr(wd), _lib.ptr(sd), _lib.ptr(bd), relu, _lib.ptr(out), n, h, w, _lib.stream()) == -1


def test_fp32_plan_takes_the_streaming_stem(monkeypatch):
    """The fp32 HRNet plan uses the dedicated first-conv kernel (MINDPOSE_FUSE_STEM, default) and agrees with the plan that runs the
    general direct kernel there to fp32 summation order."""
    x = torch.randn(3, 3, 256, 192, generator=torch.Generator().manual_seed(4)).to(DEV)
    outs, kinds = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MINDPOSE_FUSE_STEM", flag)
        net = _net("hrnet_w32", "hrnet_head")
        outs[flag] = net(x).clone()
        kinds[flag] = [e["kind"] for e in next(iter(net._plans.values())).layer_info]
    assert kinds["1"].count("stem_f32") == 1 and kinds["0"].count("stem_f32") == 0 and len(kinds["1"]) == len(kinds["0"])
    assert _nerr(outs["1"], outs["0"]) < 2e-5


SMALL_CASES = [
    # n, cin, cout, h, w, relu, n_res, k, stride - layers at the sizes a handful of crops gives them (hrnet.py:51-64, 202-241, 280-344)
    (1, 256, 256, 8, 6, True, 1, 3, 1),     # branch 3, one crop: 48 pixels = three 16-pixel tiles, 64 cin quads over eight waves
    (1, 128, 128, 16, 12, True, 1, 3, 1),   # branch 2
    (2, 64, 64, 32, 24, True, 2, 3, 1),     # branch 1, both residuals
    (1, 32, 32, 64, 48, False, 0, 3, 1),    # branch 0
    (3, 24, 40, 7, 5, True, 1, 3, 1),       # ragged: 35 pixels per image (a padded last tile), cout % 16 != 0, 6 cin quads (idle K waves)
    (2, 6, 16, 3, 20, False, 0, 3, 1),      # cin % 4 != 0 (padding channel), a tile inside one row, H smaller than the row span
    (5, 48, 96, 5, 3, True, 0, 3, 1),       # 15 pixels per image: one partly filled tile, tiles span all five rows
    (1, 32, 64, 64, 48, True, 0, 3, 2),     # exchange-unit down-path 32 -> 64 (stride 2): staged rows / columns at twice the pitch
    (2, 64, 128, 32, 24, False, 2, 3, 2),   # ... with the running sum and the identity as residuals
    (2, 16, 24, 9, 7, True, 1, 3, 2),       # stride 2 on odd extents (5 x 4 outputs)
    (1, 256, 32, 8, 6, False, 0, 1, 1),     # exchange-unit up-path 1x1 (256 -> 32 at the low resolution)
    (3, 64, 17, 12, 10, False, 0, 1, 1),    # 1x1 with 17 couts (the head's shape class), 120 pixels per image
]


@pytest.mark.parametrize("case", SMALL_CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_k{c[7]}s{c[8]}" for c in SMALL_CASES])
def test_small_problem_kernel_vs_torch_and_direct(case):
    """Forced variant 11 of mp_conv2d_fwd_variant (csrc/conv_small_f32.hip: 16 couts x 16 pixels per workgroup, the eight waves split
    the k loop and fold through LDS in wave order; 3x3 stride 1 / 2 and 1x1) on the direct kernel's packed weights - against fp64
    torch at the direct kernel's bar, against the direct kernel, bit-reproducible run to run, and as a plan entry."""
    import ctypes
    from mindpose_amd import _lib
    n, cin, cout, h, w, relu, n_res, k, stride = case
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin * 7 + cout + h + stride)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = [torch.randn(n, cout, ho, wo, generator=g) for _ in range(n_res)]
    ref = F.conv2d(x.double(), wt.double(), stride=stride, padding=pad) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    for r in res:
        ref = ref + r.double()
    if relu:
        ref = F.relu(ref)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=stride, pad_top=pad, pad_left=pad, conv_h=ho, conv_w=wo, out_h=ho,
                      out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=0)
    st = _lib.stream()
    xd, wd = x.to(DEV), wt.to(DEV)
    padc = (-cout) % 16
    sc, sh = torch.cat([scale, torch.zeros(padc)]).to(DEV), torch.cat([shift, torch.zeros(padc)]).to(DEV)
    rd = [r.to(DEV) for r in res] + [None, None]
    pk = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, k, k) // 4, device=DEV)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(pk), cout, cin, k, k, 0, 0, 0, st), "pack")

    def run(variant):
        out = torch.full((n, cout, ho, wo), float("nan"), device=DEV)
        _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), variant, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd[0]),
                                             _lib.ptr(rd[1]), _lib.ptr(out), st), f"variant {variant}")
        torch.cuda.synchronize()
        return out

    out = run(11)
    assert torch.isfinite(out).all()  # every element written
    assert _nerr(out.double().cpu(), ref) <= 2e-5
    assert _nerr(out.double().cpu(), run(-1).double().cpu()) <= 2e-5
    assert torch.equal(out, run(11))  # fixed fold order: the same bits every launch
    # the wide form (variant 12: 48 / 64 pixels per workgroup; same k order per output, so the same bits) - every map of more than
    # one pixel tile has it
    if ho * wo > 16:
        assert torch.equal(run(12), out)
    else:
        assert lib.mp_conv2d_fwd_variant(ctypes.byref(d), 12, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), None, None,
                                         _lib.ptr(out), st) == -3
    # in place on the first residual (the accumulate-into form of the exchange unit)
    if n_res:
        acc = rd[0].clone()
        _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), 11, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(acc),
                                             _lib.ptr(rd[1]), _lib.ptr(acc), st), "in place")
        torch.cuda.synchronize()
        assert torch.equal(acc, out)
    # as a plan entry
    plan = lib.mp_plan_create()
    try:
        out_p = torch.empty_like(out)
        _lib.check(lib.mp_plan_add_conv_variant(plan, ctypes.byref(d), 11, _lib.ptr(xd), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd[0]),
                                                _lib.ptr(rd[1]), _lib.ptr(out_p)), "plan add")
        _lib.check(lib.mp_plan_run(plan, st), "plan run")
        torch.cuda.synchronize()
        assert torch.equal(out_p, out)
    finally:
        lib.mp_plan_destroy(plan)


def test_small_problem_kernel_rejects_what_it_does_not_cover():
    import ctypes
    from mindpose_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(1 << 20, device=DEV)

    def rc(**kw):
        base = dict(n=1, cin=64, h=8, w=8, cout=64, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=8, conv_w=8, out_h=8, out_w=8,
                    out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
        base.update(kw)
        d = _lib.ConvDesc(**base)
        return lib.mp_conv2d_fwd_variant(ctypes.byref(d), 11, _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf), None, None,
                                         _lib.ptr(buf), _lib.stream())

    assert rc() == 0
    assert rc(kh=1, kw=1, pad_top=0, pad_left=0) == 0 and rc(stride=2, conv_h=4, conv_w=4, out_h=4, out_w=4) == 0  # built forms
    assert rc(kh=1, kw=1, pad_top=0, pad_left=0, stride=2, conv_h=4, conv_w=4, out_h=4, out_w=4) == -3    # 1x1 stride 2
    assert rc(kh=2, kw=2, pad_top=0, pad_left=0, conv_h=7, conv_w=7, out_h=7, out_w=7) == -3              # 2x2 phase convs
    assert rc(n=128, h=64, w=48, conv_h=64, conv_w=48, out_h=64, out_w=48) == -3        # a chip-filling problem: the other forms' job
    assert rc(n=6, h=32, w=24, conv_h=32, conv_w=24, out_h=32, out_w=24) == -3           # 6 images x 48 pixel tiles x 4 cout tiles = 1152 > 1024 workgroups
    assert rc(n=5, h=32, w=24, conv_h=32, conv_w=24, out_h=32, out_w=24) == 0
    assert rc(out_mul=2, out_h=16, out_w=16) == -3                                      # no scatter mapping
