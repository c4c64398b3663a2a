"""GPU parity of the fp32 Winograd F(2x2,3x3) convolution (csrc/conv_wino_f32.hip) through the C ABI.

The kernel replaces the direct kernel for the 3x3 stride-1 branch convolutions of hrnet.py:51-64 / 202-241 when the per-shape
tuner finds it faster.  Same operands and epilogue; the sums are associated differently, so the comparison is against an fp64
torch-CPU formulation at 2e-5 of the output scale (the direct kernel's own single-layer bar, tests/test_gpu_conv.py), and the
direct kernel must be within the same distance of it.  Edge cases: bands that do not divide the height, fewer than 48 tiles per
band, cout tiles with a partial second half, cout not a multiple of 16, 8 / 16 input channels (1-2 chunks), two residuals,
image-grouped bands (W % 4 != 0: the 8x6 maps) incl. a batch that does not fill the last group."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mindpose_amd import _lib  # noqa: E402
from mindpose_amd.models.layers import BatchNorm2d, Conv2d, Plan, F32_WINOGRAD  # noqa: E402

DEV = torch.device("cuda:0")

CASES = [
    # n, cin, cout, h, w, relu, res1, res2
    (3, 32, 32, 64, 48, True, True, False),     # W32 branch 0: 48 tiles per band, 16-byte epilogue
    (2, 64, 64, 32, 24, True, True, False),     # branch 1
    (2, 128, 128, 16, 12, True, False, False),  # branch 2: whole image per band, tile rows of 6 (8-byte epilogue)
    (2, 256, 32, 64, 48, True, False, False),   # transition1.0: 32 chunks
    (1, 64, 64, 64, 48, False, True, True),     # stage-1 bottleneck 3x3, two residuals, no ReLU
    (2, 8, 16, 8, 8, True, True, False),        # one chunk, one 16-channel half, 16 tiles
    (2, 16, 48, 20, 16, True, False, False),    # two chunks, second cout tile has one half, 40 tiles per band, H % R != 0
    (3, 24, 40, 12, 20, False, True, False),    # cout not a multiple of 16 (padding channels masked), 3 chunks
    (1, 32, 32, 6, 96, True, False, False),     # widest supported row: one tile row per band
    (2, 48, 96, 10, 28, True, True, False),     # bands of 3 tile rows over 5: last band partly outside the image
    (5, 256, 256, 8, 6, True, True, False),     # branch 3 (W % 4 != 0): four whole images per band, N not a multiple of 4
    (2, 16, 32, 4, 6, True, False, True),       # 6 tiles per image, group clipped to the batch
    (3, 8, 16, 4, 10, False, True, False),      # 10 tiles per image, four images per band, one left over
]


def _desc(n, cin, cout, h, w, relu):
    return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                         out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=0)


# (image-grouped bands - map widths that are not a multiple of 4 - have one team: those pairs are not collected)
@pytest.mark.parametrize("case,teams", [pytest.param(c, t, id=f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}-teams{t}") for c in CASES for t in (0, 2)
                                        if not (t and c[4] % 4)])
def test_winograd_conv_vs_fp64_and_direct(case, teams, monkeypatch):
    """teams = 2 forces the two-team workgroup (64 output channels on one shared input transform; by itself only taken when the
    launch still covers every CU) on every shape incl. cout tiles whose second team is partly or wholly past Cout."""
    n, cin, cout, h, w, relu, has_r1, has_r2 = case
    if teams:
        monkeypatch.setenv("MP_WINO_TEAMS", str(teams))
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    r1 = torch.randn(n, cout, h, w, generator=g) if has_r1 else None
    r2 = torch.randn(n, cout, h, w, generator=g) if has_r2 else None
    ref = F.conv2d(x.double(), wt.double(), padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    for r in (r1, r2):
        if r is not None:
            ref = ref + r.double()
    if relu:
        ref = F.relu(ref)
    d = _desc(n, cin, cout, h, w, relu)
    assert lib.mp_conv_winograd_supported(ctypes.byref(d)) == 0
    xd, wd, sc, sh = x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV)
    r1d, r2d = (None if r is None else r.to(DEV) for r in (r1, r2))
    st = _lib.stream()
    pu = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, device=DEV)
    _lib.check(lib.mp_conv_winograd_pack_weight(_lib.ptr(wd), _lib.ptr(pu), cout, cin, st), "pack U")
    pd = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, 3, 3) // 4, device=DEV)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(pd), cout, cin, 3, 3, 0, 0, 0, st), "pack W")
    out_w = torch.full((n, cout, h, w), float("nan"), device=DEV)  # every element must be written
    out_d = torch.empty(n, cout, h, w, device=DEV)
    _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(xd), _lib.ptr(pu), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r1d),
                                          _lib.ptr(r2d), _lib.ptr(out_w), st), "winograd")
    _lib.check(lib.mp_conv2d_fwd(ctypes.byref(d), _lib.ptr(xd), _lib.ptr(pd), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r1d), _lib.ptr(r2d),
                                 _lib.ptr(out_d), st), "direct")
    torch.cuda.synchronize()
    span = float(ref.abs().max())
    err_w = float((out_w.double().cpu() - ref).abs().max()) / span
    err_d = float((out_d.double().cpu() - ref).abs().max()) / span
    assert torch.isfinite(out_w).all()
    assert err_w <= 2e-5, (err_w, err_d)
    assert err_d <= 2e-5, (err_w, err_d)


def test_winograd_result_is_deterministic_and_in_place_safe_with_residual_alias():
    """Two launches give the same bits; res1 may alias out (the exchange-unit accumulation pattern of the plan)."""
    lib = _lib.load()
    n, c, h, w = 2, 32, 32, 24
    g = torch.Generator().manual_seed(3)
    x, wt = torch.randn(n, c, h, w, generator=g).to(DEV), (torch.randn(c, c, 3, 3, generator=g) * 0.06).to(DEV)
    ones, zeros = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    base = torch.randn(n, c, h, w, generator=g).to(DEV)
    d = _desc(n, c, c, h, w, False)
    st = _lib.stream()
    pu = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(c, c) // 4, device=DEV)
    _lib.check(lib.mp_conv_winograd_pack_weight(_lib.ptr(wt), _lib.ptr(pu), c, c, st), "pack U")
    outs = []
    for _ in range(2):
        o = torch.empty_like(base)
        _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pu), _lib.ptr(ones), _lib.ptr(zeros), _lib.ptr(base),
                                              None, _lib.ptr(o), st), "winograd")
        outs.append(o)
    acc = base.clone()
    _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pu), _lib.ptr(ones), _lib.ptr(zeros), _lib.ptr(acc), None,
                                          _lib.ptr(acc), st), "winograd in place")
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(acc, outs[0])


@pytest.mark.parametrize("bad", [dict(stride=2), dict(kh=1, kw=1, pad_top=0, pad_left=0), dict(w=6, conv_w=6, out_w=6, h=6, conv_h=6, out_h=6), dict(h=7, conv_h=7, out_h=7), dict(w=18, conv_w=18, out_w=18, h=24, conv_h=24, out_h=24),
                                 dict(cin=12), dict(out_mul=2, out_rep=2, out_h=16, out_w=16), dict(w=100, conv_w=100, out_w=100), dict(flags=2)])
def test_winograd_rejects_what_it_does_not_cover(bad):
    lib = _lib.load()
    f = dict(n=1, cin=16, h=8, w=8, cout=16, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=8, conv_w=8, out_h=8, out_w=8, out_mul=1,
             out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
    f.update(bad)
    d = _lib.ConvDesc(**f)
    assert lib.mp_conv_winograd_supported(ctypes.byref(d)) == -3  # MP_ERR_UNSUPPORTED


def test_shares_cus_flag_picks_the_one_team_workgroup_with_the_same_bits():
    """A launch large enough for the two-team workgroup (64 -> 64 at 32x24, N = 64: 256 workgroups of 64 channels): with
    MP_CONV_SHARES_CUS in the descriptor (what the training step passes) the one-team form runs - a 64-channel cout tile
    becomes two 32-channel tiles, every output is still the same chunk-ordered sum, so the results are bit-identical."""
    lib = _lib.load()
    n, c, h, w = 64, 64, 32, 24
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g).to(DEV)
    wt = (torch.randn(c, c, 3, 3, generator=g) * 0.05).to(DEV)
    sc, sh = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
    st = _lib.stream()
    pu = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(c, c) // 4, device=DEV)
    _lib.check(lib.mp_conv_winograd_pack_weight(_lib.ptr(wt), _lib.ptr(pu), c, c, st), "pack U")
    outs = []
    for flags in (0, _lib.MP_CONV_SHARES_CUS):
        d = _desc(n, c, c, h, w, True)
        d.flags = flags
        out = torch.full((n, c, h, w), float("nan"), device=DEV)
        _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pu), _lib.ptr(sc), _lib.ptr(sh), None, None,
                                              _lib.ptr(out), st), "winograd")
        outs.append(out)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x.double(), wt.double(), padding=1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    assert float((outs[0].double() - ref).abs().max() / ref.abs().max()) <= 2e-5
    assert torch.equal(outs[0], outs[1])


def test_plan_takes_the_winograd_form_where_the_tuner_picks_it_and_env_turns_it_off(monkeypatch):
    """A plan of one branch conv: with the tuner on, whichever form wins is recorded (kind 9 = Winograd) and the result stays
    within 2e-5 of fp64; MINDPOSE_WINOGRAD=0 records the direct kernel."""
    g = torch.Generator().manual_seed(11)
    n, c, h, w = 16, 64, 32, 24
    conv = Conv2d(c, c, 3, stride=1, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.04)
    bn = BatchNorm2d(c)
    x = torch.randn(n, c, h, w, generator=g)
    scale, shift = bn.folded()
    ref = F.relu(F.conv2d(x.double(), conv.weight.detach().double(), padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
    kinds = []
    for env in ("1", "0"):
        monkeypatch.setenv("MINDPOSE_WINOGRAD", env)
        plan = Plan(DEV)
        xin = x.to(DEV)
        out = plan.conv(xin, conv, bn, relu=True)
        plan.run()
        torch.cuda.synchronize()
        info = plan.entry_info(0)
        kinds.append(info["kind_id"])
        assert float((out.double().cpu() - ref).abs().max() / ref.abs().max()) <= 2e-5
        if info["kind_id"] == 9:
            assert info["variant"] == F32_WINOGRAD and info["kind"] == "conv_winograd"
    assert kinds[1] == 0  # switched off: direct kernel
    assert kinds[0] in (0, 9)


def test_winograd_data_gradient_form_vs_torch_autograd_and_batched_packing():
    """Pack mode 6 = the data-gradient form (roles swapped, taps mirrored) of a forward weight: dx = winograd(dz, U6) against
    torch's autograd; pack mode 5 == mp_conv_winograd_pack_weight; both modes as jobs of mp_conv_pack_weight_batch give the same bits."""
    import numpy as np
    from mindpose_amd.models.train_ops import _PackJob
    lib = _lib.load()
    n, cin, cout, h, w = 3, 16, 40, 12, 16
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    dz = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(x, wt.double(), padding=1).backward(dz.double())
    st = _lib.stream()
    wd, dzd = wt.to(DEV), dz.to(DEV)
    nb = lib.mp_conv_winograd_packed_weight_bytes(cin, cout)  # the data-gradient conv maps cout -> cin channels
    u6 = torch.empty(nb // 4, device=DEV)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(u6), cin, cout, 3, 3, 6, 0, 0, st), "pack mode 6")
    d = _desc(n, cout, cin, h, w, False)
    ones, zeros = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)
    dx = torch.empty(n, cin, h, w, device=DEV)
    _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(dzd), _lib.ptr(u6), _lib.ptr(ones), _lib.ptr(zeros), None, None,
                                          _lib.ptr(dx), st), "winograd dgrad")
    torch.cuda.synchronize()
    assert float((dx.double().cpu() - x.grad).abs().max() / x.grad.abs().max()) <= 2e-5
    # mode 5 == the dedicated entry
    u5a = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, device=DEV)
    u5b = torch.empty_like(u5a)
    _lib.check(lib.mp_conv_winograd_pack_weight(_lib.ptr(wd), _lib.ptr(u5a), cout, cin, st), "pack U")
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(u5b), cout, cin, 3, 3, 5, 0, 0, st), "pack mode 5")
    assert torch.equal(u5a, u5b)
    # both as batch jobs (next to a direct-form job, to exercise the block table)
    pd = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, 3, 3) // 4, device=DEV)
    pd_ref = torch.empty_like(pd)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wd), _lib.ptr(pd_ref), cout, cin, 3, 3, 0, 0, 0, st), "pack direct")
    b5, b6 = torch.zeros_like(u5a), torch.zeros_like(u6)
    jobs = [(b5, cout, cin, 5), (pd, cout, cin, 0), (b6, cin, cout, 6)]
    arr = (_PackJob * len(jobs))()
    first = np.zeros(len(jobs) + 1, dtype=np.uint32)
    for i, (buf, co, ci, mode) in enumerate(jobs):
        arr[i] = _PackJob(wd.data_ptr(), buf.data_ptr(), co, ci, 3, 3, mode, 0, 0, 0)
        cp = (co + 15) // 16 * 16
        units = (ci + 3) // 4 * 4 * cp if mode in (5, 6) else (ci + 3) // 4 * 4 * 9 * (cp // 4)
        first[i + 1] = first[i] + (units + 255) // 256
    jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    first_dev = torch.from_numpy(first.view(np.int32)).to(DEV)
    _lib.check(lib.mp_conv_pack_weight_batch(_lib.ptr(jobs_dev), _lib.ptr(first_dev), len(jobs), int(first[-1]), st), "pack batch")
    torch.cuda.synchronize()
    assert torch.equal(b5, u5a) and torch.equal(b6, u6) and torch.equal(pd, pd_ref)


def test_whole_network_with_and_without_the_winograd_form(monkeypatch):
    """HRNet-W32 at the recipe's resolution, N large enough for the tuner to time candidates: the plan recorded with the Winograd
    form available contains Winograd entries, and its heat-maps agree with the all-direct plan to 2e-5 of the output scale (each
    within 1e-3 of the oracle: tests/test_gpu_conv.py) with the same arg-max wherever the top-1 / top-2 margin exceeds that."""
    import mindpose_amd as mp
    x = torch.randn(16, 3, 256, 192, generator=torch.Generator().manual_seed(5)).to(DEV)
    outs, kinds = [], []
    for env in ("1", "0"):
        monkeypatch.setenv("MINDPOSE_WINOGRAD", env)
        net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).eval()
        outs.append(net(x).clone())
        plan = next(iter(net._plans.values()))
        kinds.append([plan.entry_info(i)["kind_id"] for i in range(len(plan))])
    assert kinds[0].count(9) > 100 and kinds[1].count(9) == 0  # 213 eligible 3x3 launches when every shape picks it
    a, b = outs
    span = float(b.abs().max())
    assert float((a - b).abs().max()) / span <= 2e-5
    n, k = a.shape[:2]
    top2 = b.reshape(n, k, -1).topk(2, dim=2).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-4 * span
    assert torch.equal(a.reshape(n, k, -1).argmax(2)[safe], b.reshape(n, k, -1).argmax(2)[safe])


@pytest.mark.parametrize("tiles", [2, 3, 8])
@pytest.mark.parametrize("case", [(5, 16, 32, 16, 24), (3, 32, 48, 12, 48), (9, 24, 16, 8, 6), (2, 8, 80, 20, 16), (7, 16, 96, 12, 16, 2)],
                         ids=lambda c: f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}")
def test_winograd_several_tiles_per_workgroup(case, tiles, monkeypatch):
    """The production launches give a workgroup up to 8 consecutive tiles (next tile's rows requested before the epilogue, zero
    halo columns and staging tables reused); small test shapes would never get more than one, so the count is forced here -
    incl. counts that do not divide the number of tiles, cout tiles changing inside a workgroup's run, image-grouped bands."""
    monkeypatch.setenv("MP_WINO_TILES", str(tiles))
    n, cin, cout, h, w = case[:5]
    if len(case) > 5:  # two-team workgroups walking several tiles (96 channels: the last cout tile has an empty second team)
        monkeypatch.setenv("MP_WINO_TEAMS", str(case[5]))
    lib = _lib.load()
    g = torch.Generator().manual_seed(n * 7 + tiles)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    r1 = torch.randn(n, cout, h, w, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
                 + r1.double())
    d = _desc(n, cin, cout, h, w, True)
    assert lib.mp_conv_winograd_supported(ctypes.byref(d)) == 0
    st = _lib.stream()
    wd = wt.to(DEV)
    pu = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, device=DEV)
    _lib.check(lib.mp_conv_winograd_pack_weight(_lib.ptr(wd), _lib.ptr(pu), cout, cin, st), "pack U")
    out = torch.full((n, cout, h, w), float("nan"), device=DEV)
    xd, sc, sh, rd = x.to(DEV), scale.to(DEV), shift.to(DEV), r1.to(DEV)  # (held: the launch is asynchronous)
    _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(xd), _lib.ptr(pu), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd), None,
                                          _lib.ptr(out), st), "winograd")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert float((out.double().cpu() - ref).abs().max() / ref.abs().max()) <= 2e-5
