"""fp16 matrix-core path (amp level O2, BASELINE.json configs[4] "fp16 MFMA"): parity of the HIP kernels against the
CPU oracle, through the C ABI.

Tolerances (fp16 has a 2^-11 relative rounding error, accumulation is fp32 on both sides):
  * single kernels vs the oracle's same-rounding-points emulation: |got - ref| <= 2^-9 |ref| + 1e-4 max|ref|
    (one fp16 ulp when an fp32 accumulation-order difference flips a rounding);
  * whole networks: the HIP heat maps must be at least as close to the fp32 oracle as the op-by-op fp16 emulation of
    the reference's amp level O2 is (x1.5 slack), and within 2e-2 of the heat-map range in absolute terms.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import mindpose_amd as mp  # noqa: E402
from mindpose_amd import _lib  # noqa: E402
from mindpose_amd.models.layers import ActC8  # noqa: E402
from oracle import nets as onets  # noqa: E402
from tests import f16_matrix as fm  # noqa: E402
from tests.f16_matrix import CONV_CASES, WREG_CASES, WREG_S2_CASES, WS_CASES  # noqa: E402

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


@pytest.mark.parametrize("c", [3, 8, 17, 48])
def test_layout_round_trip(c):
    x = torch.randn(3, c, 10, 6, generator=torch.Generator().manual_seed(c))
    a = _to_c8(x)
    assert torch.equal(_from_c8(a), _h(x))
    assert torch.equal(a.to_nchw().cpu(), _h(x))
    # padding channels of the last block are zero
    blk = a.c8_tensor.cpu().permute(0, 1, 4, 2, 3).reshape(3, -1, 10, 6)
    assert torch.all(blk[:, c:] == 0)


# the library's heuristic (-1) on every case + every (case, tile variant) pair the library serves (tests/f16_matrix.py: generated from
# mp_f16_conv_supported, so nothing is skipped and a collected pair that fails to launch FAILS)
@pytest.mark.parametrize("case,variant", [pytest.param(c, -1, id=f"case{i}-heuristic") for i, c in enumerate(CONV_CASES)]
                         + fm.served_pairs(CONV_CASES, fm.TILE_VARIANTS, fm.conv_case_desc, fm.conv_case_res))
def test_conv_f16_vs_oracle(case, variant):
    n, cin, cout, k, s, h, w, relu, n_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    pad = k // 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    res = [torch.randn(n, cout, ho, wo, generator=g) for _ in range(n_res)]
    # oracle: fp16 operands, fp32 accumulate, fp32 epilogue, one rounding
    ref = F.conv2d(_h(x), _h(wt), None, stride=s, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    for r in res:
        ref = ref + _h(r)
    if relu:
        ref = F.relu(ref)
    ref = _h(ref)

    xa = _to_c8(x)
    ra = [_to_c8(r) for r in res] + [None, None]
    out = ActC8(n, cout, ho, wo, DEV)
    nb = LIB.mp_f16_packed_weight_bytes(cout, cin, k, k)
    packed = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
    _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wt.to(DEV)), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
    padc = (-cout) % 16
    sc = torch.cat([scale, torch.zeros(padc)]).to(DEV)
    sh = torch.cat([shift, torch.zeros(padc)]).to(DEV)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=s, pad_top=pad, pad_left=pad, conv_h=ho,
                      conv_w=wo, out_h=ho, out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu),
                      flags=0)
    rc = LIB.mp_f16_conv2d_fwd(ctypes.byref(d), variant, _lib.ptr(xa), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh),
                               _lib.ptr(ra[0]), _lib.ptr(ra[1]), _lib.ptr(out), _lib.stream())
    _lib.check(rc, f"mp_f16_conv2d_fwd, variant {variant}")
    got = _from_c8(out)
    tol = ref.abs() * 2.0 ** -9 + 1e-4 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} beyond one fp16 ulp; max diff {float((got - ref).abs().max())}"
    # padding channels stay zero (the next layer's MFMA reads them)
    blk = out.c8_tensor.cpu().permute(0, 1, 4, 2, 3).reshape(n, -1, ho, wo)
    assert torch.all(blk[:, cout:] == 0)


@pytest.mark.parametrize("groups,case,variant", [pytest.param(g, *pr.values, id=f"g{g}-{pr.id}") for g in ("1", "3") for pr in
                                                 fm.served_pairs(CONV_CASES, fm.MT_VARIANTS, fm.conv_case_desc, fm.conv_case_res, MP_F16_MT_GROUPS=g)])
def test_conv_f16_multi_tile_vs_oracle(case, variant, groups, monkeypatch):
    # the persistent multi-tile kernel only engages when a workgroup gets >= 2 tiles; MP_F16_MT_GROUPS caps the number of
    # workgroups per cout tile so that small problems exercise long tile runs (1 group = every tile in one workgroup,
    # 3 groups = ragged last run)
    monkeypatch.setenv("MP_F16_MT_GROUPS", groups)
    test_conv_f16_vs_oracle(case, variant)


@pytest.mark.parametrize("case,variant", fm.served_pairs(WREG_CASES, fm.WREG_VARIANTS, fm.conv_case_desc, fm.conv_case_res))
def test_conv_f16_wreg_vs_oracle_and_bit_identical_to_tile_kernel(case, variant):
    """The weights-in-registers kernel (LDS-DMA input tile, weight fragments streamed from global memory) against the oracle,
    and bit for bit against the one-tile kernel: same k order, same epilogue arithmetic."""
    n, cin, cout, k, s, h, w, relu, n_res = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    pad = k // 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = [torch.randn(n, cout, h, w, generator=g) for _ in range(n_res)]
    xa = _to_c8(x)
    ra = [_to_c8(r) for r in res] + [None]
    nb = LIB.mp_f16_packed_weight_bytes(cout, cin, k, k)
    packed = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
    _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wt.to(DEV)), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
    sc, sh = scale.to(DEV), shift.to(DEV)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=s, pad_top=pad, pad_left=pad, conv_h=h, conv_w=w,
                      out_h=h, out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=0)

    def run(v):
        out = ActC8(n, cout, h, w, DEV)
        out.c8_tensor.fill_(float("nan"))  # every element of the output must be written
        rc = LIB.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(xa), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(ra[0]), None,
                                   _lib.ptr(out), _lib.stream())
        return rc, out

    rc, out = run(variant)
    _lib.check(rc, f"mp_f16_conv2d_fwd, variant {variant}")
    rc0, base = run(-1)
    _lib.check(rc0, "mp_f16_conv2d_fwd")
    assert torch.equal(out.c8_tensor, base.c8_tensor), "weights-in-registers kernel differs from the tile kernel"
    if n <= 8:
        ref = F.conv2d(_h(x), _h(wt), None, stride=s, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        for r in res:
            ref = ref + _h(r)
        ref = _h(F.relu(ref) if relu else ref)
        got = _from_c8(out)
        tol = ref.abs() * 2.0 ** -9 + 1e-4 * ref.abs().max()
        assert not ((got - ref).abs() > tol).any(), float((got - ref).abs().max())


@pytest.mark.parametrize("groups,case,variant", [pytest.param(g, *pr.values, id=f"g{g}-{pr.id}") for g in ("2", "5") for pr in
                                                 fm.served_pairs(WS_CASES, fm.WS_VARIANTS, fm.conv_case_desc, fm.conv_case_res, MP_F16_WS_GROUPS=g)])
def test_conv_f16_weight_stationary_vs_oracle_and_bit_identical_to_tile_kernel(case, variant, groups, monkeypatch):
    """The persistent weight-stationary kernel (conv_f16_ws.hip: weight fragments in AGPRs / VGPRs for the whole launch, pixel tiles
    through a two-stage LDS-DMA ring) against the oracle arithmetic and bit for bit against the tile kernels; `groups` workgroups
    per cout slice force tile runs of several tiles (ring parity, ragged last band, ragged last run)."""
    monkeypatch.setenv("MP_F16_WS_GROUPS", groups)
    n, cin, cout, k, s, h, w, relu, n_res = case
    g = torch.Generator().manual_seed(sum(case[:7]) + 1)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = [torch.randn(n, cout, h, w, generator=g) for _ in range(n_res)]
    xa = _to_c8(x)
    ra = [_to_c8(r) for r in res] + [None]
    nb = LIB.mp_f16_packed_weight_bytes(cout, cin, k, k)
    packed = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
    _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wt.to(DEV)), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
    padc = (-cout) % 16
    sc, sh = torch.cat([scale, torch.zeros(padc)]).to(DEV), torch.cat([shift, torch.zeros(padc)]).to(DEV)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=s, pad_top=1, pad_left=1, conv_h=h, conv_w=w,
                      out_h=h, out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=0)

    def run(v):
        out = ActC8(n, cout, h, w, DEV)
        out.c8_tensor.fill_(float("nan"))  # every element of the output must be written
        rc = LIB.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(xa), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(ra[0]), None,
                                   _lib.ptr(out), _lib.stream())
        return rc, out

    rc, out = run(variant)
    _lib.check(rc, f"mp_f16_conv2d_fwd, variant {variant}")
    rc0, base = run(-1)
    _lib.check(rc0, "mp_f16_conv2d_fwd")
    # padding channels of the last block are zero in both; everything else bit for bit
    assert torch.equal(out.c8_tensor, base.c8_tensor), "weight-stationary kernel differs from the tile kernel"
    ref = F.conv2d(_h(x), _h(wt), None, stride=s, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    for r in res:
        ref = ref + _h(r)
    ref = _h(F.relu(ref) if relu else ref)
    got = _from_c8(out)
    tol = ref.abs() * 2.0 ** -9 + 1e-4 * ref.abs().max()
    assert not ((got - ref).abs() > tol).any(), float((got - ref).abs().max())


def test_conv_f16_weight_stationary_rejects_what_it_does_not_cover():
    t = torch.zeros(1 << 16, device=DEV)

    def rc(v, **kw):
        base = dict(n=8, cin=64, h=32, w=24, cout=64, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=32, conv_w=24, out_h=32,
                    out_w=24, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
        base.update(kw)
        d = _lib.ConvDesc(**base)
        return LIB.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None, _lib.ptr(t), _lib.stream())

    assert rc(39, stride=2, conv_h=16, conv_w=12, out_h=16, out_w=12) != 0            # stride 2
    assert rc(39, kh=1, kw=1, pad_top=0, pad_left=0) != 0                              # 1x1
    assert rc(39, cin=96) != 0 and rc(41, cin=64) != 0                                 # k-steps are a template parameter
    assert rc(39, cout=48) != 0                                                        # couts not a multiple of the workgroup's slice
    assert rc(39, n=1, h=4, conv_h=4, out_h=4) != 0                                    # one tile per workgroup: nothing to amortise


def test_conv_f16_wreg_rejects_what_it_does_not_cover():
    t = torch.zeros(1 << 16, device=DEV)

    def rc(**kw):
        base = dict(n=2, cin=128, h=16, w=12, cout=128, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=16, conv_w=12, out_h=16,
                    out_w=12, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
        base.update(kw)
        d = _lib.ConvDesc(**base)
        return LIB.mp_f16_conv2d_fwd(ctypes.byref(d), 25, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None, _lib.ptr(t), _lib.stream())

    assert rc(kh=1, kw=1, pad_top=0, pad_left=0, stride=2, conv_h=8, conv_w=6, out_h=8, out_w=6) != 0   # 1x1 stride 2
    assert rc(cin=32) != 0                                            # small K with all-pixels-per-wave: not offered
    assert rc(cout=64) != 0                                           # fewer cout tiles than 4 waves x 2
    assert rc(pad_top=0, pad_left=0, conv_h=14, conv_w=10, out_h=14, out_w=10) != 0  # not a "same" convolution


def _deconv_phase_desc(n, cin, cout, h, w, py, px):
    return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=2, kw=2, stride=1, pad_top=1 - py, pad_left=1 - px, conv_h=h, conv_w=w,
                         out_h=2 * h, out_w=2 * w, out_mul=2, out_rep=1, out_off_y=py, out_off_x=px, relu=1, flags=0)


# one-tile 1 / 3 and persistent 11 / 16: the ones of them that serve all four phases of the shape below (tests/test_f16_matrix_cpu.py
# holds the floor: at least one of each family)
DECONV_VARIANTS = [-1] + [v for v in (1, 3, 11, 16) if all(fm.supported(_deconv_phase_desc(3, 64, 48, 16, 12, py, px), v, 0, 0, MP_F16_MT_GROUPS=2)
                                                         for py in (0, 1) for px in (0, 1))]


@pytest.mark.parametrize("variant", DECONV_VARIANTS)
def test_deconv_phases_f16_vs_oracle(variant, monkeypatch):
    # Conv2dTranspose(k=4, s=2, p=1) + BN + ReLU as four 2x2 sub-pixel phase convs with the strided-scatter output mapping
    monkeypatch.setenv("MP_F16_MT_GROUPS", "2")
    g = torch.Generator().manual_seed(4)
    n, cin, cout, h, w = 3, 64, 48, 16, 12
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.conv_transpose2d(_h(x), _h(wt), None, stride=2, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = _h(F.relu(ref))
    xa = _to_c8(x)
    out = ActC8(n, cout, 2 * h, 2 * w, DEV)
    padc = (-cout) % 16
    sc, sh = torch.cat([scale, torch.zeros(padc)]).to(DEV), torch.cat([shift, torch.zeros(padc)]).to(DEV)
    nb = LIB.mp_f16_packed_weight_bytes(cout, cin, 2, 2)
    ran = False
    for py in (0, 1):
        for px in (0, 1):
            packed = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
            _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wt.to(DEV)), _lib.ptr(packed), cout, cin, 2, 2, 1, py, px, _lib.stream()), "pack")
            d = _deconv_phase_desc(n, cin, cout, h, w, py, px)
            rc = LIB.mp_f16_conv2d_fwd(ctypes.byref(d), variant, _lib.ptr(xa), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), None,
                                       None, _lib.ptr(out), _lib.stream())
            _lib.check(rc, f"deconv phase ({py}, {px}), variant {variant}")
            ran = True
    assert ran
    got = _from_c8(out)
    tol = ref.abs() * 2.0 ** -9 + 1e-4 * ref.abs().max()
    assert not ((got - ref).abs() > tol).any(), float((got - ref).abs().max())


def test_conv_f16_rejects_unsupported():
    d = _lib.ConvDesc(n=1, cin=8, h=8, w=8, cout=8, kh=7, kw=7, stride=2, pad_top=3, pad_left=3, conv_h=4, conv_w=4,
                      out_h=4, out_w=4, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
    t = torch.zeros(4096, device=DEV)
    assert LIB.mp_f16_conv2d_fwd(ctypes.byref(d), -1, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None,
                                 _lib.ptr(t), _lib.stream()) != 0
    d.kh = d.kw = 3
    d.out_mul = 2  # up-sampling epilogue is not part of the fp16 kernel
    assert LIB.mp_f16_conv2d_fwd(ctypes.byref(d), -1, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None,
                                 _lib.ptr(t), _lib.stream()) != 0


@pytest.mark.parametrize("c,h,w,scales", [(32, 64, 48, (2, 4, 8)), (48, 32, 24, (2,)), (64, 16, 12, (2, 4)), (17, 8, 8, (1, 2))])
def test_fuse_sum_f16(c, h, w, scales):
    g = torch.Generator().manual_seed(c)
    n = 3
    base = torch.randn(n, c, h, w, generator=g)
    terms = [torch.randn(n, c, h // s, w // s, generator=g) for s in scales]
    ref = _h(base)
    for t, s in zip(terms, scales):
        ref = ref + F.interpolate(_h(t), scale_factor=s, mode="nearest") if s > 1 else ref + _h(t)
    ref = _h(F.relu(ref))
    args = []
    keep = []
    for i in range(3):
        if i < len(terms):
            keep.append(_to_c8(terms[i]))
            args += [_lib.ptr(keep[-1]), scales[i]]
        else:
            args += [None, 1]
    out = ActC8(n, c, h, w, DEV)
    _lib.check(LIB.mp_f16_fuse_upsample_sum(_lib.ptr(_to_c8(base)), *args, _lib.ptr(out), n, c, h, w, 1, _lib.stream()), "fuse")
    assert torch.equal(_from_c8(out), ref)  # same fp32 sums in the same order, one rounding: bit-exact


def _net(backbone, head="hrnet_head"):
    return mp.init_synthetic(mp.create_network(backbone, head), seed=0).to(DEV).eval()


def _nerr(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("backbone,head,shape", [("hrnet_w32", "hrnet_head", (2, 3, 256, 192)),
                                                 ("hrnet_w48", "hrnet_head", (1, 3, 128, 96)),
                                                 ("hrnet_w32", "hrnet_head", (3, 3, 96, 64)),
                                                 ("resnet50", "simple_baseline_head", (2, 3, 256, 192))])
def test_network_o2_vs_amp_oracle(backbone, head, shape):
    net = _net(backbone, head)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    params = {k: v.cpu() for k, v in net.state_dict().items()}
    ref32 = onets.net_forward(params, x, backbone, head)
    ref16 = onets.net_forward(params, x, backbone, head, amp=True)
    got32 = net(x.to(DEV)).cpu().clone()
    mp.models.auto_mixed_precision(net, "O2")
    got = net(x.to(DEV)).cpu().clone()
    assert got.dtype == torch.float32 and got.shape == ref32.shape
    e_hip, e_emul = _nerr(got, ref32), _nerr(ref16, ref32)
    assert e_hip <= 1.5 * e_emul + 1e-3, f"HIP fp16 {e_hip} vs op-by-op amp-O2 emulation {e_emul}"
    assert e_hip < 2e-2
    # key-point indices: equal wherever the fp32 oracle's top-1/top-2 margin exceeds the fp16 error
    n, k = got.shape[:2]
    rf = ref32.reshape(n, k, -1)
    top2 = rf.topk(2, dim=2).values
    safe = (top2[..., 0] - top2[..., 1]) > 2.5 * e_hip * ref32.abs().max()
    assert safe.float().mean() > 0.5
    assert torch.equal(got.reshape(n, k, -1).argmax(2)[safe], rf.argmax(2)[safe])
    # switching back restores the fp32 path bit for bit
    mp.models.auto_mixed_precision(net, "O0")
    assert torch.equal(net(x.to(DEV)).cpu(), got32)


def test_config5_w48_384x288_udp_dark_flip_fp16():
    # BASELINE.json configs[4] as named: HRNet-W48 384x288, UDP + DARK decode, flip test, fp16 MFMA
    from oracle import decoder as od
    from tests.golden import recipes
    from mindpose_amd.engine.inferencer.topdown_inferencer import _MultiRunNet
    net = _net("hrnet_w48")
    mp.models.auto_mixed_precision(net, "O2")
    dec = mp.create_decoder("topdown_heatmap", use_udp=True, dark_udp_refine=True, kernel_size=17).to(DEV)
    ev = mp.create_eval_network(net, dec, output_raw=True)
    mr = _MultiRunNet(ev, dec, np.array(recipes.FLIP_INDEX), shift_heatmap=False).to(DEV)
    x = torch.randn(2, 3, 384, 288, generator=torch.Generator().manual_seed(5))
    center, scale, score = (torch.from_numpy(a) for a in recipes.boxes(2, 6))
    preds, boxes = mr(x.to(DEV), center.to(DEV), scale.to(DEV), score.to(DEV))
    # decoder parity on the fp16 network's own heat maps (decode is fp32 on both sides)
    hm = net(x.to(DEV)).cpu().numpy().copy()
    hf = net(torch.flip(x, dims=[3]).to(DEV)).cpu().numpy().copy()
    avg = od.flip_aggregate(hm, hf, recipes.FLIP_INDEX, shift_heatmap=False)
    rp, rb, _ = od.decode(avg, center.numpy(), scale.numpy(), score.numpy(), use_udp=True, dark_udp_refine=True, kernel_size=17)
    assert np.array_equal(boxes.cpu().numpy(), rb)
    d = np.abs(preds.cpu().numpy()[..., :2] - rp[..., :2])
    assert np.mean(d < 1e-2) > 0.98  # DARK's Hessian solve is ill-conditioned on a few flat maps (see the fp32 test)
    # and against the fp32 network: key points within 1/4 heat-map pixel for the well-conditioned majority
    mp.models.auto_mixed_precision(net, "O0")
    p32, _ = mr(x.to(DEV), center.to(DEV), scale.to(DEV), score.to(DEV))
    px = float(scale.max()) * 200.0 / 72.0  # image pixels per heat-map pixel (UDP, 96x72 maps)
    dd = np.abs(preds.cpu().numpy()[..., :2] - p32.cpu().numpy()[..., :2])
    assert np.mean(dd < 0.25 * px) > 0.9


def test_flip_test_as_one_batched_forward_equals_two_forwards(monkeypatch):
    """The flip test as ONE forward of [crops | mirrors] (PlannedModule.forward_flip_pair, the default) against the reference's
    two forwards (topdown_inferencer.py:168-170; MINDPOSE_FLIP_BATCHED=0): the fp16 kernels are bit-identical across tile shapes
    and inference BatchNorm is per sample, so key points and boxes must agree bit for bit."""
    from tests.golden import recipes
    from mindpose_amd.engine.inferencer.topdown_inferencer import _MultiRunNet
    net = _net("hrnet_w32")
    mp.models.auto_mixed_precision(net, "O2")
    dec = mp.create_decoder("topdown_heatmap", shift_coordinate=True).to(DEV)
    ev = mp.create_eval_network(net, dec, output_raw=True)
    mr = _MultiRunNet(ev, dec, np.array(recipes.FLIP_INDEX), shift_heatmap=True).to(DEV)
    x = torch.randn(3, 3, 256, 192, generator=torch.Generator().manual_seed(9)).to(DEV)
    center, scale, score = (torch.from_numpy(a).to(DEV) for a in recipes.boxes(3, 4))
    monkeypatch.setenv("MINDPOSE_FLIP_BATCHED", "1")
    p1, b1 = mr(x, center, scale, score)
    a1 = dec.last_argmax.clone()
    monkeypatch.setenv("MINDPOSE_FLIP_BATCHED", "0")
    p0, b0 = mr(x, center, scale, score)
    assert torch.equal(a1, dec.last_argmax) and torch.equal(p1, p0) and torch.equal(b1, b0)


def _block_pair(n, c, h, w, seed):
    """Inputs of one BasicBlock plus the two-launch result (mp_f16_conv2d_fwd x 2, heuristic variant)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g)
    ws = [torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5 for _ in range(2)]
    affine = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1) for _ in range(2)]
    xa = _to_c8(x)
    packed, scs, shs = [], [], []
    for wt, (scale, shift) in zip(ws, affine):
        nb = LIB.mp_f16_packed_weight_bytes(c, c, 3, 3)
        pk = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
        _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wt.to(DEV)), _lib.ptr(pk), c, c, 3, 3, 0, 0, 0, _lib.stream()), "pack")
        packed.append(pk)
        padc = (-c) % 16
        scs.append(torch.cat([scale, torch.zeros(padc)]).to(DEV))
        shs.append(torch.cat([shift, torch.zeros(padc)]).to(DEV))
    d = _lib.ConvDesc(n=n, cin=c, h=h, w=w, cout=c, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                      out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    mid, two = ActC8(n, c, h, w, DEV), ActC8(n, c, h, w, DEV)
    _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(d), -1, _lib.ptr(xa), _lib.ptr(packed[0]), _lib.ptr(scs[0]), _lib.ptr(shs[0]),
                                     None, None, _lib.ptr(mid), _lib.stream()), "conv1")
    _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(d), -1, _lib.ptr(mid), _lib.ptr(packed[1]), _lib.ptr(scs[1]), _lib.ptr(shs[1]),
                                     _lib.ptr(xa), None, _lib.ptr(two), _lib.stream()), "conv2")
    return xa, packed, scs, shs, two


@pytest.mark.parametrize("shape", [(3, 32, 64, 48), (2, 32, 16, 12), (2, 30, 21, 17), (1, 32, 5, 40), (5, 32, 7, 8),
                                   (3, 64, 32, 24), (2, 64, 19, 13), (5, 64, 7, 8), (1, 64, 5, 40), (2, 64, 48, 36),
                                   (3, 128, 16, 12), (2, 128, 11, 9), (5, 128, 7, 8), (2, 128, 24, 18)])
@pytest.mark.parametrize("rows", [0, 4, 1])
def test_fused_basicblock_equals_two_convs(shape, rows):
    """mp_f16_basicblock_fwd keeps the intermediate tile in LDS; same operands, k order and rounding points as the two-launch
    path, so the result is bit-identical - at the HRNet branch shape, on ragged extents (partial last row band, pixel counts
    that do not fill the 16-wide MFMA tiles) and with padding channels."""
    n, c, h, w = shape
    xa, packed, scs, shs, two = _block_pair(n, c, h, w, seed=h * w + rows)
    fused = ActC8(n, c, h, w, DEV)
    fused.c8_tensor.fill_(7.0)
    rc = LIB.mp_f16_basicblock_fwd(_lib.ptr(xa), _lib.ptr(packed[0]), _lib.ptr(scs[0]), _lib.ptr(shs[0]), _lib.ptr(packed[1]),
                                   _lib.ptr(scs[1]), _lib.ptr(shs[1]), _lib.ptr(fused), n, c, h, w, rows, _lib.stream())
    if rc == -3 and rows > 0 and LIB.mp_f16_basicblock_supported(n, c, h, w) == 1:
        pytest.skip("this forced band height does not fit the kernel's tile budget for this map (rows = 0 does)")
    _lib.check(rc, "mp_f16_basicblock_fwd")
    torch.cuda.synchronize()
    assert torch.equal(fused.c8_tensor, two.c8_tensor)


def test_fused_basicblock_rejects_other_widths():
    xa, packed, scs, shs, two = _block_pair(1, 32, 8, 8, seed=1)
    args = (_lib.ptr(packed[0]), _lib.ptr(scs[0]), _lib.ptr(shs[0]), _lib.ptr(packed[1]), _lib.ptr(scs[1]), _lib.ptr(shs[1]))
    assert LIB.mp_f16_basicblock_fwd(_lib.ptr(xa), *args, _lib.ptr(two), 1, 48, 8, 8, 0, _lib.stream()) == -3  # MP_ERR_UNSUPPORTED
    assert LIB.mp_f16_basicblock_supported(4, 64, 32, 24) == 1 and LIB.mp_f16_basicblock_supported(4, 32, 64, 48) == 1
    assert LIB.mp_f16_basicblock_supported(4, 48, 32, 24) == 0 and LIB.mp_f16_basicblock_supported(4, 64, 8, 300) == 0
    assert LIB.mp_f16_basicblock_fwd(_lib.ptr(xa), *args, _lib.ptr(xa), 1, 32, 8, 8, 0, _lib.stream()) == -3  # MP_ERR_UNSUPPORTED
    assert LIB.mp_f16_basicblock_fwd(None, *args, _lib.ptr(two), 1, 32, 8, 8, 0, _lib.stream()) == -1  # MP_ERR_NULL


@pytest.mark.parametrize("case,variant", fm.served_pairs(WREG_S2_CASES, fm.WREG_S2_VARIANTS, fm.s2_case_desc, fm.s2_case_res))
def test_conv_f16_wreg_stride2_bit_identical_to_tile_kernel(case, variant):
    n, cin, cout, h, w, n_res = case
    g = torch.Generator().manual_seed(sum(case))
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = [_to_c8(torch.randn(n, cout, ho, wo, generator=g)) for _ in range(n_res)] + [None, None]
    xa = _to_c8(x)
    nb = LIB.mp_f16_packed_weight_bytes(cout, cin, 3, 3)
    packed = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
    _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wt.to(DEV)), _lib.ptr(packed), cout, cin, 3, 3, 0, 0, 0, _lib.stream()), "pack")
    padc = (-cout) % 16
    sc, sh = torch.cat([scale, torch.zeros(padc)]).to(DEV), torch.cat([shift, torch.zeros(padc)]).to(DEV)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=3, kw=3, stride=2, pad_top=1, pad_left=1, conv_h=ho, conv_w=wo, out_h=ho,
                      out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)

    def run(v):
        out = ActC8(n, cout, ho, wo, DEV)
        out.c8_tensor.fill_(float("nan"))
        rc = LIB.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(xa), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res[0]),
                                   _lib.ptr(res[1]), _lib.ptr(out), _lib.stream())
        return rc, out

    rc, out = run(variant)
    _lib.check(rc, f"mp_f16_conv2d_fwd, variant {variant}")
    rc0, base = run(-1)
    _lib.check(rc0, "mp_f16_conv2d_fwd")
    assert torch.equal(out.c8_tensor, base.c8_tensor)
    ref = F.conv2d(_h(x), _h(wt), None, stride=2, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    for r in res[:n_res]:
        ref = ref + _from_c8(r)
    ref = _h(F.relu(ref))
    got = _from_c8(out)
    tol = ref.abs() * 2.0 ** -9 + 1e-4 * ref.abs().max()
    assert not ((got - ref).abs() > tol).any(), float((got - ref).abs().max())


def _pw_conv(xa, wt, scale, shift, res, relu, n, cin, cout, h, w):
    """one 1x1 fp16 conv launch (library's own variant choice) -> (ActC8 output, packed weights, padded scale, padded shift)"""
    nb = LIB.mp_f16_packed_weight_bytes(cout, cin, 1, 1)
    pk = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
    _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wt.to(DEV)), _lib.ptr(pk), cout, cin, 1, 1, 0, 0, 0, _lib.stream()), "pack")
    sc, sh = scale.to(DEV).contiguous(), shift.to(DEV).contiguous()
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=h, conv_w=w, out_h=h,
                      out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=relu, flags=0)
    out = ActC8(n, cout, h, w, DEV)
    _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(d), -1, _lib.ptr(xa), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res) if res is not None else None,
                                     None, _lib.ptr(out), _lib.stream()), "1x1 conv")
    return out, pk, sc, sh


@pytest.mark.parametrize("relus", [(1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("shape", [(3, 64, 48), (2, 16, 12), (5, 8, 8), (2, 96, 72), (1, 4, 16)])
def test_expand_reduce_chain_equals_two_convs(shape, relus):
    """mp_f16_expand_reduce_fwd (expand conv of Bottleneck i + reduce conv of Bottleneck i + 1, hrnet.py:107-146, y kept in LDS for
    the second GEMM) against two mp_f16_conv2d_fwd launches: y and z bit-identical - HRNet's stage-1 map, W48's, small maps with
    one / few pixel tiles per image."""
    n, h, w = shape
    g = torch.Generator().manual_seed(n * h * w + relus[0])
    mid = _to_c8(torch.randn(n, 64, h, w, generator=g))
    res = _to_c8(torch.randn(n, 256, h, w, generator=g))
    w3 = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    w1 = torch.randn(64, 256, 1, 1, generator=g) * (2.0 / 256) ** 0.5
    s3, b3 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    s1, b1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    y_ref, pk3, sc3, sh3 = _pw_conv(mid, w3, s3, b3, res, relus[0], n, 64, 256, h, w)
    z_ref, pk1, sc1, sh1 = _pw_conv(y_ref, w1, s1, b1, None, relus[1], n, 256, 64, h, w)
    y, z = ActC8(n, 256, h, w, DEV), ActC8(n, 64, h, w, DEV)
    y.c8_tensor.fill_(7.0); z.c8_tensor.fill_(7.0)
    _lib.check(LIB.mp_f16_expand_reduce_fwd(_lib.ptr(mid), _lib.ptr(res), _lib.ptr(pk3), _lib.ptr(sc3), _lib.ptr(sh3), relus[0], _lib.ptr(pk1),
                                            _lib.ptr(sc1), _lib.ptr(sh1), relus[1], _lib.ptr(y), _lib.ptr(z), n, 64, 256, 64, h, w,
                                            _lib.stream()), "mp_f16_expand_reduce_fwd")
    torch.cuda.synchronize()
    assert torch.equal(y.c8_tensor, y_ref.c8_tensor)
    assert torch.equal(z.c8_tensor, z_ref.c8_tensor)


@pytest.mark.parametrize("shape", [(3, 64, 48), (2, 16, 12), (5, 8, 8), (2, 96, 72), (1, 4, 16)])
def test_down_sample_chain_equals_three_convs(shape):
    """mp_f16_ds_expand_reduce_fwd - the FIRST Bottleneck's chain launch with its down-sample conv (hrnet.py:74-81) computed inside,
    rounded to fp16 as the stand-alone conv stores it - against the three mp_f16_conv2d_fwd launches: y and z bit-identical."""
    n, h, w = shape
    g = torch.Generator().manual_seed(7 * n * h * w)
    mid, x0 = _to_c8(torch.randn(n, 64, h, w, generator=g)), _to_c8(torch.randn(n, 64, h, w, generator=g))
    w3 = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    wd = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    w1 = torch.randn(64, 256, 1, 1, generator=g) * (2.0 / 256) ** 0.5
    s3, b3 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    sd, bd = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    s1, b1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    d_ref, pkd, scd, shd = _pw_conv(x0, wd, sd, bd, None, 0, n, 64, 256, h, w)
    y_ref, pk3, sc3, sh3 = _pw_conv(mid, w3, s3, b3, d_ref, 1, n, 64, 256, h, w)
    z_ref, pk1, sc1, sh1 = _pw_conv(y_ref, w1, s1, b1, None, 1, n, 256, 64, h, w)
    y, z = ActC8(n, 256, h, w, DEV), ActC8(n, 64, h, w, DEV)
    y.c8_tensor.fill_(7.0); z.c8_tensor.fill_(7.0)
    _lib.check(LIB.mp_f16_ds_expand_reduce_fwd(_lib.ptr(mid), _lib.ptr(x0), _lib.ptr(pkd), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(pk3), _lib.ptr(sc3),
                                               _lib.ptr(sh3), 1, _lib.ptr(pk1), _lib.ptr(sc1), _lib.ptr(sh1), 1, _lib.ptr(y), _lib.ptr(z), n, 64,
                                               256, 64, h, w, _lib.stream()), "mp_f16_ds_expand_reduce_fwd")
    torch.cuda.synchronize()
    assert torch.equal(y.c8_tensor, y_ref.c8_tensor)
    assert torch.equal(z.c8_tensor, z_ref.c8_tensor)
    assert LIB.mp_f16_ds_expand_reduce_fwd(_lib.ptr(mid), None, _lib.ptr(pkd), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(pk3), _lib.ptr(sc3),
                                           _lib.ptr(sh3), 1, _lib.ptr(pk1), _lib.ptr(sc1), _lib.ptr(sh1), 1, _lib.ptr(y), _lib.ptr(z), n, 64, 256,
                                           64, h, w, _lib.stream()) == -1  # no block input: MP_ERR_NULL


def test_expand_reduce_chain_rejects_what_it_is_not_built_for():
    a = ActC8(1, 256, 8, 8, DEV)
    p = _lib.ptr(a)
    f = torch.zeros(256, device=DEV)
    args = lambda cm, ce, cr, h, w: (p, p, p, _lib.ptr(f), _lib.ptr(f), 1, p, _lib.ptr(f), _lib.ptr(f), 1, p, p, 1, cm, ce, cr, h, w, _lib.stream())  # noqa: E731
    assert LIB.mp_f16_expand_reduce_fwd(*args(32, 128, 32, 8, 8)) == -3   # other widths: MP_ERR_UNSUPPORTED
    assert LIB.mp_f16_expand_reduce_fwd(*args(64, 256, 64, 8, 6)) == -3   # 48 pixels: a tile would straddle images
    assert LIB.mp_f16_expand_reduce_fwd(None, *args(64, 256, 64, 8, 8)[1:]) == -1  # MP_ERR_NULL


@pytest.mark.parametrize("backbone,shape", [("hrnet_w32", (3, 3, 256, 192)), ("hrnet_w48", (2, 3, 128, 96))])
def test_network_with_the_stage1_chain_launch_equals_the_two_launch_plan(backbone, shape, monkeypatch):
    """The amp-O2 plan with the expand + reduce chain launches of stage 1 (MINDPOSE_FUSE_PWCHAIN, default) against the plan with one
    launch per conv: heat-maps bit for bit, and the chain entries are really in the plan."""
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    outs, kinds = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MINDPOSE_FUSE_PWCHAIN", flag)
        net = _net(backbone)
        mp.models.auto_mixed_precision(net, "O2")
        outs[flag] = net(x).clone()
        plan = next(iter(net._plans.values())) if hasattr(net, "_plans") else None
        kinds[flag] = [e["kind"] for e in plan.layer_info] if plan is not None else None
    assert torch.equal(outs["1"], outs["0"])
    if kinds["1"] is not None:
        # three chain launches, the first with the block's down-sample conv inside: its launch and three reduce convs are gone
        assert kinds["1"].count("pwchain_f16") == 3 and kinds["0"].count("pwchain_f16") == 0
        assert len(kinds["1"]) == len(kinds["0"]) - 4
    # the form before the down-sample conv moved inside (dual 1x1 launch + identity chain): same bits again
    monkeypatch.setenv("MINDPOSE_FUSE_PWCHAIN", "1")
    monkeypatch.setenv("MINDPOSE_FUSE_PWCHAIN_DS", "0")
    net = _net(backbone)
    mp.models.auto_mixed_precision(net, "O2")
    assert torch.equal(net(x), outs["1"])
    plan = next(iter(net._plans.values())) if hasattr(net, "_plans") else None
    if plan is not None:
        assert [e["kind"] for e in plan.layer_info].count("pwchain_f16") == 4


@pytest.mark.parametrize("shape", [(3, 256, 192), (2, 384, 288), (2, 64, 64), (5, 8, 32), (1, 6, 96)])
@pytest.mark.parametrize("relu", [1, 0])
def test_stem_conv_from_the_fp32_image(shape, relu):
    """mp_f16_stem_conv_fwd (first conv of HRNet, hrnet.py:377-385, read from the fp32 NCHW image with (tap, channel) as the k axis)
    against (a) an fp64 convolution of the fp16-rounded image and weights: within one fp16 rounding of the output, and (b) the layout
    pass + general fp16 conv it replaces: equal except for rare one-ulp flips (another summation order of the same 27 products)."""
    n, h, w = shape
    g = torch.Generator().manual_seed(n * h + w + relu)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    out = ActC8(n, 64, h // 2, w // 2, DEV)
    out.c8_tensor.fill_(7.0)
    xd, wd, sd, bd = x.to(DEV), wt.to(DEV).contiguous(), scale.to(DEV), shift.to(DEV)
    _lib.check(LIB.mp_f16_stem_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(sd), _lib.ptr(bd), relu, _lib.ptr(out), n, h, w, _lib.stream()), "stem")
    got = _from_c8(out).cpu().double()
    ref = F.conv2d(x.half().double(), wt.half().double(), stride=2, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if relu:
        ref = ref.clamp_min(0)
    tol = 2.0 ** -10 * ref.abs().clamp_min(2.0 ** -14) + 1e-6 * ref.abs().max()  # one fp16 rounding + fp32 accumulation
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() / tol).max())
    # the two launches it replaces
    xa = _to_c8(x)
    nb = LIB.mp_f16_packed_weight_bytes(64, 3, 3, 3)
    pk = torch.empty(nb // 2, device=DEV, dtype=torch.float16)
    _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(wd), _lib.ptr(pk), 64, 3, 3, 3, 0, 0, 0, _lib.stream()), "pack")
    d = _lib.ConvDesc(n=n, cin=3, h=h, w=w, cout=64, kh=3, kw=3, stride=2, pad_top=1, pad_left=1, conv_h=h // 2, conv_w=w // 2, out_h=h // 2,
                      out_w=w // 2, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=relu, flags=0)
    two = ActC8(n, 64, h // 2, w // 2, DEV)
    _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(d), -1, _lib.ptr(xa), _lib.ptr(pk), _lib.ptr(sd), _lib.ptr(bd), None, None, _lib.ptr(two),
                                     _lib.stream()), "general conv")
    a, b = out.c8_tensor.float(), two.c8_tensor.float()
    assert float((a != b).float().mean()) < 2e-3
    # (near a zero of the pre-activation the two summation orders differ by fp32 rounding of O(1) terms, not by an fp16 ulp of the result)
    assert bool(((a - b).abs() <= 2.0 ** -10 * b.abs() + 4e-6 * float(b.abs().max())).all())


def test_stem_conv_rejects_other_geometries():
    x = torch.zeros(1, 3, 8, 40, device=DEV)
    wd, f = torch.zeros(64, 3, 3, 3, device=DEV), torch.zeros(64, device=DEV)
    out = ActC8(1, 64, 4, 20, DEV)
    assert LIB.mp_f16_stem_conv_fwd(_lib.ptr(x), _lib.ptr(wd), _lib.ptr(f), _lib.ptr(f), 1, _lib.ptr(out), 1, 8, 40, _lib.stream()) == -3   # w % 32
    assert LIB.mp_f16_stem_conv_fwd(_lib.ptr(x), _lib.ptr(wd), _lib.ptr(f), _lib.ptr(f), 1, _lib.ptr(out), 1, 7, 64, _lib.stream()) == -3   # odd h
    assert LIB.mp_f16_stem_conv_fwd(None, _lib.ptr(wd), _lib.ptr(f), _lib.ptr(f), 1, _lib.ptr(out), 1, 8, 64, _lib.stream()) == -1


def test_network_with_the_fused_stem_agrees_with_the_layout_pass_plan(monkeypatch):
    """amp-O2 HRNet-W32 with the first conv reading the fp32 image (MINDPOSE_FUSE_STEM, default) against layout pass + general conv:
    the stem is really in the plan, the layout pass is gone, heat-maps agree to fp16 noise of one early layer."""
    x = torch.randn(3, 3, 256, 192, generator=torch.Generator().manual_seed(4)).to(DEV)
    outs, kinds = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MINDPOSE_FUSE_STEM", flag)
        net = _net("hrnet_w32")
        mp.models.auto_mixed_precision(net, "O2")
        outs[flag] = net(x).clone()
        kinds[flag] = [e["kind"] for e in next(iter(net._plans.values())).layer_info]
    assert kinds["1"].count("stem_f16") == 1 and "to_c8" not in kinds["1"] and kinds["0"].count("to_c8") == 1
    # rare one-ulp flips in the first layer travel through ~100 fp16 layers of a randomly initialised net: two fp16 evaluations of one
    # graph, a fraction of either one's distance to the fp32 oracle (test_network_o2_vs_amp_oracle: up to 2e-2)
    assert _nerr(outs["1"], outs["0"]) < 1e-2


@pytest.mark.parametrize("shape", [(3, 64, 48), (2, 96, 72), (5, 8, 8)])
def test_dual_pointwise_launch_equals_two_convs(shape):
    """mp_f16_dual_pw_fwd (down-sample conv 64 -> 256 without ReLU + reduce conv 64 -> 64 with ReLU of stage 1's first Bottleneck,
    hrnet.py:74-81, 107-123, on ONE staged input tile) against two mp_f16_conv2d_fwd launches: both outputs bit-identical."""
    n, h, w = shape
    g = torch.Generator().manual_seed(n * h * w)
    x = _to_c8(torch.randn(n, 64, h, w, generator=g))
    wa = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    wb = torch.randn(64, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    sa, ba = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    sb, bb = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    ya_ref, pka, sca, sha = _pw_conv(x, wa, sa, ba, None, 0, n, 64, 256, h, w)
    zb_ref, pkb, scb, shb = _pw_conv(x, wb, sb, bb, None, 1, n, 64, 64, h, w)
    ya, zb = ActC8(n, 256, h, w, DEV), ActC8(n, 64, h, w, DEV)
    ya.c8_tensor.fill_(7.0); zb.c8_tensor.fill_(7.0)
    _lib.check(LIB.mp_f16_dual_pw_fwd(_lib.ptr(x), _lib.ptr(pka), _lib.ptr(sca), _lib.ptr(sha), 0, _lib.ptr(pkb), _lib.ptr(scb), _lib.ptr(shb), 1,
                                      _lib.ptr(ya), _lib.ptr(zb), n, 64, 256, 64, h, w, _lib.stream()), "mp_f16_dual_pw_fwd")
    torch.cuda.synchronize()
    assert torch.equal(ya.c8_tensor, ya_ref.c8_tensor) and torch.equal(zb.c8_tensor, zb_ref.c8_tensor)
