"""BatchNorm fused out of the amp-O2 training step (round 3): the conv launches produce the BatchNorm partial sums in their
epilogues (csrc/conv_f16_dev.h), the BatchNorm passes run as apply-only kernels, the autograd node `Chain16Fn` wires it up.

Bars: the conv OUTPUT of a statistics build is bit-identical to the plain launch (mode 2: times the ReLU mask); the partial
sums equal fp64 sums of that output to 1e-5 relative (fp32 partials over <= a few thousand values, fp64 combination); the
apply-only BatchNorm passes agree with the reduction-pass kernels to one fp16 ulp / 1e-4 on the parameter gradients; a whole
HRNet-W32 step through the fused chains agrees with the per-cell step (loss 1e-4 relative, gradient cosine > 0.9995) and is
bit-reproducible run to run.  Reference semantics: hrnet.py:51-64 (conv -> BatchNorm -> ReLU), tools/train.py:170-181 (amp O2).
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from mindpose_amd import _lib  # noqa: E402
from mindpose_amd.models.layers import ActC8, F16_VARIANTS  # noqa: E402
from tests import f16_matrix as fm  # noqa: E402

DEV = torch.device("cuda:0")
LIB = _lib.load()


def _to_c8(x):
    n, c, h, w = x.shape
    a = ActC8(n, c, h, w, DEV)
    _lib.check(LIB.mp_f16_to_c8(_lib.ptr(x.to(DEV).contiguous()), _lib.ptr(a), n, c, h, w, _lib.stream()), "to_c8")
    return a


def _from_c8(a):
    n, c, h, w = a.shape
    out = torch.empty(n, c, h, w, device=DEV)
    _lib.check(LIB.mp_f16_from_c8(_lib.ptr(a), _lib.ptr(out), n, c, h, w, _lib.stream()), "from_c8")
    return out


def _desc(n, cin, h, w, cout, k, s):
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=s, pad_top=pad, pad_left=pad, conv_h=ho, conv_w=wo,
                         out_h=ho, out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0), ho, wo


def _pack(w):
    cout, cin, k, _ = w.shape
    packed = torch.empty(LIB.mp_f16_packed_weight_bytes(cout, cin, k, k) // 2, device=DEV, dtype=torch.float16)
    _lib.check(LIB.mp_f16_pack_weight(_lib.ptr(w.to(DEV).contiguous()), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
    return packed


def _sums_from_partials(part, c, n_parts):
    """[C8][n_parts][8][2] fp32 -> per-channel (sum a, sum b) in fp64."""
    c8 = (c + 7) // 8
    p = part.double().reshape(c8, n_parts, 8, 2).sum(dim=1).reshape(c8 * 8, 2)[:c]
    return p[:, 0].cpu(), p[:, 1].cpu()


CASES = [
    # n, cin, cout, k, s, h, w
    (5, 32, 32, 3, 1, 64, 48),     # W32 branch 0 (multi-tile runs when N is large enough)
    (6, 64, 64, 3, 1, 32, 24),
    (6, 128, 128, 3, 1, 16, 12),
    (9, 256, 256, 3, 1, 8, 6),     # image-grouped tiles, ragged last group
    (3, 64, 256, 1, 1, 64, 48),    # stage-1 1x1
    (3, 256, 64, 1, 1, 32, 24),
    (3, 32, 64, 3, 2, 64, 48),     # fuse down-sampling conv
    (2, 3, 64, 3, 2, 64, 48),      # stem
    (3, 48, 48, 3, 1, 24, 18),     # W48 widths (48 couts: odd cout-tile counts)
    (2, 40, 24, 3, 1, 9, 7),       # ragged everything, cout % 16 != 0
    (40, 32, 32, 3, 1, 64, 48),    # > 512 partial slots on the one-tile kernels: the fold launch
]


@pytest.mark.parametrize("case", CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_k{c[3]}s{c[4]}_{c[5]}x{c[6]}" for c in CASES])
def test_conv_epilogue_statistics_forward(case, monkeypatch):
    n, cin, cout, k, s, h, w = case
    monkeypatch.setenv("MP_F16_MT_GROUPS", "7")  # persistent kernels: several tiles per workgroup even at this size
    monkeypatch.setenv("MP_F16_WS_GROUPS", "7")  # ... and the weight-stationary ones (variants 37 - 44 engage only with >= 2 tiles per workgroup)
    g = torch.Generator().manual_seed(sum(case))
    x = _to_c8(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    packed = _pack(wt)
    d, ho, wo = _desc(n, cin, h, w, cout, k, s)
    c16 = (cout + 15) // 16 * 16
    ones, zeros = torch.ones(c16, device=DEV), torch.zeros(c16, device=DEV)
    tested = 0
    for v in range(F16_VARIANTS):
        z0 = ActC8(n, cout, ho, wo, DEV)
        if not fm.supported(d, v, 0, 0):  # the library's own answer: a variant it serves must launch (no silent drop-outs)
            continue
        _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None, None,
                                         _lib.ptr(z0), _lib.stream()), f"plain launch, variant {v}")
        n_parts = LIB.mp_f16_conv_stats_parts(ctypes.byref(d), v)
        assert n_parts > 0, f"variant {v} runs this shape but has no statistics build"
        if not fm.supported(d, v, 0, 1):  # weight-stationary builds that do not fit the register file with the statistics (ws_build_fits)
            assert v in fm.WS_VARIANTS, f"variant {v}: only weight-stationary shapes may leave out a statistics build"
            continue
        c8 = (cout + 7) // 8
        part = torch.full((c8 * n_parts * 16,), float("nan"), device=DEV)
        st = _lib.ConvStats(mode=1, relu=0, partials=part.data_ptr(), partials_bytes=part.numel() * 4)
        z1 = ActC8(n, cout, ho, wo, DEV)
        _lib.check(LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                               _lib.ptr(z1), ctypes.byref(st), _lib.stream()), f"stats launch, variant {v}")
        assert torch.equal(z1.c8_tensor, z0.c8_tensor), f"variant {v}: output of the statistics build differs"
        zf = _from_c8(z0).double()
        ref_a, ref_b = zf.sum(dim=(0, 2, 3)).cpu(), (zf * zf).sum(dim=(0, 2, 3)).cpu()
        got_a, got_b = _sums_from_partials(part, cout, n_parts)
        assert torch.isfinite(part).all(), f"variant {v}: a partial slot was never written"
        scale_a = zf.abs().sum(dim=(0, 2, 3)).cpu().clamp_min(1e-30)
        assert ((got_a - ref_a).abs() / scale_a).max() < 1e-5, f"variant {v}: sum"
        assert ((got_b - ref_b).abs() / ref_b.clamp_min(1e-30)).max() < 1e-5, f"variant {v}: sum of squares"
        # too small a buffer is refused, not overrun
        st_small = _lib.ConvStats(mode=1, relu=0, partials=part.data_ptr(), partials_bytes=part.numel() * 4 - 4)
        assert LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                           _lib.ptr(z1), ctypes.byref(st_small), _lib.stream()) != 0
        tested += 1
    assert tested >= 3


@pytest.mark.parametrize("relu", [0, 1])
@pytest.mark.parametrize("case", [c for c in CASES if c[4] == 1 and c[1] == c[2]][:5] + [(3, 64, 256, 1, 1, 64, 48)],
                         ids=lambda c: f"n{c[0]}_{c[1]}to{c[2]}_k{c[3]}_{c[5]}x{c[6]}")
def test_conv_epilogue_statistics_backward(case, relu, monkeypatch):
    """Mode 2: a data-gradient-shaped launch (residual gradient in res1) masks its output with y > 0 and sums g, g * z."""
    n, cin, cout, k, s, h, w = case
    monkeypatch.setenv("MP_F16_MT_GROUPS", "5")
    monkeypatch.setenv("MP_F16_WS_GROUPS", "5")
    g = torch.Generator().manual_seed(sum(case) + relu)
    x = _to_c8(torch.randn(n, cin, h, w, generator=g))
    packed = _pack(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    d, ho, wo = _desc(n, cin, h, w, cout, k, s)
    res = _to_c8(torch.randn(n, cout, ho, wo, generator=g))
    zt = torch.randn(n, cout, ho, wo, generator=g) * 1.5 + 0.2
    yt = torch.relu(torch.randn(n, cout, ho, wo, generator=g))  # about half the positions closed
    z, y = _to_c8(zt), _to_c8(yt)
    c16 = (cout + 15) // 16 * 16
    ones, zeros = torch.ones(c16, device=DEV), torch.zeros(c16, device=DEV)
    tested = 0
    for v in range(F16_VARIANTS):
        o0 = ActC8(n, cout, ho, wo, DEV)
        if not fm.supported(d, v, 1, 0):
            continue
        _lib.check(LIB.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), _lib.ptr(res), None,
                                         _lib.ptr(o0), _lib.stream()), f"plain launch, variant {v}")
        n_parts = LIB.mp_f16_conv_stats_parts(ctypes.byref(d), v)
        assert n_parts > 0
        if not fm.supported(d, v, 1, 2):
            assert v in fm.WS_VARIANTS, f"variant {v}: only weight-stationary shapes may leave out a statistics build"
            continue
        c8 = (cout + 7) // 8
        part = torch.full((c8 * n_parts * 16,), float("nan"), device=DEV)
        st = _lib.ConvStats(mode=2, relu=relu, partials=part.data_ptr(), partials_bytes=part.numel() * 4, z=_lib.ptr(z),
                            y=_lib.ptr(y) if relu else None)
        o1 = ActC8(n, cout, ho, wo, DEV)
        _lib.check(LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros),
                                               _lib.ptr(res), _lib.ptr(o1), ctypes.byref(st), _lib.stream()), f"mode 2, variant {v}")
        mask = (y.c8_tensor > 0) if relu else torch.ones_like(y.c8_tensor, dtype=torch.bool)
        want = torch.where(mask, o0.c8_tensor, torch.zeros_like(o0.c8_tensor))
        assert torch.equal(o1.c8_tensor, want), f"variant {v}: stored tensor is not the masked gradient"
        gf, zf = _from_c8(o1).double(), _from_c8(z).double()
        ref_a, ref_b = gf.sum(dim=(0, 2, 3)).cpu(), (gf * zf).sum(dim=(0, 2, 3)).cpu()
        got_a, got_b = _sums_from_partials(part, cout, n_parts)
        sa, sb = gf.abs().sum(dim=(0, 2, 3)).cpu().clamp_min(1e-30), (gf * zf).abs().sum(dim=(0, 2, 3)).cpu().clamp_min(1e-30)
        assert ((got_a - ref_a).abs() / sa).max() < 1e-5 and ((got_b - ref_b).abs() / sb).max() < 1e-5, f"variant {v}"
        tested += 1
    assert tested >= 3


@pytest.mark.parametrize("c,h,w,relu,with_res,n_parts", [(32, 64, 48, True, True, 37), (64, 16, 12, True, False, 5),
                                                         (256, 8, 6, True, False, 1), (48, 24, 18, False, False, 700), (64, 16, 12, True, False, 1500),
                                                         (17, 8, 6, False, False, 3)])
def test_bn_apply_only_passes_vs_reduction_passes(c, h, w, relu, with_res, n_parts):
    """mp_f16_bn_train_fwd_stats / _bwd_stats fed with EXACT partial sums (made here in fp64, split over n_parts slots) against the
    two-pass kernels on the same tensors."""
    g = torch.Generator().manual_seed(c + h)
    n = 6
    zt = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
    z = _to_c8(zt)
    res = _to_c8(torch.randn(n, c, h, w, generator=g)) if with_res else None
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.1).to(DEV)
    nb = LIB.mp_bn_workspace_bytes(c)
    ws = torch.empty(nb // 4 + 1, device=DEV)
    c8 = (c + 7) // 8

    def partials(a, b):
        """per-channel sums a, b ([c] fp64) spread over n_parts slots with uneven weights (the fold must add them all)."""
        wts = torch.rand(n_parts, generator=g).double() + 0.1
        wts = (wts / wts.sum()).to(DEV)
        full = torch.zeros(c8 * 8, 2, dtype=torch.float64, device=DEV)
        full[:c, 0], full[:c, 1] = a, b
        p = (full.reshape(c8, 1, 8, 2) * wts.reshape(1, n_parts, 1, 1)).float().contiguous()
        return p.reshape(-1)

    # ---- forward
    zf = _from_c8(z).double()
    pf = partials(zf.sum(dim=(0, 2, 3)), (zf * zf).sum(dim=(0, 2, 3)))
    outs = []
    for fused in (False, True):
        y = ActC8(n, c, h, w, DEV)
        mean, invstd = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        mm, mv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        if fused:
            _lib.check(LIB.mp_f16_bn_train_fwd_stats(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(res), _lib.ptr(y), _lib.ptr(mean),
                                                     _lib.ptr(invstd), _lib.ptr(mm), _lib.ptr(mv), n, c, h * w, 1e-5, 0.9, int(relu),
                                                     _lib.ptr(pf), n_parts, _lib.ptr(ws), nb, _lib.stream()), "fwd_stats")
        else:
            _lib.check(LIB.mp_f16_bn_train_fwd(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(res), _lib.ptr(y), _lib.ptr(mean),
                                               _lib.ptr(invstd), _lib.ptr(mm), _lib.ptr(mv), n, c, h * w, 1e-5, 0.9, int(relu), _lib.ptr(ws),
                                               nb, _lib.stream()), "fwd")
        outs.append((y, mean, invstd, mm, mv))
    (y0, m0, i0, mm0, mv0), (y1, m1, i1, mm1, mv1) = outs
    assert torch.allclose(m1, m0, rtol=1e-5, atol=1e-6) and torch.allclose(i1, i0, rtol=1e-5)
    assert torch.allclose(mm1, mm0, rtol=1e-5, atol=1e-6) and torch.allclose(mv1, mv0, rtol=1e-5)
    a, b = _from_c8(y0), _from_c8(y1)
    tol = a.abs() * 2.0 ** -9 + 2e-4 * a.abs().max()
    assert ((a - b).abs() <= tol).all()
    assert (y0.c8_tensor != y1.c8_tensor).float().mean() < 1e-3  # statistics differ in the last bits: a rare one-ulp flip at most

    # ---- backward: the pre-masked gradient g and its sums against the two-pass kernel given dy, y
    dyt = torch.randn(n, c, h, w, generator=g)
    dy = _to_c8(dyt)
    dz0, dr0 = ActC8(n, c, h, w, DEV), (ActC8(n, c, h, w, DEV) if with_res else None)
    dg0, db0 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    _lib.check(LIB.mp_f16_bn_train_bwd(_lib.ptr(dy), _lib.ptr(z), _lib.ptr(y0), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(m0), _lib.ptr(i0),
                                       _lib.ptr(dz0), _lib.ptr(dr0), _lib.ptr(dg0), _lib.ptr(db0), None, None, n, c, h * w, int(relu),
                                       _lib.ptr(ws), nb, _lib.stream()), "bwd")
    gm = ActC8(n, c, h, w, DEV)
    gm.c8_tensor.copy_(torch.where(y0.c8_tensor > 0, dy.c8_tensor, torch.zeros_like(dy.c8_tensor)) if relu else dy.c8_tensor)
    gf = _from_c8(gm).double()
    pb = partials(gf.sum(dim=(0, 2, 3)), (gf * zf).sum(dim=(0, 2, 3)))
    dz1 = ActC8(n, c, h, w, DEV)
    dg1, db1 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    acc_g, acc_b = torch.full((c,), 1.0, device=DEV), torch.full((c,), 2.0, device=DEV)
    _lib.check(LIB.mp_f16_bn_train_bwd_stats(_lib.ptr(gm), _lib.ptr(z), _lib.ptr(gamma), _lib.ptr(m0), _lib.ptr(i0), _lib.ptr(dz1), _lib.ptr(dg1),
                                             _lib.ptr(db1), _lib.ptr(acc_g), _lib.ptr(acc_b), n, c, h * w, _lib.ptr(pb), n_parts, _lib.ptr(ws),
                                             nb, _lib.stream()), "bwd_stats")
    assert torch.equal(acc_g, dg1 + 1.0) and torch.equal(acc_b, db1 + 2.0)
    assert torch.allclose(db1, db0, rtol=1e-4, atol=1e-4 * float(db0.abs().max()))
    assert torch.allclose(dg1, dg0, rtol=1e-4, atol=1e-4 * float(dg0.abs().max()))
    a, b = _from_c8(dz0), _from_c8(dz1)
    tol = a.abs() * 2.0 ** -9 + 2e-4 * a.abs().max()
    assert ((a - b).abs() <= tol).all()
    if with_res:  # the residual branch's gradient IS the pre-masked gradient
        assert torch.equal(dr0.c8_tensor, gm.c8_tensor)


@pytest.mark.parametrize("k,relu", [(2, 1), (3, 1), (4, 0)])
@pytest.mark.parametrize("c,h,w", [(32, 64, 48), (128, 16, 12), (40, 9, 7)])
def test_fan_out_sum_with_statistics(c, h, w, k, relu):
    """mp_f16_sum_tensors_stats == mp_sum_tensors followed by the mask, bit for bit; sums of g, g * z."""
    g = torch.Generator().manual_seed(c + k)
    n = 5
    ops = [_to_c8(torch.randn(n, c, h, w, generator=g)) for _ in range(k)]
    z, y = _to_c8(torch.randn(n, c, h, w, generator=g) + 0.3), _to_c8(torch.relu(torch.randn(n, c, h, w, generator=g)))
    ref = ActC8(n, c, h, w, DEV)
    ptrs = [_lib.ptr(o) for o in ops] + [None] * (4 - k)
    _lib.check(LIB.mp_sum_tensors(*ptrs, _lib.ptr(ref), ref.c8_tensor.numel() * 2, 1, _lib.stream()), "sum")
    want = torch.where(y.c8_tensor > 0, ref.c8_tensor, torch.zeros_like(ref.c8_tensor)) if relu else ref.c8_tensor
    n_parts = LIB.mp_f16_ew_stats_parts(n, c, h * w)
    c8 = (c + 7) // 8
    part = torch.full((c8 * n_parts * 16,), float("nan"), device=DEV)
    out = ActC8(n, c, h, w, DEV)
    _lib.check(LIB.mp_f16_sum_tensors_stats(*ptrs, _lib.ptr(out), _lib.ptr(z), _lib.ptr(y) if relu else None, relu, n, c, h * w, _lib.ptr(part),
                                            part.numel() * 4, _lib.stream()), "sum stats")
    assert torch.equal(out.c8_tensor, want)
    gf, zf = _from_c8(out).double(), _from_c8(z).double()
    got_a, got_b = _sums_from_partials(part, c, n_parts)
    ref_a, ref_b = gf.sum(dim=(0, 2, 3)).cpu(), (gf * zf).sum(dim=(0, 2, 3)).cpu()
    sa, sb = gf.abs().sum(dim=(0, 2, 3)).cpu().clamp_min(1e-30), (gf * zf).abs().sum(dim=(0, 2, 3)).cpu().clamp_min(1e-30)
    assert ((got_a - ref_a).abs() / sa).max() < 1e-5 and ((got_b - ref_b).abs() / sb).max() < 1e-5
    assert LIB.mp_f16_sum_tensors_stats(*ptrs, _lib.ptr(out), _lib.ptr(z), None, 1, n, c, h * w, _lib.ptr(part), part.numel() * 4,
                                        _lib.stream()) == -1  # relu without y: MP_ERR_NULL


@pytest.mark.parametrize("s,relu_t", [(1, 1), (2, 0), (4, 0), (8, 0)])
def test_exchange_unit_backward_term_with_statistics(s, relu_t):
    """mp_f16_fuse_sum_bwd_term_stats == the same term of mp_f16_fuse_upsample_sum_bwd (times the term's own ReLU mask)."""
    g = torch.Generator().manual_seed(s)
    n, c, h, w = 3, 32, 32, 24
    dy, outp = _to_c8(torch.randn(n, c, h, w, generator=g)), _to_c8(torch.relu(torch.randn(n, c, h, w, generator=g)))
    lh, lw = h // s, w // s
    z, y = _to_c8(torch.randn(n, c, lh, lw, generator=g)), _to_c8(torch.relu(torch.randn(n, c, lh, lw, generator=g)))
    ref = ActC8(n, c, lh, lw, DEV)
    args = [None, 1, None, 1, None, 1]
    if s == 1:
        _lib.check(LIB.mp_f16_fuse_upsample_sum_bwd(_lib.ptr(dy), _lib.ptr(outp), _lib.ptr(ref), *args, n, c, h, w, 1, _lib.stream()), "bwd")
    else:
        args[0], args[1] = _lib.ptr(ref), s
        _lib.check(LIB.mp_f16_fuse_upsample_sum_bwd(_lib.ptr(dy), _lib.ptr(outp), None, *args, n, c, h, w, 1, _lib.stream()), "bwd")
    want = torch.where(y.c8_tensor > 0, ref.c8_tensor, torch.zeros_like(ref.c8_tensor)) if relu_t else ref.c8_tensor
    n_parts = LIB.mp_f16_fuse_term_stats_parts(n, c, h, w, s)
    part = torch.full((c // 8 * n_parts * 16,), float("nan"), device=DEV)
    got = ActC8(n, c, lh, lw, DEV)
    _lib.check(LIB.mp_f16_fuse_sum_bwd_term_stats(_lib.ptr(dy), _lib.ptr(outp), _lib.ptr(got), s, n, c, h, w, 1, _lib.ptr(z),
                                                  _lib.ptr(y) if relu_t else None, relu_t, _lib.ptr(part), part.numel() * 4, _lib.stream()),
               "term stats")
    assert torch.equal(got.c8_tensor, want)
    gf, zf = _from_c8(got).double(), _from_c8(z).double()
    got_a, got_b = _sums_from_partials(part, c, n_parts)
    ref_a, ref_b = gf.sum(dim=(0, 2, 3)).cpu(), (gf * zf).sum(dim=(0, 2, 3)).cpu()
    sa, sb = gf.abs().sum(dim=(0, 2, 3)).cpu().clamp_min(1e-30), (gf * zf).abs().sum(dim=(0, 2, 3)).cpu().clamp_min(1e-30)
    assert ((got_a - ref_a).abs() / sa).max() < 1e-5 and ((got_b - ref_b).abs() / sb).max() < 1e-5


@pytest.mark.parametrize("case", [(6, 32, 32, 3, 1, 64, 48, 8), (5, 64, 64, 3, 1, 32, 24, 5), (7, 128, 128, 3, 1, 16, 12, 8),
                                  (9, 256, 256, 3, 1, 8, 6, 3), (4, 64, 256, 1, 1, 32, 24, 2), (3, 32, 64, 3, 2, 64, 48, 4),
                                  (2, 40, 24, 3, 1, 9, 7, 1)], ids=lambda c: f"n{c[0]}_{c[1]}to{c[2]}_k{c[3]}s{c[4]}_{c[5]}x{c[6]}_jobs{c[7]}")
def test_grouped_weight_gradient_equals_per_layer_launches(case):
    """mp_f16_conv_wgrad_grouped (1 .. 8 layers of one shape, blockIdx.z = layer) against mp_f16_conv_wgrad per layer: the same
    products, another split-K partition (fp32 summation order: 2e-5 of the gradient's scale), accumulation into the destination,
    bit-reproducible; and against torch autograd for the first layer."""
    import torch.nn.functional as F
    n, cin, cout, k, s, h, w, jobs = case
    g = torch.Generator().manual_seed(sum(case))
    d, ho, wo = _desc(n, cin, h, w, cout, k, s)
    xs = [torch.randn(n, cin, h, w, generator=g) for _ in range(jobs)]
    dzs = [torch.randn(n, cout, ho, wo, generator=g) for _ in range(jobs)]
    xa, dza = [_to_c8(t) for t in xs], [_to_c8(t) for t in dzs]
    ref = []
    for j in range(jobs):
        dw = torch.full((cout, cin, k, k), 0.5, device=DEV)
        wsb = LIB.mp_f16_conv_wgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(wsb // 4, device=DEV)
        _lib.check(LIB.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(xa[j]), _lib.ptr(dza[j]), _lib.ptr(dw), 1.0, 1, _lib.ptr(ws), wsb,
                                         _lib.stream()), "wgrad")
        ref.append(dw)
    wsb = LIB.mp_f16_conv_wgrad_grouped_workspace_bytes(ctypes.byref(d), jobs)
    assert wsb > 0

    def grouped():
        outs = [torch.full((cout, cin, k, k), 0.5, device=DEV) for _ in range(jobs)]
        ws = torch.empty(wsb // 4, device=DEV)
        arr = ctypes.c_void_p * jobs
        _lib.check(LIB.mp_f16_conv_wgrad_grouped(ctypes.byref(d), arr(*[_lib.ptr(t) for t in xa]), arr(*[_lib.ptr(t) for t in dza]),
                                                 arr(*[_lib.ptr(t) for t in outs]), jobs, 1.0, 1, _lib.ptr(ws), wsb, _lib.stream()), "grouped")
        return outs

    got, again = grouped(), grouped()
    for j in range(jobs):
        scale = float(ref[j].abs().max())
        assert float((got[j] - ref[j]).abs().max()) <= 2e-5 * scale, j
        assert torch.equal(got[j], again[j])
    # torch autograd on the fp16-rounded operands (layer 0); the 0.5 the destination held is still there
    xt = xs[0].half().float()
    wt = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(xt, wt, stride=s, padding=k // 2).backward(dzs[0].half().float())
    assert torch.allclose(got[0].cpu() - 0.5, wt.grad, rtol=1e-3, atol=1e-4 * float(wt.grad.abs().max()))
    arr1 = ctypes.c_void_p * 9
    assert LIB.mp_f16_conv_wgrad_grouped(ctypes.byref(d), arr1(), arr1(), arr1(), 9, 1.0, 1, None, 0, _lib.stream()) != 0  # > 8 layers


def _step(fused, monkeypatch, backbone="hrnet_w32", head="hrnet_head", size=(2, 64, 64)):
    import mindpose_amd as mp
    from mindpose_amd.utils import AdamWeightDecay
    monkeypatch.setenv("MINDPOSE_BN_FUSE", "1" if fused else "0")
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network(backbone, head), seed=0).to(DEV).train()
    mp.models.auto_mixed_precision(net, "O2")
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
    g = torch.Generator().manual_seed(3)
    n, h, w = size
    x = torch.randn(n, 3, h, w, generator=g).to(DEV)
    kp = (torch.rand(n, 17, 3, generator=g) * torch.tensor([float(w), float(h), 2.0])).to(DEV)
    target, weight = mp.TopDownGenerateTarget(config=dict(image_size=[w, h], heatmap_size=[w // 4, h // 4]), sigma=2.0)(kp)
    opt.zero_grad()
    loss = nwl(x, target, weight)
    (loss * 1024.0).backward()
    stats = {k: v.clone() for k, v in net.state_dict().items() if k.endswith(("moving_mean", "moving_variance"))}
    return float(loss.detach()), opt.grads.arena.clone(), stats


def test_single_chain_statistics_match_reduction_path(monkeypatch):
    """One conv -> BatchNorm -> ReLU group, fused against per-cell, on the SAME input: the statistics (seen through the moving
    averages) agree to fp32 summation order, the outputs except for rare one-ulp flips, the gradients to 1e-3."""
    from mindpose_amd.models import train_ops as T
    from mindpose_amd.models.layers import BatchNorm2d, Conv2d
    torch.manual_seed(2)
    cv, bn = Conv2d(32, 64, 3, padding=1).to(DEV), BatchNorm2d(64).to(DEV)
    torch.nn.init.normal_(cv.weight, std=(2.0 / (9 * 32)) ** 0.5)
    x = torch.randn(16, 32, 32, 24, device=DEV)
    out = {}
    for fused in (True, False):
        monkeypatch.setenv("MINDPOSE_BN_FUSE", "1" if fused else "0")
        bn.moving_mean.zero_()
        bn.moving_variance.fill_(1.0)
        for p in list(cv.parameters()) + list(bn.parameters()):
            p.grad = None
        xi = T.to_c8(x.clone().requires_grad_(True))
        y = T.conv_bn_act(xi, cv, bn, relu=True)
        T.from_c8(y, 64).square().mean().backward()
        out[fused] = (y.detach().clone(), bn.moving_mean.clone(), bn.moving_variance.clone(), cv.weight.grad.clone(), bn.gamma.grad.clone(),
                      bn.beta.grad.clone())
    (y1, mm1, mv1, dw1, dg1, db1), (y0, mm0, mv0, dw0, dg0, db0) = out[True], out[False]
    assert torch.allclose(mm1, mm0, rtol=2e-6, atol=1e-7) and torch.allclose(mv1, mv0, rtol=2e-6)
    assert (y1 != y0).float().mean() < 1e-3
    assert float((y1.float() - y0.float()).abs().max()) <= 2.0 ** -9 * float(y0.float().abs().max())
    for a, b in ((dw1, dw0), (dg1, dg0), (db1, db0)):
        assert float((a - b).norm() / b.norm()) < 1e-3


@pytest.mark.parametrize("backbone,head,size", [("hrnet_w32", "hrnet_head", (3, 128, 96)), ("resnet50", "simple_baseline_head", (2, 64, 64))])
def test_fused_chain_step_vs_per_cell_step(backbone, head, size, monkeypatch):
    """Whole step.  The BACKWARD pieces alone (MINDPOSE_BN_FUSE_PARTS=30: gradients pre-masked and reduced by the launches that
    produce them) reproduce the per-cell gradients to summation order; the forward statistics come out in another summation order,
    which flips rare fp16 roundings of the BatchNorm outputs - through ~100 layers the two steps are then two fp16 evaluations of
    one graph, as far apart as the HIP step and the oracle's emulation are (tests/test_gpu_train_full.py: 0.99)."""
    l0, g0, s0 = _step(False, monkeypatch, backbone, head, size)
    monkeypatch.setenv("MINDPOSE_BN_FUSE_PARTS", "30")
    lb, gb, _ = _step(True, monkeypatch, backbone, head, size)
    assert lb == l0
    assert float(torch.nn.functional.cosine_similarity(gb.double(), g0.double(), dim=0)) > 0.99995
    assert float((gb - g0).norm() / g0.norm()) < 1e-2
    monkeypatch.setenv("MINDPOSE_BN_FUSE_PARTS", "31")
    l1, g1, s1 = _step(True, monkeypatch, backbone, head, size)
    assert abs(l1 - l0) <= 1e-3 * abs(l0), (l1, l0)
    cos = float(torch.nn.functional.cosine_similarity(g1.double(), g0.double(), dim=0))
    assert cos > 0.985, cos
    for k in s0:
        assert torch.allclose(s1[k], s0[k], rtol=5e-3, atol=5e-4), k
    # fixed partitions, fixed orders, no atomics: the fused step is bit-reproducible
    l2, g2, _ = _step(True, monkeypatch, backbone, head, size)
    assert l2 == l1 and torch.equal(g2, g1)


def test_grouped_batchnorm_apply_entries_equal_the_single_entries():
    """mp_f16_bn_train_{fwd,bwd}_stats_grouped on four tensors of different shapes == four single calls, bit for bit."""
    import ctypes
    from mindpose_amd.models import train_ops as T
    lib = LIB
    torch.manual_seed(5)
    shapes = [(6, 32, 16, 12), (6, 64, 8, 6), (6, 128, 4, 3), (6, 17, 10, 6)]
    fwd_jobs, bwd_jobs, singles = [], [], []
    keep = []
    for (n, c, h, w) in shapes:
        z = _to_c8(torch.randn(n, c, h, w) * 2 + 0.3)
        res = _to_c8(torch.randn(n, c, h, w))
        gy = _to_c8(torch.randn(n, c, h, w))
        c8 = (c + 7) // 8
        n_parts = 5
        # partial sums as a conv epilogue would leave them: random splits of the true sums over 5 slots
        zf = _from_c8(z)
        tot = torch.stack([zf.sum((0, 2, 3)), (zf * zf).sum((0, 2, 3))], 1)  # [c, 2]
        gf = _from_c8(gy)
        totb = torch.stack([gf.sum((0, 2, 3)), (gf * zf).sum((0, 2, 3))], 1)
        def slots(t):
            wgt = torch.rand(n_parts, device=DEV); wgt = wgt / wgt.sum()
            part = torch.zeros(c8, n_parts, 8, 2, device=DEV)
            for ch in range(c):
                part[ch // 8, :, ch % 8, :] = wgt[:, None] * t[ch][None, :]
            return part.contiguous()
        pf, pb = slots(tot), slots(totb)
        gamma, beta = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.1
        ws, ws_bytes = T._bn16_workspace(lib, c, DEV)
        ws = ws.clone()  # one workspace per job: the jobs of a grouped launch run concurrently
        out = {}
        for tag in ("one", "grp"):
            out[tag] = dict(y=torch.zeros_like(z.c8_tensor), mean=torch.zeros(c, device=DEV), invstd=torch.zeros(c, device=DEV),
                            mm=torch.zeros(c, device=DEV), mv=torch.ones(c, device=DEV), dz=torch.zeros_like(z.c8_tensor),
                            dgamma=torch.zeros(c, device=DEV), dbeta=torch.zeros(c, device=DEV))
        o = out["one"]
        _lib.check(lib.mp_f16_bn_train_fwd_stats(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(res), _lib.ptr(o["y"]), _lib.ptr(o["mean"]),
                                                 _lib.ptr(o["invstd"]), _lib.ptr(o["mm"]), _lib.ptr(o["mv"]), n, c, h * w, 1e-5, 0.9, 1,
                                                 _lib.ptr(pf), n_parts, _lib.ptr(ws), ws_bytes, _lib.stream()), "fwd one")
        _lib.check(lib.mp_f16_bn_train_bwd_stats(_lib.ptr(gy), _lib.ptr(z), _lib.ptr(gamma), _lib.ptr(o["mean"]), _lib.ptr(o["invstd"]),
                                                 _lib.ptr(o["dz"]), _lib.ptr(o["dgamma"]), _lib.ptr(o["dbeta"]), None, None, n, c, h * w,
                                                 _lib.ptr(pb), n_parts, _lib.ptr(ws), ws_bytes, _lib.stream()), "bwd one")
        q = out["grp"]
        fwd_jobs.append(_lib.BnFwdJob(z=_lib.ptr(z), gamma=_lib.ptr(gamma), beta=_lib.ptr(beta), res=_lib.ptr(res), y=_lib.ptr(q["y"]),
                                      save_mean=_lib.ptr(q["mean"]), save_invstd=_lib.ptr(q["invstd"]), moving_mean=_lib.ptr(q["mm"]),
                                      moving_var=_lib.ptr(q["mv"]), partials=_lib.ptr(pf), workspace=_lib.ptr(ws), workspace_bytes=ws_bytes,
                                      n=n, c=c, hw=h * w, relu=1, n_parts=n_parts, reserved=0))
        bwd_jobs.append(_lib.BnBwdJob(g=_lib.ptr(gy), z=_lib.ptr(z), gamma=_lib.ptr(gamma), save_mean=_lib.ptr(o["mean"]),
                                      save_invstd=_lib.ptr(o["invstd"]), dz=_lib.ptr(q["dz"]), dgamma=_lib.ptr(q["dgamma"]),
                                      dbeta=_lib.ptr(q["dbeta"]), dgamma_acc=None, dbeta_acc=None, partials=_lib.ptr(pb),
                                      workspace=_lib.ptr(ws), workspace_bytes=ws_bytes, n=n, c=c, hw=h * w, n_parts=n_parts))
        singles.append(out)
        keep.append((z, res, gy, pf, pb, gamma, beta, ws))
    fa = (_lib.BnFwdJob * 4)(*fwd_jobs)
    ba = (_lib.BnBwdJob * 4)(*bwd_jobs)
    _lib.check(lib.mp_f16_bn_train_fwd_stats_grouped(fa, 4, 1e-5, 0.9, _lib.stream()), "fwd grouped")
    _lib.check(lib.mp_f16_bn_train_bwd_stats_grouped(ba, 4, _lib.stream()), "bwd grouped")
    torch.cuda.synchronize()
    for out in singles:
        for k in out["one"]:
            assert torch.equal(out["one"][k], out["grp"][k]), k
    # argument checks
    assert lib.mp_f16_bn_train_fwd_stats_grouped(fa, 5, 1e-5, 0.9, _lib.stream()) != 0
    assert lib.mp_f16_bn_train_bwd_stats_grouped(ba, 0, _lib.stream()) != 0


def test_link_is_dropped_when_a_tensor_has_two_consumers(monkeypatch):
    """The cross-node hand-over (the next chain's data gradient reduces for the previous chain's last BatchNorm) is only valid for a
    single consumer: with two, autograd sums two gradients and the BatchNorm must do its own reduction."""
    import mindpose_amd as mp
    from mindpose_amd.models import train_ops as T
    from mindpose_amd.models.layers import BatchNorm2d, Conv2d
    torch.manual_seed(1)

    def mk(cin, cout):
        cv, bn = Conv2d(cin, cout, 3, padding=1), BatchNorm2d(cout)
        torch.nn.init.normal_(cv.weight, std=(2.0 / (9 * cin)) ** 0.5)
        return cv.to(DEV), bn.to(DEV)

    (c0, b0), (c1, b1), (c2, b2) = mk(16, 32), mk(32, 32), mk(32, 32)
    x = torch.randn(2, 16, 16, 12, device=DEV)

    def run(two_consumers, fused):
        monkeypatch.setenv("MINDPOSE_BN_FUSE", "1" if fused else "0")
        for m in (c0, b0, c1, b1, c2, b2):
            for p in m.parameters():
                p.grad = None
        a = T.conv_bn_act(T.to_c8(x), c0, b0, relu=True)
        link = getattr(a, "_mp_bn_link", None)
        assert (link is not None) == fused
        u = T.conv_bn_act(a, c1, b1, relu=True)
        out = T.from_c8(u, 32)
        if two_consumers:
            out = out + T.from_c8(T.conv_bn_act(a, c2, b2, relu=True), 32)
        out.square().mean().backward()
        if fused:
            assert link.claimed == (2 if two_consumers else 1)
            assert (link.partials is not None) == (not two_consumers)
        return torch.cat([p.grad.flatten() for m in (c0, b0, c1, b1) for p in m.parameters()])

    for two in (False, True):
        gf, gr = run(two, True), run(two, False)
        cos = float(torch.nn.functional.cosine_similarity(gf.double(), gr.double(), dim=0))
        assert cos > 0.9999, (two, cos)


def test_link_hand_over_is_checked_against_the_gradient_tensor(monkeypatch):
    """A consumer that never announces itself (a plain torch op on the channel-blocked activation: an auxiliary loss, a hook) leaves
    ``claimed`` at 1, but autograd then SUMS its gradient with the chain's pre-masked one.  The producer must notice that the
    gradient it receives is not the tensor the consumer's data-gradient launch wrote, and reduce for its BatchNorm itself."""
    from mindpose_amd.models import train_ops as T
    from mindpose_amd.models.layers import BatchNorm2d, Conv2d
    torch.manual_seed(2)

    def mk(cin, cout):
        cv, bn = Conv2d(cin, cout, 3, padding=1), BatchNorm2d(cout)
        torch.nn.init.normal_(cv.weight, std=(2.0 / (9 * cin)) ** 0.5)
        return cv.to(DEV), bn.to(DEV)

    (c0, b0), (c1, b1) = mk(16, 32), mk(32, 32)
    x = torch.randn(2, 16, 16, 12, device=DEV)

    def run(aux, fused):
        monkeypatch.setenv("MINDPOSE_BN_FUSE", "1" if fused else "0")
        for m in (c0, b0, c1, b1):
            for p in m.parameters():
                p.grad = None
        a = T.conv_bn_act(T.to_c8(x), c0, b0, relu=True)
        link = getattr(a, "_mp_bn_link", None)
        loss = T.from_c8(T.conv_bn_act(a, c1, b1, relu=True), 32).square().mean()
        if aux:
            loss = loss + 0.37 * a.float().square().mean()  # plain torch on the c8 tensor: no _claim
        loss.backward()
        if fused:
            assert link.claimed == 1 and link.partials is not None  # the consumer did hand over ...
        return torch.cat([p.grad.flatten() for m in (c0, b0, c1, b1) for p in m.parameters()])

    for aux in (False, True):
        gf, gr = run(aux, True), run(aux, False)
        cos = float(torch.nn.functional.cosine_similarity(gf.double(), gr.double(), dim=0))
        assert cos > 0.9999, (aux, cos)  # ... and with the extra consumer the producer did not use it


# ---- BatchNorm apply on the consumer's operand (round 4): conv1 -> bn1 -> relu -> conv2 without a pass over y1 ------------------
PRE_CASES = [
    # n, cin, cout, h, w
    (5, 32, 32, 64, 48),    # W32 branch 0
    (6, 64, 64, 32, 24),
    (6, 128, 128, 16, 12),
    (3, 48, 48, 24, 18),    # W48 width: padding planes behind Cin (48 -> 64)
    (3, 40, 64, 21, 13),    # ragged rows (last tile short), cin not a multiple of 16
    (2, 96, 96, 48, 36),
]


@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("case", PRE_CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}" for c in PRE_CASES])
def test_conv_applies_the_batchnorm_below_on_its_operand(case, relu):
    """mp_f16_bn_train_finalize + mp_f16_conv2d_fwd_stats(pre_scale / pre_shift / pre_out) against the apply pass
    (mp_f16_bn_train_fwd_stats) followed by the same conv launch: activation tensor, conv output, its partial sums, saved statistics
    and moving averages all bit-identical, on every variant that takes the form (hrnet.py:67-72; nn.BatchNorm2d in training mode)."""
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(sum(case) + relu)
    z = _to_c8(torch.randn(n, cin, h, w, generator=g) * 1.7 + 0.3)
    # partial sums of z as a producing conv would leave them: [C8][n_parts][8][2], here three slots splitting the batch
    zf = _from_c8(z)
    c8 = (cin + 7) // 8
    n_parts_in = 3
    part_in = torch.zeros(c8, n_parts_in, 8, 2, device=DEV)
    for sl, idx in enumerate(torch.arange(n).chunk(n_parts_in)):
        zz = zf[idx.to(DEV)]
        a = torch.zeros(c8 * 8, device=DEV); b = torch.zeros(c8 * 8, device=DEV)
        a[:cin] = zz.sum(dim=(0, 2, 3)); b[:cin] = (zz * zz).sum(dim=(0, 2, 3))
        part_in[:, sl, :, 0] = a.reshape(c8, 8); part_in[:, sl, :, 1] = b.reshape(c8, 8)
    part_in = part_in.reshape(-1).contiguous()
    gamma, beta = torch.rand(cin, generator=g).to(DEV) + 0.5, (torch.randn(cin, generator=g) * 0.2).to(DEV)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    packed = _pack(wt)
    d, ho, wo = _desc(n, cin, h, w, cout, 3, 1)
    c16 = (cout + 15) // 16 * 16
    ones, zeros = torch.ones(c16, device=DEV), torch.zeros(c16, device=DEV)
    wsb = LIB.mp_bn_workspace_bytes(2048)
    ws = torch.zeros(wsb // 4 + 1, device=DEV)

    # route A: the apply pass, then the conv on the activation
    mean_a, inv_a = torch.empty(cin, device=DEV), torch.empty(cin, device=DEV)
    mm_a, mv_a = torch.zeros(cin, device=DEV), torch.ones(cin, device=DEV)
    y_a = ActC8(n, cin, h, w, DEV)
    _lib.check(LIB.mp_f16_bn_train_fwd_stats(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), None, _lib.ptr(y_a), _lib.ptr(mean_a), _lib.ptr(inv_a),
                                             _lib.ptr(mm_a), _lib.ptr(mv_a), n, cin, h * w, 1e-5, 0.9, relu, _lib.ptr(part_in), n_parts_in,
                                             _lib.ptr(ws), wsb, _lib.stream()), "apply")
    # route B: statistics only ...
    mean_b, inv_b = torch.empty(cin, device=DEV), torch.empty(cin, device=DEV)
    mm_b, mv_b = torch.zeros(cin, device=DEV), torch.ones(cin, device=DEV)
    sc, sh = torch.full((c8 * 8,), float("nan"), device=DEV), torch.full((c8 * 8,), float("nan"), device=DEV)
    _lib.check(LIB.mp_f16_bn_train_finalize(_lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean_b), _lib.ptr(inv_b), _lib.ptr(mm_b), _lib.ptr(mv_b), n,
                                            cin, h * w, 1e-5, 0.9, _lib.ptr(part_in), n_parts_in, _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(ws), wsb,
                                            _lib.stream()), "finalize")
    assert torch.equal(mean_b, mean_a) and torch.equal(inv_b, inv_a) and torch.equal(mm_b, mm_a) and torch.equal(mv_b, mv_a)
    assert torch.isfinite(sc).all() and torch.isfinite(sh).all() and (sc[cin:] == 0).all() and (sh[cin:] == 0).all()
    tested = 0
    for v in range(F16_VARIANTS):
        if not LIB.mp_f16_conv_pre_supported(ctypes.byref(d), v):
            continue
        n_parts = LIB.mp_f16_conv_stats_parts(ctypes.byref(d), v)
        assert n_parts > 0
        co8 = (cout + 7) // 8
        pa = torch.full((co8 * n_parts * 16,), float("nan"), device=DEV)
        pb = torch.full((co8 * n_parts * 16,), float("nan"), device=DEV)
        z2a, z2b = ActC8(n, cout, ho, wo, DEV), ActC8(n, cout, ho, wo, DEV)
        st = _lib.ConvStats(mode=1, relu=0, partials=pa.data_ptr(), partials_bytes=pa.numel() * 4)
        _lib.check(LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(y_a), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                               _lib.ptr(z2a), ctypes.byref(st), _lib.stream()), f"plain route, variant {v}")
        # ... and the conv applies them on the raw tensor, writing the activation on the way
        y_b = ActC8(n, cin, h, w, DEV)
        y_b.c8_tensor.fill_(float("nan"))
        st = _lib.ConvStats(mode=1, relu=0, partials=pb.data_ptr(), partials_bytes=pb.numel() * 4, pre_scale=_lib.ptr(sc), pre_shift=_lib.ptr(sh),
                            pre_out=_lib.ptr(y_b), pre_relu=relu)
        _lib.check(LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(z), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                               _lib.ptr(z2b), ctypes.byref(st), _lib.stream()), f"operand route, variant {v}")
        assert torch.equal(y_b.c8_tensor, y_a.c8_tensor), f"variant {v}: activation written by the conv differs from the apply pass"
        assert torch.equal(z2b.c8_tensor, z2a.c8_tensor), f"variant {v}: conv output"
        assert torch.equal(pb, pa), f"variant {v}: partial sums"
        # without an activation tensor the launch still convolves the same operand
        z2c = ActC8(n, cout, ho, wo, DEV)
        st = _lib.ConvStats(mode=1, relu=0, partials=pb.data_ptr(), partials_bytes=pb.numel() * 4, pre_scale=_lib.ptr(sc), pre_shift=_lib.ptr(sh),
                            pre_relu=relu)
        _lib.check(LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(z), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                               _lib.ptr(z2c), ctypes.byref(st), _lib.stream()), f"operand route without pre_out, variant {v}")
        assert torch.equal(z2c.c8_tensor, z2a.c8_tensor)
        tested += 1
    assert tested >= 1, "no variant takes the BatchNorm on its operand for this shape"
    # the form is refused where it is not built: backward statistics, other kernel families, stride 2
    st = _lib.ConvStats(mode=1, relu=0, partials=pb.data_ptr(), partials_bytes=pb.numel() * 4, pre_scale=_lib.ptr(sc), pre_shift=_lib.ptr(sh))
    z2d = ActC8(n, cout, ho, wo, DEV)
    for v in range(F16_VARIANTS):
        if not LIB.mp_f16_conv_pre_supported(ctypes.byref(d), v):
            assert LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(z), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                               _lib.ptr(z2d), ctypes.byref(st), _lib.stream()) != 0, v
    st = _lib.ConvStats(mode=1, relu=0, partials=pb.data_ptr(), partials_bytes=pb.numel() * 4, pre_scale=_lib.ptr(sc))  # shift missing
    assert LIB.mp_f16_conv2d_fwd_stats(ctypes.byref(d), 34, _lib.ptr(z), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros), None,
                                       _lib.ptr(z2d), ctypes.byref(st), _lib.stream()) != 0


@pytest.mark.parametrize("backbone,head,size", [("hrnet_w32", "hrnet_head", (10, 128, 96)), ("resnet50", "simple_baseline_head", (12, 64, 64))])
def test_step_with_batchnorm_on_the_operand_equals_the_apply_pass_step(backbone, head, size, monkeypatch):
    """The whole amp-O2 step with MINDPOSE_BN_PRE=1 (chains leave bn1's apply to conv2's operand staging) against MINDPOSE_BN_PRE=0:
    the operand route is actually taken (batches large enough for the tuner to pick variants: below 2^26 MACs a layer keeps the
    library's heuristic and the apply pass), agrees with the apply-pass step to summation order and repeats bit for bit."""
    from mindpose_amd.models import train_ops as T
    monkeypatch.setenv("MINDPOSE_BN_PRE", "0")
    l0, g0, s0 = _step(True, monkeypatch, backbone, head, size)
    monkeypatch.setenv("MINDPOSE_BN_PRE", "1")
    taken = []
    run = T._run_bn_fwd_job
    monkeypatch.setattr(T, "_run_bn_fwd_job", lambda lib, j: (taken.append(j.get("pre") is not None), run(lib, j))[1])
    l1, g1, s1 = _step(True, monkeypatch, backbone, head, size)
    assert sum(taken) >= 4, f"{sum(taken)} of {len(taken)} BatchNorms went the operand route"
    # conv2 now runs on another tile variant than the apply-pass step tuned for it: same conv bits, but ITS partial sums are cut
    # differently, so bn2's statistics come out in another fp32 summation order - two fp16 evaluations of one graph (cf.
    # test_fused_chain_step_vs_per_cell_step); the launch-level test above pins the route itself bit for bit
    assert abs(l1 - l0) <= 1e-3 * abs(l0), (l1, l0)
    cos = float(torch.nn.functional.cosine_similarity(g1.double(), g0.double(), dim=0))
    assert cos > 0.985, cos
    for k in s0:
        assert torch.allclose(s1[k], s0[k], rtol=5e-3, atol=5e-4), k
    l2, g2, _ = _step(True, monkeypatch, backbone, head, size)  # fixed partitions and orders: bit-reproducible
    assert l2 == l1 and torch.equal(g2, g1)
