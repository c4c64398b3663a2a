"""GPU parity: decoder / flip aggregation / target generation / loss through the C ABI vs the CPU
oracle and the committed golden fixtures.  Bit-exact for indices, boxes, plain/shift coordinates,
aggregated heat-maps and plain targets; stated tolerances elsewhere."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mindpose_amd as mp  # noqa: E402
from oracle import decoder as od  # noqa: E402
from oracle import loss as ol  # noqa: E402
from oracle import target as ot  # noqa: E402
from tests.golden import recipes  # noqa: E402
from tests.golden.gen_golden import DECODER_CASES, decoder_inputs  # noqa: E402
from tests.golden_io import TARGET_CASES, load_npz, load_target_case  # noqa: E402

DEV = "cuda:0"


def _dec(kw):
    kw = dict(kw)
    if "shift_coord" in kw:
        kw["shift_coordinate"] = kw.pop("shift_coord")
    return mp.create_decoder("topdown_heatmap", **kw).to(DEV)


def _cuda(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in arrs]


@pytest.mark.parametrize("case", DECODER_CASES, ids=[c[0] for c in DECODER_CASES])
def test_decoder_vs_golden_and_oracle(case):
    name, kind, shape, seed, kw = case
    g = load_npz("decoder.npz")
    hm, center, scale, score = decoder_inputs(kind, shape, seed)
    dec = _dec(kw)
    preds, boxes = dec(*_cuda(hm, center, scale, score))
    preds, boxes = preds.cpu().numpy(), boxes.cpu().numpy()
    idx = dec.last_argmax.cpu().numpy()
    assert preds.shape == shape[:2] + (3,) and boxes.shape == (shape[0], 6)
    # bit-exact: arg-max indices, max values, boxes
    assert np.array_equal(idx, g[name + "/idx"])
    assert np.array_equal(preds[..., 2], g[name + "/preds"][..., 2])
    assert np.array_equal(boxes, g[name + "/boxes"])
    if not kw.get("dark_udp_refine"):
        # plain / +-0.25 shift coordinates: same fp32 expression, no FMA contraction -> bit-exact
        assert np.array_equal(preds, g[name + "/preds"])
    else:
        # DARK: blur summation order + log differ by ulps and inv(Hessian) amplifies them; the refined
        # offset is compared in heat-map pixels on well-conditioned joints (Gaussian blobs)
        ref = g[name + "/preds"]
        if kind == "uniform":
            # U(0,1) noise maps: the Hessian of log(blur) is nearly singular on some joints, so
            # inv(Hessian) amplifies last-ulp blur differences without bound there: require 99 % of the
            # coordinates within 1e-3 relative and every one within 10 %
            close = np.isclose(preds[..., :2], ref[..., :2], rtol=1e-3, atol=0.5)
            assert close.mean() >= 0.99
            np.testing.assert_allclose(preds[..., :2], ref[..., :2], rtol=0.1, atol=2.0)
        else:
            mask = np.ones(shape[:2], dtype=bool)
            if shape[0] > 1:
                mask[1, :4] = False  # constant / one-hot / negative maps: singular Hessian
            np.testing.assert_allclose(preds[mask][:, :2], ref[mask][:, :2], rtol=1e-4, atol=2e-2)


def test_decoder_large_batch_argmax_bit_exact():
    # BASELINE-size batch (N=128): indices vs numpy argmax, property: decoded max == map max
    rng = np.random.default_rng(9)
    hm = rng.random((128, 17, 64, 48), dtype=np.float32)
    center, scale, score = recipes.boxes(128, 10)
    dec = _dec({})
    preds, _ = dec(*_cuda(hm, center, scale, score))
    idx = dec.last_argmax.cpu().numpy()
    assert np.array_equal(idx, hm.reshape(128, 17, -1).argmax(2).astype(np.int32))
    assert np.array_equal(preds.cpu().numpy()[..., 2], hm.reshape(128, 17, -1).max(2))


def test_decoder_errors():
    with pytest.raises(ValueError):
        mp.create_decoder("topdown_heatmap", shift_coordinate=True, dark_udp_refine=True)
    dec = _dec({})
    with pytest.raises(mp._lib.MindposeHipError):
        dec(torch.zeros(1, 17, 64, 48), torch.zeros(1, 2), torch.ones(1, 2), torch.ones(1))  # CPU tensors
    with pytest.raises(ValueError):
        dec(torch.zeros(2, 17, 64, 48, device=DEV), torch.zeros(1, 2, device=DEV), torch.ones(2, 2, device=DEV),
            torch.ones(2, device=DEV))


@pytest.mark.parametrize("shift", [False, True])
def test_flip_aggregate_decode_fused(shift):
    g = load_npz("flip.npz")
    tag = "shift" if shift else "noshift"
    h = recipes.blob_heatmaps(3, 17, 64, 48, 301)
    hf = recipes.blob_heatmaps(3, 17, 64, 48, 302)
    center, scale, score = recipes.boxes(3, 1301)
    dec = _dec(dict(shift_coord=True))
    th, thf, tc, ts, tsc = _cuda(h, hf, center, scale, score)
    fi = torch.tensor(recipes.FLIP_INDEX)
    (preds, boxes), avg = dec.decode_flip_aggregated(th, thf, fi, shift, tc, ts, tsc, return_heatmap=True)
    ref_avg = od.flip_aggregate(h, hf, recipes.FLIP_INDEX, shift_heatmap=shift)
    assert np.array_equal(avg.cpu().numpy(), ref_avg)  # (a+b)*0.5: bit-exact
    assert np.array_equal(dec.last_argmax.cpu().numpy(), g[tag + "/idx"])
    assert np.array_equal(preds.cpu().numpy(), g[tag + "/preds"])
    assert np.array_equal(boxes.cpu().numpy(), g[tag + "/boxes"])
    # without materialising the average
    preds2, boxes2 = dec.decode_flip_aggregated(th, thf, fi, shift, tc, ts, tsc)
    assert torch.equal(preds2, preds) and torch.equal(boxes2, boxes)
    # stand-alone aggregation entry point
    lib = mp._lib.load()
    out = torch.empty_like(th)
    fi_d = fi.to(DEV, torch.int32)
    mp._lib.check(lib.mp_flip_aggregate(th.data_ptr(), thf.data_ptr(), fi_d.data_ptr(), out.data_ptr(), 3, 17, 64, 48,
                                        int(shift), None), "mp_flip_aggregate")
    assert np.array_equal(out.cpu().numpy(), ref_avg)


@pytest.mark.parametrize("name", TARGET_CASES)
def test_target_vs_reference_golden(name):
    c = load_target_case(name)
    cfg = dict(image_size=c["image_size"], heatmap_size=c["heatmap_size"])
    if c["joint_weights"] is not None:
        cfg["joint_weights"] = c["joint_weights"].tolist()
    t = mp.TopDownGenerateTarget(is_train=True, config=cfg, sigma=c["sigma"], use_udp=c["use_udp"],
                                 use_different_joint_weights=c["joint_weights"] is not None)
    target, weight = t(torch.from_numpy(c["keypoints"]).to(DEV))
    target, weight = target.cpu().numpy(), weight.cpu().numpy()
    assert np.array_equal(weight, c["target_weight"])
    assert np.array_equal(target != 0, c["target"] != 0)  # identical support (window placement, rounding rule)
    if not c["use_udp"]:
        assert np.array_equal(target.view(np.uint32), c["target"].view(np.uint32))  # bit-exact
    else:
        # UDP evaluates exp() in fp64 on the device and rounds to fp32: <= 1 fp32 ulp from numpy's fp64 exp
        ulp = np.abs(target.view(np.int32).astype(np.int64) - c["target"].view(np.int32).astype(np.int64))
        assert ulp.max() <= 1
        assert (ulp > 0).mean() < 1e-3


def test_target_full_batch_properties():
    # BASELINE config-4 distribution, N=128: every stamped map peaks at 1.0 (plain) at the rounded joint;
    # weights in {0, vis}; oracle equality on a subset
    rng = np.random.default_rng(1000)
    n = 128
    kp = np.empty((n, 17, 3), dtype=np.float32)
    kp[..., 0] = rng.uniform(-20, 212, (n, 17))
    kp[..., 1] = rng.uniform(-20, 276, (n, 17))
    kp[..., 2] = (rng.uniform(size=(n, 17)) < 0.7)
    t = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    target, weight = t(torch.from_numpy(kp).to(DEV))
    target, weight = target.cpu().numpy(), weight.cpu().numpy()
    rt, rw = ot.generate_target(kp[:16], [192, 256], [48, 64], 2.0)
    assert np.array_equal(target[:16], rt) and np.array_equal(weight[:16], rw)
    assert set(np.unique(weight)) <= {0.0, 1.0}
    assert target.max() == 1.0 and target.min() == 0.0
    empty = target.reshape(n, 17, -1).max(2) == 0
    assert np.array_equal(empty, weight == 0)  # vis in {0,1}: a map is empty iff its weight is 0


def test_target_transform_entry_and_errors():
    t = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]))
    out = t.transform({"keypoints": np.array([[10.0, 40.0, 1.0]], dtype=np.float32)})
    assert out["target"].shape == (1, 64, 48) and out["target"][0, 10, 2] == 1.0  # round(2.5) == 2
    with pytest.raises(ValueError):
        mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]),
                                 use_different_joint_weights=True)


@pytest.mark.parametrize("name", list(recipes.LOSS_CASES))
def test_loss_fwd_bwd(name):
    g = load_npz("loss.npz")
    shape, seed = recipes.LOSS_CASES[name]
    pred, target, w = recipes.loss_inputs(shape, seed)
    tp, tt, tw = _cuda(pred, target, w)
    tp.requires_grad_(True)
    plain = mp.create_loss("joint_mse")(tp, tt)
    weighted = mp.create_loss("joint_mse", use_target_weight=True)(tp, tt, tw)
    assert plain.numel() == 1 and weighted.numel() == 1
    # tolerance: heat-maps/loss within 1e-3 (BASELINE.json); we hold 1e-6 relative
    assert abs(float(plain.detach()) - float(g[name + "/loss_plain"])) <= 1e-6 * abs(float(g[name + "/loss_plain"]))
    assert abs(float(weighted.detach()) - float(g[name + "/loss_weighted"])) <= 1e-6 * abs(float(g[name + "/loss_weighted"]))
    (weighted * 3.0).backward()
    ref = ol.joints_mse_grad(pred, target, w, use_target_weight=True, grad_out=3.0)
    np.testing.assert_allclose(tp.grad.cpu().numpy(), ref, rtol=1e-6, atol=1e-12)
    # determinism: two runs are bit-identical
    again = mp.create_loss("joint_mse", use_target_weight=True)(tp.detach(), tt, tw)
    assert float(again) == float(weighted.detach())


def test_loss_known_answer_and_full_size():
    pred = torch.ones(128, 17, 64, 48, device=DEV)
    target = torch.zeros_like(pred)
    w = torch.full((128, 17), 0.5, device=DEV)
    assert float(mp.create_loss("joint_mse", use_target_weight=True)(pred, target, w)) == 0.5
    assert float(mp.create_loss("joint_mse")(pred, target)) == 1.0
    with pytest.raises(ValueError):
        mp.create_loss("joint_mse", use_target_weight=True)(pred, target)


def test_maxpool_same():
    from oracle import nets as onets
    x = torch.randn(2, 5, 12, 10)
    out = torch.empty(2, 5, 6, 5, device=DEV)
    lib = mp._lib.load()
    xd = x.to(DEV)
    mp._lib.check(lib.mp_maxpool3x3s2_same(xd.data_ptr(), out.data_ptr(), 2, 5, 12, 10, None), "maxpool")
    assert torch.equal(out.cpu(), onets.maxpool3x3s2_same(x))


@pytest.mark.parametrize("case", [c for c in DECODER_CASES if c[4].get("dark_udp_refine")], ids=lambda c: c[0])
def test_dark_refine_intermediates_vs_oracle(case):
    """top_down_decoder.py:171-205 term by term, on EVERY map (noise maps and singular joints included): the 3x3 neighbourhood
    of log(clip(blur(h))), the gradient and the Hessian entries of the kernel (mp_decode_topdown_debug: same kernel, extra output)
    against the oracle at 1e-5; the final 2x2 solve is checked as a solve (residual of (H + 1e-7 I) delta = g on the kernel's own
    terms), which holds however ill-conditioned H is - so a wrong tap, pad rule or clip bound cannot hide behind conditioning."""
    import ctypes
    name, kind, shape, seed, kw = case
    hm, center, scale, score = decoder_inputs(kind, shape, seed)
    n, k, h, w = shape
    lib = mp._lib.load()
    dec = _dec(kw)
    thm, tc, ts, tsc = _cuda(hm, center, scale, score)
    preds = torch.empty(n, k, 3, device=DEV)
    boxes = torch.empty(n, 6, device=DEV)
    idx = torch.empty(n, k, dtype=torch.int32, device=DEV)
    terms = torch.full((n, k, 16), float("nan"), device=DEV)
    blur = dec._blur_on(torch.device(DEV))
    # to_original=0: coordinates stay in heat-map pixels, so the refinement offset itself is visible
    mp._lib.check(lib.mp_decode_topdown_debug(thm.data_ptr(), tc.data_ptr(), ts.data_ptr(), tsc.data_ptr(), preds.data_ptr(),
                                              boxes.data_ptr(), idx.data_ptr(), n, k, h, w, mp._lib.MP_REFINE_DARK,
                                              int(dec.use_udp), 0, float(dec.pixel_std), blur.data_ptr(), int(dec.kernel_size),
                                              terms.data_ptr(), mp._lib.stream()), "mp_decode_topdown_debug")
    t = terms.cpu().numpy()
    coords, _, ref_idx = od.get_max_preds(hm)
    assert np.array_equal(idx.cpu().numpy(), ref_idx.astype(np.int32))
    ref = {}
    od.dark_udp_refine_coords(coords, hm, dec.kernel_size, terms=ref)
    nb = ref["neighbourhood"]
    # the reference gathers 7 of the 9 positions (not the two anti-diagonal corners); all 9 are compared
    np.testing.assert_allclose(t[..., :9], nb, rtol=1e-5, atol=2e-6)
    mag = np.abs(nb).max(axis=-1) + 1e-30  # derivatives are differences of these values: tolerance relative to their size
    for j, key in ((9, "dx"), (10, "dy"), (11, "dxx"), (12, "dyy"), (13, "dxy")):
        err = np.abs(t[..., j] - ref[key]) / mag
        assert err.max() < 1e-5, (key, float(err.max()))
    # the solve, on the kernel's own terms, in fp64: (H + 1e-7 I) delta = g  with  delta = arg-max - refined
    dx, dy, dxx, dyy, dxy = (t[..., j].astype(np.float64) for j in (9, 10, 11, 12, 13))
    delta = coords.astype(np.float64) - preds.cpu().numpy()[..., :2].astype(np.float64)
    ha, hd = dxx + np.float64(np.float32(1e-7)), dyy + np.float64(np.float32(1e-7))
    rx = ha * delta[..., 0] + dxy * delta[..., 1] - dx
    ry = dxy * delta[..., 0] + hd * delta[..., 1] - dy
    # fp32 adjugate solve + the subtraction from the arg-max coordinate (ulp of a coordinate up to 64 ~ 4e-6 per unit of H)
    bound = (1e-4 * (np.abs(ha) * np.abs(delta[..., 0]) + np.abs(dxy) * (np.abs(delta[..., 0]) + np.abs(delta[..., 1]))
                     + np.abs(hd) * np.abs(delta[..., 1]) + np.abs(dx) + np.abs(dy))
             + 8e-6 * (np.abs(ha) + np.abs(hd) + 2 * np.abs(dxy)) * 64 + 1e-12)
    finite = np.isfinite(delta).all(axis=-1)
    assert (np.abs(rx)[finite] <= bound[finite]).all() and (np.abs(ry)[finite] <= bound[finite]).all()
    # where the determinant vanishes exactly the reference's inv() is singular too; everything else is finite
    assert (finite | (t[..., 14] == 0)).all()
    # and the product entry point returns exactly the same coordinates as the debug entry
    preds2 = torch.empty_like(preds)
    mp._lib.check(lib.mp_decode_topdown(thm.data_ptr(), tc.data_ptr(), ts.data_ptr(), tsc.data_ptr(), preds2.data_ptr(),
                                        boxes.data_ptr(), idx.data_ptr(), n, k, h, w, mp._lib.MP_REFINE_DARK, int(dec.use_udp), 0,
                                        float(dec.pixel_std), blur.data_ptr(), int(dec.kernel_size), mp._lib.stream()), "decode")
    assert torch.equal(preds2.nan_to_num(123.0), preds.nan_to_num(123.0))


@pytest.mark.parametrize("shape", [(3, 3, 256, 192), (2, 3, 17, 13), (1, 5, 8, 4)])
def test_flip_width_bit_exact(shape):
    """mp_flip_width == ops.ReverseV2 on the width axis (topdown_inferencer.py:168-170), 16-byte and scalar paths."""
    lib = mp._lib.load()
    x = torch.randn(*shape, device=DEV)
    out = torch.full_like(x, float("nan"))
    n, c, h, w = shape
    mp._lib.check(lib.mp_flip_width(x.data_ptr(), out.data_ptr(), n, c, h, w, mp._lib.stream()), "mp_flip_width")
    assert torch.equal(out, torch.flip(x, dims=[3]))
    assert lib.mp_flip_width(x.data_ptr(), x.data_ptr(), n, c, h, w, mp._lib.stream()) == -3  # in place: unsupported
