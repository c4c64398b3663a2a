"""oracle/decoder.py: frozen goldens, an independent torch-CPU formulation, hand known-answers
(SURVEY.md 8c).  The reference's own tests for this path are shape-only -> parity unpinned."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import decoder as od
from tests.golden import recipes
from tests.golden.gen_golden import DECODER_CASES, decoder_inputs
from tests.golden_io import load_npz


@pytest.mark.parametrize("case", DECODER_CASES, ids=[c[0] for c in DECODER_CASES])
def test_oracle_matches_frozen_golden(case):
    name, kind, shape, seed, kw = case
    g = load_npz("decoder.npz")
    hm, center, scale, score = decoder_inputs(kind, shape, seed)
    preds, boxes, idx = od.decode(hm, center, scale, score, **kw)
    assert np.array_equal(idx, g[name + "/idx"])
    np.testing.assert_allclose(preds, g[name + "/preds"], rtol=1e-5, atol=1e-4)
    np.testing.assert_array_equal(boxes, g[name + "/boxes"])


def test_reference_test_shapes():
    # tests/models/decoders/test_top_down_decoder.py:8-47 of the reference: (8,17,48,64) -> (8,17,3),(8,6)
    hm = recipes.uniform_heatmaps(8, 17, 48, 64, 1)
    center, scale, score = recipes.boxes(8, 2)
    for kw in (dict(), dict(shift_coord=True), dict(use_udp=True, dark_udp_refine=True)):
        preds, boxes, _ = od.decode(hm, center, scale, score, **kw)
        assert preds.shape == (8, 17, 3) and boxes.shape == (8, 6)
    with pytest.raises(ValueError):
        od.decode(hm, center, scale, score, shift_coord=True, dark_udp_refine=True)


def test_one_hot_decodes_exactly_and_identity_transform():
    h, w = 64, 48
    hm = np.zeros((1, 3, h, w), dtype=np.float32)
    pts = [(5, 7), (0, 0), (47, 63)]
    for j, (x, y) in enumerate(pts):
        hm[0, j, y, x] = 1.0
    # scale=(W/200,H/200), center=(W/2,H/2) makes _transform_preds the identity (non-UDP)
    center = np.array([[w / 2, h / 2]], dtype=np.float32)
    scale = np.array([[w / 200.0, h / 200.0]], dtype=np.float32)
    preds, boxes, idx = od.decode(hm, center, scale, np.array([0.9], dtype=np.float32))
    for j, (x, y) in enumerate(pts):
        assert preds[0, j].tolist() == [float(x), float(y), 1.0]
        assert idx[0, j] == y * w + x
    assert boxes[0].tolist() == [24.0, 32.0, np.float32(0.24), np.float32(0.32),
                                 float(np.float32(0.24) * np.float32(200) * (np.float32(0.32) * np.float32(200))),
                                 np.float32(0.9)]


def test_tie_takes_first_index_and_negative_maps_not_masked():
    hm = recipes.blob_heatmaps(2, 17, 64, 48, 5)
    center, scale, score = recipes.boxes(2, 6)
    preds, _, idx = od.decode(hm, center, scale, score, to_original=False)
    assert idx[1, 0] == 0
    assert idx[1, 1] == 5 * 48 + 7
    assert idx[1, 2] == 47
    assert idx[1, 3] == 3 * 48 + 3 and preds[1, 3, 2] == -0.5


def _torch_decode(hm, center, scale, score, shift=False, dark=False, use_udp=False, k=11, pixel_std=200.0):
    """Independent formulation: torch.max + conv2d(groups) + closed-form 2x2 inverse."""
    t = torch.from_numpy(hm)
    n, c, h, w = t.shape
    maxv, idx = t.reshape(n, c, -1).max(dim=2)
    x = (idx % w).float()
    y = torch.div(idx, w, rounding_mode="floor").float()
    if shift:
        tp = F.pad(t, (1, 1, 1, 1))
        ii = torch.arange(n)[:, None].expand(n, c)
        jj = torch.arange(c)[None, :].expand(n, c)
        yi, xi = y.long(), x.long()
        dx = tp[ii, jj, yi + 1, xi + 2] - tp[ii, jj, yi + 1, xi]
        dy = tp[ii, jj, yi + 2, xi + 1] - tp[ii, jj, yi, xi + 1]
        dx = torch.where((xi >= 1) & (xi <= w - 2), dx, torch.zeros_like(dx))
        dy = torch.where((yi >= 1) & (yi <= h - 2), dy, torch.zeros_like(dy))
        x = x + 0.25 * torch.sign(dx)
        y = y + 0.25 * torch.sign(dy)
    if dark:
        ker = torch.from_numpy(od.create_gaussian_kernel(k))[None, None].repeat(c, 1, 1, 1)
        b = F.conv2d(t, ker, padding=k // 2, groups=c)
        b = torch.log(b.clamp(0.001, 50))
        b = F.pad(b, (1, 1, 1, 1))
        ii = torch.arange(n)[:, None].expand(n, c)
        jj = torch.arange(c)[None, :].expand(n, c)
        yi, xi = y.long() + 1, x.long() + 1
        g = lambda oy, ox: b[ii, jj, yi + oy, xi + ox]
        dx = 0.5 * (g(0, 1) - g(0, -1))
        dy = 0.5 * (g(1, 0) - g(-1, 0))
        dxx = g(0, 1) - 2 * g(0, 0) + g(0, -1)
        dyy = g(1, 0) - 2 * g(0, 0) + g(-1, 0)
        dxy = 0.5 * (g(1, 1) - g(0, 1) - g(1, 0) + 2 * g(0, 0) - g(0, -1) - g(-1, 0) + g(-1, -1))
        a, d = dxx + 1e-7, dyy + 1e-7
        det = a * d - dxy * dxy
        x = x - (d * dx - dxy * dy) / det
        y = y - (-dxy * dx + a * dy) / det
    s = torch.from_numpy(scale) * pixel_std
    den_x, den_y = (w - 1.0, h - 1.0) if use_udp else (float(w), float(h))
    cen = torch.from_numpy(center)
    X = x * (s[:, 0:1] / den_x) + cen[:, 0:1] - s[:, 0:1] * 0.5
    Y = y * (s[:, 1:2] / den_y) + cen[:, 1:2] - s[:, 1:2] * 0.5
    return torch.stack([X, Y, maxv], dim=2).numpy(), idx.numpy()


@pytest.mark.parametrize("mode", ["plain", "shift", "dark", "dark_udp"])
def test_oracle_vs_independent_torch_formulation(mode):
    hm = recipes.blob_heatmaps(4, 17, 64, 48, 77)
    center, scale, score = recipes.boxes(4, 78)
    kw = dict(plain={}, shift=dict(shift_coord=True), dark=dict(dark_udp_refine=True),
              dark_udp=dict(dark_udp_refine=True, use_udp=True))[mode]
    preds, _, idx = od.decode(hm, center, scale, score, **kw)
    tp, tidx = _torch_decode(hm, center, scale, score, shift=mode == "shift", dark=mode.startswith("dark"),
                             use_udp=mode == "dark_udp")
    assert np.array_equal(idx, tidx)
    # constant / one-hot / negative maps (sample 1, joints 0-3) have a singular Hessian: DARK there is
    # ill-conditioned by construction, compare the well-posed joints only
    mask = np.ones(preds.shape[:2], dtype=bool)
    if mode.startswith("dark"):
        mask[1, :4] = False
    np.testing.assert_allclose(preds[mask], tp[mask], rtol=2e-4, atol=2e-3)


def test_flip_aggregate_and_index():
    assert od.flip_index_from_pairs(recipes.FLIP_PAIRS).tolist() == recipes.FLIP_INDEX
    g = load_npz("flip.npz")
    h = recipes.blob_heatmaps(3, 17, 64, 48, 301)
    hf = recipes.blob_heatmaps(3, 17, 64, 48, 302)
    for shift, tag in ((False, "noshift"), (True, "shift")):
        avg = od.flip_aggregate(h, hf, recipes.FLIP_INDEX, shift_heatmap=shift)
        np.testing.assert_array_equal(avg[:, :, ::7, ::5], g[tag + "/avg_sample"])
        # hand check of one element: joint 1 <- flipped joint 2, mirrored column (shifted by one)
        x = 10
        src_col = 48 - 1 - (x - 1 if shift else x)
        assert avg[0, 1, 3, x] == np.float32((h[0, 1, 3, x] + hf[0, 2, 3, src_col]) * np.float32(0.5))
        if shift:
            assert avg[0, 1, 3, 0] == np.float32((h[0, 1, 3, 0] + hf[0, 2, 3, 47]) * np.float32(0.5))
