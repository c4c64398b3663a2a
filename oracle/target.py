"""CPU oracle: Gaussian heat-map target generation (TEST INFRASTRUCTURE, see oracle/__init__.py).

numpy restatement of ``TopDownGenerateTarget._encoding`` / ``_udp_encoding``
(/root/reference/mindpose/data/transform/topdown_transform.py:324-375, :377-430), batched
over samples.  PINNED: compared bit-for-bit with golden vectors produced by the reference's
own implementation (tests/golden/gen_golden.py, tests/test_oracle_target.py).
"""
import math

import numpy as np


def gaussian_patch(sigma):
    """The precomputed un-normalised patch of ``_encoding`` (topdown_transform.py:335-344):
    side ``len(arange(0, 2*3*sigma+1))``, centre ``size // 2``, fp32, centre value 1."""
    tmp_size = sigma * 3
    size = 2 * tmp_size + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, None]
    x0 = y0 = size // 2
    g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
    return g  # fp32 under numpy >= 2 (python-float scalars are weak)


def _window(mu, tmp_size, extent):
    """Bounds of one axis: (ul, br, g_lo, g_hi, img_lo, img_hi) - topdown_transform.py:353-365."""
    ul = int(mu - tmp_size)
    br = int(mu + tmp_size + 1)
    g_lo, g_hi = max(0, -ul), min(br, extent) - ul
    i_lo, i_hi = max(0, ul), min(br, extent)
    return ul, br, g_lo, g_hi, i_lo, i_hi


def generate_target(keypoints, image_size, heatmap_size, sigma=2.0, use_udp=False,
                    joint_weights=None):
    """keypoints [N,K,3] (x, y, vis) in input-image px -> (target [N,K,H,W] fp32,
    target_weight [N,K] fp32).  ``image_size`` / ``heatmap_size`` are (W, H) like the yaml."""
    keypoints = np.asarray(keypoints, dtype=np.float32)
    n, k, _ = keypoints.shape
    img = np.array(image_size)
    w, h = int(heatmap_size[0]), int(heatmap_size[1])
    tmp_size = sigma * 3
    size = 2 * tmp_size + 1
    xs = np.arange(0, size, 1, np.float32)
    ys = xs[:, None]
    x0 = y0 = size // 2
    g_plain = gaussian_patch(sigma)

    if use_udp:
        feat_stride = (img - 1.0) / (np.array([w, h]) - 1.0)  # :398
    else:
        feat_stride = img / np.array([w, h])  # :349 (float64)

    target = np.zeros((n, k, h, w), dtype=np.float32)
    weight = np.zeros((n, k), dtype=np.float32)
    for b in range(n):
        for j in range(k):
            weight[b, j] = keypoints[b, j, 2]
            fx = keypoints[b, j, 0] / feat_stride[0]  # fp32 scalar / fp64 scalar -> fp64
            fy = keypoints[b, j, 1] / feat_stride[1]
            if use_udp:
                mu_x, mu_y = int(fx + 0.5), int(fy + 0.5)  # :399-400 (truncation)
            else:
                mu_x, mu_y = round(fx), round(fy)  # :350-351 (half-to-even)
            ulx, brx, gx0, gx1, ix0, ix1 = _window(mu_x, tmp_size, w)
            uly, bry, gy0, gy1, iy0, iy1 = _window(mu_y, tmp_size, h)
            if ulx >= w or uly >= h or brx < 0 or bry < 0:  # :355-357
                weight[b, j] = 0
                continue
            if weight[b, j] > 0.5:
                if use_udp:
                    x0p = x0 + fx - mu_x  # fp64 sub-pixel centre, :407-410
                    y0p = y0 + fy - mu_y
                    g = np.exp(-((xs - x0p) ** 2 + (ys - y0p) ** 2) / (2 * sigma ** 2))
                else:
                    g = g_plain
                target[b, j, iy0:iy1, ix0:ix1] = g[gy0:gy1, gx0:gx1]
    if joint_weights is not None:
        weight = np.multiply(weight, np.array(joint_weights)).astype(np.float32)
    return target, weight


def known_answers():
    """Hand known-answers of SURVEY.md 8c: centre value 1, corner exp(-72/8) for sigma=2,
    round(2.5) == 2."""
    g = gaussian_patch(2.0)
    return {
        "centre": float(g[6, 6]),
        "corner": float(g[0, 0]),
        "corner_expected": math.exp(-72.0 / 8.0),
        "round_2_5": round(2.5),
    }
