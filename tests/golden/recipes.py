"""Seeded input recipes shared by ``gen_golden.py`` (writer) and the parity tests (readers).

Large random inputs are regenerated from a seed (numpy ``default_rng`` streams are stable);
the committed ``.npz`` fixtures hold the small inputs and the EXPECTED OUTPUTS.
"""
import numpy as np

FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
FLIP_INDEX = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]


def keypoint_sets(n, k, image_w, image_h, seed):
    """[n,k,3] fp32 keypoints: uniform incl. out-of-bounds, on-edge, .5-ties (half-to-even for the
    plain encoder at stride 4), invisible joints."""
    rng = np.random.default_rng(seed)
    kp = np.empty((n, k, 3), dtype=np.float32)
    kp[..., 0] = rng.uniform(-40, image_w + 40, size=(n, k))
    kp[..., 1] = rng.uniform(-40, image_h + 40, size=(n, k))
    kp[..., 2] = (rng.uniform(size=(n, k)) < 0.7).astype(np.float32)
    # sample 0: exact edges / corners
    edge = [(0, 0), (image_w - 1, image_h - 1), (image_w, image_h), (0, image_h - 1), (image_w - 1, 0),
            (-1, -1), (image_w + 23, 10), (10, image_h + 23), (-28, 10), (10, -28), (-24, -24),
            (image_w + 24, image_h + 24), (-27.9, 5), (5, -27.9), (image_w + 27.9, 5), (2, 2), (1.99, 2.01)]
    for j in range(min(k, len(edge))):
        kp[0, j] = (edge[j][0], edge[j][1], 1.0)
    # sample 1: x/stride lands on .5 -> round-half-to-even (stride 4: x = 4*m + 2)
    for j in range(k):
        kp[1, j] = (4.0 * j + 2.0, 4.0 * (2 * j) + 2.0, 1.0)
    # sample 2: visibility values around the 0.5 threshold and >1 (COCO vis=2)
    vis = [0.0, 0.4, 0.5, 0.50001, 0.6, 1.0, 2.0]
    for j in range(k):
        kp[2, j, 2] = vis[j % len(vis)]
        kp[2, j, 0] = 20.0 + 9.0 * j
        kp[2, j, 1] = 30.0 + 11.0 * j
    return kp


def uniform_heatmaps(n, k, h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.random((n, k, h, w), dtype=np.float32)


def boxes(n, seed):
    rng = np.random.default_rng(seed)
    center = rng.uniform(0, 400, size=(n, 2)).astype(np.float32)
    scale = rng.uniform(0.3, 3, size=(n, 2)).astype(np.float32)
    score = rng.uniform(0, 1, size=(n,)).astype(np.float32)
    return center, scale, score


def blob_heatmaps(n, k, h, w, seed, sigma=2.0):
    """Gaussian-blob heat-maps with sub-pixel centres + small positive noise; the first joints of
    sample 0 put the peak on every border / corner, sample 1 holds exact ties and constant maps."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    hm = np.empty((n, k, h, w), dtype=np.float32)
    cx = rng.uniform(2, w - 3, size=(n, k)).astype(np.float32)
    cy = rng.uniform(2, h - 3, size=(n, k)).astype(np.float32)
    border = [(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1), (0, h // 2), (w - 1, h // 2), (w // 2, 0),
              (w // 2, h - 1), (1, 1), (w - 2, h - 2), (0.4, 5.3), (w - 1.4, 7.7)]
    for j in range(min(k, len(border))):
        cx[0, j], cy[0, j] = border[j]
    amp = rng.uniform(0.3, 1.0, size=(n, k)).astype(np.float32)
    for b in range(n):
        for j in range(k):
            g = np.exp(-((xs - cx[b, j]) ** 2 + (ys - cy[b, j]) ** 2) / np.float32(2 * sigma * sigma))
            hm[b, j] = amp[b, j] * g
    hm += rng.uniform(0, 0.01, size=hm.shape).astype(np.float32)
    if n > 1:
        hm[1, 0] = 0.25                      # constant map -> idx 0
        hm[1, 1] = 0.0
        hm[1, 1, 5, 7] = 1.0
        hm[1, 1, 9, 3] = 1.0                 # tie -> first flat index (5*w+7)
        hm[1, 2] = 0.0
        hm[1, 2, h - 1, w - 1] = 0.5
        hm[1, 2, 0, w - 1] = 0.5             # tie across rows -> (0, w-1)
        hm[1, 3] = -1.0                      # all-negative map (no maxval>0 masking in the reference)
        hm[1, 3, 3, 3] = -0.5
    return hm


def loss_inputs(shape, seed):
    """pred, target [N,K,H,W] and weights [N,K] (one weight forced to 0)."""
    rng = np.random.default_rng(seed)
    pred = rng.random(shape, dtype=np.float32)
    target = rng.random(shape, dtype=np.float32)
    w = rng.random(shape[:2], dtype=np.float32)
    w[0, 0] = 0.0
    return pred, target, w


LOSS_CASES = {"ref": ((4, 12, 32, 32), 401), "coco": ((6, 17, 64, 48), 402)}
