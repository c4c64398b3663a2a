"""CPU oracle: top-down heat-map decoder (TEST INFRASTRUCTURE, see oracle/__init__.py).

numpy fp32 restatement of ``TopDownHeatMapDecoder``
(/root/reference/mindpose/models/decoders/top_down_decoder.py:14-215).
PARITY UNPINNED by the reference (MindSpore ops, shape-only tests); MindSpore
semantics assumed here are listed at each function.
"""
import numpy as np

F32 = np.float32


def create_gaussian_kernel(kernel_size):
    """top_down_decoder.py:207-215 - normalised k x k Gaussian, sigma = 0.3*((k-1)/2-1)+0.8."""
    sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    xs = np.arange(-(kernel_size - 1) // 2, (kernel_size - 1) // 2 + 1, 1)
    ys = xs[:, None]
    kernel = np.exp(-(xs ** 2 + ys ** 2) / (2 * sigma ** 2))
    kernel = kernel / kernel.sum()
    return kernel.astype(F32)  # float64 -> fp32 cast, as ms.Tensor(kernel, dtype=float32)


def get_max_preds(heatmap):
    """top_down_decoder.py:96-116.

    ``ops.max(axis=2)`` -> (index, value); ties resolve to the FIRST (lowest) flat index
    [MS-knowledge]; numpy ``argmax`` has the same rule.  No ``maxval > 0`` masking.
    Returns coords [N,K,2] fp32 (x, y), maxvals [N,K,1] fp32, idx [N,K] int64.
    """
    n, k, h, w = heatmap.shape
    flat = heatmap.reshape(n, k, -1)
    idx = np.argmax(flat, axis=2)
    maxvals = np.take_along_axis(flat, idx[..., None], axis=2).astype(F32)
    preds = np.repeat(idx[..., None], 2, axis=2).astype(F32)  # :109 cast(tile(idx)) -> fp32
    preds[:, :, 0] = preds[:, :, 0] % F32(w)  # :111
    preds[:, :, 1] = np.floor(preds[:, :, 1] / F32(w))  # :112
    return preds, maxvals, idx


def shift_coordinate(coords, heatmap, idx):
    """top_down_decoder.py:118-141 - +-0.25 px shift by the sign of the central difference.

    dx is defined for 1 <= x <= W-2 (any y) and dy for 1 <= y <= H-2 (any x); elsewhere 0.
    """
    n, k, h, w = heatmap.shape
    dx = np.zeros_like(heatmap)
    dy = np.zeros_like(heatmap)
    dx[:, :, :, 1:-1] = heatmap[:, :, :, 2:] - heatmap[:, :, :, :-2]
    dy[:, :, 1:-1, :] = heatmap[:, :, 2:, :] - heatmap[:, :, :-2, :]
    sx = np.sign(dx).reshape(n, k, -1)
    sy = np.sign(dy).reshape(n, k, -1)
    off_x = np.take_along_axis(sx, idx[..., None], axis=2)[..., 0] * F32(0.25)
    off_y = np.take_along_axis(sy, idx[..., None], axis=2)[..., 0] * F32(0.25)
    out = coords.copy()
    out[..., 0] += off_x
    out[..., 1] += off_y
    return out


def _depthwise_blur_same(heatmap, kernel):
    """``ops.conv2d(group=K, pad_mode='same')`` with one shared k x k kernel: zero 'same' padding,
    cross-correlation (top_down_decoder.py:174-175).  Taps accumulated row-major in fp32."""
    k = kernel.shape[0]
    r = k // 2
    n, c, h, w = heatmap.shape
    padded = np.zeros((n, c, h + 2 * r, w + 2 * r), dtype=F32)
    padded[:, :, r:r + h, r:r + w] = heatmap
    out = np.zeros_like(heatmap)
    for i in range(k):
        for j in range(k):
            out += kernel[i, j] * padded[:, :, i:i + h, j:j + w]
    return out


def dark_udp_refine_coords(coords, heatmap, kernel_size, terms=None):
    """top_down_decoder.py:171-205 - DARK / UDP second-order Taylor refinement.

    blur -> clip[1e-3, 50] -> log -> zero-pad by 1 (value 0 in log space) -> 7 gathers ->
    coords -= inv(Hessian + 1e-7 I) @ grad.  The reference builds the flat gather index in
    fp32 (exact only while N*K*(H+2)*(W+2) < 2^24); the oracle uses integers and asserts
    that bound so both agree.
    """
    n, k, h, w = heatmap.shape
    assert n * k * (h + 2) * (w + 2) < 2 ** 24, "reference fp32 flat index would lose exactness"
    kernel = create_gaussian_kernel(kernel_size)
    hm = _depthwise_blur_same(heatmap.astype(F32), kernel)
    hm = np.clip(hm, F32(0.001), F32(50))
    hm = np.log(hm).astype(F32)
    hm = np.pad(hm, ((0, 0), (0, 0), (1, 1), (1, 1)))
    hm = hm.reshape(-1)

    index = coords[..., 0] + 1 + (coords[..., 1] + 1) * (w + 2)
    index = index.astype(np.int64)
    index = index + (w + 2) * (h + 2) * np.arange(0, n * k, 1).reshape(-1, k)
    index = index.reshape(-1, 1)
    i_ = hm[index]
    ix1 = hm[index + 1]
    iy1 = hm[index + w + 2]
    ix1y1 = hm[index + w + 3]
    ix1_y1_ = hm[index - w - 3]
    ix1_ = hm[index - 1]
    iy1_ = hm[index - 2 - w]

    dx = F32(0.5) * (ix1 - ix1_)
    dy = F32(0.5) * (iy1 - iy1_)
    derivative = np.concatenate([dx, dy], axis=1).reshape(n, k, 2, 1)

    dxx = ix1 - 2 * i_ + ix1_
    dyy = iy1 - 2 * i_ + iy1_
    dxy = F32(0.5) * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_)
    if terms is not None:
        # the intermediates, for the kernel's debug output (mp_decode_topdown_debug): 3x3 log-blur neighbourhood row-major
        # (the two corners the reference never gathers are filled in too), gradient, Hessian entries
        nb = np.stack([hm[index + oy * (w + 2) + ox] for oy in (-1, 0, 1) for ox in (-1, 0, 1)], axis=-1).reshape(n, k, 9)
        terms.update(neighbourhood=nb.astype(F32), dx=dx.reshape(n, k), dy=dy.reshape(n, k), dxx=dxx.reshape(n, k),
                     dyy=dyy.reshape(n, k), dxy=dxy.reshape(n, k))
    hessian = np.concatenate([dxx, dxy, dxy, dyy], axis=1).reshape(n, k, 2, 2).astype(F32)
    hessian = np.linalg.inv(hessian + np.eye(2, dtype=F32) * F32(1e-7)).astype(F32)
    delta = np.matmul(hessian, derivative.astype(F32))[..., 0]
    return (coords - delta).astype(F32)


def transform_preds(coords, center, scale, heatmap_shape, pixel_std=200.0, use_udp=False):
    """top_down_decoder.py:143-169 - heat-map px -> image px."""
    h, w = heatmap_shape
    scale = (scale * F32(pixel_std)).astype(F32)
    if use_udp:
        scale_x = scale[:, 0:1] / F32(w - 1.0)
        scale_y = scale[:, 1:2] / F32(h - 1.0)
    else:
        scale_x = scale[:, 0:1] / F32(w)
        scale_y = scale[:, 1:2] / F32(h)
    out = np.ones_like(coords)
    out[:, :, 0] = coords[:, :, 0] * scale_x + center[:, 0:1] - scale[:, 0:1] * F32(0.5)
    out[:, :, 1] = coords[:, :, 1] * scale_y + center[:, 1:2] - scale[:, 1:2] * F32(0.5)
    return out.astype(F32)


def decode(heatmap, center, scale, score, pixel_std=200.0, to_original=True,
           shift_coord=False, use_udp=False, dark_udp_refine=False, kernel_size=11):
    """``TopDownHeatMapDecoder.construct`` top_down_decoder.py:72-94.

    Returns (all_preds [N,K,3] = (x, y, maxval), all_boxes [N,6], idx [N,K] int64).
    """
    if dark_udp_refine and shift_coord:
        raise ValueError("`udp_refine` and `shift_coordinate` cannot be `true` in the same time.")
    heatmap = np.ascontiguousarray(heatmap, dtype=F32)
    center = np.asarray(center, dtype=F32)
    scale = np.asarray(scale, dtype=F32)
    score = np.asarray(score, dtype=F32)
    n, k, h, w = heatmap.shape
    coords, maxvals, idx = get_max_preds(heatmap)
    if shift_coord:
        coords = shift_coordinate(coords, heatmap, idx)
    elif dark_udp_refine:
        coords = dark_udp_refine_coords(coords, heatmap, kernel_size)
    if to_original:
        coords = transform_preds(coords, center, scale, (h, w), pixel_std, use_udp)
    all_preds = np.zeros((n, k, 3), dtype=F32)
    all_boxes = np.zeros((n, 6), dtype=F32)
    all_preds[:, :, 0:2] = coords[:, :, 0:2]
    all_preds[:, :, 2:3] = maxvals
    all_boxes[:, 0:2] = center[:, 0:2]
    all_boxes[:, 2:4] = scale[:, 0:2]
    all_boxes[:, 4] = np.prod(scale * F32(pixel_std), axis=1)
    all_boxes[:, 5] = score
    return all_preds, all_boxes, idx


def flip_back(flipped_heatmap, flip_index, shift_heatmap=False):
    """``_MultiRunNet._flip_back`` / ``_shift_heatmap``
    (/root/reference/mindpose/engine/inferencer/topdown_inferencer.py:180-187)."""
    out = flipped_heatmap[:, np.asarray(flip_index), ...][..., ::-1].copy()
    if shift_heatmap:
        out[..., 1:] = out[..., :-1].copy()
    return out


def flip_aggregate(heatmap, flipped_heatmap, flip_index, shift_heatmap=False):
    """``(heatmap + flip_back(flipped)) * 0.5`` topdown_inferencer.py:171-176."""
    fb = flip_back(np.asarray(flipped_heatmap, dtype=F32), flip_index, shift_heatmap)
    return ((np.asarray(heatmap, dtype=F32) + fb) * F32(0.5)).astype(F32)


def flip_index_from_pairs(flip_pairs):
    """topdown_inferencer.py:78-80."""
    fi = np.array(flip_pairs)[:, ::-1].flatten()
    return np.insert(fi, 0, 0)
