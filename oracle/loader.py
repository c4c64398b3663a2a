"""CPU oracle: top-down loader geometry + the crop (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, in numpy,
  * ``TopDownBoxToCenterScale._xywh2cs``  /root/reference/mindpose/data/transform/topdown_transform.py:131-154 (eval mode)
  * ``get_affine_transform`` / ``_get_3rd_point`` / ``rotate_point``  transform/utils.py:44-147
  * ``get_warp_matrix`` (UDP)  transform/utils.py:150-181
  * ``cv2.warpAffine(..., flags=cv2.INTER_LINEAR)`` as TopDownAffine calls it (:211-216, :249-254) followed by
    ``vision.Normalize`` + ``vision.HWC2CHW`` (data_factory.py:129-133).

Pinning: ``xywh2cs`` and ``get_warp_matrix`` are bit-exact against golden vectors produced by the reference's own numpy
code (tests/golden/geometry.npz).  ``get_affine_transform`` needs ``cv2.getAffineTransform`` and ``warp_affine`` restates
OpenCV's fixed-point INTER_LINEAR path from knowledge of imgproc/imgwarp.cpp - cv2 is not installed: PARITY UNPINNED for
those two (checked against analytic known answers and a float bilinear interpolation instead).
"""
import numpy as np


def xywh2cs(x, y, w, h, image_size, pixel_std=200.0, scale_padding=1.25):
    """:131-154 with is_train=False.  image_size = [w, h]."""
    aspect_ratio = image_size[0] / image_size[1]
    center = np.array([x + w * 0.5, y + h * 0.5], dtype=np.float32)
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w / pixel_std, h / pixel_std], dtype=np.float32)
    scale = scale * scale_padding
    return center, scale


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def get_affine_transform_cv(src3, dst3):
    """cv2.getAffineTransform: the 2x3 map sending three source points to three destination points (6x6 system solved in
    float64, as OpenCV does)."""
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[i, 0:2], a[i, 2] = src3[i], 1.0
        a[i + 3, 3:5], a[i + 3, 5] = src3[i], 1.0
        b[i], b[i + 3] = dst3[i, 0], dst3[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=False, pixel_std=200.0):
    """utils.py:44-103."""
    scale_tmp = scale * pixel_std
    shift = np.array(shift)
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    pt = [0.0, src_w * -0.5]
    src_dir = [pt[0] * cs - pt[1] * sn, pt[0] * sn + pt[1] * cs]
    dst_dir = np.array([0.0, dst_w * -0.5])
    src = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    src[2, :] = _third_point(src[0, :], src[1, :])
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    dst[2, :] = _third_point(dst[0, :], dst[1, :])
    if inv:
        return get_affine_transform_cv(np.float32(dst), np.float32(src))
    return get_affine_transform_cv(np.float32(src), np.float32(dst))


def get_warp_matrix(theta, size_input, size_dst, size_target):
    """utils.py:150-181."""
    theta = np.deg2rad(theta)
    matrix = np.zeros((2, 3), dtype=np.float32)
    scale_x = size_dst[0] / size_target[0]
    scale_y = size_dst[1] / size_target[1]
    matrix[0, 0] = np.cos(theta) * scale_x
    matrix[0, 1] = -np.sin(theta) * scale_x
    matrix[0, 2] = scale_x * (-0.5 * size_input[0] * np.cos(theta) + 0.5 * size_input[1] * np.sin(theta) + 0.5 * size_target[0])
    matrix[1, 0] = np.sin(theta) * scale_y
    matrix[1, 1] = np.cos(theta) * scale_y
    matrix[1, 2] = scale_y * (-0.5 * size_input[0] * np.sin(theta) - 0.5 * size_input[1] * np.cos(theta) + 0.5 * size_target[1])
    return matrix


def warp_affine(image, trans, out_w, out_h):
    """cv2.warpAffine(image, trans, (out_w, out_h), flags=cv2.INTER_LINEAR), BORDER_CONSTANT 0, uint8 HWC in / out.
    OpenCV's fixed-point scheme: AB_BITS=10, INTER_BITS=5, 15-bit weights (exact products for bilinear)."""
    img = np.asarray(image)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, c = img.shape
    m = np.array(trans, dtype=np.float64).reshape(2, 3).copy()
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = a11, m[0, 1] * -d, m[1, 0] * -d, a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    xs = np.arange(out_w, dtype=np.float64)
    ys = np.arange(out_h, dtype=np.float64)
    adelta = np.rint(m[0, 0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(m[1, 0] * xs * 1024.0).astype(np.int64)
    x0 = np.rint((m[0, 1] * ys + m[0, 2]) * 1024.0).astype(np.int64) + 16
    y0 = np.rint((m[1, 1] * ys + m[1, 2]) * 1024.0).astype(np.int64) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    acc = np.zeros((out_h, out_w, c), np.int64)
    for dy, dx, wgt in ((0, 0, (32 - fx) * (32 - fy) * 32), (0, 1, fx * (32 - fy) * 32), (1, 0, (32 - fx) * fy * 32),
                        (1, 1, fx * fy * 32)):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        px = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        acc += np.where(ok[..., None], px, 0) * wgt[..., None]
    return np.minimum((acc + (1 << 14)) >> 15, 255).astype(np.uint8)


def normalize_chw(image_u8, mean, std):
    """vision.Normalize(mean, std) + vision.HWC2CHW: float32 (x - mean) / std per channel."""
    x = image_u8.astype(np.float32)
    out = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(out.transpose(2, 0, 1), dtype=np.float32)


def crop(image_u8, center, scale, rotation, image_size, use_udp=False, pixel_std=200.0,
         mean=(0.485 * 255, 0.456 * 255, 0.406 * 255), std=(0.229 * 255, 0.224 * 255, 0.255 * 255)):
    """TopDownAffine (+UDP) -> Normalize -> HWC2CHW for one box.  image_size = [w, h]."""
    image_size = np.asarray(image_size)
    if use_udp:
        trans = get_warp_matrix(rotation, center * 2.0, image_size - 1.0, scale * pixel_std)
    else:
        trans = get_affine_transform(center, scale, rotation, image_size, pixel_std=pixel_std)
    warped = warp_affine(image_u8, trans, int(image_size[0]), int(image_size[1]))
    return normalize_chw(warped, mean, std), trans
