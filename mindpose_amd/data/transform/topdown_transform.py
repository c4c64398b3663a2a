"""Top-down loader transforms on the MI355X HIP path: box -> centre/scale, the affine crop, Gaussian targets.


Host mirror of the reference's ``TopDownGenerateTarget``
(mindpose/data/transform/topdown_transform.py:264-430): same constructor / config keys / error
behaviour, but BATCHED and on the device - keypoints [N,K,3] in, target [N,K,H,W] + weights [N,K] out,
written once by one block per (sample, joint) - instead of a 17-iteration Python loop per sample in
dataset worker processes.

``TopDownBoxToCenterScale`` and ``TopDownAffine`` keep the reference's names, config keys and per-sample
``transform(state)`` contract (topdown_transform.py:97-262); the geometry is the reference's own numpy arithmetic on the
host (a handful of flops per box), the pixel work - cv2.warpAffine + Normalize + HWC2CHW - is ONE fused HIP pass over
a batch of boxes (``TopDownAffine.crop_batch`` -> ``mp_warp_affine``) that writes straight into the network's NCHW fp32
input buffer, instead of three per-sample passes in dataset worker processes.
"""
import ctypes
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ... import _lib
from ...register import register


@register("transform", extra_name="topdown_generate_target")
class TopDownGenerateTarget:
    def __init__(self, is_train: bool = True, config: Optional[Dict[str, Any]] = None, sigma: float = 2.0,
                 use_different_joint_weights: bool = False, use_udp: bool = False) -> None:
        self.is_train = is_train
        self.config = config if config else dict()
        self._transform_cfg = self.load_transform_cfg()
        self.sigma = sigma
        self.use_different_joint_weights = use_different_joint_weights
        self.use_udp = use_udp
        if self.use_different_joint_weights and self._transform_cfg["joint_weights"] is None:
            raise ValueError("`joint_weights` must be provided if `use_different_joint_weights` is True.")
        self._patch_cache: Dict[str, torch.Tensor] = {}
        self._jw_cache: Dict[str, torch.Tensor] = {}

    def load_transform_cfg(self) -> Dict[str, Any]:
        """Subset of topdown_transform.py:60-94 that target generation reads."""
        cfg = dict()
        cfg["image_size"] = np.array(self.config["image_size"])
        cfg["heatmap_size"] = np.array(self.config["heatmap_size"])
        assert len(cfg["image_size"]) == 2
        assert len(cfg["heatmap_size"]) == 2
        cfg["joint_weights"] = np.array(self.config["joint_weights"]) if "joint_weights" in self.config else None
        return cfg

    # -- host-side constants computed with the reference's own numpy expressions ------------------
    def _feat_stride(self) -> Tuple[float, float]:
        image_size = self._transform_cfg["image_size"]
        w, h = self._transform_cfg["heatmap_size"]
        if self.use_udp:
            fs = (image_size - 1.0) / (np.array([w, h]) - 1.0)  # :398
        else:
            fs = image_size / np.array([w, h])  # :349
        return float(fs[0]), float(fs[1])

    def _gaussian_patch(self) -> np.ndarray:
        """:335-344 - the un-normalised patch, bit-identical to the reference on the same numpy."""
        tmp_size = self.sigma * 3
        size = 2 * tmp_size + 1
        x = np.arange(0, size, 1, np.float32)
        y = x[:, None]
        x0, y0 = size // 2, size // 2
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * self.sigma ** 2))
        return np.ascontiguousarray(g, dtype=np.float32)

    def generate(self, keypoints: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """keypoints [N,K,3] (x, y, visibility) in input-image px, CUDA -> (target [N,K,H,W], weight [N,K])."""
        lib = _lib.load()
        kp = _lib.require_cuda_f32(keypoints, "keypoints")
        if kp.dim() != 3 or kp.shape[2] != 3:
            raise ValueError("keypoints must be [N,K,3]")
        n, k, _ = kp.shape
        w, h = (int(v) for v in self._transform_cfg["heatmap_size"])
        dev = kp.device
        key = str(dev)
        patch, side = None, 0
        if not self.use_udp:
            if key not in self._patch_cache:
                self._patch_cache[key] = torch.from_numpy(self._gaussian_patch()).to(dev)
            patch = self._patch_cache[key]
            side = patch.shape[0]
        jw = None
        if self.use_different_joint_weights:
            if key not in self._jw_cache:
                jwn = np.asarray(self._transform_cfg["joint_weights"], dtype=np.float64).reshape(-1)
                if jwn.size != k:
                    raise ValueError("joint_weights must have K entries")
                self._jw_cache[key] = torch.from_numpy(jwn).to(dev)
            jw = self._jw_cache[key]
        fsx, fsy = self._feat_stride()
        target = torch.empty(n, k, h, w, device=dev, dtype=torch.float32)
        weight = torch.empty(n, k, device=dev, dtype=torch.float32)
        _lib.check(lib.mp_gaussian_target(_lib.ptr(kp), _lib.ptr(patch), side, _lib.ptr(jw), _lib.ptr(target),
                                          _lib.ptr(weight), n, k, h, w, fsx, fsy, float(self.sigma),
                                          int(self.use_udp), _lib.stream()), "mp_gaussian_target")
        return target, weight

    def transform(self, state: Dict[str, Any]) -> Dict[str, Any]:
        """Reference-shaped entry (:305-322): ``state["keypoints"]`` [K,3] (numpy or tensor) ->
        ``dict(target [K,H,W], target_weight [K])`` as device tensors."""
        kp = state["keypoints"]
        if not torch.is_tensor(kp):
            kp = torch.as_tensor(np.asarray(kp, dtype=np.float32)).cuda()
        target, weight = self.generate(kp[None])
        return dict(target=target[0], target_weight=weight[0])

    def __call__(self, keypoints: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.generate(keypoints)


def _load_full_cfg(config: Dict[str, Any]) -> Dict[str, Any]:
    """TopDownTransform.load_transform_cfg (topdown_transform.py:60-94): same keys, same derived flip_index."""
    cfg = dict()
    cfg["image_size"] = np.array(config["image_size"])
    cfg["heatmap_size"] = np.array(config["heatmap_size"])
    assert len(cfg["image_size"]) == 2
    assert len(cfg["heatmap_size"]) == 2
    flip_pairs = np.array(config["flip_pairs"])
    if len(flip_pairs.shape) == 2:
        flip_index = flip_pairs[:, ::-1].flatten()
        flip_index = np.insert(flip_index, 0, 0)
    else:
        flip_index = flip_pairs
    cfg["flip_pairs"] = flip_pairs
    cfg["flip_index"] = flip_index
    cfg["upper_body_ids"] = np.array(config["upper_body_ids"])
    cfg["pixel_std"] = float(config["pixel_std"])
    cfg["scale_padding"] = float(config["scale_padding"])
    cfg["joint_weights"] = np.array(config["joint_weights"]) if "joint_weights" in config else None
    return cfg


@register("transform", extra_name="topdown_box_to_center_scale")
class TopDownBoxToCenterScale:
    """Box (x, y, w, h) -> centre and scale (topdown_transform.py:97-154); host numpy, the reference's expressions."""

    def __init__(self, is_train: bool = True, config: Optional[Dict[str, Any]] = None) -> None:
        self.is_train = is_train
        self.config = config if config else dict()
        self._transform_cfg = _load_full_cfg(self.config)

    def transform(self, state: Dict[str, Any]) -> Dict[str, Any]:
        center, scale = self._xywh2cs(*state["boxes"])
        return dict(center=center, scale=scale)

    def _xywh2cs(self, x: float, y: float, w: float, h: float) -> Tuple[np.ndarray, np.ndarray]:
        aspect_ratio = self._transform_cfg["image_size"][0] / self._transform_cfg["image_size"][1]
        center = np.array([x + w * 0.5, y + h * 0.5], dtype=np.float32)
        if self.is_train and np.random.rand() < 0.3:  # random centre shift for the training set (:140-141)
            center += np.random.uniform(-0.2, 0.2, size=2) * [w, h]
        if w > aspect_ratio * h:
            h = w * 1.0 / aspect_ratio
        elif w < aspect_ratio * h:
            w = h * aspect_ratio
        pixel_std = self._transform_cfg["pixel_std"]
        scale_padding = self._transform_cfg["scale_padding"]
        scale = np.array([w / pixel_std, h / pixel_std], dtype=np.float32)
        scale = scale * scale_padding
        return center, scale

    def transform_batch(self, boxes: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """boxes [N,4] -> (centers [N,2], scales [N,2]), one ``_xywh2cs`` per row."""
        cs = [self._xywh2cs(*b) for b in np.asarray(boxes)]
        return np.stack([c for c, _ in cs]), np.stack([s for _, s in cs])


def _get_3rd_point(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    direction = a - b
    return b + np.array([-direction[1], direction[0]], dtype=np.float32)


def get_affine_transform(center: np.ndarray, scale: np.ndarray, rot: float, output_size, shift=(0.0, 0.0),
                         inv: bool = False, pixel_std: float = 200.0) -> np.ndarray:
    """transform/utils.py:44-103.  ``cv2.getAffineTransform`` is the exact 3-point solve in float64."""
    assert len(center) == 2
    assert len(scale) == 2
    assert len(output_size) == 2
    assert len(shift) == 2
    scale_tmp = scale * pixel_std
    shift = np.array(shift)
    src_w = scale_tmp[0]
    dst_w = output_size[0]
    dst_h = output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p0, p1 = 0.0, src_w * -0.5
    src_dir = [p0 * cs - p1 * sn, p0 * sn + p1 * cs]
    dst_dir = np.array([0.0, dst_w * -0.5])
    src = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    src[2, :] = _get_3rd_point(src[0, :], src[1, :])
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    dst[2, :] = _get_3rd_point(dst[0, :], dst[1, :])
    a, b = (dst, src) if inv else (src, dst)
    sys6 = np.zeros((6, 6), np.float64)
    rhs = np.zeros(6, np.float64)
    for i in range(3):
        sys6[i, 0:2], sys6[i, 2] = a[i], 1.0
        sys6[i + 3, 3:5], sys6[i + 3, 5] = a[i], 1.0
        rhs[i], rhs[i + 3] = b[i, 0], b[i, 1]
    return np.linalg.solve(sys6, rhs).reshape(2, 3)


def get_warp_matrix(theta: float, size_input: np.ndarray, size_dst: np.ndarray, size_target: np.ndarray) -> np.ndarray:
    """transform/utils.py:150-181 (UDP)."""
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


@register("transform", extra_name="topdown_affine")
class TopDownAffine:
    """Affine crop of one instance (topdown_transform.py:157-262): matrix on the host, pixels on the GPU."""

    NORMALIZE_MEAN = (0.485, 0.456, 0.406)  # data_factory.py:78-79 (the reference's std really ends in 0.255)
    NORMALIZE_STD = (0.229, 0.224, 0.255)

    def __init__(self, is_train: bool = True, config: Optional[Dict[str, Any]] = None, use_udp: bool = False) -> None:
        self.is_train = is_train
        self.config = config if config else dict()
        self._transform_cfg = _load_full_cfg(self.config)
        self.use_udp = use_udp

    def get_matrix(self, center: np.ndarray, scale: np.ndarray, rotation: float = 0.0) -> np.ndarray:
        """The 2x3 source -> crop matrix of ``_affine`` (:198-207) or ``_udp_affine`` (:236-245)."""
        image_size = self._transform_cfg["image_size"]
        pixel_std = self._transform_cfg["pixel_std"]
        if self.use_udp:
            return get_warp_matrix(rotation, center * 2.0, image_size - 1.0, scale * pixel_std)
        return get_affine_transform(center, scale, rotation, image_size, pixel_std=pixel_std)

    def _launch(self, images: Sequence[torch.Tensor], index: Sequence[int], mats: np.ndarray, normalize: bool,
                out: Optional[torch.Tensor], mean, std, flips=None) -> torch.Tensor:
        lib = _lib.load()
        w, h = (int(v) for v in self._transform_cfg["image_size"])
        n = len(index)
        dev = images[0].device
        for im in images:
            if not im.is_cuda or im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous():
                raise _lib.MindposeHipError("source images must be contiguous CUDA uint8 tensors [H, W, 3] (no CPU fallback)")
        base = images[0].data_ptr()
        offs = torch.tensor([images[i].data_ptr() - base for i in index], dtype=torch.int64, device=dev)
        hw = torch.tensor([[images[i].shape[0], images[i].shape[1]] for i in index], dtype=torch.int32, device=dev)
        tr = torch.from_numpy(np.ascontiguousarray(mats, dtype=np.float64).reshape(n, 6)).to(dev)
        fl = None if flips is None else torch.tensor([int(bool(f)) for f in flips], dtype=torch.int32, device=dev)
        if out is None:
            out = (torch.empty(n, 3, h, w, device=dev, dtype=torch.float32) if normalize
                   else torch.empty(n, h, w, 3, device=dev, dtype=torch.uint8))
        want = (n, 3, h, w) if normalize else (n, h, w, 3)
        if tuple(out.shape) != want or not out.is_contiguous() or out.dtype != (torch.float32 if normalize else torch.uint8):
            raise ValueError(f"out must be a contiguous {want} tensor")
        m3 = (ctypes.c_float * 3)(*[float(np.float32(v * 255.0)) for v in mean])
        s3 = (ctypes.c_float * 3)(*[float(np.float32(v * 255.0)) for v in std])
        _lib.check(lib.mp_warp_affine(base, _lib.ptr(offs), _lib.ptr(hw), _lib.ptr(fl), _lib.ptr(tr), _lib.ptr(out), n, h, w, int(normalize),
                                      m3, s3, _lib.stream()), "mp_warp_affine")
        return out

    def crop_batch(self, images: Union[torch.Tensor, Sequence[torch.Tensor]], centers: np.ndarray, scales: np.ndarray,
                   rotations: Optional[np.ndarray] = None, image_index: Optional[Sequence[int]] = None,
                   out: Optional[torch.Tensor] = None, normalize_mean=NORMALIZE_MEAN, normalize_std=NORMALIZE_STD,
                   flips: Optional[Sequence[bool]] = None) -> Tuple[torch.Tensor, np.ndarray]:
        """warpAffine + Normalize + HWC2CHW for N boxes in one launch.

        images: one CUDA uint8 [H,W,3] tensor or a list of them; ``image_index[i]`` = the image box i lives in.
        ``flips[i]``: box i is cropped from the horizontally flipped image (training augmentation; ``centers`` are then
        the already mirrored centres, as TopDownHorizontalRandomFlip returns them) - applied while sampling.
        Returns (crops [N,3,h,w] fp32 CUDA - ``out`` when given, e.g. the network's input buffer -, matrices [N,2,3])."""
        if torch.is_tensor(images):
            images = [images]
        n = len(centers)
        index = list(image_index) if image_index is not None else [0] * n
        rot = np.zeros(n) if rotations is None else np.asarray(rotations)
        mats = np.stack([self.get_matrix(np.asarray(centers[i]), np.asarray(scales[i]), float(rot[i])) for i in range(n)])
        return self._launch(images, index, mats, True, out, normalize_mean, normalize_std, flips), mats

    def transform(self, state: Dict[str, Any]) -> Dict[str, Any]:
        """Per-sample contract of the reference (:183-262): required keys image, center, scale, rotation, keypoints
        (optional); returns the warped uint8 HWC image (CUDA tensor) and the transformed keypoints."""
        trans = self.get_matrix(state["center"], state["scale"], state["rotation"])
        image = state["image"]
        if not torch.is_tensor(image):
            raise _lib.MindposeHipError("TopDownAffine warps on the GPU: pass the image as a CUDA uint8 tensor")
        out = dict()
        out["image"] = self._launch([image], [0], trans[None], False, None, self.NORMALIZE_MEAN, self.NORMALIZE_STD)[0]
        if "keypoints" in state:
            out["keypoints"] = self.transform_keypoints(state["keypoints"], trans)
        return out

    def transform_keypoints(self, kp: np.ndarray, trans: np.ndarray) -> np.ndarray:
        """Key points through the crop matrix, in place (:204-207 visible joints only; UDP :242-245 all joints)."""
        if self.use_udp:
            kp[:, 0:2] = np.dot(np.concatenate((kp[:, 0:2], np.ones((kp.shape[0], 1), dtype=np.float32)), axis=-1), trans.T)
        else:
            for i in range(kp.shape[0]):
                if kp[i, 2] > 0.0:
                    kp[i, 0:2] = np.array(trans) @ np.array([kp[i, 0], kp[i, 1], 1.0])
        return kp


def fliplr_joints(keypoints: np.ndarray, img_width: int, flip_pairs=None, flip_index: Optional[np.ndarray] = None) -> np.ndarray:
    """Mirror key points horizontally and swap left / right joints (transform/utils.py:7-41)."""
    assert img_width > 0
    assert flip_pairs is not None or flip_index is not None
    if flip_pairs is not None:
        flipped = keypoints.copy()
        for left, right in flip_pairs:
            flipped[..., left, :] = keypoints[..., right, :]
            flipped[..., right, :] = keypoints[..., left, :]
    else:
        flipped = keypoints[..., flip_index, :]
    flipped[..., 0] = img_width - 1 - flipped[..., 0]
    return flipped


@register("transform", extra_name="topdown_horizontal_random_flip")
class TopDownHorizontalRandomFlip:
    """Random horizontal flip of image, key points and box centre (topdown_transform.py:433-488).  The image may be a
    numpy HWC array or a CUDA tensor (flipped on the device); the random draw and the label arithmetic are the reference's."""

    def __init__(self, is_train: bool = True, config: Optional[Dict[str, Any]] = None, flip_prob: float = 0.5) -> None:
        self.is_train = is_train
        self.config = config if config else dict()
        self._transform_cfg = _load_full_cfg(self.config)
        self.flip_prob = flip_prob

    def transform(self, state: Dict[str, Any]) -> Dict[str, Any]:
        image, keypoints, center = state["image"], state["keypoints"], state["center"]
        if np.random.rand() <= self.flip_prob:
            image = torch.flip(image, dims=[1]) if torch.is_tensor(image) else image[:, ::-1]
            keypoints = fliplr_joints(keypoints, image.shape[1], flip_index=self._transform_cfg["flip_index"])
            center[0] = image.shape[1] - center[0]
        return dict(image=image, keypoints=keypoints, center=center)


@register("transform", extra_name="topdown_halfbody_transform")
class TopDownHalfBodyTransform:
    """Keep only the upper or the lower body at random (topdown_transform.py:490-606); host numpy, same random draws."""

    def __init__(self, is_train: bool = True, config: Optional[Dict[str, Any]] = None, num_joints_half_body: int = 8,
                 prob_half_body: float = 0.3, scale_padding: float = 1.5) -> None:
        self.is_train = is_train
        self.config = config if config else dict()
        self._transform_cfg = _load_full_cfg(self.config)
        self.num_joints_half_body = num_joints_half_body
        self.prob_half_body = prob_half_body
        self.scale_padding = scale_padding

    def half_body_transform(self, keypoints: np.ndarray, num_joints: int = 17):
        upper, lower = [], []
        for joint_id in range(num_joints):
            if keypoints[joint_id][2] > 0:
                (upper if joint_id in self._transform_cfg["upper_body_ids"] else lower).append(keypoints[joint_id])
        if np.random.randn() < 0.5 and len(upper) > 2:  # (the reference really draws from randn here)
            selected = upper
        elif len(lower) > 2:
            selected = lower
        else:
            selected = upper
        if len(selected) < 2:
            return None, None
        selected = np.array(selected, dtype=np.float32)
        center = selected.mean(axis=0)[:2]
        left_top = np.amin(selected, axis=0)
        right_bottom = np.amax(selected, axis=0)
        w = right_bottom[0] - left_top[0]
        h = right_bottom[1] - left_top[1]
        aspect_ratio = self._transform_cfg["image_size"][0] / self._transform_cfg["image_size"][1]
        if w > aspect_ratio * h:
            h = w * 1.0 / aspect_ratio
        elif w < aspect_ratio * h:
            w = h * aspect_ratio
        scale = np.array([w / self._transform_cfg["pixel_std"], h / self._transform_cfg["pixel_std"]], dtype=np.float32)
        return center, scale * self.scale_padding

    def transform(self, state: Dict[str, Any]) -> Dict[str, Any]:
        keypoints = state["keypoints"]
        if np.sum(keypoints[:, 2]) > self.num_joints_half_body and np.random.rand() < self.prob_half_body:
            c, s = self.half_body_transform(keypoints, num_joints=keypoints.shape[0])
            if c is not None and s is not None:
                return dict(center=c, scale=s)
        return dict()


@register("transform", extra_name="topdown_randomscale_rotation")
class TopDownRandomScaleRotation:
    """Random scale and rotation of the box (topdown_transform.py:608-667); host numpy, same random draws."""

    def __init__(self, is_train: bool = True, config: Optional[Dict[str, Any]] = None, rot_factor: float = 40.0,
                 scale_factor: float = 0.5, rot_prob: float = 0.6) -> None:
        self.is_train = is_train
        self.config = config if config else dict()
        self._transform_cfg = _load_full_cfg(self.config)
        self.rot_factor, self.scale_factor, self.rot_prob = rot_factor, scale_factor, rot_prob

    def transform(self, state: Dict[str, Any]) -> Dict[str, Any]:
        s = state["scale"]
        sf, rf = self.scale_factor, self.rot_factor
        s_factor = np.clip(np.random.randn() * sf + 1, 1 - sf, 1 + sf, dtype=np.float32)
        s = s * s_factor
        r_factor = np.clip(np.random.randn() * rf, -rf * 2, rf * 2, dtype=np.float32)
        r = r_factor if np.random.rand() <= self.rot_prob else np.float32(0.0)
        return dict(scale=s, rotation=r)
