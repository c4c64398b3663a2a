"""Top-down record store.  Contract kept from mindpose/data/dataset/topdown.py:8-137: constructor arguments,
the two hooks a format class fills in (``load_dataset_cfg`` / ``load_dataset``), the column tuple ``__getitem__`` hands to the
pipeline, and the box clipping rule ``_sanitize_bbox`` (pinned bit-exact by tests/golden/dataset.npz).

Own structure: records live COLUMN-wise in a `RecordTable` (one numpy array per field, image paths interned once per image),
because the batched GPU pipeline (data_factory.py) consumes whole index ranges - boxes [B, 4], key points [B, K, 3] - rather than
one Python dict per sample; a dict view (`record`) exists for the per-sample transform list.  The column tuple is assembled from
the declared column list of ``column_names.py``, so the order has one source of truth."""
import copy
import logging
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from ..column_names import COLUMN_MAP

_PLACEHOLDER = np.float32(0)  # columns that later transforms fill (center, scale, target, target_weight)


def clip_boxes_to_image(xywh: np.ndarray, img_width: float, img_height: float) -> Tuple[np.ndarray, np.ndarray]:
    """Vectorised box rule of topdown.py:123-137 on an [M, 4] float64 array of (x, y, w, h).

    The near corner is clamped to >= 0; the far corner is near + (extent - 1, never negative), clamped to the last pixel; the
    result is (near, far - near).  Returns the clipped boxes and the mask of boxes with positive extent on both axes.  Same
    IEEE operations, in the same order, as the scalar expression - hence bit-exact for float64 inputs."""
    xywh = np.asarray(xywh, dtype=np.float64).reshape(-1, 4)
    near = np.maximum(0.0, xywh[:, :2])
    far = np.minimum(np.array([img_width - 1, img_height - 1], dtype=np.float64), near + np.maximum(0.0, xywh[:, 2:] - 1))
    return np.concatenate([near, far - near], axis=1), np.all(far > near, axis=1)


class RecordTable:
    """Columnar list of top-down records.  Fields: ``image`` (index into ``image_files``), ``boxes`` [R, 4] (x, y, w, h),
    ``bbox_ids``, ``bbox_scores``, ``rotation``, and - ground truth only - ``keypoints`` [R, K, 3]."""

    def __init__(self, image_files: Sequence[str], image: Iterable[int], boxes, bbox_ids, bbox_scores, keypoints=None) -> None:
        self.image_files = list(image_files)
        self.image = np.asarray(list(image), dtype=np.int64)
        n = len(self.image)
        self.boxes = np.asarray(boxes, dtype=np.float64).reshape(n, 4)
        self.bbox_ids = np.asarray(bbox_ids, dtype=np.int64).reshape(n)
        self.bbox_scores = np.asarray(bbox_scores, dtype=np.float64).reshape(n)
        self.rotation = np.zeros(n, dtype=np.float64)
        if keypoints is not None:
            keypoints = np.asarray(keypoints).reshape(n, -1, 3) if n else np.zeros((0, 0, 3))
        self.keypoints = keypoints

    def __len__(self) -> int:
        return len(self.image)

    def image_file(self, idx: int) -> str:
        return self.image_files[self.image[idx]]

    def row(self, idx: int) -> Dict[str, Any]:
        """Dict view of one record (keys as coco_topdown.py:64-74); numbers come back as Python / numpy scalars."""
        rec = dict(image_file=self.image_file(idx), rotation=self.rotation[idx].item(), boxes=self.boxes[idx].tolist(),
                   bbox_ids=int(self.bbox_ids[idx]), bbox_scores=self.bbox_scores[idx].item())
        if self.keypoints is not None:
            rec["keypoints"] = self.keypoints[idx]
        return rec


class ImagePath(str):
    """The ``image`` column as a path whose bytes have not been read yet (`TopDownDataset.lazy_image`)."""

    def read(self) -> np.ndarray:
        return np.fromfile(str(self), dtype=np.uint8)


class TopDownDataset:
    """Args (topdown.py:41-67): image_root, annotation_file, is_train, num_joints, use_gt_bbox_for_val, detection_file, config.

    Items - training: (image, center, scale, boxes, keypoints, rotation, target, target_weight);
    evaluation: (image, center, scale, rotation, image_file, boxes, bbox_ids, bbox_scores) - `column_names.COLUMN_MAP`.
    A format class implements ``load_dataset_cfg() -> dict`` and ``load_dataset() -> RecordTable``."""

    def __init__(self, image_root: str, annotation_file: Optional[str] = None, is_train: bool = False, num_joints: int = 17,
                 use_gt_bbox_for_val: bool = False, detection_file: Optional[str] = None,
                 config: Optional[Dict[str, Any]] = None) -> None:
        needs_detections = not is_train and not use_gt_bbox_for_val
        if annotation_file is None and needs_detections:
            raise ValueError("For evaluation, `detection_file` must be provided when `use_gt_bbox_for_val` is `False`")
        self.image_root, self.annotation_file, self.detection_file = image_root, annotation_file, detection_file
        self.is_train, self.use_gt_bbox_for_val, self.num_joints = is_train, use_gt_bbox_for_val, num_joints
        self.config = dict(config) if config else {}
        self._dataset_cfg = self.load_dataset_cfg()
        self._table = self.load_dataset()
        self._columns = COLUMN_MAP["topdown"]["train" if is_train else "val"]
        logging.info(f"Number of records in dataset: {len(self._table)}")

    # -- hooks of a dataset format ---------------------------------------------------------------------------------------------
    def load_dataset_cfg(self) -> Dict[str, Any]:
        raise NotImplementedError("Child class must implement this method.")

    def load_dataset(self) -> RecordTable:
        raise NotImplementedError("Child class must implement this method.")

    # -- access ----------------------------------------------------------------------------------------------------------------
    @property
    def table(self) -> RecordTable:
        return self._table

    def __len__(self) -> int:
        return len(self._table)

    def record(self, idx: int) -> Dict[str, Any]:
        """The raw record the column tuple is built from; the batched GPU pipeline reads records directly."""
        return self._table.row(idx)

    def _column_sources(self, idx: int) -> Dict[str, Callable[[], Any]]:
        t = self._table
        return {
            # ENCODED bytes: decoding is a pipeline step.  ``lazy_image`` (set by a pipeline whose codec runs in worker processes): only
            # the path travels, the worker reads the file itself
            "image": (lambda: ImagePath(t.image_file(idx))) if getattr(self, "lazy_image", False) else (lambda: np.fromfile(t.image_file(idx), dtype=np.uint8)),
            "image_file": lambda: t.image_file(idx),
            "boxes": lambda: t.boxes[idx].astype(np.float32),
            "keypoints": lambda: t.keypoints[idx].astype(np.float32),
            "rotation": lambda: np.float32(t.rotation[idx]),
            "bbox_ids": lambda: np.int32(t.bbox_ids[idx]),
            "bbox_scores": lambda: np.float32(t.bbox_scores[idx]),
        }

    def __getitem__(self, idx: int) -> tuple:
        produce = self._column_sources(idx)
        return tuple(produce[name]() if name in produce else _PLACEHOLDER for name in self._columns)

    # -- box rule --------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _sanitize_bbox(annos: List[Dict], img_width: int, img_height: int) -> List[Dict[str, Any]]:
        """Annotations whose box survives `clip_boxes_to_image` and whose ``area`` (when present) is positive, as deep copies
        carrying the clipped box; annotations without a box are dropped (topdown.py:123-137)."""
        boxed = [a for a in annos if "bbox" in a]
        if not boxed:
            return []
        clipped, positive = clip_boxes_to_image(np.array([a["bbox"] for a in boxed], dtype=np.float64), img_width, img_height)
        kept = []
        for anno, box, ok in zip(boxed, clipped, positive):
            if ok and anno.get("area", 1) > 0:
                clone = copy.deepcopy(anno)
                clone["bbox"] = box.tolist()
                kept.append(clone)
        return kept
