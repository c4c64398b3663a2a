"""Record loader base of the top-down path (reference: mindpose/data/dataset/topdown.py:8-137).

Host Python like the reference (disk / JSON I/O is not GPU work): a dataset is a list of records; ``__getitem__`` hands out
the reference's column tuple - the ENCODED image bytes first (decoding is a pipeline step, data_factory.py:116-118) and
placeholders for the columns later transforms fill (``data/column_names.py``)."""
import logging
from copy import deepcopy
from typing import Any, Dict, List, Optional

import numpy as np


class TopDownDataset:
    """Args (topdown.py:41-67): image_root, annotation_file, is_train, num_joints, use_gt_bbox_for_val, detection_file, config.

    Items - training: (image, center, scale, boxes, keypoints, rotation, target, target_weight);
    evaluation: (image, center, scale, rotation, image_file, boxes, bbox_ids, bbox_scores).
    Child classes implement ``load_dataset_cfg`` and ``load_dataset``."""

    def __init__(self, image_root: str, annotation_file: Optional[str] = None, is_train: bool = False, num_joints: int = 17,
                 use_gt_bbox_for_val: bool = False, detection_file: Optional[str] = None,
                 config: Optional[Dict[str, Any]] = None) -> None:
        self.image_root = image_root
        self.annotation_file = annotation_file
        self.is_train = is_train
        self.num_joints = num_joints
        self.use_gt_bbox_for_val = use_gt_bbox_for_val
        self.detection_file = detection_file
        self.config = config if config else dict()

        if self.annotation_file is None:
            if not self.is_train and not self.use_gt_bbox_for_val:
                raise ValueError("For evaluation, `detection_file` must be provided when `use_gt_bbox_for_val` is `False`")

        self._dataset_cfg = self.load_dataset_cfg()
        self._dataset = self.load_dataset()
        logging.info(f"Number of records in dataset: {len(self._dataset)}")

    def load_dataset_cfg(self) -> Dict[str, Any]:
        raise NotImplementedError("Child class must implement this method.")

    def load_dataset(self) -> List[Dict[str, Any]]:
        """Records with the keys image_file, boxes (x, y, w, h), keypoints [K, 3] (ground truth only), rotation, bbox_ids,
        bbox_scores (1 for ground truth)."""
        raise NotImplementedError("Child class must implement this method.")

    def __len__(self) -> int:
        return len(self._dataset)

    def record(self, idx: int) -> Dict[str, Any]:
        """The raw record (what ``__getitem__`` is built from); the batched GPU pipeline reads records directly."""
        return self._dataset[idx]

    def __getitem__(self, idx: int):
        record = self._dataset[idx]
        image = np.fromfile(record["image_file"], dtype=np.uint8)
        if self.is_train:
            return (
                image,
                np.float32(0),  # placeholder for center
                np.float32(0),  # placeholder for scale
                np.asarray(record["boxes"], dtype=np.float32),
                np.asarray(record["keypoints"], dtype=np.float32),
                np.float32(record["rotation"]),
                np.float32(0),  # placeholder for target
                np.float32(0),  # placeholder for target_weight
            )
        return (
            image,
            np.float32(0),  # placeholder for center
            np.float32(0),  # placeholder for scale
            np.float32(record["rotation"]),
            record["image_file"],
            np.asarray(record["boxes"], dtype=np.float32),
            np.int32(record["bbox_ids"]),
            np.float32(record["bbox_scores"]),
        )

    @staticmethod
    def _sanitize_bbox(annos: List[Dict], img_width: int, img_height: int) -> List[Dict[str, Any]]:
        """Clip every box to the image and drop the degenerate ones (topdown.py:123-137): a box keeps its top-left corner
        clamped to >= 0, its far corner clamped to the last pixel; kept when the clipped box has positive extent and the
        annotation's ``area`` (when present) is positive.  Pinned bit-exact by tests/golden/dataset.npz."""
        valid_annos = []
        for anno in annos:
            if "bbox" not in anno:
                continue
            x, y, w, h = anno["bbox"]
            x1 = max(0, x)
            y1 = max(0, y)
            x2 = min(img_width - 1, x1 + max(0, w - 1))
            y2 = min(img_height - 1, y1 + max(0, h - 1))
            if ("area" not in anno or anno["area"] > 0) and x2 > x1 and y2 > y1:
                valid_anno = deepcopy(anno)
                valid_anno["bbox"] = [x1, y1, x2 - x1, y2 - y1]
                valid_annos.append(valid_anno)
        return valid_annos
