"""COCO-format top-down records without pycocotools (reference: mindpose/data/dataset/coco_topdown.py:14-174).

The reference indexes the annotation file with ``pycocotools.coco.COCO`` and uses four of its calls (``imgs``, ``getImgIds``,
``getAnnIds(imgIds=, iscrowd=False)``, ``loadAnns``); `_CocoIndex` below builds the same index from plain ``json`` - images and
annotations in FILE ORDER, which is the order pycocotools' dictionaries iterate in - so the record lists come out in the
reference's order (bbox_ids count per image for ground truth, globally for detections)."""
import json
import os
from collections import defaultdict
from typing import Any, Dict, List

import numpy as np

from ...register import register
from .topdown import RecordTable, TopDownDataset, clip_boxes_to_image


class _CocoIndex:
    """The slice of ``pycocotools.coco.COCO`` the loader needs."""

    def __init__(self, annotation_file: str) -> None:
        with open(annotation_file, "r") as f:
            dataset = json.load(f)
        if not isinstance(dataset, dict):
            raise ValueError(f"annotation file format {type(dataset)} not supported")
        self.dataset = dataset
        self.imgs: Dict[int, Dict[str, Any]] = {}
        self.anns: Dict[int, Dict[str, Any]] = {}
        self.img_to_anns: Dict[int, List[Dict[str, Any]]] = defaultdict(list)
        for ann in dataset.get("annotations", []):
            self.img_to_anns[ann["image_id"]].append(ann)
            self.anns[ann["id"]] = ann
        for img in dataset.get("images", []):
            self.imgs[img["id"]] = img

    def get_img_ids(self) -> List[int]:
        return list(self.imgs.keys())

    def load_img(self, img_id: int) -> Dict[str, Any]:
        return self.imgs[img_id]

    def anns_of(self, img_id: int, iscrowd: bool = False) -> List[Dict[str, Any]]:
        """``loadAnns(getAnnIds(imgIds=img_id, iscrowd=iscrowd))``: the image's annotations in file order, crowd flag equal."""
        return [a for a in self.img_to_anns.get(img_id, []) if a.get("iscrowd", 0) == iscrowd]


@register("dataset", extra_name="coco_topdown")
class COCOTopDownDataset(TopDownDataset):
    """COCO person key points as top-down records.  Two sources, chosen as coco_topdown.py:55-62 does: the annotation file's
    own boxes (training, or evaluation with ``use_gt_bbox_for_val``) or a detector's boxes (``detection_file``).

    Filtering rules kept from coco_topdown.py:86-160.  Ground truth: non-crowd annotations only; the box must survive
    `clip_boxes_to_image` with a positive ``area``; at least one labelled key point (some non-zero entry, and ``num_keypoints``
    not 0 when the field exists); visibility 2 is stored as 1; ``bbox_ids`` restart at 0 for every image.  Detections: category 1
    with ``score >= config["det_bbox_thr"]``; ``bbox_ids`` number the kept boxes across the whole file.  Every record has rotation
    0; ground-truth boxes score 1."""

    def load_dataset_cfg(self) -> Dict[str, Any]:
        return {"det_bbox_thr": float(self.config["det_bbox_thr"])}  # KeyError when absent, like the reference

    def load_dataset(self) -> RecordTable:
        self.coco = _CocoIndex(self.annotation_file)
        image_ids = self.coco.get_img_ids()  # file order
        self._image_slot = {image_id: slot for slot, image_id in enumerate(image_ids)}
        paths = [os.path.join(self.image_root, self.coco.load_img(image_id)["file_name"]) for image_id in image_ids]
        from_annotations = self.is_train or self.use_gt_bbox_for_val
        return self._ground_truth_table(image_ids, paths) if from_annotations else self._detection_table(paths)

    # -- ground truth: one pass over the index, image by image -----------------------------------------------------------------
    def _ground_truth_table(self, image_ids: List[int], paths: List[str]) -> RecordTable:
        slots: List[int] = []
        ranks: List[int] = []
        boxes: List[np.ndarray] = []
        joints: List[np.ndarray] = []
        for slot, image_id in enumerate(image_ids):
            meta = self.coco.load_img(image_id)
            people = [a for a in self.coco.anns_of(image_id, iscrowd=False) if "bbox" in a]
            if not people:
                continue
            clipped, positive = clip_boxes_to_image(np.array([a["bbox"] for a in people], dtype=np.float64), meta["width"], meta["height"])
            rank = 0
            for anno, box, ok in zip(people, clipped, positive):
                if not (ok and anno.get("area", 1) > 0 and self._is_labelled(anno)):
                    continue
                kp = np.array(anno["keypoints"]).reshape(-1, 3)
                kp[:, 2] = np.minimum(1, kp[:, 2])
                slots.append(slot), ranks.append(rank), boxes.append(box), joints.append(kp)
                rank += 1
        return RecordTable(paths, slots, boxes, ranks, np.ones(len(slots)), keypoints=joints)

    @staticmethod
    def _is_labelled(anno: Dict[str, Any]) -> bool:
        return "keypoints" in anno and max(anno["keypoints"]) != 0 and anno.get("num_keypoints", 1) != 0

    # -- detector boxes: vectorised over the whole result file -----------------------------------------------------------------
    def _detection_table(self, paths: List[str]) -> RecordTable:
        with open(self.detection_file, "r") as f:
            detections = json.load(f)
        category = np.array([d["category_id"] for d in detections], dtype=np.int64)
        score = np.array([d["score"] for d in detections], dtype=np.float64)
        keep = np.flatnonzero((category == 1) & ~(score < self._dataset_cfg["det_bbox_thr"]))
        slots = [self._image_slot[detections[i]["image_id"]] for i in keep]
        boxes = [detections[i]["bbox"] for i in keep]
        return RecordTable(paths, slots, boxes, np.arange(len(keep)), score[keep])
