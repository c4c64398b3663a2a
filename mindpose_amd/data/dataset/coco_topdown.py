"""COCO-format top-down records without pycocotools (reference: mindpose/data/dataset/coco_topdown.py:14-174).

The reference indexes the annotation file with ``pycocotools.coco.COCO`` and uses four of its calls (``imgs``, ``getImgIds``,
``getAnnIds(imgIds=, iscrowd=False)``, ``loadAnns``); `_CocoIndex` below builds the same index from plain ``json`` - images and
annotations in FILE ORDER, which is the order pycocotools' dictionaries iterate in - so the record lists come out in the
reference's order (bbox_ids count per image for ground truth, globally for detections)."""
import json
import os
from collections import defaultdict
from typing import Any, Dict, List, Tuple

import numpy as np

from ...register import register
from .topdown import TopDownDataset


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
    """Ground-truth records (training, or ``use_gt_bbox_for_val``) or detector boxes above ``config["det_bbox_thr"]``
    (evaluation with ``detection_file``): coco_topdown.py:44-163."""

    def load_dataset_cfg(self) -> Dict[str, Any]:
        dataset_cfg = dict()
        dataset_cfg["det_bbox_thr"] = float(self.config["det_bbox_thr"])
        return dataset_cfg

    def load_dataset(self) -> List[Dict[str, Any]]:
        self.coco = _CocoIndex(self.annotation_file)
        self.id2name, self.name2id = self._get_mapping_id_name(self.coco.imgs)
        if self.is_train or self.use_gt_bbox_for_val:
            return self._load_coco_keypoint_annotations()
        return self._load_coco_detection_result()

    def _load_coco_keypoint_annotations(self) -> List[Dict[str, Any]]:
        self.img_ids = self.coco.get_img_ids()
        gt_db = []
        for img_id in self.img_ids:
            gt_db.extend(self._load_coco_keypoint_annotations_per_img(img_id))
        return gt_db

    def _load_coco_detection_result(self) -> List[Dict[str, Any]]:
        """Person detections (``category_id == 1``) with ``score >= det_bbox_thr``; ``bbox_ids`` count the kept boxes (:86-118)."""
        with open(self.detection_file, "r") as f:
            all_boxes = json.load(f)
        bbox_id = 0
        kpt_db = []
        for det_res in all_boxes:
            if det_res["category_id"] != 1:
                continue
            image_file = os.path.join(self.image_root, self.id2name[det_res["image_id"]])
            box = det_res["bbox"]
            score = det_res["score"]
            if score < self._dataset_cfg["det_bbox_thr"]:
                continue
            kpt_db.append({"image_file": image_file, "rotation": 0, "boxes": box, "bbox_ids": bbox_id, "bbox_scores": score})
            bbox_id += 1
        return kpt_db

    def _load_coco_keypoint_annotations_per_img(self, img_id: int) -> List[Dict[str, Any]]:
        """Non-crowd annotations with a sane box and at least one labelled key point; visibility 2 -> 1 (:120-160)."""
        img_ann = self.coco.load_img(img_id)
        img_width = img_ann["width"]
        img_height = img_ann["height"]
        annos = self.coco.anns_of(img_id, iscrowd=False)  # no need to train crowd instances
        annos = self._sanitize_bbox(annos, img_width, img_height)
        bbox_id = 0
        rec = []
        for anno in annos:
            if "keypoints" not in anno:
                continue
            if max(anno["keypoints"]) == 0:
                continue
            if "num_keypoints" in anno and anno["num_keypoints"] == 0:
                continue
            keypoints = np.array(anno["keypoints"]).reshape(-1, 3)
            keypoints[:, 2] = np.minimum(1, keypoints[:, 2])
            image_file = os.path.join(self.image_root, self.id2name[img_id])
            rec.append({"image_file": image_file, "keypoints": keypoints, "rotation": 0, "boxes": anno["bbox"], "bbox_ids": bbox_id,
                        "bbox_scores": 1.0})
            bbox_id += 1
        return rec

    @staticmethod
    def _get_mapping_id_name(imgs: Dict[int, Dict[str, Any]]) -> Tuple[Dict[int, str], Dict[str, int]]:
        id2name, name2id = {}, {}
        for image_id, image in imgs.items():
            file_name = image["file_name"]
            id2name[image_id] = file_name
            name2id[file_name] = image_id
        return id2name, name2id
