"""Top-down evaluator: rescoring + OKS NMS + COCO key-point AP, host Python like the reference.

Mirror of ``TopDownEvaluator`` (mindpose/engine/evaluator/topdown_evaluator.py:12-148) and its base
(engine/evaluator/evaluator.py:15-180): same constructor, config keys, record format, result file, and the same ten
``(name, value)`` statistics - with the COCO annotation file read by ``json`` and the OKS-AP computed by
``coco_eval.coco_keypoint_eval`` instead of pycocotools.
"""
import json
import os
from collections import defaultdict
from typing import Any, Dict, List, Optional, Set, Union

import numpy as np

from ...register import register
from ...utils.nms import oks_nms, soft_oks_nms
from .coco_eval import STATS_NAMES, coco_keypoint_eval


@register("evaluator", extra_name="topdown")
class TopDownEvaluator:
    SUPPORT_METRICS = {"AP"}

    def __init__(self, annotation_file: str, metric: Union[str, List[str]] = "AP", num_joints: int = 17,
                 config: Optional[Dict[str, Any]] = None, remove_result_file: bool = True,
                 result_path: str = "./result_keypoints.json") -> None:
        self.annotation_file = annotation_file
        self.num_joints = num_joints
        self.config = config if config else dict()
        self._metrics = set(metric) if isinstance(metric, list) else set([metric])
        for single_metric in self._metrics:
            if single_metric not in self.SUPPORT_METRICS:
                raise KeyError(f"metric {single_metric} is not supported")
        self._evaluation_cfg = self.load_evaluation_cfg()
        with open(annotation_file) as f:
            self.coco = json.load(f)
        self.id2name = {im["id"]: im["file_name"] for im in self.coco["images"]}
        self.name2id = {im["file_name"]: im["id"] for im in self.coco["images"]}
        cats = sorted(self.coco["categories"], key=lambda c: c["id"])
        self.classes = ["__background__"] + [c["name"] for c in cats]
        self._class_to_coco_ind = {c["name"]: c["id"] for c in cats}
        self.remove_result_file = remove_result_file
        self.result_path = result_path

    @property
    def metrics(self) -> Set[str]:
        return self._metrics

    def load_evaluation_cfg(self) -> Dict[str, Any]:
        cfg = dict()
        cfg["vis_thr"] = self.config["vis_thr"]
        cfg["oks_thr"] = self.config["oks_thr"]
        cfg["use_nms"] = self.config["use_nms"]
        cfg["soft_nms"] = self.config["soft_nms"]
        cfg["sigmas"] = np.array(self.config["sigmas"])
        return cfg

    def __call__(self, inference_result) -> Dict[str, Any]:
        return self.eval(inference_result)

    def eval(self, inference_result: List[Dict[str, Any]]) -> Dict[str, Any]:
        """records -> per-image person lists -> rescoring -> (soft) OKS NMS -> result file -> the ten COCO statistics."""
        people = self._group_by_image(inference_result)
        survivors = [self._suppress(self._rescore(persons)) for persons in people.values()]
        results = self._dump_results(survivors, self.result_path)
        stats = dict(self._do_python_keypoint_eval(results))
        missing = [m for m in self.metrics if m not in stats]
        if missing:
            raise ValueError(f"`{missing[0]}` is not in the returned result `{stats.keys()}`")
        if self.remove_result_file:
            os.remove(self.result_path)
        return stats

    def _group_by_image(self, records) -> Dict[int, List[Dict[str, Any]]]:
        """One list per image id, sorted by bbox_id with repeated boxes removed (:77-94, :134-148)."""
        people: Dict[int, List[Dict[str, Any]]] = defaultdict(list)
        for rec in records:
            image_id = self.name2id[os.path.basename(rec["image_path"])]
            box = rec["box"]
            people[image_id].append(dict(keypoints=rec["pred"], center=box[0:2], scale=box[2:4], area=box[4], score=box[5],
                                         image_id=image_id, bbox_id=rec["bbox_id"]))
        for image_id, persons in people.items():
            persons = sorted(persons, key=lambda person: person["bbox_id"])
            people[image_id] = [q for k, q in enumerate(persons) if k == 0 or q["bbox_id"] != persons[k - 1]["bbox_id"]]
        return people

    def _rescore(self, persons: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        """score = box score x mean confidence of the key points above ``vis_thr`` (:101-113); a python-float running sum in
        joint order, as the reference accumulates it."""
        vis_thr = self._evaluation_cfg["vis_thr"]
        for person in persons:
            total, count = 0, 0
            for joint in range(self.num_joints):
                conf = person["keypoints"][joint][2]
                if conf > vis_thr:
                    total, count = total + conf, count + 1
            person["score"] = (total / count if count else total) * person["score"]
        return persons

    def _suppress(self, persons: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        cfg = self._evaluation_cfg
        if not cfg["use_nms"]:
            return persons
        pick = (soft_oks_nms if cfg["soft_nms"] else oks_nms)(persons, cfg["oks_thr"], sigmas=np.asarray(cfg["sigmas"]))
        return [persons[k] for k in pick]

    def _dump_results(self, keypoints, res_file: str) -> List[Dict[str, Any]]:
        """COCO result entries of the (single) person category, written like evaluator.py:89-131 does."""
        cat_id = self._class_to_coco_ind[self.classes[1]]
        entries: List[Dict[str, Any]] = []
        for persons in keypoints:
            if not persons:
                continue
            flat = np.array([person["keypoints"] for person in persons]).reshape(-1, self.num_joints * 3)
            for person, row in zip(persons, flat):
                entries.append({"image_id": person["image_id"], "category_id": cat_id, "keypoints": row.tolist(),
                                "score": float(person["score"]), "center": np.asarray(person.get("center", -1)).tolist(),
                                "scale": np.asarray(person.get("scale", -1)).tolist()})
        with open(res_file, "w") as f:
            json.dump(entries, f, sort_keys=True, indent=4)
        return entries

    def _do_python_keypoint_eval(self, results):
        cat_id = self._class_to_coco_ind[self.classes[1]]
        gts = [a for a in self.coco["annotations"] if a.get("category_id", cat_id) == cat_id]
        stats = coco_keypoint_eval(gts, results, image_ids=sorted(self.id2name))
        return list(zip(STATS_NAMES, stats))
