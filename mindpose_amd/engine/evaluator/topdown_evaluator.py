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
        kpts = defaultdict(list)
        for record in inference_result:
            image_id = self.name2id[os.path.basename(record["image_path"])]
            kpts[image_id].append({"keypoints": record["pred"], "center": record["box"][0:2], "scale": record["box"][2:4],
                                   "area": record["box"][4], "score": record["box"][5], "image_id": image_id,
                                   "bbox_id": record["bbox_id"]})
        kpts = self._sort_and_unique_bboxes(kpts)
        vis_thr, oks_thr = self._evaluation_cfg["vis_thr"], self._evaluation_cfg["oks_thr"]
        valid_kpts = []
        for image_id in kpts.keys():
            img_kpts = kpts[image_id]
            for n_p in img_kpts:  # rescoring: box score x mean score of the key points above vis_thr (:101-113)
                box_score = n_p["score"]
                kpt_score, valid_num = 0, 0
                for n_jt in range(0, self.num_joints):
                    t_s = n_p["keypoints"][n_jt][2]
                    if t_s > vis_thr:
                        kpt_score = kpt_score + t_s
                        valid_num = valid_num + 1
                if valid_num != 0:
                    kpt_score = kpt_score / valid_num
                n_p["score"] = kpt_score * box_score
            if self._evaluation_cfg["use_nms"]:
                nms = soft_oks_nms if self._evaluation_cfg["soft_nms"] else oks_nms
                keep = nms(img_kpts, oks_thr, sigmas=np.asarray(self._evaluation_cfg["sigmas"]))
                valid_kpts.append([img_kpts[_keep] for _keep in keep])
            else:
                valid_kpts.append(img_kpts)
        results = self._write_coco_keypoint_results(valid_kpts, self.result_path)
        name_value = dict(self._do_python_keypoint_eval(results))
        for name in self.metrics:
            if name not in name_value:
                raise ValueError(f"`{name}` is not in the returned result `{name_value.keys()}`")
        if self.remove_result_file:
            os.remove(self.result_path)
        return name_value

    def _sort_and_unique_bboxes(self, kpts, key: str = "bbox_id"):
        for img_id, persons in kpts.items():
            num = len(persons)
            kpts[img_id] = sorted(kpts[img_id], key=lambda x: x[key])
            for i in range(num - 1, 0, -1):
                if kpts[img_id][i][key] == kpts[img_id][i - 1][key]:
                    del kpts[img_id][i]
        return kpts

    def _write_coco_keypoint_results(self, keypoints, res_file: str) -> List[Dict[str, Any]]:
        cls = self.classes[1]
        cat_id = self._class_to_coco_ind[cls]
        cat_results = []
        for img_kpts in keypoints:
            if not img_kpts:
                continue
            key_points = np.array([k["keypoints"] for k in img_kpts]).reshape(-1, self.num_joints * 3)
            cat_results.extend({"image_id": k["image_id"], "category_id": cat_id, "keypoints": kp.tolist(),
                                "score": float(k["score"]), "center": np.asarray(k.get("center", -1)).tolist(),
                                "scale": np.asarray(k.get("scale", -1)).tolist()} for k, kp in zip(img_kpts, key_points))
        with open(res_file, "w") as f:
            json.dump(cat_results, f, sort_keys=True, indent=4)
        return cat_results

    def _do_python_keypoint_eval(self, results):
        cat_id = self._class_to_coco_ind[self.classes[1]]
        gts = [a for a in self.coco["annotations"] if a.get("category_id", cat_id) == cat_id]
        stats = coco_keypoint_eval(gts, results, image_ids=sorted(self.id2name))
        return list(zip(STATS_NAMES, stats))
