"""Top-down evaluator: rescoring + OKS NMS + COCO key-point AP without pycocotools.  The AP protocol is
PARITY UNPINNED (no pycocotools here); the cases below have hand-computable answers."""
import json

import numpy as np
import pytest

import mindpose_amd as mp
from mindpose_amd.engine.evaluator import coco_keypoint_eval
from mindpose_amd.utils.nms import COCO_SIGMAS


def _person(rng, cx, cy, size):
    kp = np.zeros((17, 3))
    kp[:, 0] = cx + rng.uniform(-0.4, 0.4, 17) * size
    kp[:, 1] = cy + rng.uniform(-0.5, 0.5, 17) * size
    kp[:, 2] = 2
    return kp


def _gt(ann_id, image_id, kp, size):
    x0, y0 = kp[:, 0].min(), kp[:, 1].min()
    return dict(id=ann_id, image_id=image_id, category_id=1, keypoints=kp.reshape(-1).tolist(), num_keypoints=17,
                area=float(size * size), bbox=[float(x0), float(y0), float(kp[:, 0].max() - x0), float(kp[:, 1].max() - y0)], iscrowd=0)


def _shift_for_oks(target, area):
    """Uniform displacement d of every key point that gives exactly this OKS against itself... solved numerically."""
    lo, hi = 0.0, 500.0
    for _ in range(80):
        d = 0.5 * (lo + hi)
        e = d * d / (COCO_SIGMAS * 2) ** 2 / (area + np.spacing(1)) / 2
        lo, hi = (d, hi) if np.mean(np.exp(-e)) > target else (lo, d)
    return lo


def test_coco_keypoint_ap_hand_cases():
    rng = np.random.RandomState(0)
    size = 150.0  # area 22500 > 96^2: "large"
    kps = [_person(rng, 200 + 300 * i, 300, size) for i in range(3)]
    gts = [_gt(i + 1, i + 1, kps[i], size) for i in range(3)]
    perfect = [dict(image_id=i + 1, category_id=1, keypoints=kps[i].reshape(-1).tolist(), score=0.9 - 0.1 * i) for i in range(3)]
    s = coco_keypoint_eval(gts, perfect)
    assert s[0] == s[1] == s[2] == s[4] == 1.0 and s[5] == s[9] == 1.0 and s[3] == -1.0 and s[8] == -1.0  # no medium gts
    # every detection at OKS 0.62: matched at thresholds 0.50, 0.55, 0.60 only -> AP = AR = 3/10, AP.5 = 1, AP.75 = 0
    d = _shift_for_oks(0.62, size * size)
    shifted = []
    for i in range(3):
        kp = kps[i].copy()
        kp[:, 0] += d
        shifted.append(dict(image_id=i + 1, category_id=1, keypoints=kp.reshape(-1).tolist(), score=0.5))
    s = coco_keypoint_eval(gts, shifted)
    assert abs(s[0] - 0.3) < 1e-9 and s[1] == 1.0 and s[2] == 0.0 and abs(s[5] - 0.3) < 1e-9
    # one image, one gt: a higher-scored false positive ahead of the true positive -> precision 1/2 at every recall level
    far = kps[0].copy()
    far[:, 0] += 2000
    two = [dict(image_id=1, category_id=1, keypoints=far.reshape(-1).tolist(), score=0.9),
           dict(image_id=1, category_id=1, keypoints=kps[0].reshape(-1).tolist(), score=0.8)]
    s = coco_keypoint_eval(gts[:1], two)
    assert abs(s[0] - 0.5) < 1e-9 and s[5] == 1.0
    # a missed person: recall 2/3, precision 1 up to recall 2/3 -> AP = 67 of the 101 recall thresholds
    s = coco_keypoint_eval(gts, perfect[:2])
    assert abs(s[5] - 2 / 3) < 1e-12 and abs(s[0] - 67 / 101) < 1e-12
    # crowd / unlabeled ground truth is ignored, detections matched to it are neither TP nor FP
    crowd = dict(gts[2], iscrowd=1)
    s = coco_keypoint_eval(gts[:2] + [crowd], perfect)
    assert s[0] == 1.0 and s[5] == 1.0


def test_topdown_evaluator_end_to_end(tmp_path):
    rng = np.random.RandomState(1)
    size = 150.0
    images = [dict(id=10 + i, file_name=f"{i:012d}.jpg") for i in range(3)]
    kps = [_person(rng, 300, 300, size) for _ in range(3)]
    anns = [_gt(i + 1, 10 + i, kps[i], size) for i in range(3)]
    ann_file = tmp_path / "person_keypoints.json"
    ann_file.write_text(json.dumps(dict(images=images, annotations=anns, categories=[dict(id=1, name="person")])))
    cfg = dict(vis_thr=0.2, oks_thr=0.9, use_nms=True, soft_nms=False, sigmas=COCO_SIGMAS.tolist())
    ev = mp.TopDownEvaluator(str(ann_file), metric="AP", config=cfg, result_path=str(tmp_path / "res.json"), remove_result_file=False)
    assert mp.entrypoint("evaluator", "topdown") is mp.TopDownEvaluator
    records = []
    for i in range(3):
        pred = kps[i].copy()
        pred[:, 2] = 0.8  # key-point confidences
        box = np.array([300, 300, 1.0, 1.0, size * size, 0.95], np.float32)
        records.append(dict(pred=pred.astype(np.float32), box=box, image_path=f"/data/val2017/{i:012d}.jpg", bbox_id=i))
        # a near-duplicate detection of the same person with a lower box score: OKS NMS must drop it;
        # and a verbatim repeat of the same bbox_id: removed by _sort_and_unique_bboxes
        dup = pred.copy()
        dup[:, :2] += 0.5
        records.append(dict(pred=dup.astype(np.float32), box=box * np.array([1, 1, 1, 1, 1, 0.5], np.float32),
                            image_path=f"/data/val2017/{i:012d}.jpg", bbox_id=100 + i))
        records.append(records[-2])
    out = ev(records)
    assert set(out) == {"AP", "AP .5", "AP .75", "AP (M)", "AP (L)", "AR", "AR .5", "AR .75", "AR (M)", "AR (L)"}
    assert out["AP"] == 1.0 and out["AR"] == 1.0
    res = json.loads((tmp_path / "res.json").read_text())
    assert len(res) == 3 and all(abs(r["score"] - 0.8 * 0.95) < 1e-6 for r in res)  # rescoring = mean kpt score x box score
    with pytest.raises(KeyError):
        mp.TopDownEvaluator(str(ann_file), metric="PCK", config=cfg)


def test_engine_factories_merge_configs(tmp_path, caplog):
    import inspect
    images = [dict(id=1, file_name="a.jpg")]
    ann_file = tmp_path / "ann.json"
    ann_file.write_text(json.dumps(dict(images=images, annotations=[], categories=[dict(id=1, name="person")])))
    ev = mp.create_evaluator(str(ann_file), name="topdown", metric="AP", config=dict(vis_thr=0.2, oks_thr=0.9, use_nms=True),
                             dataset_config=dict(soft_nms=False, sigmas=COCO_SIGMAS.tolist(), vis_thr=0.3))
    assert isinstance(ev, mp.TopDownEvaluator) and ev._evaluation_cfg["vis_thr"] == 0.3  # dataset config wins, with a warning
    assert any("Duplicated keys" in r.message for r in caplog.records)
    inf = mp.create_inferencer(net=object(), name="topdown_heatmap", config=dict(has_heatmap_output=True, hflip_tta=False, shift_heatmap=False),
                               dataset_config=dict(flip_pairs=[[1, 2]]))
    assert isinstance(inf, mp.TopDownHeatMapInferencer)
    assert list(inspect.signature(mp.create_inferencer).parameters) == ["net", "name", "config", "dataset_config", "kwargs"]
    assert list(inspect.signature(mp.create_evaluator).parameters) == ["annotation_file", "name", "metric", "config",
                                                                         "dataset_config", "kwargs"]
