"""Dataset record loader + ``create_dataset`` / ``create_pipeline`` plumbing, CPU part.

``_sanitize_bbox`` is pinned bit-exact by goldens produced by the reference's own function (tests/golden/gen_golden.py
``gen_dataset``: mindpose/data/dataset/topdown.py imports only numpy).  The COCO record lists cannot be produced by the reference
here (coco_topdown.py needs pycocotools): hand cases for every rule of coco_topdown.py:64-163.
"""
import json
import os

import numpy as np
import pytest

import mindpose_amd as mp
from mindpose_amd.data.data_factory import ShardedDataset, _convert_names_to_transform
from mindpose_amd.data.dataset import COCOTopDownDataset, TopDownDataset
from tests.golden_io import load_npz


def test_sanitize_bbox_vs_reference_golden():
    z = load_npz("dataset.npz")
    for case in range(3):
        w, h = (int(v) for v in z[f"c{case}_size"])
        boxes, area = z[f"c{case}_boxes"], z[f"c{case}_area"]
        annos = []
        for i in range(len(boxes)):
            a = dict(id=i)
            if z[f"c{case}_has_bbox"][i]:
                a["bbox"] = [float(v) for v in boxes[i]]
            if z[f"c{case}_has_area"][i]:
                a["area"] = float(area[i])
            annos.append(a)
        kept = TopDownDataset._sanitize_bbox(annos, w, h)
        assert [a["id"] for a in kept] == z[f"c{case}_kept_ids"].tolist()
        got = np.array([a["bbox"] for a in kept], dtype=np.float64).reshape(-1, 4)
        assert np.array_equal(got, z[f"c{case}_kept_boxes"])  # bit-exact
        assert annos[int(kept[0]["id"])] is not kept[0]         # deep copies: the source annotations stay untouched


def _kp(vis):
    k = np.zeros((17, 3))
    k[:, 0], k[:, 1], k[:, 2] = np.arange(17) + 10, np.arange(17) + 20, vis
    return k.reshape(-1).tolist()


def _write_coco(tmp_path):
    images = [dict(id=7, file_name="b.jpg", width=100, height=80), dict(id=3, file_name="a.jpg", width=64, height=64),
              dict(id=9, file_name="c.jpg", width=50, height=50)]
    anns = [
        dict(id=1, image_id=3, category_id=1, iscrowd=0, bbox=[5, 5, 20, 30], area=600, num_keypoints=17, keypoints=_kp(2)),
        dict(id=2, image_id=7, category_id=1, iscrowd=1, bbox=[5, 5, 20, 30], area=600, num_keypoints=17, keypoints=_kp(2)),   # crowd
        dict(id=3, image_id=7, category_id=1, iscrowd=0, bbox=[5, 5, 20, 30], area=600, num_keypoints=0, keypoints=[0] * 51),   # no key points
        dict(id=4, image_id=7, category_id=1, iscrowd=0, bbox=[-4, 10, 30, 40], area=900, num_keypoints=5, keypoints=_kp(1)),   # clipped box
        dict(id=5, image_id=7, category_id=1, iscrowd=0, bbox=[10, 10, 1, 40], area=40, num_keypoints=5, keypoints=_kp(1)),     # degenerate
        dict(id=6, image_id=7, category_id=1, iscrowd=0, bbox=[20, 10, 30, 40], area=0, num_keypoints=5, keypoints=_kp(1)),     # area 0
        dict(id=7, image_id=7, category_id=1, iscrowd=0, bbox=[30, 20, 30, 40], area=50, keypoints=_kp(2)),                     # no num_keypoints
        dict(id=8, image_id=3, category_id=1, iscrowd=0, bbox=[1, 2, 30, 40], area=50),                                          # no keypoints key
        dict(id=9, image_id=7, category_id=1, iscrowd=0, bbox=[1, 2, 30, 40], area=50, num_keypoints=3, keypoints=[0] * 51),    # all zero
    ]
    path = os.path.join(tmp_path, "ann.json")
    with open(path, "w") as f:
        json.dump(dict(images=images, annotations=anns, categories=[dict(id=1, name="person")]), f)
    return path


def test_coco_ground_truth_records(tmp_path):
    ann = _write_coco(tmp_path)
    ds = COCOTopDownDataset("/img", ann, is_train=True, config=dict(det_bbox_thr=0.0))
    recs = [ds.record(i) for i in range(len(ds))]
    # images in FILE order (7, 3, 9), annotations per image in file order; bbox_ids restart per image
    assert [(os.path.basename(r["image_file"]), r["bbox_ids"]) for r in recs] == [("b.jpg", 0), ("b.jpg", 1), ("a.jpg", 0)]
    assert recs[0]["boxes"] == [0, 10, 29, 39]      # annotation 4: x clamped to 0, far corner = x1 + w - 1
    assert recs[1]["boxes"] == [30, 20, 29, 39]     # annotation 7 (no num_keypoints field): kept
    assert recs[2]["boxes"] == [5, 5, 19, 29]
    assert all(r["rotation"] == 0 and r["bbox_scores"] == 1.0 for r in recs)
    assert recs[2]["keypoints"].shape == (17, 3) and set(recs[2]["keypoints"][:, 2].tolist()) == {1}   # visibility 2 -> 1
    assert recs[0]["keypoints"][:, :2].tolist() == np.array(_kp(1)).reshape(17, 3)[:, :2].tolist()
    # use_gt_bbox_for_val takes the same branch
    dv = COCOTopDownDataset("/img", ann, is_train=False, use_gt_bbox_for_val=True, config=dict(det_bbox_thr=0.0))
    assert len(dv) == 3


def test_coco_detection_records_and_errors(tmp_path):
    ann = _write_coco(tmp_path)
    dets = [dict(image_id=7, category_id=1, bbox=[1, 2, 3, 4], score=0.9), dict(image_id=3, category_id=2, bbox=[1, 2, 3, 4], score=0.99),
            dict(image_id=3, category_id=1, bbox=[5, 6, 7, 8], score=0.1), dict(image_id=9, category_id=1, bbox=[9, 9, 9, 9], score=0.3),
            dict(image_id=3, category_id=1, bbox=[2, 2, 2, 2], score=0.29)]
    det_file = os.path.join(tmp_path, "det.json")
    with open(det_file, "w") as f:
        json.dump(dets, f)
    ds = COCOTopDownDataset("/img", ann, is_train=False, detection_file=det_file, config=dict(det_bbox_thr=0.3))
    recs = [ds.record(i) for i in range(len(ds))]
    assert [(os.path.basename(r["image_file"]), r["bbox_ids"], r["bbox_scores"]) for r in recs] == [("b.jpg", 0, 0.9), ("c.jpg", 1, 0.3)]
    assert "keypoints" not in recs[0] and recs[0]["boxes"] == [1, 2, 3, 4]
    with pytest.raises(ValueError, match="detection_file"):
        COCOTopDownDataset("/img", None, is_train=False, config=dict(det_bbox_thr=0.0))
    with pytest.raises(KeyError):
        COCOTopDownDataset("/img", ann, is_train=True, config=dict())  # det_bbox_thr is required, as in the reference


def test_getitem_columns(tmp_path):
    ann = _write_coco(tmp_path)
    for name in ("a.jpg", "b.jpg", "c.jpg"):
        with open(os.path.join(tmp_path, name), "wb") as f:
            f.write(b"\xff\xd8payload-" + name.encode())
    tr = COCOTopDownDataset(str(tmp_path), ann, is_train=True, config=dict(det_bbox_thr=0.0))
    item = tr[0]
    assert len(item) == 8 and item[0].dtype == np.uint8 and item[0].tobytes() == b"\xff\xd8payload-b.jpg"
    assert item[3].dtype == np.float32 and item[3].tolist() == [0, 10, 29, 39] and item[4].shape == (17, 3) and item[4].dtype == np.float32
    va = COCOTopDownDataset(str(tmp_path), ann, is_train=False, use_gt_bbox_for_val=True, config=dict(det_bbox_thr=0.0))
    item = va[2]
    assert len(item) == 8 and item[4].endswith("a.jpg") and item[6] == 0 and item[6].dtype == np.int32 and item[7] == np.float32(1.0)


def test_create_dataset_sharding_and_registry(tmp_path):
    ann = _write_coco(tmp_path)
    full = mp.create_dataset(str(tmp_path), ann, is_train=False, use_gt_bbox_for_val=True, detection_file=None, config=dict(det_bbox_thr=0.0))
    assert isinstance(full, ShardedDataset) and len(full) == 3 and full.indices().tolist() == [0, 1, 2]
    assert full.column_names == ["image", "center", "scale", "rotation", "image_file", "boxes", "bbox_ids", "bbox_scores"]
    shards = [mp.create_dataset(str(tmp_path), ann, is_train=False, use_gt_bbox_for_val=True, device_num=2, rank_id=r,
                                config=dict(det_bbox_thr=0.0)) for r in range(2)]
    assert [s.indices().tolist() for s in shards] == [[0, 2], [1, 0]]  # round robin, wrapped to equal length
    tr = mp.create_dataset(str(tmp_path), ann, is_train=True, config=dict(det_bbox_thr=0.0))
    e0 = tr.indices().tolist()
    list(iter(tr.source and []))  # no I/O
    tr.epoch += 1
    assert sorted(e0) == [0, 1, 2] and sorted(tr.indices().tolist()) == [0, 1, 2]
    with pytest.raises(ValueError):
        mp.create_dataset(str(tmp_path), ann, dataset_format="coco_bottomup", config=dict(det_bbox_thr=0.0))
    with pytest.raises(ValueError):
        ShardedDataset([], [], False, num_shards=2, shard_id=2)


def test_transform_list_from_names():
    cfg = dict(image_size=[192, 256], heatmap_size=[48, 64], pixel_std=200.0, scale_padding=1.25, flip_pairs=[[1, 2]], upper_body_ids=[0, 1, 2])
    ts = _convert_names_to_transform(["topdown_box_to_center_scale", {"topdown_horizontal_random_flip": {"flip_prob": 0.25}},
                                      {"topdown_affine": None}, {"topdown_generate_target": {"sigma": 3.0, "use_udp": True}}],
                                     is_train=True, config=cfg)
    assert [type(t).__name__ for t in ts] == ["TopDownBoxToCenterScale", "TopDownHorizontalRandomFlip", "TopDownAffine", "TopDownGenerateTarget"]
    assert ts[1].flip_prob == 0.25 and ts[3].sigma == 3.0 and ts[3].use_udp and all(t.is_train for t in ts)
    with pytest.raises(ValueError):
        _convert_names_to_transform(["topdown_no_such_transform"], config=cfg)
    with pytest.raises(ValueError):
        mp.create_pipeline(ShardedDataset([], ["image"], False), [], method="bottomup")
