"""Generate the committed golden fixtures under tests/golden/.

Run ONLY in the build container (needs /root/reference):  python tests/golden/gen_golden.py

* ``target_*.npz``  - outputs of the REFERENCE's own numpy ``TopDownGenerateTarget``
  (mindpose/data/transform/topdown_transform.py:264-430), loaded by file path so that
  ``mindpose/__init__.py`` (which imports MindSpore) is bypassed.  ``cv2`` is absent here; the
  module only calls ``cv2.setNumThreads`` at import (:29) and the target code never touches it,
  so an empty ``cv2`` module object is registered for the import.  No reference source or
  bytecode is written anywhere; only inputs and expected outputs are saved.
* ``decoder_*.npz``, ``loss.npz``, ``flip.npz`` - outputs of the CPU oracle restatement
  (MindSpore is not installable, so the reference itself cannot produce them: parity unpinned).
  They freeze the oracle so later edits cannot silently change it.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden import recipes  # noqa: E402

REF = "/root/reference/mindpose"


def load_reference_target_class():
    for name in ["mindpose", "mindpose.data", "mindpose.data.transform"]:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    sys.modules["cv2"] = cv2

    def load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    load("mindpose.register", REF + "/register.py")
    load("mindpose.data.column_names", REF + "/data/column_names.py")
    load("mindpose.data.transform.transform", REF + "/data/transform/transform.py")
    load("mindpose.data.transform.utils", REF + "/data/transform/utils.py")
    tt = load("mindpose.data.transform.topdown_transform", REF + "/data/transform/topdown_transform.py")
    return tt.TopDownGenerateTarget


def sparse_pack(target):
    """Store only the non-zero entries (targets are ~98 % zeros)."""
    flat = target.reshape(-1)
    nz = np.flatnonzero(flat)
    return nz.astype(np.int32), flat[nz]


def gen_targets():
    cls = load_reference_target_class()
    cases = [
        ("target_plain_64x48", dict(image_size=[192, 256], heatmap_size=[48, 64]), 2.0, False, None, 101),
        ("target_udp_64x48", dict(image_size=[192, 256], heatmap_size=[48, 64]), 2.0, True, None, 102),
        ("target_plain_96x72_s3", dict(image_size=[288, 384], heatmap_size=[72, 96]), 3.0, False, None, 103),
        ("target_udp_96x72_s3", dict(image_size=[288, 384], heatmap_size=[72, 96]), 3.0, True, None, 104),
        ("target_plain_jw", dict(image_size=[192, 256], heatmap_size=[48, 64]), 2.0, False,
         [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.2, 1.2, 1.5, 1.5, 1.0, 1.0, 1.2, 1.2, 1.5, 1.5], 105),
    ]
    for name, geom, sigma, udp, jw, seed in cases:
        cfg = dict(geom, flip_pairs=recipes.FLIP_PAIRS, upper_body_ids=list(range(11)),
                   pixel_std=200.0, scale_padding=1.25)
        if jw is not None:
            cfg["joint_weights"] = jw
        t = cls(is_train=True, config=cfg, sigma=sigma, use_udp=udp,
                use_different_joint_weights=jw is not None)
        n = 24
        kp = recipes.keypoint_sets(n, 17, geom["image_size"][0], geom["image_size"][1], seed)
        targets, weights = [], []
        for b in range(n):
            out = t.transform({"keypoints": kp[b]})
            targets.append(np.asarray(out["target"], dtype=np.float32))
            weights.append(np.asarray(out["target_weight"]))
        target = np.stack(targets)
        weight = np.stack(weights)
        nz_idx, nz_val = sparse_pack(target)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), keypoints=kp, target_shape=np.array(target.shape),
            target_nz_idx=nz_idx, target_nz_val=nz_val, target_weight=weight.astype(np.float32),
            weight_dtype=str(weight.dtype), sigma=sigma, use_udp=udp,
            image_size=np.array(geom["image_size"]), heatmap_size=np.array(geom["heatmap_size"]),
            joint_weights=np.array(jw if jw is not None else [], dtype=np.float64),
            source="reference:mindpose/data/transform/topdown_transform.py TopDownGenerateTarget "
                   "(numpy %s)" % np.__version__)
        print(name, target.shape, "nnz", nz_idx.size, "weights>0", int((weight > 0).sum()))


DECODER_CASES = [
    # name, heatmap recipe, (n,k,h,w), seed, kwargs
    ("uniform_plain", "uniform", (8, 17, 64, 48), 201, dict()),
    ("uniform_shift", "uniform", (8, 17, 64, 48), 202, dict(shift_coord=True)),
    ("uniform_dark_udp", "uniform", (8, 17, 64, 48), 203, dict(use_udp=True, dark_udp_refine=True)),
    ("refshape_plain", "uniform", (8, 17, 48, 64), 204, dict()),  # the reference test's own shape
    ("blob_plain", "blob", (4, 17, 64, 48), 211, dict()),
    ("blob_shift", "blob", (4, 17, 64, 48), 211, dict(shift_coord=True)),
    ("blob_dark", "blob", (4, 17, 64, 48), 211, dict(dark_udp_refine=True)),
    ("blob_dark_udp", "blob", (4, 17, 64, 48), 211, dict(use_udp=True, dark_udp_refine=True)),
    ("blob_udp_plain", "blob", (4, 17, 64, 48), 211, dict(use_udp=True)),
    ("blob_noorig_shift", "blob", (4, 17, 64, 48), 211, dict(shift_coord=True, to_original=False)),
    ("blob96_dark_udp_k17", "blob3", (3, 17, 96, 72), 212, dict(use_udp=True, dark_udp_refine=True, kernel_size=17)),
    ("blob_k5_odd", "blob", (2, 5, 20, 14), 213, dict(shift_coord=True)),
]


def decoder_inputs(kind, shape, seed):
    n, k, h, w = shape
    if kind == "uniform":
        hm = recipes.uniform_heatmaps(n, k, h, w, seed)
    elif kind == "blob":
        hm = recipes.blob_heatmaps(n, k, h, w, seed, sigma=2.0)
    else:
        hm = recipes.blob_heatmaps(n, k, h, w, seed, sigma=3.0)
    center, scale, score = recipes.boxes(n, seed + 1000)
    return hm, center, scale, score


def gen_decoder():
    from oracle import decoder as od
    out = {}
    for name, kind, shape, seed, kw in DECODER_CASES:
        hm, center, scale, score = decoder_inputs(kind, shape, seed)
        preds, boxes, idx = od.decode(hm, center, scale, score, **kw)
        out[name + "/preds"] = preds
        out[name + "/boxes"] = boxes
        out[name + "/idx"] = idx.astype(np.int32)
        print("decoder", name, preds.shape, float(np.abs(preds).max()))
    np.savez_compressed(os.path.join(HERE, "decoder.npz"), source="oracle/decoder.py (parity unpinned)", **out)


def gen_flip():
    from oracle import decoder as od
    out = {}
    for shift in (False, True):
        h = recipes.blob_heatmaps(3, 17, 64, 48, 301)
        hf = recipes.blob_heatmaps(3, 17, 64, 48, 302)
        center, scale, score = recipes.boxes(3, 1301)
        avg = od.flip_aggregate(h, hf, recipes.FLIP_INDEX, shift_heatmap=shift)
        preds, boxes, idx = od.decode(avg, center, scale, score, shift_coord=True)
        tag = "shift" if shift else "noshift"
        out[tag + "/avg_checksum"] = np.array([avg.astype(np.float64).sum(), np.abs(avg).astype(np.float64).sum()])
        out[tag + "/avg_sample"] = avg[:, :, ::7, ::5].copy()
        out[tag + "/preds"] = preds
        out[tag + "/boxes"] = boxes
        out[tag + "/idx"] = idx.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "flip.npz"), source="oracle/decoder.py (parity unpinned)", **out)
    print("flip done")


def gen_loss():
    from oracle import loss as ol
    out = {}
    for name, (shape, seed) in recipes.LOSS_CASES.items():
        pred, target, w = recipes.loss_inputs(shape, seed)
        out[name + "/loss_plain"] = ol.joints_mse(pred, target)
        out[name + "/loss_weighted"] = ol.joints_mse(pred, target, w, use_target_weight=True)
        g = ol.joints_mse_grad(pred, target, w, use_target_weight=True)
        out[name + "/grad_sample"] = g[:, :, ::8, ::8].copy()
        out[name + "/grad_abs_sum"] = np.abs(g).astype(np.float64).sum()
    np.savez_compressed(os.path.join(HERE, "loss.npz"), source="oracle/loss.py (parity unpinned)", **out)
    print("loss done")


def gen_geometry():
    """Loader geometry computed by the REFERENCE's own numpy code (no cv2 call on these paths):
    TopDownBoxToCenterScale._xywh2cs in eval mode (topdown_transform.py:131-154) and get_warp_matrix (utils.py:150-181)."""
    load_reference_target_class()
    tt = sys.modules["mindpose.data.transform.topdown_transform"]
    ut = sys.modules["mindpose.data.transform.utils"]
    rng = np.random.RandomState(77)
    out = {}
    for name, image_size in (("256x192", [192, 256]), ("384x288", [288, 384])):
        cfg = dict(image_size=image_size, heatmap_size=[image_size[0] // 4, image_size[1] // 4], flip_pairs=recipes.FLIP_PAIRS,
                   upper_body_ids=list(range(11)), pixel_std=200.0, scale_padding=1.25)
        t = tt.TopDownBoxToCenterScale(is_train=False, config=cfg)
        boxes = np.concatenate([rng.uniform(0, 500, (40, 2)), rng.uniform(5, 400, (40, 2))], axis=1).astype(np.float32)
        boxes[0] = [10, 20, 96, 128]   # exactly the aspect ratio
        boxes[1] = [0, 0, 300, 10]     # very wide
        boxes[2] = [5.5, 7.25, 3, 400]  # very tall
        cs = [t.transform(dict(boxes=b)) for b in boxes]
        out[f"boxes_{name}"] = boxes
        out[f"center_{name}"] = np.stack([c["center"] for c in cs])
        out[f"scale_{name}"] = np.stack([c["scale"] for c in cs])
        mats = []
        args = []
        for i in range(24):
            theta = float(rng.uniform(-45, 45)) if i % 3 else 0.0
            center = rng.uniform(50, 400, 2).astype(np.float32)
            scale = rng.uniform(0.3, 3.0, 2).astype(np.float32)
            size = np.array(image_size)
            mats.append(ut.get_warp_matrix(theta, center * 2.0, size - 1.0, scale * 200.0))
            args.append([theta, *center, *scale])
        out[f"warp_args_{name}"] = np.array(args, np.float64)
        out[f"warp_matrix_{name}"] = np.stack(mats)
    # training-time augmentations (numpy + np.random only): seeded runs of the reference classes
    cfg = dict(image_size=[192, 256], heatmap_size=[48, 64], flip_pairs=recipes.FLIP_PAIRS, upper_body_ids=list(range(11)),
               pixel_std=200.0, scale_padding=1.25)
    kps = rng.uniform(0, 300, (30, 17, 3)).astype(np.float32)
    kps[..., 2] = (rng.rand(30, 17) > 0.25).astype(np.float32)
    kps[3, :, 2] = 0
    kps[4, 6:, 2] = 0  # upper body only
    scales = rng.uniform(0.4, 2.5, (30, 2)).astype(np.float32)
    out["aug_keypoints"], out["aug_scales"] = kps, scales
    hb = tt.TopDownHalfBodyTransform(is_train=True, config=cfg)
    rs = tt.TopDownRandomScaleRotation(is_train=True, config=cfg)
    np.random.seed(4321)
    hb_c, hb_s, hb_hit, rs_s, rs_r = [], [], [], [], []
    for i in range(30):
        o = hb.transform(dict(keypoints=kps[i].copy()))
        hb_hit.append(1 if o else 0)
        hb_c.append(o.get("center", np.zeros(2, np.float32)))
        hb_s.append(o.get("scale", np.zeros(2, np.float32)))
        o = rs.transform(dict(scale=scales[i].copy()))
        rs_s.append(o["scale"])
        rs_r.append(o["rotation"])
    out["aug_halfbody_hit"] = np.array(hb_hit)
    out["aug_halfbody_center"], out["aug_halfbody_scale"] = np.stack(hb_c), np.stack(hb_s)
    out["aug_rs_scale"], out["aug_rs_rotation"] = np.stack(rs_s), np.array(rs_r, np.float32)
    out["aug_fliplr_index"] = ut.fliplr_joints(kps, 192, flip_index=hb._transform_cfg["flip_index"])
    out["aug_fliplr_pairs"] = ut.fliplr_joints(kps, 192, flip_pairs=recipes.FLIP_PAIRS)
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out)
    print("geometry.npz", {k: v.shape for k, v in out.items()})


def nms_cases():
    """Seeded person lists: clusters of near-duplicate poses (so that NMS has something to suppress) + distinct ones."""
    rng = np.random.RandomState(2024)
    cases = []
    for n_people, dup in ((1, 0), (6, 3), (12, 8), (25, 10), (40, 30)):
        people = []
        base = rng.uniform(50, 400, (n_people, 17, 2))
        for p in range(n_people):
            kp = np.concatenate([base[p], rng.uniform(0.05, 1.0, (17, 1))], axis=1)
            people.append(dict(keypoints=kp.astype(np.float32), score=float(rng.uniform(0.1, 1.0)), area=float(rng.uniform(2e3, 6e4))))
        for _ in range(dup):
            src = people[rng.randint(0, n_people)]
            kp = src["keypoints"].copy()
            kp[:, :2] += rng.normal(0, rng.choice([1.0, 4.0, 12.0]), (17, 2)).astype(np.float32)
            people.append(dict(keypoints=kp, score=float(rng.uniform(0.1, 1.0)), area=src["area"] * float(rng.uniform(0.8, 1.2))))
        cases.append(people)
    return cases


def gen_nms():
    """OKS NMS outputs of the REFERENCE's own numpy module (mindpose/utils/nms.py, no MindSpore / cv2 import)."""
    spec = importlib.util.spec_from_file_location("ref_nms", REF + "/utils/nms.py")
    nms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nms)
    out = {}
    for ci, people in enumerate(nms_cases()):
        out[f"c{ci}_keypoints"] = np.stack([p["keypoints"] for p in people])
        out[f"c{ci}_score"] = np.array([p["score"] for p in people])
        out[f"c{ci}_area"] = np.array([p["area"] for p in people])
        kpts = np.array([p["keypoints"].flatten() for p in people])
        areas = out[f"c{ci}_area"]
        out[f"c{ci}_iou0"] = nms.oks_iou(kpts[0], kpts[1:], areas[0], areas[1:])
        out[f"c{ci}_iou0_vis"] = nms.oks_iou(kpts[0], kpts[1:], areas[0], areas[1:], None, 0.4)
        for thr in (0.5, 0.9):
            tag = str(thr).replace(".", "")
            out[f"c{ci}_keep_{tag}"] = np.asarray(nms.oks_nms(people, thr), dtype=np.int64)
            out[f"c{ci}_keep_vis_{tag}"] = np.asarray(nms.oks_nms(people, thr, None, 0.4), dtype=np.int64)
            out[f"c{ci}_soft_{tag}"] = np.asarray(nms.soft_oks_nms(people, thr), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **out)
    print("nms.npz", len(out), "arrays")


def gen_dataset():
    """``TopDownDataset._sanitize_bbox`` (mindpose/data/dataset/topdown.py:123-137) from the reference itself: the module imports
    only numpy / logging / copy, so it loads by file path.  Inputs: boxes around and across the image border, zero / negative
    extents, annotations without ``bbox`` or with ``area`` <= 0; outputs: which annotations survive and their clipped boxes."""
    spec = importlib.util.spec_from_file_location("ref_topdown_dataset", REF + "/data/dataset/topdown.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.RandomState(7)
    out = {}
    for case, (w, h) in enumerate([(640, 480), (333, 500), (64, 64)]):
        n = 40
        boxes = np.stack([rng.uniform(-0.3 * w, 1.1 * w, n), rng.uniform(-0.3 * h, 1.1 * h, n), rng.uniform(-5, 0.8 * w, n),
                          rng.uniform(-5, 0.8 * h, n)], axis=1)
        boxes[::7] = np.round(boxes[::7])          # integer boxes (COCO has both)
        boxes[3] = [10.0, 10.0, 1.0, 50.0]         # one pixel wide: x2 == x1 -> dropped
        boxes[4] = [w - 1.0, 5.0, 30.0, 30.0]      # starts on the last column
        area = rng.uniform(-1.0, 50.0, n)          # some <= 0
        has_area = rng.rand(n) < 0.8
        has_bbox = rng.rand(n) < 0.9
        annos = []
        for i in range(n):
            a = dict(id=i)
            if has_bbox[i]:
                a["bbox"] = [float(v) for v in boxes[i]]
            if has_area[i]:
                a["area"] = float(area[i])
            annos.append(a)
        kept = mod.TopDownDataset._sanitize_bbox(annos, w, h)
        out[f"c{case}_size"] = np.array([w, h])
        out[f"c{case}_boxes"], out[f"c{case}_area"] = boxes, area
        out[f"c{case}_has_area"], out[f"c{case}_has_bbox"] = has_area, has_bbox
        out[f"c{case}_kept_ids"] = np.array([a["id"] for a in kept], dtype=np.int64)
        out[f"c{case}_kept_boxes"] = np.array([a["bbox"] for a in kept], dtype=np.float64).reshape(-1, 4)
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), source="reference mindpose/data/dataset/topdown.py::_sanitize_bbox", **out)


def gen_helpers():
    """Point helpers of the REFERENCE's mindpose/data/transform/utils.py (numpy only): affine_transform, rotate_point,
    warp_affine_joints, pad_to_same, transform_keypoints on seeded inputs."""
    load_reference_target_class()
    ut = sys.modules["mindpose.data.transform.utils"]
    rng = np.random.RandomState(31)
    out = {}
    mats = rng.uniform(-2, 2, (12, 2, 3))
    pts = rng.uniform(-100, 400, (12, 2)).astype(np.float32)
    angles = rng.uniform(-3.2, 3.2, 12)
    out["mats"], out["pts"], out["angles"] = mats, pts, angles
    out["affine"] = np.stack([ut.affine_transform(tuple(p), m) for p, m in zip(pts, mats)])
    out["rotated"] = np.array([ut.rotate_point(tuple(p), a) for p, a in zip(pts, angles)])
    joints = rng.uniform(0, 300, (5, 17, 2)).astype(np.float32)
    out["joints"] = joints
    out["warped_joints"] = ut.warp_affine_joints(joints, mats[0])
    ragged = [rng.uniform(0, 1, sh).astype(np.float32) for sh in ((3, 5), (1, 7), (4, 2))]
    for i, (a, b) in enumerate(zip(ragged, ut.pad_to_same(ragged))):
        out[f"ragged_{i}"], out[f"padded_{i}"] = a, b
    coords = [rng.uniform(0, 64, (17, 3, 3)).astype(np.float32), np.zeros((17, 0, 3), np.float32), rng.uniform(0, 64, (17, 1, 3)).astype(np.float32)]
    center = rng.uniform(50, 300, (3, 2)).astype(np.float32)
    scale = rng.uniform(0.5, 2.0, (3, 2)).astype(np.float32)
    hm = np.array([[48, 64], [48, 64], [72, 96]], np.float32)
    res = ut.transform_keypoints(coords, center, scale, hm)
    out["tk_center"], out["tk_scale"], out["tk_heatmap_shape"] = center, scale, hm
    for i, (a, b) in enumerate(zip(coords, res)):
        out[f"tk_in_{i}"], out[f"tk_out_{i}"] = a, b
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
    print("helpers.npz", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["targets", "decoder", "flip", "loss", "geometry", "nms", "helpers", "dataset"]
    if "dataset" in which:
        gen_dataset()
    if "helpers" in which:
        gen_helpers()
    if "nms" in which:
        gen_nms()
    if "geometry" in which:
        gen_geometry()
    if "targets" in which:
        gen_targets()
    if "decoder" in which:
        gen_decoder()
    if "flip" in which:
        gen_flip()
    if "loss" in which:
        gen_loss()
