"""The evaluation protocol of tools/eval.py:55-103 as ONE chain on synthetic COCO-format data (X3 groundwork):

    save_checkpoint -> load_checkpoint -> load_param_into_net -> TopDownBoxToCenterScale -> TopDownAffine.crop_batch (HIP)
    -> TopDownHeatMapInferencer with flip test (HIP network x2 + fused aggregate/decode) -> TopDownEvaluator -> OKS-AP

COCO itself and the released checkpoint are not available offline, so the chain is pinned two ways:
  * a heat-map painter in the network's place (Gaussian targets at known key points, through the same inferencer / decoder /
    evaluator): detections are the ground truth re-encoded through the decoder's inverse -> AP = AR = 1.0 exactly, and a
    controlled perturbation (4 of 16 people moved away, their box scores lowest) -> AP = 76/101, AR = 0.75 by hand;
  * the real HRNet-W32 (synthetic weights through the .ckpt writer / reader) against the ORACLE chain (oracle crop, network,
    flip aggregation, decoder) used as ground truth -> AP = 1.0.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mindpose_amd as mp  # noqa: E402
from mindpose_amd.engine.inferencer.topdown_inferencer import COCO_FLIP_INDEX, COCO_FLIP_PAIRS  # noqa: E402
from mindpose_amd.engine.evaluator.coco_eval import KPT_OKS_SIGMAS  # noqa: E402

DEV = torch.device("cuda:0")
DATA_CFG = dict(image_size=[192, 256], heatmap_size=[48, 64], pixel_std=200.0, scale_padding=1.25, flip_pairs=COCO_FLIP_PAIRS,
                upper_body_ids=list(range(11)))
EVAL_CFG = dict(vis_thr=0.2, oks_thr=0.9, use_nms=True, soft_nms=False, sigmas=KPT_OKS_SIGMAS.tolist())
N_IMAGES, PER_IMAGE = 8, 2


def _scene(seed=0):
    """8 images of 480x640 with two well-separated person boxes each."""
    rng = np.random.RandomState(seed)
    images = [rng.randint(0, 256, (480, 640, 3)).astype(np.uint8) for _ in range(N_IMAGES)]
    boxes, image_index = [], []
    for i in range(N_IMAGES):
        for p in range(PER_IMAGE):
            x0 = 20 + 320 * p + rng.uniform(0, 30)
            y0 = 30 + rng.uniform(0, 40)
            boxes.append([x0, y0, rng.uniform(150, 220), rng.uniform(300, 380)])
            image_index.append(i)
    return images, np.array(boxes, np.float32), np.array(image_index)


def _write_annotations(path, boxes, image_index, keypoints):
    """COCO person_keypoints json: every joint labelled visible (v = 2), area = the box area."""
    coco = dict(images=[dict(id=100 + i, file_name=f"{i:012d}.jpg", height=480, width=640) for i in range(N_IMAGES)],
                categories=[dict(id=1, name="person", supercategory="person")], annotations=[])
    for j, (box, im) in enumerate(zip(boxes, image_index)):
        kp = np.concatenate([np.asarray(keypoints[j], np.float64)[:, :2], np.full((17, 1), 2.0)], axis=1)
        coco["annotations"].append(dict(id=j + 1, image_id=100 + int(im), category_id=1, iscrowd=0, num_keypoints=17,
                                        bbox=[float(v) for v in box], area=float(box[2] * box[3]),
                                        keypoints=kp.reshape(-1).tolist()))
    with open(path, "w") as f:
        json.dump(coco, f)


def _batches(crops, centers, scales, scores, image_index, batch=8):
    for s in range(0, len(centers), batch):
        e = min(s + batch, len(centers))
        yield dict(image=crops[s:e], center=torch.from_numpy(centers[s:e]).to(DEV), scale=torch.from_numpy(scales[s:e]).to(DEV),
                   bbox_scores=torch.from_numpy(scores[s:e]).to(DEV),
                   image_file=[f"/data/val2017/{int(i):012d}.jpg" for i in image_index[s:e]], bbox_ids=list(range(s, e)))


class _Painter(torch.nn.Module):
    """Stands where ``Net`` stands: returns Gaussian heat-maps (mp_gaussian_target) for the key points queued for the batch -
    crop-pixel coordinates - and, on every second call (the flip-test's second run), the heat-maps of the mirrored crop."""

    def __init__(self):
        super().__init__()
        self.target = mp.TopDownGenerateTarget(is_train=False, config=DATA_CFG, sigma=2.0)
        self.queue, self.calls, self.flip_index = [], 0, torch.tensor(COCO_FLIP_INDEX, device=DEV)

    def forward(self, image):
        second = self.calls % 2 == 1
        self.calls += 1
        kp = self.queue[0] if not second else self.queue.pop(0)
        heat, _ = self.target(kp)
        assert heat.shape[0] == image.shape[0]
        if second:  # what a network would produce for the mirrored crop: flip_back(second) == first
            heat = heat[:, self.flip_index].flip(3).contiguous()
        return heat


def test_ap_chain_with_known_keypoints_and_controlled_perturbation(tmp_path):
    images, boxes, image_index = _scene(1)
    n = len(boxes)
    b2cs = mp.TopDownBoxToCenterScale(is_train=False, config=DATA_CFG)
    centers, scales = b2cs.transform_batch(boxes)
    aff = mp.TopDownAffine(is_train=False, config=DATA_CFG)
    crops, _ = aff.crop_batch([torch.from_numpy(im).to(DEV) for im in images], centers, scales, image_index=image_index)
    assert crops.shape == (n, 3, 256, 192)
    # key points on the heat-map grid (crop pixel = 4 x heat-map pixel), GT = the decoder's back-projection of those points
    rng = np.random.RandomState(2)
    hxy = np.stack([rng.randint(4, 44, (n, 17)), rng.randint(4, 60, (n, 17))], axis=-1).astype(np.float32)
    moved = [3, 6, 9, 12]
    hxy[moved] = np.stack([rng.randint(28, 44, (4, 17)), rng.randint(36, 60, (4, 17))], axis=-1)  # lower-right quadrant
    crop_kp = np.concatenate([hxy * 4.0, np.ones((n, 17, 1), np.float32)], axis=-1)
    s = scales * np.float32(200.0)
    gt = np.empty((n, 17, 2), np.float32)
    gt[..., 0] = hxy[..., 0] * (s[:, 0:1] / np.float32(48)) + centers[:, 0:1] - s[:, 0:1] * np.float32(0.5)
    gt[..., 1] = hxy[..., 1] * (s[:, 1:2] / np.float32(64)) + centers[:, 1:2] - s[:, 1:2] * np.float32(0.5)
    ann = os.path.join(tmp_path, "person_keypoints_synth.json")
    _write_annotations(ann, boxes, image_index, gt)

    decoder = mp.create_decoder("topdown_heatmap").to(DEV)
    inf_cfg = dict(has_heatmap_output=True, hflip_tta=True, shift_heatmap=False, flip_pairs=COCO_FLIP_PAIRS)
    evaluator = mp.TopDownEvaluator(ann, metric="AP", num_joints=17, config=EVAL_CFG, result_path=os.path.join(tmp_path, "res.json"))

    def run(kp_per_person, scores):
        painter = _Painter()
        painter.queue = [torch.from_numpy(kp_per_person[s:s + 8]).to(DEV) for s in range(0, n, 8)]
        eval_net = mp.create_eval_network(painter, decoder, output_raw=True)
        inferencer = mp.TopDownHeatMapInferencer(eval_net, config=inf_cfg, decoder=decoder)
        records = inferencer(_batches(crops, centers, scales, scores, image_index))
        assert len(records) == n and painter.calls == 2 * ((n + 7) // 8)
        return records, evaluator(records)

    records, stats = run(crop_kp, np.ones(n, np.float32))
    pred = np.array([r["pred"] for r in records], np.float32)
    assert np.array_equal(pred[..., :2], gt) and np.all(pred[..., 2] == 1.0)  # exact: peak value 1 on a grid point
    for key in ("AP", "AP .5", "AP .75", "AR", "AR .5", "AR .75"):
        assert stats[key] == pytest.approx(1.0, abs=1e-12), (key, stats[key])

    # perturbation: people 3, 6, 9, 12 are detected far from where they are, with the lowest box scores -> 12 true positives ranked
    # first, recall tops out at 12/16 with precision 1: 76 of the 101 recall points (0.00 ... 0.75) carry precision 1 at every
    # OKS threshold -> AP = 76/101, AR = 0.75
    kp_bad = crop_kp.copy()
    kp_bad[moved, :, 0:2] = 16.0  # every joint at heat-map pixel (4, 4): >= 24 heat-map pixels (~ 115 image px) from each true joint,
    # a confident, wrong detection whose OKS against its own ground truth is < 0.2
    scores = np.ones(n, np.float32)
    scores[moved] = 1e-3
    _, stats_bad = run(kp_bad, scores)
    for key in ("AP", "AP .5", "AP .75"):
        assert stats_bad[key] == pytest.approx(76.0 / 101.0, abs=1e-9), (key, stats_bad[key])
    for key in ("AR", "AR .5", "AR .75"):
        assert stats_bad[key] == pytest.approx(0.75, abs=1e-12), (key, stats_bad[key])


def test_ap_chain_real_network_through_ckpt_vs_oracle_chain(tmp_path):
    from mindpose_amd.utils import load_checkpoint, load_param_into_net, save_checkpoint
    from oracle import decoder as od
    from oracle import loader as ol
    from oracle import nets as onets

    images, boxes, image_index = _scene(3)
    n = len(boxes)
    # weights -> .ckpt -> a fresh network (tools/eval.py:64-68)
    src = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=7)
    ckpt = os.path.join(tmp_path, "hrnet_w32_synth.ckpt")
    save_checkpoint({k: v.detach().cpu().numpy() for k, v in src.state_dict().items()}, ckpt)
    params = load_checkpoint(ckpt)
    net = mp.create_network("hrnet_w32", "hrnet_head")
    not_loaded = load_param_into_net(net, params)
    assert not not_loaded
    net = net.to(DEV).eval()
    for k, v in src.state_dict().items():
        assert torch.equal(net.state_dict()[k].cpu(), v), k

    # tools/eval.py:30-52: the records come from the detection file through create_dataset, the crops from create_pipeline
    # (Decode -> topdown_box_to_center_scale -> topdown_affine -> Normalize -> HWC2CHW, the pixel steps as one HIP launch per batch)
    scores = np.linspace(0.5, 1.0, n).astype(np.float32)
    img_dir = os.path.join(tmp_path, "val2017")
    os.makedirs(img_dir)
    for i, im in enumerate(images):
        with open(os.path.join(img_dir, f"{i:012d}.jpg"), "wb") as f:  # a .npy payload under the annotation's file name (no JPEG
            np.save(f, im)                                               # encoder offline; the pipeline's decoder sniffs the magic)
    ann0 = os.path.join(tmp_path, "person_keypoints_images_only.json")
    _write_annotations(ann0, boxes, image_index, np.zeros((n, 17, 2), np.float32))
    det_file = os.path.join(tmp_path, "detections.json")
    dets = [dict(image_id=100 + int(im), category_id=1, bbox=[float(v) for v in box], score=float(sc))
            for box, im, sc in zip(boxes, image_index, scores)]
    dets.insert(3, dict(image_id=100, category_id=1, bbox=[1.0, 1.0, 50.0, 50.0], score=0.01))   # below det_bbox_thr
    dets.insert(5, dict(image_id=101, category_id=3, bbox=[1.0, 1.0, 50.0, 50.0], score=0.99))   # not a person
    with open(det_file, "w") as f:
        json.dump(dets, f)
    dataset = mp.create_dataset(img_dir, ann0, dataset_format="coco_topdown", is_train=False, use_gt_bbox_for_val=False,
                                detection_file=det_file, num_workers=2, config=dict(det_bbox_thr=0.3))
    assert len(dataset) == n
    pipeline = mp.create_pipeline(dataset, ["topdown_box_to_center_scale", "topdown_affine"], method="topdown", batch_size=8,
                                  is_train=False, num_workers=2, config=DATA_CFG)
    batches = list(pipeline)
    assert len(batches) == len(pipeline) == 2 and list(batches[0]) == ["image", "image_file", "boxes", "bbox_ids", "center", "scale", "bbox_scores"]
    # the same crops / geometry as the transforms called directly
    b2cs = mp.TopDownBoxToCenterScale(is_train=False, config=DATA_CFG)
    centers, scales = b2cs.transform_batch(boxes)
    aff = mp.TopDownAffine(is_train=False, config=DATA_CFG)
    crops, _ = aff.crop_batch([torch.from_numpy(im).to(DEV) for im in images], centers, scales, image_index=image_index)
    assert torch.equal(torch.cat([b["image"] for b in batches]), crops)
    assert np.array_equal(torch.cat([b["center"] for b in batches]).cpu().numpy(), centers)
    assert np.array_equal(torch.cat([b["scale"] for b in batches]).cpu().numpy(), scales)
    assert np.array_equal(np.concatenate([b["bbox_ids"] for b in batches]), np.arange(n))
    assert np.array_equal(np.concatenate([b["boxes"] for b in batches]), boxes)
    assert batches[1]["image_file"][-1] == os.path.join(img_dir, f"{N_IMAGES - 1:012d}.jpg")
    decoder = mp.create_decoder("topdown_heatmap", shift_coordinate=True).to(DEV)
    eval_net = mp.create_eval_network(net, decoder, output_raw=True)
    inf_cfg = dict(has_heatmap_output=True, hflip_tta=True, shift_heatmap=True, flip_pairs=COCO_FLIP_PAIRS)
    inferencer = mp.TopDownHeatMapInferencer(eval_net, config=inf_cfg, decoder=decoder)
    records = inferencer(pipeline)

    # the oracle chain on the same files: crop (cv2.warpAffine restatement + Normalize + HWC2CHW), network twice, flip-back with
    # the one-pixel shift, average, arg-max + shift decode
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}
    ref_crops = np.stack([ol.crop(images[image_index[j]], centers[j], scales[j], 0.0, [192, 256])[0] for j in range(n)])
    assert np.array_equal(crops.cpu().numpy(), ref_crops)
    x = torch.from_numpy(ref_crops)
    h = onets.net_forward(sd, x, "hrnet_w32", "hrnet_head").numpy()
    hf = onets.net_forward(sd, torch.flip(x, dims=[3]), "hrnet_w32", "hrnet_head").numpy()
    avg = od.flip_aggregate(h, hf, COCO_FLIP_INDEX, shift_heatmap=True)
    ref_preds, ref_boxes, _ = od.decode(avg, centers, scales, scores, shift_coord=True)
    pred = np.array([r["pred"] for r in records], np.float32)
    box = np.array([r["box"] for r in records], np.float32)
    assert np.array_equal(box, ref_boxes)
    # heat-maps agree to ~3e-6 of their range; a key point only moves where two pixels tie within that - report how many did
    same = np.isclose(pred[..., :2], ref_preds[..., :2], rtol=0, atol=1e-3).all(axis=-1)
    assert same.mean() > 0.99, f"{(~same).sum()} of {same.size} key points moved"
    np.testing.assert_allclose(pred[..., 2], ref_preds[..., 2], rtol=1e-4, atol=1e-5)

    ann = os.path.join(tmp_path, "person_keypoints_from_oracle.json")
    _write_annotations(ann, boxes, image_index, ref_preds)
    evaluator = mp.TopDownEvaluator(ann, metric="AP", num_joints=17, config=dict(EVAL_CFG, vis_thr=-1e9, use_nms=False),
                                    result_path=os.path.join(tmp_path, "res2.json"))
    stats = evaluator(records)
    if same.all():
        assert stats["AP"] == pytest.approx(1.0, abs=1e-12) and stats["AR"] == pytest.approx(1.0, abs=1e-12)
    else:  # a tied arg-max moved one key point of one person: at most that person drops below the strictest OKS thresholds
        assert stats["AP"] > 0.9 and stats["AP .5"] == pytest.approx(1.0, abs=1e-12)
