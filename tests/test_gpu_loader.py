"""Fused crop kernel (warpAffine + Normalize + HWC2CHW) against the CPU oracle, through the C ABI.
Integer pixel values are compared bit-exact; the normalised fp32 output is the same IEEE expression on both sides, so it
is compared bit-exact too."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import mindpose_amd as mp  # noqa: E402
from oracle import loader as ol  # noqa: E402
from tests.golden import recipes  # noqa: E402

DEV = torch.device("cuda:0")
CFG = dict(heatmap_size=[48, 64], flip_pairs=recipes.FLIP_PAIRS, upper_body_ids=list(range(11)), pixel_std=200.0,
           scale_padding=1.25)


def _images(rng):
    return [rng.randint(0, 256, s).astype(np.uint8) for s in ((480, 640, 3), (333, 500, 3), (97, 61, 3))]


@pytest.mark.parametrize("use_udp", [False, True])
@pytest.mark.parametrize("image_size", [[192, 256], [288, 384]])
def test_crop_batch_bit_exact_vs_oracle(use_udp, image_size):
    rng = np.random.RandomState(11 + use_udp)
    imgs = _images(rng)
    dimgs = [torch.from_numpy(i).to(DEV) for i in imgs]
    n = 24
    index = rng.randint(0, len(imgs), n)
    boxes = []
    for i in range(n):
        h, w, _ = imgs[index[i]].shape
        bw, bh = rng.uniform(10, w * 1.2), rng.uniform(10, h * 1.2)
        boxes.append([rng.uniform(-0.3 * w, w), rng.uniform(-0.3 * h, h), bw, bh])  # partly / mostly outside included
    boxes = np.array(boxes, np.float32)
    b2cs = mp.TopDownBoxToCenterScale(is_train=False, config=dict(CFG, image_size=image_size))
    centers, scales = b2cs.transform_batch(boxes)
    rot = np.where(np.arange(n) % 3 == 0, 0.0, rng.uniform(-40, 40, n))
    aff = mp.TopDownAffine(is_train=False, config=dict(CFG, image_size=image_size), use_udp=use_udp)
    crops, mats = aff.crop_batch(dimgs, centers, scales, rot, image_index=index)
    assert crops.shape == (n, 3, image_size[1], image_size[0]) and crops.dtype == torch.float32
    got = crops.cpu().numpy()
    for i in range(n):
        ref, trans = ol.crop(imgs[index[i]], centers[i], scales[i], float(rot[i]), image_size, use_udp=use_udp)
        assert np.array_equal(np.asarray(mats[i], np.float64), np.asarray(trans, np.float64))
        assert np.array_equal(got[i], ref), f"crop {i}: {np.abs(got[i] - ref).max()}"


def test_transform_per_sample_contract_and_errors():
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (200, 300, 3)).astype(np.uint8)
    aff = mp.TopDownAffine(is_train=False, config=dict(CFG, image_size=[192, 256]))
    kp = np.concatenate([rng.uniform(0, 200, (17, 2)), (rng.rand(17, 1) > 0.3)], axis=1).astype(np.float32)
    center, scale = np.array([150.0, 100.0], np.float32), np.array([0.9, 1.2], np.float32)
    out = aff.transform(dict(image=torch.from_numpy(img).to(DEV), center=center, scale=scale, rotation=15.0, keypoints=kp.copy()))
    trans = ol.get_affine_transform(center, scale, 15.0, np.array([192, 256]))
    assert out["image"].dtype == torch.uint8 and tuple(out["image"].shape) == (256, 192, 3)
    assert np.array_equal(out["image"].cpu().numpy(), ol.warp_affine(img, trans, 192, 256))
    want = kp.copy()
    for i in range(17):
        if want[i, 2] > 0:
            want[i, :2] = trans @ np.array([want[i, 0], want[i, 1], 1.0])
    assert np.array_equal(out["keypoints"], want)
    with pytest.raises(mp._lib.MindposeHipError):
        aff.transform(dict(image=img, center=center, scale=scale, rotation=0.0))  # numpy image: no CPU fallback
    with pytest.raises(mp._lib.MindposeHipError):
        aff.crop_batch(torch.from_numpy(img), center[None], scale[None])  # CPU tensor


def test_boxes_to_keypoints_end_to_end():
    # detector boxes -> centre/scale -> fused crop written straight into the plan's input buffer -> network -> decode
    from oracle import decoder as od
    from oracle import nets as onets
    rng = np.random.RandomState(9)
    img = rng.randint(0, 256, (360, 480, 3)).astype(np.uint8)
    boxes = np.array([[40, 30, 120, 260], [200, 50, 180, 200], [10, 10, 400, 300]], np.float32)
    cfg = dict(CFG, image_size=[192, 256])
    centers, scales = mp.TopDownBoxToCenterScale(is_train=False, config=cfg).transform_batch(boxes)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).eval()
    dec = mp.create_decoder("topdown_heatmap", shift_coordinate=True).to(DEV)
    ev = mp.create_eval_network(net, dec, output_raw=True)
    buf = net.input_buffer((3, 3, 256, 192), DEV)
    crops, _ = mp.TopDownAffine(is_train=False, config=cfg).crop_batch(torch.from_numpy(img).to(DEV), centers, scales, out=buf)
    assert crops.data_ptr() == buf.data_ptr()
    score = torch.ones(3, device=DEV)
    (preds, pboxes), hm = ev(buf, torch.from_numpy(centers).to(DEV), torch.from_numpy(scales).to(DEV), score)
    x = np.stack([ol.crop(img, centers[i], scales[i], 0.0, [192, 256])[0] for i in range(3)])
    ref_hm = onets.net_forward({k: v.cpu() for k, v in net.state_dict().items()}, torch.from_numpy(x), "hrnet_w32", "hrnet_head")
    assert float((hm.cpu() - ref_hm).abs().max() / ref_hm.abs().max()) < 1e-3
    rp, rb, _ = od.decode(hm.cpu().numpy(), centers, scales, np.ones(3, np.float32), shift_coord=True)
    assert np.array_equal(preds.cpu().numpy(), rp) and np.array_equal(pboxes.cpu().numpy(), rb)


def test_training_augmentation_chain_flip_in_kernel():
    # flip -> half-body / random scale-rotation -> fused crop with the flip applied while sampling == crop of a flipped copy
    rng = np.random.RandomState(21)
    img = rng.randint(0, 256, (240, 320, 3)).astype(np.uint8)
    cfg = dict(CFG, image_size=[192, 256])
    np.random.seed(99)
    fl = mp.TopDownHorizontalRandomFlip(True, cfg, flip_prob=0.5)
    rs = mp.TopDownRandomScaleRotation(True, cfg)
    aff = mp.TopDownAffine(True, cfg)
    centers, scales, rots, flips = [], [], [], []
    for i in range(12):
        kp = np.concatenate([rng.uniform(0, 240, (17, 2)), np.ones((17, 1))], axis=1).astype(np.float32)
        c0 = np.array([rng.uniform(60, 260), rng.uniform(60, 180)], np.float32)
        st = fl.transform(dict(image=img, keypoints=kp, center=c0.copy()))
        flips.append(st["image"] is not img)
        o = rs.transform(dict(scale=np.array([0.8, 1.1], np.float32)))
        centers.append(st["center"]); scales.append(o["scale"]); rots.append(float(o["rotation"]))
    assert any(flips) and not all(flips)
    dimg = torch.from_numpy(img).to(DEV)
    crops, _ = aff.crop_batch(dimg, np.stack(centers), np.stack(scales), np.array(rots), flips=flips)
    got = crops.cpu().numpy()
    for i in range(12):
        src = np.ascontiguousarray(img[:, ::-1]) if flips[i] else img
        ref, _ = ol.crop(src, centers[i], scales[i], rots[i], [192, 256])
        assert np.array_equal(got[i], ref), i


def test_training_pipeline_batches_match_per_sample_transforms(tmp_path):
    """create_dataset / create_pipeline in training mode (data_factory.py:116-151 with the recipe's transform list): the batched
    GPU pipeline (one crop launch + one target launch per batch) against the SAME transforms applied sample by sample with the
    same random draws - crops and targets bit-exact - plus shuffling, drop_remainder and the final columns."""
    import json
    import os
    from oracle import loader as ol
    from oracle import target as otarget
    rng = np.random.RandomState(5)
    n_img = 7
    cfg = dict(image_size=[192, 256], heatmap_size=[48, 64], pixel_std=200.0, scale_padding=1.25, upper_body_ids=list(range(11)),
               flip_pairs=[[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]], det_bbox_thr=0.0)
    images, anns = [], []
    for i in range(n_img):
        im = rng.randint(0, 256, (120 + 8 * i, 160, 3)).astype(np.uint8)
        with open(os.path.join(tmp_path, f"{i}.jpg"), "wb") as f:
            np.save(f, im)
        images.append(dict(id=i + 1, file_name=f"{i}.jpg", width=160, height=120 + 8 * i))
        kp = np.concatenate([rng.uniform(20, 100, (17, 2)), rng.randint(0, 3, (17, 1))], axis=1)
        kp[kp[:, 2] == 0, :2] = 0
        anns.append(dict(id=i + 1, image_id=i + 1, category_id=1, iscrowd=0, bbox=[15.0, 12.0, 100.0, 90.0], area=9000.0,
                         num_keypoints=int((kp[:, 2] > 0).sum()), keypoints=kp.reshape(-1).tolist()))
    ann = os.path.join(tmp_path, "train.json")
    with open(ann, "w") as f:
        json.dump(dict(images=images, annotations=anns, categories=[dict(id=1, name="person")]), f)
    names = ["topdown_box_to_center_scale", {"topdown_horizontal_random_flip": {"flip_prob": 0.5}}, "topdown_halfbody_transform",
             "topdown_randomscale_rotation", "topdown_affine", {"topdown_generate_target": {"sigma": 2.0}}]
    ds = mp.create_dataset(str(tmp_path), ann, is_train=True, config=cfg)
    pipe = mp.create_pipeline(ds, names, batch_size=3, is_train=True, config=cfg)
    assert len(pipe) == 2  # 7 samples, batch 3, drop_remainder
    order = ds.indices().tolist()
    assert sorted(order) == list(range(n_img)) and order != list(range(n_img))
    np.random.seed(11)
    batches = list(pipe)
    assert len(batches) == 2 and list(batches[0]) == ["image", "target", "target_weight"]
    assert batches[0]["image"].shape == (3, 3, 256, 192) and batches[0]["target"].shape == (3, 17, 64, 48)
    # the same samples, one by one, with the same draws: host transforms of the package + the oracle's crop / target
    from mindpose_amd.data.data_factory import _convert_names_to_transform
    ts = _convert_names_to_transform(names, is_train=True, config=cfg)
    np.random.seed(11)
    k = 0
    for b in batches:
        for j in range(3):
            rec = ds.source.record(order[k])
            k += 1
            img = np.load(rec["image_file"])
            st = dict(image=img, boxes=np.asarray(rec["boxes"], np.float32), keypoints=np.asarray(rec["keypoints"], np.float32),
                      rotation=np.float32(rec["rotation"]))
            for t in ts[:4]:
                st.update(t.transform(st))
            flipped = st["image"].strides[1] < 0
            src = np.ascontiguousarray(st["image"])
            crop, trans = ol.crop(src, st["center"], st["scale"], float(st["rotation"]), [192, 256])
            assert np.array_equal(b["image"][j].cpu().numpy(), crop), (k, flipped)
            kp = ts[4].transform_keypoints(st["keypoints"].copy(), ts[4].get_matrix(st["center"], st["scale"], st["rotation"]))
            tgt, wgt = otarget.generate_target(kp[None], (192, 256), (48, 64), sigma=2.0)
            assert np.array_equal(b["target"][j].cpu().numpy(), tgt[0]) and np.array_equal(b["target_weight"][j].cpu().numpy(), wgt[0])


def test_prefetched_batches_equal_the_synchronous_ones(tmp_path):
    """`create_pipeline(prefetch=2)` - the batches prepared ahead by a background thread on a side stream, one pinned upload per
    batch - against `prefetch=0`: same samples, same order of the np.random draws, so every column is bit-equal; JPEG payloads
    (PIL decode in the thread pool), several epochs, the consumer running kernels of its own between batches."""
    import json
    import os
    from PIL import Image
    rng = np.random.RandomState(8)
    cfg = dict(image_size=[192, 256], heatmap_size=[48, 64], pixel_std=200.0, scale_padding=1.25, upper_body_ids=list(range(11)),
               flip_pairs=[[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]], det_bbox_thr=0.0)
    images, anns = [], []
    for i in range(11):
        h, w = 120 + 8 * i, 160 + 4 * i
        Image.fromarray(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(os.path.join(tmp_path, f"{i}.jpg"), quality=92)
        images.append(dict(id=i + 1, file_name=f"{i}.jpg", width=w, height=h))
        kp = np.concatenate([rng.uniform(20, 100, (17, 2)), rng.randint(0, 3, (17, 1))], axis=1)
        kp[kp[:, 2] == 0, :2] = 0
        anns.append(dict(id=i + 1, image_id=i + 1, category_id=1, iscrowd=0, bbox=[15.0, 12.0, 100.0, 90.0], area=9000.0,
                         num_keypoints=int((kp[:, 2] > 0).sum()), keypoints=kp.reshape(-1).tolist()))
    ann = os.path.join(tmp_path, "train.json")
    with open(ann, "w") as f:
        json.dump(dict(images=images, annotations=anns, categories=[dict(id=1, name="person")]), f)
    names = ["topdown_box_to_center_scale", {"topdown_horizontal_random_flip": {"flip_prob": 0.5}}, "topdown_halfbody_transform",
             "topdown_randomscale_rotation", "topdown_affine", {"topdown_generate_target": {"sigma": 2.0}}]

    def epochs(prefetch):
        ds = mp.create_dataset(str(tmp_path), ann, is_train=True, config=cfg)
        pipe = mp.create_pipeline(ds, names, batch_size=4, is_train=True, num_workers=3, config=cfg, prefetch=prefetch)
        assert pipe.prefetch == prefetch
        np.random.seed(23)
        got = []
        busy = torch.randn(512, 512, device=DEV)
        for _ in range(3):
            for b in pipe:
                busy = busy @ busy * 1e-3  # the consumer's own work on its stream while the next batch is prepared
                got.append({k: v.clone() for k, v in b.items()})
        return got

    a, b = epochs(0), epochs(2)
    assert len(a) == len(b) == 6  # 11 samples, batch 4, drop_remainder, 3 epochs (a new shuffle each)
    for x, y in zip(a, b):
        assert list(x) == list(y) == ["image", "target", "target_weight"]
        for k in x:
            assert torch.equal(x[k], y[k]), k
    assert not torch.equal(a[0]["image"], a[2]["image"])  # epochs differ

    # an abandoned iteration (break) stops the producer thread
    import threading
    ds = mp.create_dataset(str(tmp_path), ann, is_train=True, config=cfg)
    pipe = mp.create_pipeline(ds, names, batch_size=2, is_train=True, config=cfg, prefetch=2)
    it = iter(pipe)
    next(it)
    it.close()
    assert not [t for t in threading.enumerate() if t.name == "mindpose-loader-prefetch" and t.is_alive()]
