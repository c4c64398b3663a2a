"""Loader geometry on the CPU: the oracle and the host mirror against golden vectors produced by the
reference's own numpy code, plus known answers for the parts that need cv2 in the reference (parity unpinned there)."""
import os

import numpy as np
import pytest

import mindpose_amd as mp
from oracle import loader as ol
from tests.golden import recipes

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry.npz"))
CFG = dict(heatmap_size=[48, 64], flip_pairs=recipes.FLIP_PAIRS, upper_body_ids=list(range(11)), pixel_std=200.0,
           scale_padding=1.25)


@pytest.mark.parametrize("name,image_size", [("256x192", [192, 256]), ("384x288", [288, 384])])
def test_box_to_center_scale_bit_exact_vs_reference(name, image_size):
    boxes = G[f"boxes_{name}"]
    t = mp.TopDownBoxToCenterScale(is_train=False, config=dict(CFG, image_size=image_size))
    c, s = t.transform_batch(boxes)
    assert c.dtype == np.float32 and s.dtype == np.float32
    assert np.array_equal(c, G[f"center_{name}"]) and np.array_equal(s, G[f"scale_{name}"])
    for i, b in enumerate(boxes):
        oc, os_ = ol.xywh2cs(*b, np.array(image_size))
        assert np.array_equal(oc, G[f"center_{name}"][i]) and np.array_equal(os_, G[f"scale_{name}"][i])
        st = t.transform(dict(boxes=b))
        assert np.array_equal(st["center"], oc) and np.array_equal(st["scale"], os_)


@pytest.mark.parametrize("name,image_size", [("256x192", [192, 256]), ("384x288", [288, 384])])
def test_udp_warp_matrix_bit_exact_vs_reference(name, image_size):
    t = mp.TopDownAffine(is_train=False, config=dict(CFG, image_size=image_size), use_udp=True)
    for args, want in zip(G[f"warp_args_{name}"], G[f"warp_matrix_{name}"]):
        theta, center, scale = float(args[0]), args[1:3].astype(np.float32), args[3:5].astype(np.float32)
        size = np.array(image_size)
        assert np.array_equal(ol.get_warp_matrix(theta, center * 2.0, size - 1.0, scale * 200.0), want)
        assert np.array_equal(t.get_matrix(center, scale, theta), want)


def test_affine_transform_known_answers():
    # rot = 0: the map is a uniform scale dst_w / src_w about the box centre onto the crop centre (utils.py:44-103)
    center, scale = np.array([320.0, 240.0], np.float32), np.array([1.2, 1.6], np.float32)
    m = ol.get_affine_transform(center, scale, 0.0, np.array([192, 256]))
    k = 192.0 / (1.2 * 200.0)
    assert np.allclose(m, [[k, 0, 96 - k * 320], [0, k, 128 - k * 240]], atol=1e-9)
    t = mp.TopDownAffine(is_train=False, config=dict(CFG, image_size=[192, 256]))
    assert np.array_equal(t.get_matrix(center, scale, 0.0), m)
    # rot = 90 degrees: centre still maps to the crop centre, the linear part is k * R(-90deg) in image axes
    m = ol.get_affine_transform(center, scale, 90.0, np.array([192, 256]))
    assert np.allclose(m @ np.array([320.0, 240.0, 1.0]), [96.0, 128.0], atol=1e-4)
    assert np.allclose(m[:, :2], k * np.array([[0.0, 1.0], [-1.0, 0.0]]), atol=1e-6)
    # inv=True is the inverse map
    mi = ol.get_affine_transform(center, scale, 30.0, np.array([192, 256]), inv=True)
    mf = ol.get_affine_transform(center, scale, 30.0, np.array([192, 256]))
    full = np.vstack([mf, [0, 0, 1]]) @ np.vstack([mi, [0, 0, 1]])
    assert np.allclose(full, np.eye(3), atol=1e-5)
    assert np.array_equal(mp.get_affine_transform(center, scale, 30.0, np.array([192, 256]), inv=True), mi)


def test_warp_affine_oracle_properties():
    from scipy.ndimage import map_coordinates
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (60, 80, 3)).astype(np.uint8)
    # identity and integer translations are exact copies with a zero border
    assert np.array_equal(ol.warp_affine(img, [[1, 0, 0], [0, 1, 0]], 80, 60), img)
    sh = ol.warp_affine(img, [[1, 0, 5], [0, 1, -3]], 80, 60)
    assert np.array_equal(sh[:57, 5:], img[3:, :75]) and not sh[:, :5].any() and not sh[57:].any()
    # a half-pixel shift averages neighbours with round-half-up: (a + b + 1) >> 1
    hs = ol.warp_affine(img, [[1, 0, 0.5], [0, 1, 0]], 80, 60)
    want = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1
    assert np.array_equal(hs[:, 1:], want)
    # a general affine map agrees with float bilinear interpolation up to the 1/32-pixel coordinate quantisation
    smooth = (np.add.outer(np.arange(60) * 2.0, np.arange(80) * 1.5)[..., None] + np.array([0, 20, 40])).astype(np.uint8)
    m = np.array([[0.9, 0.2, 3.3], [-0.15, 1.1, 4.7]])
    got = ol.warp_affine(smooth, m, 64, 48).astype(np.float64)
    inv = np.linalg.inv(np.vstack([m, [0, 0, 1]]))
    ys, xs = np.mgrid[0:48, 0:64]
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    inside = (sx >= 0) & (sx <= 79) & (sy >= 0) & (sy <= 59)
    for c in range(3):
        ref = map_coordinates(smooth[..., c].astype(np.float64), [sy, sx], order=1, mode="constant")
        assert np.abs(got[..., c] - ref)[inside].max() <= 1.0
    # Normalize + HWC2CHW
    chw = ol.normalize_chw(img, [100.0, 110.0, 120.0], [50.0, 60.0, 70.0])
    assert chw.shape == (3, 60, 80) and chw.dtype == np.float32
    assert chw[1, 7, 9] == (np.float32(img[7, 9, 1]) - np.float32(110.0)) / np.float32(60.0)


def test_training_augmentations_bit_exact_vs_reference():
    """Half-body, random scale / rotation (same np.random draws in the same order) and fliplr_joints against seeded runs of
    the reference's own classes."""
    cfg = dict(CFG, image_size=[192, 256])
    kps, scales = G["aug_keypoints"], G["aug_scales"]
    hb = mp.TopDownHalfBodyTransform(is_train=True, config=cfg)
    rs = mp.TopDownRandomScaleRotation(is_train=True, config=cfg)
    np.random.seed(4321)
    hits = 0
    for i in range(30):
        o = hb.transform(dict(keypoints=kps[i].copy()))
        assert (1 if o else 0) == int(G["aug_halfbody_hit"][i])
        if o:
            hits += 1
            assert np.array_equal(o["center"], G["aug_halfbody_center"][i]) and np.array_equal(o["scale"], G["aug_halfbody_scale"][i])
        o = rs.transform(dict(scale=scales[i].copy()))
        assert np.array_equal(o["scale"], G["aug_rs_scale"][i]) and o["rotation"] == G["aug_rs_rotation"][i]
        assert o["scale"].dtype == np.float32 and np.asarray(o["rotation"]).dtype == np.float32
    assert 3 <= hits <= 20 and (G["aug_rs_rotation"] == 0).any() and (G["aug_rs_rotation"] != 0).any()
    fi = hb._transform_cfg["flip_index"]
    assert np.array_equal(mp.fliplr_joints(kps, 192, flip_index=fi), G["aug_fliplr_index"])
    assert np.array_equal(mp.fliplr_joints(kps, 192, flip_pairs=recipes.FLIP_PAIRS), G["aug_fliplr_pairs"])
    # the flip transform: label arithmetic as the reference (centre mirrored about the image WIDTH, key points about W - 1)
    fl = mp.TopDownHorizontalRandomFlip(is_train=True, config=cfg, flip_prob=1.0)
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    o = fl.transform(dict(image=img, keypoints=kps[0, :, :].copy(), center=np.array([2.0, 1.0], np.float32)))
    assert np.array_equal(o["image"], img[:, ::-1]) and o["center"][0] == 4.0
    assert np.array_equal(o["keypoints"], mp.fliplr_joints(kps[0], 6, flip_index=fi))
    assert mp.entrypoint("transform", "topdown_halfbody_transform") is mp.TopDownHalfBodyTransform


def test_point_helpers_match_reference_goldens():
    """affine_transform / rotate_point / warp_affine_joints / pad_to_same / transform_keypoints against outputs of the reference's
    own mindpose/data/transform/utils.py (tests/golden/helpers.npz, made by gen_golden.py helpers)."""
    import os
    from mindpose_amd.data.transform import utils as U
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "helpers.npz"))
    for p, m, want in zip(g["pts"], g["mats"], g["affine"]):
        assert np.array_equal(U.affine_transform(tuple(p), m), want)
    for p, a, want in zip(g["pts"], g["angles"], g["rotated"]):
        assert np.array_equal(np.array(U.rotate_point(tuple(p), a)), want)
    got = U.warp_affine_joints(g["joints"], g["mats"][0])
    assert got.dtype == g["warped_joints"].dtype and np.array_equal(got, g["warped_joints"])
    padded = U.pad_to_same([g[f"ragged_{i}"] for i in range(3)])
    for i, b in enumerate(padded):
        assert np.array_equal(b, g[f"padded_{i}"])
    res = U.transform_keypoints([g[f"tk_in_{i}"] for i in range(3)], g["tk_center"], g["tk_scale"], g["tk_heatmap_shape"])
    for i, b in enumerate(res):
        assert b.dtype == g[f"tk_out_{i}"].dtype and np.array_equal(b, g[f"tk_out_{i}"])
    with np.testing.assert_raises(AssertionError):
        U.affine_transform((1.0, 2.0, 3.0), g["mats"][0])


def test_codec_processes_decode_into_shared_memory_slots():
    """`_DecodeProcesses` (data/decode_worker.py as child processes): every payload comes back as a view of the shared block equal to
    the in-process decode - JPEG and .npy payloads, ragged sizes, several regions in flight - an image larger than a slot is handed
    back (None) for the caller to decode, a corrupt payload raises, and the block is unlinked on close."""
    import io
    import os
    from PIL import Image
    from mindpose_amd.data.data_factory import _DecodeProcesses, _decode
    rng = np.random.RandomState(0)
    payloads = []
    for i in range(21):
        im = rng.randint(0, 256, (60 + i, 80 + 2 * i, 3)).astype(np.uint8)
        b = io.BytesIO()
        if i % 3 == 0:
            np.save(b, im)
        else:
            Image.fromarray(im).save(b, format="JPEG", quality=90)
        payloads.append(np.frombuffer(b.getvalue(), np.uint8))
    b = io.BytesIO()
    np.save(b, rng.randint(0, 256, (400, 300, 3)).astype(np.uint8))  # 360 kB: larger than the 256 kB slots below
    payloads.append(np.frombuffer(b.getvalue(), np.uint8))
    codec = _DecodeProcesses(workers=3, batch=32, slot_bytes=256 << 10, regions=2)
    name = codec.shm.name
    try:
        tickets = [codec.submit(payloads), codec.submit(payloads[::-1])]  # two regions in flight
        for ticket, pay in zip(tickets, (payloads, payloads[::-1])):
            images = codec.collect(ticket)
            for p, im in zip(pay, images):
                ref = _decode(p)
                if ref.nbytes > codec.slot_bytes:
                    assert im is None
                else:
                    assert np.array_equal(im, ref)
        with pytest.raises(ValueError):
            codec.collect(codec.submit([np.frombuffer(b"not an image at all", np.uint8)]))
    finally:
        codec.close()
    assert not os.path.exists(os.path.join("/dev/shm", name.lstrip("/")))
