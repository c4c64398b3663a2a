"""OKS NMS against golden vectors produced by the reference's own mindpose/utils/nms.py: OKS values, keep
indices and their order are bit-exact."""
import os

import numpy as np

from mindpose_amd.utils import nms

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nms.npz"))


def _people(ci):
    kps, sc, ar = G[f"c{ci}_keypoints"], G[f"c{ci}_score"], G[f"c{ci}_area"]
    return [dict(keypoints=kps[i], score=float(sc[i]), area=float(ar[i])) for i in range(len(sc))]


def test_oks_nms_matches_reference():
    n_cases = len([k for k in G.files if k.endswith("_score")])
    assert n_cases == 5
    for ci in range(n_cases):
        people = _people(ci)
        kpts = np.array([p["keypoints"].flatten() for p in people])
        areas = G[f"c{ci}_area"]
        got = nms.oks_iou(kpts[0], kpts[1:], areas[0], areas[1:])
        assert got.dtype == np.float32 and np.array_equal(got, G[f"c{ci}_iou0"])
        assert np.array_equal(nms.oks_iou(kpts[0], kpts[1:], areas[0], areas[1:], None, 0.4), G[f"c{ci}_iou0_vis"])
        for tag, thr in (("05", 0.5), ("09", 0.9)):
            assert np.array_equal(np.asarray(nms.oks_nms(people, thr), dtype=np.int64), G[f"c{ci}_keep_{tag}"])
            assert np.array_equal(np.asarray(nms.oks_nms(people, thr, None, 0.4), dtype=np.int64), G[f"c{ci}_keep_vis_{tag}"])
            assert np.array_equal(np.asarray(nms.soft_oks_nms(people, thr), dtype=np.int64), G[f"c{ci}_soft_{tag}"])
    # the interesting cases really suppress something, and the empty list is handled like the reference
    assert len(G["c3_keep_05"]) < len(G["c3_score"]) and len(G["c4_soft_05"]) == 20
    assert nms.oks_nms([], 0.9) == [] and nms.soft_oks_nms([], 0.9) == []
