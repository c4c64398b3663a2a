"""OKS NMS for the top-down evaluator - host numpy, as in the reference.

Restates mindpose/utils/nms.py:7-190 (``oks_iou``, ``oks_nms``, ``soft_oks_nms``) with the per-detection Python loop of
``oks_iou`` vectorised over detections.  Pinned bit-exact (OKS values, keep indices and order) by golden vectors produced
by the reference's own module (tests/golden/nms.npz).  A person instance list is tiny (<= a few dozen boxes per image), so
this stays on the host.
"""
from typing import Any, Dict, List, Optional

import numpy as np

COCO_SIGMAS = np.array([0.26, 0.25, 0.25, 0.35, 0.35, 0.79, 0.79, 0.72, 0.72, 0.62, 0.62, 1.07, 1.07, 0.87, 0.87, 0.89, 0.89]) / 10.0


def oks_iou(g: np.ndarray, d: np.ndarray, a_g: float, a_d: np.ndarray, sigmas: Optional[np.ndarray] = None,
            vis_thr: Optional[float] = None) -> np.ndarray:
    """OKS between one flattened key-point vector ``g`` [3K] and detections ``d`` [N,3K] (nms.py:7-68).

    With ``vis_thr`` the reference masks with ``list(vg > thr) and list(vd > thr)`` - a Python ``and`` of two lists, i.e.
    the DETECTION's visibility mask alone; that behaviour is kept."""
    if sigmas is None:
        sigmas = COCO_SIGMAS
    d = np.asarray(d)
    ious = np.zeros(len(d), dtype=np.float32)
    if len(d) == 0:
        return ious
    key_vars = (sigmas * 2) ** 2
    xg, yg = g[0::3], g[1::3]
    dx = d[:, 0::3] - xg
    dy = d[:, 1::3] - yg
    e = (dx ** 2 + dy ** 2) / key_vars / ((a_g + np.asarray(a_d)[:, None]) / 2 + np.spacing(1)) / 2
    if vis_thr is None:
        ious[:] = np.sum(np.exp(-e), axis=1) / e.shape[1]
        return ious
    vis = d[:, 2::3] > vis_thr
    for n_d in range(len(d)):  # ragged selections: keep the reference's summation order per detection
        sel = e[n_d][vis[n_d]]
        ious[n_d] = np.sum(np.exp(-sel)) / len(sel) if sel.size != 0 else 0.0
    return ious


def _gather(kpts_db: List[Dict[str, Any]]):
    scores = np.array([k["score"] for k in kpts_db])
    kpts = np.array([np.asarray(k["keypoints"]).flatten() for k in kpts_db])
    areas = np.array([k["area"] for k in kpts_db])
    return scores, kpts, areas


def oks_nms(kpts_db: List[Dict[str, Any]], thr: float, sigmas: Optional[np.ndarray] = None,
            vis_thr: Optional[float] = None) -> np.ndarray:
    """Greedy OKS NMS: keep the best-scored instance, drop the ones whose OKS with it exceeds ``thr`` (nms.py:71-108)."""
    if not kpts_db:
        return []
    scores, kpts, areas = _gather(kpts_db)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        rest = order[1:]
        ovr = oks_iou(kpts[i], kpts[rest], areas[i], areas[rest], sigmas, vis_thr)
        order = rest[np.where(ovr <= thr)[0]]
    return np.array(keep)


def _rescore(overlap: np.ndarray, scores: np.ndarray, thr: float, key_type: str = "gaussian") -> np.ndarray:
    assert len(overlap) == len(scores)
    assert key_type in ["gaussian", "linear"]
    if key_type == "linear":
        inds = np.where(overlap >= thr)[0]
        scores[inds] = scores[inds] * (1 - overlap[inds])
    else:
        scores = scores * np.exp(-(overlap ** 2) / thr)
    return scores


def soft_oks_nms(kpts_db: List[Dict[str, Any]], thr: float, max_dets: int = 20, sigmas: Optional[np.ndarray] = None,
                 vis_thr: Optional[float] = None) -> np.ndarray:
    """Soft OKS NMS: Gaussian score decay instead of removal, at most ``max_dets`` kept (nms.py:139-190)."""
    if not kpts_db:
        return []
    scores, kpts, areas = _gather(kpts_db)
    order = scores.argsort()[::-1]
    scores = scores[order]
    keep = np.zeros(max_dets, dtype=np.intp)
    cnt = 0
    while order.size > 0 and cnt < max_dets:
        i = order[0]
        rest = order[1:]
        ovr = oks_iou(kpts[i], kpts[rest], areas[i], areas[rest], sigmas, vis_thr)
        scores = _rescore(ovr, scores[1:], thr)
        tmp = scores.argsort()[::-1]
        order = rest[tmp]
        scores = scores[tmp]
        keep[cnt] = i
        cnt += 1
    return keep[:cnt]
