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


def _columns(kpts_db: List[Dict[str, Any]]):
    """Person list -> (scores [N], flattened key points [N, 3K], areas [N])."""
    score = np.array([person["score"] for person in kpts_db])
    flat = np.array([np.asarray(person["keypoints"]).flatten() for person in kpts_db])
    area = np.array([person["area"] for person in kpts_db])
    return score, flat, area


def oks_nms(kpts_db: List[Dict[str, Any]], thr: float, sigmas: Optional[np.ndarray] = None,
            vis_thr: Optional[float] = None) -> np.ndarray:
    """Greedy OKS NMS (nms.py:71-108): walk the instances by descending score, keep the head of the queue and drop every
    queued instance whose OKS with it exceeds ``thr``.  Returns the kept indices in that order."""
    if not kpts_db:
        return []
    score, flat, area = _columns(kpts_db)
    queue = score.argsort()[::-1]
    kept = []
    while queue.size > 0:
        head, queue = queue[0], queue[1:]
        kept.append(head)
        overlap = oks_iou(flat[head], flat[queue], area[head], area[queue], sigmas, vis_thr)
        queue = queue[np.where(overlap <= thr)[0]]
    return np.array(kept)


def _decay(overlap: np.ndarray, scores: np.ndarray, thr: float, key_type: str = "gaussian") -> np.ndarray:
    """Soft-NMS score decay (nms.py:111-136): gaussian exp(-oks^2 / thr), or linear (1 - oks) above the threshold."""
    assert len(overlap) == len(scores)
    assert key_type in ["gaussian", "linear"]
    if key_type == "gaussian":
        return scores * np.exp(-(overlap ** 2) / thr)
    hit = np.where(overlap >= thr)[0]
    scores[hit] = scores[hit] * (1 - overlap[hit])
    return scores


def soft_oks_nms(kpts_db: List[Dict[str, Any]], thr: float, max_dets: int = 20, sigmas: Optional[np.ndarray] = None,
                 vis_thr: Optional[float] = None) -> np.ndarray:
    """Soft OKS NMS (nms.py:139-190): nothing is removed; after each pick the remaining scores decay with their OKS to the
    pick and the queue is re-sorted; at most ``max_dets`` picks."""
    if not kpts_db:
        return []
    score, flat, area = _columns(kpts_db)
    queue = score.argsort()[::-1]
    live = score[queue]
    picked = np.zeros(max_dets, dtype=np.intp)
    n = 0
    while queue.size > 0 and n < max_dets:
        head, queue = queue[0], queue[1:]
        overlap = oks_iou(flat[head], flat[queue], area[head], area[queue], sigmas, vis_thr)
        live = _decay(overlap, live[1:], thr)
        resort = live.argsort()[::-1]
        queue, live = queue[resort], live[resort]
        picked[n] = head
        n += 1
    return picked[:n]
