"""COCO key-point OKS-AP without pycocotools.

Restates the ``COCOeval(cocoGt, cocoDt, "keypoints")`` protocol that the reference's evaluator drives
(mindpose/engine/evaluator/evaluator.py:133-160: evaluate / accumulate / summarize): OKS matching per image at IoU
thresholds 0.50:0.05:0.95, area ranges all / medium / large, maxDets 20, 101-point interpolated precision, the ten
summary numbers AP, AP .5, AP .75, AP (M), AP (L), AR, AR .5, AR .75, AR (M), AR (L).
PARITY UNPINNED: pycocotools is not installed here; the protocol is restated from knowledge of cocoeval.py and checked
with hand-computable cases (tests/test_evaluator_cpu.py).
"""
from collections import defaultdict
from typing import Any, Dict, List, Sequence

import numpy as np

KPT_OKS_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
AREA_RNG = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]  # all, medium, large
MAX_DETS = 20
STATS_NAMES = ["AP", "AP .5", "AP .75", "AP (M)", "AP (L)", "AR", "AR .5", "AR .75", "AR (M)", "AR (L)"]


def _prepare_dt(results: Sequence[Dict[str, Any]]) -> List[Dict[str, Any]]:
    """COCO.loadRes for key points: area / bbox from the key-point extent, running ids."""
    out = []
    for i, r in enumerate(results):
        d = dict(r)
        kp = np.asarray(d["keypoints"], dtype=np.float64)
        x, y = kp[0::3], kp[1::3]
        x0, x1, y0, y1 = np.min(x), np.max(x), np.min(y), np.max(y)
        d["area"] = float((x1 - x0) * (y1 - y0))
        d["bbox"] = [x0, y0, x1 - x0, y1 - y0]
        d["id"] = i + 1
        out.append(d)
    return out


def _oks_matrix(gts, dts, sigmas):
    if len(gts) == 0 or len(dts) == 0:
        return np.zeros((len(dts), len(gts)))
    ious = np.zeros((len(dts), len(gts)))
    vars_ = (sigmas * 2) ** 2
    k = len(sigmas)
    for j, gt in enumerate(gts):
        g = np.array(gt["keypoints"], dtype=np.float64)
        xg, yg, vg = g[0::3], g[1::3], g[2::3]
        k1 = np.count_nonzero(vg > 0)
        bb = gt["bbox"]
        x0, x1 = bb[0] - bb[2], bb[0] + bb[2] * 2
        y0, y1 = bb[1] - bb[3], bb[1] + bb[3] * 2
        for i, dt in enumerate(dts):
            d = np.array(dt["keypoints"], dtype=np.float64)
            xd, yd = d[0::3], d[1::3]
            if k1 > 0:
                dx, dy = xd - xg, yd - yg
            else:  # no labelled key point: distance to the doubled box
                z = np.zeros(k)
                dx = np.max((z, x0 - xd), axis=0) + np.max((z, xd - x1), axis=0)
                dy = np.max((z, y0 - yd), axis=0) + np.max((z, yd - y1), axis=0)
            e = (dx ** 2 + dy ** 2) / vars_ / (gt["area"] + np.spacing(1)) / 2
            if k1 > 0:
                e = e[vg > 0]
            ious[i, j] = np.sum(np.exp(-e)) / e.shape[0]
    return ious


def _evaluate_img(gts, dts, ious_full, a_rng):
    """One (image, area range): greedy matching per IoU threshold.  dts are already sorted by score and cut to MAX_DETS."""
    if len(gts) == 0 and len(dts) == 0:
        return None
    g_ignore = np.array([1 if (g["_ig"] or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gts], dtype=int)
    gtind = np.argsort(g_ignore, kind="mergesort")
    g_ignore = g_ignore[gtind]
    iscrowd = [int(gts[i].get("iscrowd", 0)) for i in gtind]
    ious = ious_full[:, gtind] if ious_full.size else ious_full
    T, G, D = len(IOU_THRS), len(gts), len(dts)
    gtm = np.zeros((T, G))
    dtm = np.zeros((T, D))
    dt_ig = np.zeros((T, D))
    if G and D:
        for ti, t in enumerate(IOU_THRS):
            for di in range(D):
                iou = min([t, 1 - 1e-10])
                m = -1
                for gi in range(G):
                    if gtm[ti, gi] > 0 and not iscrowd[gi]:
                        continue
                    if m > -1 and g_ignore[m] == 0 and g_ignore[gi] == 1:
                        break
                    if ious[di, gi] < iou:
                        continue
                    iou = ious[di, gi]
                    m = gi
                if m == -1:
                    continue
                dt_ig[ti, di] = g_ignore[m]
                dtm[ti, di] = gts[gtind[m]]["id"]
                gtm[ti, m] = dts[di]["id"]
    a = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dts]).reshape((1, D))
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
    return dict(dtm=dtm, dt_ig=dt_ig, g_ignore=g_ignore, scores=[d["score"] for d in dts])


def coco_keypoint_eval(gt_annotations: Sequence[Dict[str, Any]], results: Sequence[Dict[str, Any]], image_ids=None,
                       sigmas: np.ndarray = KPT_OKS_SIGMAS) -> List[float]:
    """gt_annotations: COCO ``annotations`` entries of ONE category (image_id, keypoints [3K], num_keypoints, area, bbox,
    iscrowd); results: entries with image_id, keypoints [3K], score.  Returns the ten summary numbers (-1 = undefined)."""
    gts_by_img, dts_by_img = defaultdict(list), defaultdict(list)
    for g in gt_annotations:
        g = dict(g)
        g["_ig"] = bool(g.get("ignore", 0)) or bool(g.get("iscrowd", 0)) or g.get("num_keypoints", 1) == 0
        gts_by_img[g["image_id"]].append(g)
    for d in _prepare_dt(results):
        dts_by_img[d["image_id"]].append(d)
    if image_ids is None:
        image_ids = sorted(set(gts_by_img) | set(dts_by_img))
    T, R, A = len(IOU_THRS), len(REC_THRS), len(AREA_RNG)
    precision = -np.ones((T, R, A))
    recall = -np.ones((T, A))
    per_img = {}
    for img in image_ids:
        dts = dts_by_img.get(img, [])
        order = np.argsort([-d["score"] for d in dts], kind="mergesort")
        dts = [dts[i] for i in order[:MAX_DETS]]
        per_img[img] = (gts_by_img.get(img, []), dts, _oks_matrix(gts_by_img.get(img, []), dts, sigmas))
    for ai, a_rng in enumerate(AREA_RNG):
        evs = [e for e in (_evaluate_img(*per_img[img], a_rng) for img in image_ids) if e is not None]
        if not evs:
            continue
        scores = np.concatenate([e["scores"] for e in evs])
        inds = np.argsort(-scores, kind="mergesort")
        dtm = np.concatenate([e["dtm"] for e in evs], axis=1)[:, inds]
        dt_ig = np.concatenate([e["dt_ig"] for e in evs], axis=1)[:, inds]
        g_ig = np.concatenate([e["g_ignore"] for e in evs])
        npig = np.count_nonzero(g_ig == 0)
        if npig == 0:
            continue
        tps = np.logical_and(dtm, np.logical_not(dt_ig))
        fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
        tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
        fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
        for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
            nd = len(tp)
            rc = tp / npig
            pr = tp / (fp + tp + np.spacing(1))
            q = np.zeros((R,))
            recall[t, ai] = rc[-1] if nd else 0
            pr = pr.tolist()
            for i in range(nd - 1, 0, -1):
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            idx = np.searchsorted(rc, REC_THRS, side="left")
            for ri, pi in enumerate(idx):
                if pi < nd:
                    q[ri] = pr[pi]
            precision[t, :, ai] = q

    def _ap(ai, t=None):
        s = precision[:, :, ai] if t is None else precision[t:t + 1, :, ai]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0

    def _ar(ai, t=None):
        s = recall[:, ai] if t is None else recall[t:t + 1, ai]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0

    t50, t75 = 0, int(np.where(np.isclose(IOU_THRS, 0.75))[0][0])
    return [_ap(0), _ap(0, t50), _ap(0, t75), _ap(1), _ap(2), _ar(0), _ar(0, t50), _ar(0, t75), _ar(1), _ar(2)]
