"""Metric primitives of the OpenTAL evaluation (reference AFSD/evaluation/utils_eval.py:20-133) plus the three
scikit-learn curve metrics the reference calls (eval_detection.py:459-491: average_precision_score, roc_auc_score,
roc_curve), restated on numpy so that the evaluation has no dependency beyond numpy.  CPU code, like the reference's:
the detections it consumes are what the HIP inference path (csrc/infer.hip) writes."""
import numpy as np


def interpolated_prec_rec(prec, rec):
    """Interpolated AP, VOC 2011 style (utils_eval.py:20-29)."""
    mprec = np.hstack([[0], prec, [0]])
    mrec = np.hstack([[0], rec, [1]])
    mprec = np.maximum.accumulate(mprec[::-1])[::-1]
    idx = np.where(mrec[1::] != mrec[0:-1])[0] + 1
    return np.sum((mrec[idx] - mrec[idx - 1]) * mprec[idx])


def segment_iou(target_segment, candidate_segments):
    """tIoU of one [start, end] against N candidates (utils_eval.py:32-58)."""
    tt1 = np.maximum(target_segment[0], candidate_segments[:, 0])
    tt2 = np.minimum(target_segment[1], candidate_segments[:, 1])
    inter = (tt2 - tt1).clip(0)
    union = (candidate_segments[:, 1] - candidate_segments[:, 0]) + (target_segment[1] - target_segment[0]) - inter
    return inter.astype(float) / union


def wrapper_segment_iou(target_segments, candidate_segments):
    """(n candidates) x (m targets) tIoU matrix (utils_eval.py:61-83)."""
    if candidate_segments.ndim != 2 or target_segments.ndim != 2:
        raise ValueError('Dimension of arguments is incorrect')
    return np.stack([segment_iou(t, candidate_segments) for t in target_segments], 1)


def open_set_detection_rate(preds, pred_cls, gt_cls):
    """Area under the correct-classification-rate / false-positive-rate curve (utils_eval.py:86-133).
    preds (N,) confidence of being known; pred_cls (N,) predicted class > 0; gt_cls (N,) 0 = unknown, > 0 known."""
    known = gt_cls > 0
    x1, x2 = preds[known], preds[gt_cls == 0]
    m_x1 = (pred_cls[known] == gt_cls[known]).astype(float)
    k_target = np.concatenate((m_x1, np.zeros(len(x2))))
    u_target = np.concatenate((np.zeros(len(x1)), np.ones(len(x2))))
    predict = np.concatenate((x1, x2))
    n = len(preds)
    idx = predict.argsort()
    s_k, s_u = k_target[idx], u_target[idx]
    CCR, FPR = np.zeros(n + 2), np.zeros(n + 2)
    if n > 1:
        # cut-off k (k = 0 .. n-2): correct knowns strictly above it, unknowns at or above it
        suffix_k = np.concatenate((np.cumsum(s_k[::-1])[::-1], [0.0]))
        suffix_u = np.cumsum(s_u[::-1])[::-1]
        CCR[:n - 1] = suffix_k[1:n] / float(len(x1)) if len(x1) > 0 else 1.0
        FPR[:n - 1] = suffix_u[:n - 1] / float(len(x2)) if len(x2) > 0 else 0.0
    CCR[n + 1], FPR[n + 1] = 1.0, 1.0
    order = np.lexsort((CCR, FPR))[::-1]            # descending by FPR, then CCR: sorted(zip(FPR, CCR), reverse=True)
    f, c = FPR[order], CCR[order]
    oscr = float(np.sum((f[:-1] - f[1:]) * (c[:-1] + c[1:]) / 2.0))
    return oscr, FPR.tolist(), CCR.tolist()


# ----------------------------------------------------------------------------- curve metrics (scikit-learn semantics)
def _binary_clf_curve(labels, scores):
    """False / true positive counts at every distinct score, highest first (sklearn.metrics._ranking)."""
    labels = np.asarray(labels) == 1
    scores = np.asarray(scores, dtype=np.float64)
    order = np.argsort(scores, kind="mergesort")[::-1]
    scores, labels = scores[order], labels[order]
    distinct = np.where(np.diff(scores))[0]
    ends = np.r_[distinct, labels.size - 1]
    tps = np.cumsum(labels, dtype=np.float64)[ends]
    fps = 1 + ends - tps
    return fps, tps, scores[ends]


def roc_curve(labels, scores, drop_intermediate=True):
    fps, tps, thr = _binary_clf_curve(labels, scores)
    if drop_intermediate and len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps, fps, thr = np.r_[0, tps], np.r_[0, fps], np.r_[np.inf, thr]
    fpr = fps / fps[-1] if fps[-1] > 0 else np.full_like(fps, np.nan)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.full_like(tps, np.nan)
    return fpr, tpr, thr


def roc_auc_score(labels, scores):
    fpr, tpr, _ = roc_curve(labels, scores)
    return float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) / 2.0))


def average_precision_score(labels, scores):
    """sum_n (R_n - R_{n-1}) P_n over the distinct-threshold precision/recall points."""
    fps, tps, _ = _binary_clf_curve(labels, scores)
    ps = tps + fps
    precision = np.divide(tps, ps, out=np.zeros_like(tps), where=ps != 0)
    recall = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps)
    # sklearn reverses the curve and appends (recall 0, precision 1); AP = -sum(diff(recall) * precision[:-1])
    precision = np.hstack((precision[::-1], [1.0]))
    recall = np.hstack((recall[::-1], [0.0]))
    return float(-np.sum(np.diff(recall) * precision[:-1]))
