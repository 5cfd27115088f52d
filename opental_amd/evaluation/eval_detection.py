"""Detection evaluation of OpenTAL: closed-set mAP and the open-set metrics AUROC / AUPR / FAR@95 / OSDR from a
result JSON (reference AFSD/evaluation/eval_detection.py: ANETdetection :26-320, compute_average_precision_detection
:323-402, split_results_by_gt :405-456, compute_auc_scores :459-491, compute_osdr_scores :494-510).  SURVEY 8f rank 3.

Same class name, constructor arguments and `evaluate(type=...)` results as the reference, on plain numpy arrays
instead of pandas frames + joblib (the reference walks data frames row by row; here a prediction row is an index into
parallel arrays).  Pinned against the imported reference by oracle/pin_evaluation.py -> tests/golden/eval_*.json.

Not restated: the Wilderness-Impact variant (`evaluate('WI')`) -- the reference's own drivers have it commented out
(thumos14/eval_open.py:77-80) -- and the plotting helpers (save_curve_data).
"""
import json

import numpy as np

from .utils_eval import (average_precision_score, interpolated_prec_rec, open_set_detection_rate, roc_auc_score,
                         roc_curve, segment_iou)


class ANETdetection(object):
    GROUND_TRUTH_FIELDS = ['database']
    PREDICTION_FIELDS = ['results', 'version', 'external_data']

    def __init__(self, ground_truth_filename=None, prediction_filename=None, cls_idx_detection=None,
                 ground_truth_fields=GROUND_TRUTH_FIELDS, prediction_fields=PREDICTION_FIELDS,
                 tiou_thresholds=np.linspace(0.5, 0.95, 10), ood_threshold=None, ood_scoring='confidence',
                 subset=['validation'], openset=False, draw_auc=False, curve_data_path=None, verbose=False,
                 check_status=False, dataset='thumos14'):
        if not ground_truth_filename:
            raise IOError('Please input a valid ground truth file.')
        if not prediction_filename:
            raise IOError('Please input a valid prediction file.')
        if check_status:
            raise NotImplementedError("check_status fetches a blocked-video list over the network")
        self.subset = subset
        self.tiou_thresholds = tiou_thresholds
        self.ood_threshold = ood_threshold
        self.ood_scoring = ood_scoring
        self.openset = openset
        self.draw_auc = draw_auc
        self.curve_data_path = curve_data_path
        self.verbose = verbose
        self.gt_fields = ground_truth_fields
        self.pred_fields = prediction_fields
        self.ap = None
        assert dataset in ['thumos14', 'anet', 'thumos_anet']
        self.dataset = dataset
        self.blocked_videos = list()
        self.activity_index = self.get_activity_index(cls_idx_detection)
        self.ground_truth, self.video_lst = self._import_ground_truth(ground_truth_filename)
        self.prediction = self._import_prediction(prediction_filename)
        if self.openset:
            self.stats = {}

    def get_activity_index(self, class_info_path):
        """eval_detection.py:87-99: class name -> 1..K; 0 is reserved for '__unknown__' in the open-set protocol."""
        class_to_idx = {}
        if self.openset:
            class_to_idx['__unknown__'] = 0
        with open(class_info_path, 'r') as f:
            lines = [l for l in f.read().splitlines() if l.strip()]
        for idx, line in enumerate(lines):
            name = line.split()[1] if self.dataset in ['thumos14', 'thumos_anet'] else line.strip()
            class_to_idx[name] = idx + 1
        return class_to_idx

    @staticmethod
    def _load_json(path, required, what):
        with open(path, 'r') as fobj:
            data = json.load(fobj)
        if any(field not in data for field in required):
            raise IOError('Please input a valid %s file.' % what)
        return data

    def _import_ground_truth(self, ground_truth_filename):
        """eval_detection.py:101-147 -> column arrays.  Every annotation of every video of the requested subsets is one row;
        a label outside the known classes is 0 ('__unknown__') in the open-set protocol and an error in the closed-set one."""
        database = self._load_json(ground_truth_filename, self.gt_fields, 'ground truth')['database']
        rows = [(vid, ann) for vid, v in database.items()
                if v['subset'] in self.subset and vid not in self.blocked_videos for ann in v['annotations']]
        if not self.openset:
            for _, ann in rows:
                assert ann['label'] in self.activity_index, 'Ground truth json contains invalid class: %s' % (ann['label'])
        video_lst = [vid for vid, _ in rows]
        seg = np.array([ann['segment'][:2] for _, ann in rows], dtype=np.float64).reshape(-1, 2)
        gt = {'video-id': np.array(video_lst, dtype=object), 't-start': seg[:, 0].copy(), 't-end': seg[:, 1].copy(),
              'label': np.array([self.activity_index.get(ann['label'], 0) for _, ann in rows], dtype=np.int64)}
        return gt, video_lst

    # out-of-distribution score of one detection under each scoring rule (eval_detection.py:183-196); a detection whose
    # score falls below ood_threshold is relabelled '__unknown__' in the open-set protocol
    OOD_SCORES = {
        'uncertainty': lambda r: r['uncertainty'],
        'confidence': lambda r: 1 - r['score'],
        'uncertainty_actionness': lambda r: r['uncertainty'] * r['actionness'],
        'a_by_inv_u': lambda r: r['actionness'] / (1 - r['uncertainty'] + 1e-6),
        'u_by_inv_a': lambda r: r['uncertainty'] / (1 - r['actionness'] + 1e-6),
        'half_au': lambda r: 0.5 * (r['actionness'] + 1) * r['uncertainty'],
    }

    def _import_prediction(self, prediction_filename):
        """eval_detection.py:149-219 -> column arrays: detections of videos without ground truth and detections of classes
        outside the index are dropped, in file order."""
        results = self._load_json(prediction_filename, self.pred_fields, 'prediction')['results']
        if self.ood_scoring not in self.OOD_SCORES:
            raise NotImplementedError(self.ood_scoring)
        ood_of = self.OOD_SCORES[self.ood_scoring]
        known_videos = set(self.video_lst) - set(self.blocked_videos)
        rows = [(vid, r) for vid, dets in results.items() if vid in known_videos
                for r in dets if r['label'] in self.activity_index]
        ood = np.array([ood_of(r) for _, r in rows], dtype=np.float64)
        label = np.array([self.activity_index[r['label']] for _, r in rows], dtype=np.int64)
        if self.openset and self.ood_threshold is not None:
            label[ood < self.ood_threshold] = self.activity_index['__unknown__']
        seg = np.array([r['segment'][:2] for _, r in rows], dtype=np.float64).reshape(-1, 2)
        return {'video-id': np.array([vid for vid, _ in rows], dtype=object), 't-start': seg[:, 0].copy(),
                't-end': seg[:, 1].copy(), 'label': label,
                'score': np.array([r['score'] for _, r in rows], dtype=np.float64), 'ood_score': ood}

    @staticmethod
    def _rows(table, mask):
        return {k: v[mask] for k, v in table.items()}

    def wrapper_compute_average_precision(self):
        """eval_detection.py:236-261.  One column per entry of activity_index; in the open-set protocol the
        '__unknown__' entry (index 0) lands in the LAST column (`ap[:, cidx - 1]` with cidx = 0), as in the reference."""
        ap = np.zeros((len(self.tiou_thresholds), len(self.activity_index)))
        for cidx in self.activity_index.values():
            gt = self._rows(self.ground_truth, self.ground_truth['label'] == cidx)
            if len(gt['label']) == 0:
                raise KeyError(cidx)        # the reference's groupby().get_group raises for a class without ground truth
            pred = self._rows(self.prediction, self.prediction['label'] == cidx)
            ap[:, cidx - 1] = compute_average_precision_detection(gt, pred, self.tiou_thresholds)
        return ap

    def pre_evaluate(self):
        unique_videos = sorted(set(self.video_lst))
        self.eval_data = split_results_by_gt(self.prediction, self.ground_truth, unique_videos, self.tiou_thresholds)

    def evaluate(self, type='AP'):
        if type == 'AP':
            self.ap = self.wrapper_compute_average_precision()
            self.mAP = self.ap.mean(axis=1)
            self.average_mAP = self.mAP.mean()
            return self.mAP, self.average_mAP, self.ap
        if type == 'AUC':
            pred_scores, _, gt_labels = self.eval_data
            self.au_roc, self.au_pr, self.far_95, _, _ = compute_auc_scores(pred_scores, gt_labels, self.tiou_thresholds)
            return self.au_roc, self.au_pr, self.far_95
        if type == 'OSDR':
            pred_scores, pred_labels, gt_labels = self.eval_data
            self.osdr, _ = compute_osdr_scores(pred_scores, pred_labels, gt_labels, self.tiou_thresholds)
            return self.osdr
        raise NotImplementedError(type)


def compute_average_precision_detection(ground_truth, prediction, tiou_thresholds=np.linspace(0.5, 0.95, 10)):
    """AP of one class at every tIoU threshold (eval_detection.py:323-402): predictions in decreasing score order, each
    matched to the not-yet-taken ground truth of its video with the highest tIoU above the threshold."""
    nthr = len(tiou_thresholds)
    ap = np.zeros(nthr)
    n = len(prediction['score'])
    if n == 0:
        return ap
    ngt = len(ground_truth['t-start'])
    npos = float(ngt)
    lock_gt = np.ones((nthr, ngt)) * -1
    sort_idx = prediction['score'].argsort()[::-1]
    vid, ps, pe = prediction['video-id'][sort_idx], prediction['t-start'][sort_idx], prediction['t-end'][sort_idx]
    tp = np.zeros((nthr, n))
    fp = np.zeros((nthr, n))
    by_video = {}
    for j, v in enumerate(ground_truth['video-id']):
        by_video.setdefault(v, []).append(j)
    gseg = np.stack([ground_truth['t-start'], ground_truth['t-end']], 1)
    for idx in range(n):
        rows = by_video.get(vid[idx])
        if rows is None:
            fp[:, idx] = 1
            continue
        rows = np.asarray(rows)
        tiou_arr = segment_iou(np.array([ps[idx], pe[idx]]), gseg[rows])
        order = tiou_arr.argsort()[::-1]
        for tidx, thr in enumerate(tiou_thresholds):
            for jdx in order:
                if tiou_arr[jdx] < thr:
                    fp[tidx, idx] = 1
                    break
                if lock_gt[tidx, rows[jdx]] >= 0:
                    continue
                tp[tidx, idx] = 1
                lock_gt[tidx, rows[jdx]] = idx
                break
            if fp[tidx, idx] == 0 and tp[tidx, idx] == 0:
                fp[tidx, idx] = 1
    tp_cumsum = np.cumsum(tp, axis=1).astype(float)
    fp_cumsum = np.cumsum(fp, axis=1).astype(float)
    recall = tp_cumsum / npos
    precision = tp_cumsum / (tp_cumsum + fp_cumsum)
    for tidx in range(nthr):
        ap[tidx] = interpolated_prec_rec(precision[tidx, :], recall[tidx, :])
    return ap


def split_results_by_gt(prediction_all, ground_truth_all, video_list, tiou_thresholds=np.linspace(0.5, 0.95, 10)):
    """Sort every prediction into background / known / unknown by the ground truth it matches (eval_detection.py:405-456);
    predictions are visited in file order within a video, and a ground truth can be taken once per threshold."""
    keys = ('bg', 'known', 'unknown')
    pred_scores = [{k: [] for k in keys} for _ in tiou_thresholds]
    pred_labels = [{k: [] for k in keys} for _ in tiou_thresholds]
    gt_labels = [{k: [] for k in keys} for _ in tiou_thresholds]
    gt_by_video, pred_by_video = {}, {}
    for j, v in enumerate(ground_truth_all['video-id']):
        gt_by_video.setdefault(v, []).append(j)
    for j, v in enumerate(prediction_all['video-id']):
        pred_by_video.setdefault(v, []).append(j)
    gseg = np.stack([ground_truth_all['t-start'], ground_truth_all['t-end']], 1)
    for video_name in video_list:
        prows = pred_by_video.get(video_name)
        if not prows:
            continue
        grows = np.asarray(gt_by_video[video_name])
        glab = ground_truth_all['label'][grows]
        lock_gt = np.ones((len(tiou_thresholds), len(grows))) * -1
        for idx, p in enumerate(prows):
            ood_score = prediction_all['ood_score'][p]
            label_pred = prediction_all['label'][p]
            tiou_arr = segment_iou(np.array([prediction_all['t-start'][p], prediction_all['t-end'][p]]), gseg[grows])
            order = tiou_arr.argsort()[::-1]
            for tidx, thr in enumerate(tiou_thresholds):
                for jdx in order:
                    if tiou_arr[jdx] < thr:
                        pred_scores[tidx]['bg'].append(ood_score)
                        pred_labels[tidx]['bg'].append(label_pred)
                        gt_labels[tidx]['bg'].append(-1.0)
                        break
                    if lock_gt[tidx, jdx] >= 0:
                        continue
                    label_gt = int(glab[jdx])
                    kind = 'unknown' if label_gt == 0 else 'known'
                    pred_scores[tidx][kind].append(ood_score)
                    pred_labels[tidx][kind].append(label_pred)
                    gt_labels[tidx][kind].append(label_gt)
                    lock_gt[tidx, jdx] = idx
                    break
    return pred_scores, pred_labels, gt_labels


def compute_auc_scores(pred_scores, gt_labels, tiou_thresholds=np.linspace(0.5, 0.95, 10), vis=False):
    """AUROC / AUPR / FAR@95 of known-vs-unknown among the matched foreground predictions (eval_detection.py:459-491);
    unknown is the positive class, the score is the out-of-distribution score."""
    auc_pr = np.zeros((len(tiou_thresholds),), dtype=np.float32)
    auc_roc = np.zeros((len(tiou_thresholds),), dtype=np.float32)
    far_95 = np.zeros((len(tiou_thresholds),), dtype=np.float32)
    for tidx in range(len(tiou_thresholds)):
        preds = pred_scores[tidx]['known'] + pred_scores[tidx]['unknown']
        labels_cls = gt_labels[tidx]['known'] + gt_labels[tidx]['unknown']
        labels = 1 - np.array(labels_cls).astype(bool).astype(int)
        if len(preds) > 0:
            auc_pr[tidx] = average_precision_score(labels, preds)
            auc_roc[tidx] = roc_auc_score(labels, preds) if len(set(labels.tolist())) > 1 else 0
            fpr, tpr, _ = roc_curve(labels, preds)
            far_95[tidx] = fpr[np.abs(tpr - 0.95).argmin()]
    return auc_roc, auc_pr, far_95, None, None


def compute_osdr_scores(pred_scores, pred_labels, gt_labels, tiou_thresholds=np.linspace(0.5, 0.95, 10), vis=False):
    """Open-set detection rate: area under the CCR-FPR curve (eval_detection.py:494-510)."""
    osdr = np.zeros((len(tiou_thresholds),), dtype=np.float32)
    for tidx in range(len(tiou_thresholds)):
        preds = 1 - np.array(pred_scores[tidx]['known'] + pred_scores[tidx]['unknown'])
        pred_cls = np.array(pred_labels[tidx]['known'] + pred_labels[tidx]['unknown'])
        gt_cls = np.array(gt_labels[tidx]['known'] + gt_labels[tidx]['unknown'])
        if len(preds) > 0:
            osdr[tidx], _, _ = open_set_detection_rate(preds, pred_cls, gt_cls)
    return osdr, None
