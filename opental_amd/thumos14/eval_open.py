"""Evaluation driver with the reference's command line (AFSD/thumos14/eval_open.py; the closed-set branch is what
AFSD/thumos14/eval.py does for one split): per split, closed-set mAP or -- with --open_set -- AUROC / AUPR / FAR@95 /
OSDR of the result JSON written by thumos14/test.py, the per-split text files `eval.txt` / `eval_open.txt` in the
reference's format, and mean +- 1.96 sigma / sqrt(n) over the splits.

    python -m opental_amd.thumos14.eval_open output/opental/split_{id:d}/detection_results.json \\
        datasets/thumos14/annotations_open/split_{id:d}/known_gt.json --cls_idx_known .../action_known.txt \\
        --all_splits 0 1 2 --open_set --ood_scoring uncertainty_actionness
"""
import argparse
import os

import numpy as np

from ..evaluation.eval_detection import ANETdetection


def write_eval_open(eval_file, tious, far_95, auc_ROC, auc_PR, OSDR):
    with open(eval_file, 'w') as f:
        for (tiou, far, auc_roc, auc_pr, osdr) in zip(tious, far_95, auc_ROC, auc_PR, OSDR):
            f.writelines(f"tIoU={tiou}: far@95={far:.5f}, auc_roc={auc_roc:.5f}, auc_pr={auc_pr:.5f}, osdr={osdr:.5f}\n")
        f.writelines(f"Average FAR@95: {far_95.mean():.5f}, Average AUC_ROC: {auc_ROC.mean():.5f}, "
                     f"Average AUC_PR: {auc_PR.mean():.5f}, Average OSDR: {OSDR.mean():.5f}\n")


def write_eval_closed(eval_file, tious, mAPs, average_mAP):
    with open(eval_file, 'w') as f:
        for (tiou, mAP) in zip(tious, mAPs):
            f.writelines(f"tIoU={tiou}: mAP={mAP:.5f}\n")
        f.writelines(f"Average mAP: {average_mAP:.5f}\n")


def get_mean_std(data, axis=0):
    """Mean and 95 % confidence half-width over splits (eval_open.py:104-108)."""
    mean = np.array(data).mean(axis=axis)
    std = np.array(data).std(axis=axis) / np.sqrt(len(data)) * 1.96
    return mean, std


def evaluate_split(pred_file, gt_file, cls_idx_known, tious, subset, open_set, ood_scoring='confidence', dataset='thumos14',
                   write=True):
    det = ANETdetection(ground_truth_filename=gt_file, prediction_filename=pred_file, cls_idx_detection=cls_idx_known,
                        subset=subset, openset=open_set, ood_scoring=ood_scoring, tiou_thresholds=tious, dataset=dataset)
    if open_set:
        det.pre_evaluate()
        auc_ROC, auc_PR, far_95 = det.evaluate(type='AUC')
        OSDR = det.evaluate(type='OSDR')
        if write:
            write_eval_open(os.path.join(os.path.dirname(pred_file), 'eval_open.txt'), tious, far_95, auc_ROC, auc_PR, OSDR)
        return {'far_95': far_95, 'auc_roc': auc_ROC, 'auc_pr': auc_PR, 'osdr': OSDR}
    mAPs, average_mAP, _ = det.evaluate(type='AP')
    if write:
        write_eval_closed(os.path.join(os.path.dirname(pred_file), 'eval.txt'), tious, mAPs, average_mAP)
    return {'mAP': mAPs, 'average_mAP': average_mAP}


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('output_json', type=str)
    parser.add_argument('gt_json', type=str, default='datasets/thumos14/annotations/thumos_gt.json', nargs='?')
    parser.add_argument('--cls_idx_known', type=str)
    parser.add_argument('--all_splits', nargs='+', type=int)
    parser.add_argument('--open_set', action='store_true')
    parser.add_argument('--dataset', type=str, default='thumos14', choices=['thumos14', 'thumos_anet'])
    parser.add_argument('--ood_scoring', type=str, default='confidence',
                        choices=['uncertainty', 'confidence', 'uncertainty_actionness', 'a_by_inv_u', 'u_by_inv_a', 'half_au'])
    args = parser.parse_args(argv)
    tious = np.linspace(0.5, 0.95, 10) if args.dataset == 'thumos_anet' else [0.3, 0.4, 0.5, 0.6, 0.7]
    subset = ['test', 'validation'] if args.dataset == 'thumos_anet' else ['test']
    per_split = []
    for split in args.all_splits:
        gt_file = args.gt_json if args.open_set else args.gt_json.format(id=split)
        per_split.append(evaluate_split(args.output_json.format(id=split), gt_file, args.cls_idx_known.format(id=split),
                                        tious, subset, args.open_set, args.ood_scoring, args.dataset))
    names = (('far_95', 'FAR@95'), ('auc_roc', 'AUC_ROC'), ('auc_pr', 'AUC_PR'), ('osdr', 'OSDR')) if args.open_set else (('mAP', 'mAP'),)
    for key, title in names:
        mean, std = get_mean_std([r[key] for r in per_split])
        avg_mean, avg_std = get_mean_std([np.mean(r[key]) for r in per_split])
        for tiou, m, s in zip(tious, mean, std):
            print(f"{title}(tIoU={tiou}): mean={m:.5f}, std={s:.5f}")
        print(f"Average {title} = {avg_mean:.5f} ({avg_std:.5f})\n")
    return per_split


if __name__ == '__main__':
    main()
