"""oracle/pin_evaluation.py -- TEST INFRASTRUCTURE, runs in the BUILD container only (needs /root/reference).

Pins opental_amd/evaluation (closed-set mAP; open-set AUROC / AUPR / FAR@95 / OSDR) against the reference's own
AFSD/evaluation/eval_detection.ANETdetection, imported from /root/reference and run on a seeded synthetic ground-truth /
detection pair, and commits that pair plus the reference's results as fixtures:

    tests/golden/eval_classes.txt, eval_gt_open.json, eval_gt_closed.json, eval_pred.json   (inputs)
    tests/golden/eval_expected.json                                                          (reference outputs)

    python -m oracle.pin_evaluation

The reference uses `np.float` (removed in numpy 1.24; the reference pins numpy of 2021): the alias is restored in THIS
process only, before its module is imported.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"       # child processes too (joblib workers of the evaluation code)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import numpy as np

TIOUS = [0.3, 0.4, 0.5, 0.6, 0.7]


def synth(seed=7, n_videos=20, n_classes=20, n_known=15):
    rs = np.random.RandomState(seed)
    names = [f"Action{c:02d}" for c in range(n_classes)]
    known = names[:n_known]
    database, results = {}, {}
    for v in range(n_videos):
        vid = f"video_test_{v:07d}"
        dur = float(rs.uniform(60, 240))
        anns = []
        for _ in range(rs.randint(2, 8)):
            length = rs.uniform(2.0, 25.0)
            start = rs.uniform(0, dur - length)
            anns.append({"label": names[rs.randint(0, n_classes)], "segment": [round(float(start), 3), round(float(start + length), 3)]})
        database[vid] = {"subset": "test", "duration": dur, "annotations": anns}
        dets = []
        for a in anns:
            s, e = a["segment"]
            is_known = a["label"] in known
            for _ in range(rs.randint(1, 5)):
                jit = (e - s) * rs.uniform(0.0, 0.45)
                ds, de = s + rs.normal(0, jit), e + rs.normal(0, jit)
                if de <= ds + 0.2:
                    de = ds + 0.2
                if is_known and rs.rand() < 0.8:
                    lab = a["label"]
                else:
                    lab = known[rs.randint(0, n_known)]
                unct = float(np.clip(rs.beta(2, 5) if is_known else rs.beta(4, 3), 1e-4, 1 - 1e-4))
                dets.append({"label": lab, "score": float(np.clip(rs.beta(5, 2) * (1 - 0.5 * unct), 1e-4, 1.0)),
                             "segment": [float(ds), float(de)], "uncertainty": unct, "actionness": float(rs.beta(5, 2))})
        for _ in range(rs.randint(20, 40)):         # background detections
            length = rs.uniform(1.0, 20.0)
            start = rs.uniform(0, dur - length)
            dets.append({"label": known[rs.randint(0, n_known)], "score": float(rs.beta(1.5, 6)),
                         "segment": [float(start), float(start + length)], "uncertainty": float(rs.beta(3, 3)),
                         "actionness": float(rs.beta(2, 4))})
        order = rs.permutation(len(dets))
        results[vid] = [dets[i] for i in order]
    # a video with detections but without annotations in the subset, and a detection of a class outside the list
    results["video_validation_0000001"] = [{"label": known[0], "score": 0.5, "segment": [1.0, 2.0], "uncertainty": 0.5, "actionness": 0.5}]
    results[next(iter(database))].append({"label": "NotAClass", "score": 0.9, "segment": [1.0, 5.0], "uncertainty": 0.1, "actionness": 0.9})
    closed = {vid: dict(v, annotations=[a for a in v["annotations"] if a["label"] in known]) for vid, v in database.items()}
    classes = "".join(f"{7 + 3 * i} {n}\n" for i, n in enumerate(known))
    return classes, {"database": database}, {"database": closed}, {"version": "synthetic", "results": results, "external_data": {}}


def main():
    os.makedirs(GOLD, exist_ok=True)
    classes, gt_open, gt_closed, pred = synth()
    paths = {k: os.path.join(GOLD, k) for k in ("eval_classes.txt", "eval_gt_open.json", "eval_gt_closed.json", "eval_pred.json")}
    open(paths["eval_classes.txt"], "w").write(classes)
    for k, obj in (("eval_gt_open.json", gt_open), ("eval_gt_closed.json", gt_closed), ("eval_pred.json", pred)):
        json.dump(obj, open(paths[k], "w"))

    np.float = float                      # see the module docstring
    sys.path.insert(0, REF)
    from AFSD.evaluation.eval_detection import ANETdetection as RefDet
    from opental_amd.evaluation.eval_detection import ANETdetection as OurDet

    def both(**kw):
        return RefDet(**kw), OurDet(**kw)

    expected, report = {}, []
    common = dict(prediction_filename=paths["eval_pred.json"], cls_idx_detection=paths["eval_classes.txt"],
                  subset=["test"], tiou_thresholds=TIOUS, dataset="thumos14")
    # closed set: mAP (thumos14/eval_open.py:95-101)
    ref, our = both(ground_truth_filename=paths["eval_gt_closed.json"], openset=False, **common)
    import joblib
    with joblib.parallel_backend("threading"):       # worker PROCESSES would not see the np.float alias
        r_map, r_avg, r_ap = ref.evaluate(type="AP")
    o_map, o_avg, o_ap = our.evaluate(type="AP")
    assert np.abs(r_ap - o_ap).max() < 1e-12, np.abs(r_ap - o_ap).max()
    expected["closed"] = {"mAP": r_map.tolist(), "average_mAP": float(r_avg), "ap": r_ap.tolist()}
    report.append(f"evaluation: closed-set AP matrix {r_ap.shape} max|ref-ours| = {np.abs(r_ap - o_ap).max():.1e}; mAP {[round(float(x), 4) for x in r_map]}")
    # open set, every OOD scoring rule: AUROC / AUPR / FAR@95 / OSDR (thumos14/eval_open.py:71-93)
    expected["open"] = {}
    for scoring in ("uncertainty", "confidence", "uncertainty_actionness", "a_by_inv_u", "u_by_inv_a", "half_au"):
        ref, our = both(ground_truth_filename=paths["eval_gt_open.json"], openset=True, ood_scoring=scoring, **common)
        ref.pre_evaluate(); our.pre_evaluate()
        r = list(ref.evaluate(type="AUC")) + [ref.evaluate(type="OSDR")]
        o = list(our.evaluate(type="AUC")) + [our.evaluate(type="OSDR")]
        d = max(float(np.abs(a - b).max()) for a, b in zip(r, o))
        assert d < 1e-6, (scoring, d)
        n_fg = [len(ref.eval_data[0][t]["known"]) + len(ref.eval_data[0][t]["unknown"]) for t in range(len(TIOUS))]
        expected["open"][scoring] = {"auc_roc": r[0].tolist(), "auc_pr": r[1].tolist(), "far_95": r[2].tolist(),
                                     "osdr": r[3].tolist(), "matched_foreground": n_fg}
        report.append(f"evaluation: open-set {scoring}: max|ref-ours| = {d:.1e}; AUROC {[round(float(x), 4) for x in r[0]]} OSDR {[round(float(x), 4) for x in r[3]]}")
    # open set with a rejection threshold: AP including the '__unknown__' column
    ref, our = both(ground_truth_filename=paths["eval_gt_open.json"], openset=True, ood_scoring="uncertainty",
                    ood_threshold=0.3, **common)
    with joblib.parallel_backend("threading"):
        r_map, r_avg, r_ap = ref.evaluate(type="AP")
    o_map, o_avg, o_ap = our.evaluate(type="AP")
    assert np.abs(r_ap - o_ap).max() < 1e-12
    expected["open_ap_threshold_0.3"] = {"mAP": r_map.tolist(), "ap_unknown_column": r_ap[:, -1].tolist()}
    report.append(f"evaluation: open-set AP with ood_threshold 0.3: max|ref-ours| = {np.abs(r_ap - o_ap).max():.1e}")
    json.dump(expected, open(os.path.join(GOLD, "eval_expected.json"), "w"), indent=1)
    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
