"""oracle/pin_anet_inference.py -- TEST INFRASTRUCTURE, runs in the BUILD container only (needs /root/reference).

Expected outputs of the ActivityNet inference post-processing -- decode_prediction, filtering, softnms_v2,
get_video_prediction of the reference's AFSD/anet/test.py (:97-201), imported from /root/reference -- on seeded synthetic
head outputs (`synthetic_heads`, regenerated identically by tests/test_anet_gpu.py), one 768-frame clip per video as in
anet/test.py:71-80.  Writes tests/golden/anet_inference.npz (the proposal lists).

    python -m oracle.pin_anet_inference
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"       # child processes too (joblib workers of the evaluation code)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import numpy as np
import torch

from oracle import afsd_oracle as O
from oracle import arch

VIDEOS = (("v_aaa", 11, 25.0, 29.7, -2.0), ("v_bbb", 12, 6.0, 120.0, -3.0), ("v_ccc", 13, 12.5, 61.44, -4.0))   # name, seed, fps, duration (s), mean quality logit


def synthetic_heads(seed, center_mean):
    """One clip's head outputs (batch 1) with realistic ranges: loc in frames, logits N(0,2)."""
    rs = np.random.RandomState(seed)
    cfg = arch.ANET
    pri = O.priors_all(cfg)
    K = pri.shape[0]
    stride = np.array([cfg["fpn_strides"][int(l)] for l in pri[:, 1]], np.float32)
    f = lambda a: torch.from_numpy(a.astype(np.float32))
    return {"priors": pri,
            "loc": f(rs.uniform(0.5, 6.0, (1, K, 2)) * stride[None, :, None]),
            "conf": f(rs.normal(0, 2.0, (1, K, cfg["num_classes"]))),
            "prop_loc": f(rs.normal(0, 0.3, (1, K, 2))),
            "prop_conf": f(rs.normal(0, 2.0, (1, K, cfg["num_classes"]))),
            "center": f(rs.normal(center_mean, 1.0, (1, K, 1))),
            "act": f(rs.normal(0.5, 1.0, (1, K, 1))),
            "prop_act": f(rs.normal(0.5, 1.0, (1, K, 1)))}


def main():
    sys.path.insert(0, REF)
    sys.argv = ["pin", os.path.join(REF, "configs/anet_opental.yaml"), "--open_set", "--split", "0"]
    fake = types.ModuleType("boundary_max_pooling_cuda")
    fake.forward = fake.backward = None
    sys.modules["boundary_max_pooling_cuda"] = fake
    torch.Tensor.cuda = lambda self, *a, **k: self
    import AFSD.anet.test as ref
    from AFSD.anet.BDNet import DirichletLayer

    class cfg:
        pass
    cfg.num_classes, cfg.clip_length, cfg.use_edl, cfg.os_head = 150, 768, True, True
    cfg.top_k, cfg.nms_sigma = 5000, 0.85                          # configs/anet_opental.yaml: testing
    cfg.idx_to_class = {i + 1: f"class_{i:03d}" for i in range(150)}
    out_layer = DirichletLayer(evidence="exp", dim=-1)
    expected = {}
    for name, seed, fps, duration, cm in VIDEOS:
        od = synthetic_heads(seed, cm)
        od["unct"] = out_layer.compute_uncertainty(od["conf"])
        od["prop_unct"] = out_layer.compute_uncertainty(od["prop_conf"])
        seg, scores, unct, actn = ref.decode_prediction(od, cfg, out_layer)
        output = [[] for _ in range(cfg.num_classes)]
        for cl in range(cfg.num_classes):
            rows = ref.filtering(seg, scores[cl], unct, actn, 0, fps, cfg)
            if rows is not None:
                output[cl].append(rows)
        props = ref.get_video_prediction(output, duration, cfg, cls_rng=range(cfg.num_classes))
        names = {v: k for k, v in cfg.idx_to_class.items()}
        expected[name + "_class"] = np.array([names[p["label"]] for p in props], np.int16)
        expected[name + "_rows"] = np.array([[p["segment"][0], p["segment"][1], p["score"], p["uncertainty"], p["actionness"]]
                                             for p in props], np.float64).reshape(-1, 5)
        expected[name + "_meta"] = np.array([seed, fps, duration, cm], np.float64)
        print(f"anet inference: {name}: {sum(len(o) > 0 for o in output)} classes with candidates, {len(props)} proposals")
    np.savez_compressed(os.path.join(GOLD, "anet_inference.npz"), **expected)
    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
        f.write("anet inference: reference decode_prediction / filtering / softnms_v2 / get_video_prediction on seeded heads -> "
                + ", ".join(f"{k[:-5]}: {len(v)} proposals" for k, v in expected.items() if k.endswith("_rows")) + "\n")


if __name__ == "__main__":
    main()
