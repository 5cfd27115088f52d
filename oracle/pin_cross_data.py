"""oracle/pin_cross_data.py -- TEST INFRASTRUCTURE, runs in the BUILD container only (needs /root/reference).

Pins SURVEY 8(f) rank 4 -- the threshold step and the cross-dataset inference driver -- against the reference's OWN
functions, run end to end with a stand-in detector (oracle/fake_heads.FakeNet: head outputs are a function of the window
content) on seeded synthetic videos:
  * AFSD/thumos14/test_cross_data.py: get_offsets :49-56, prepare_data / prepare_anet_clip :59-89, test_anet :278-331
    (decode, filtering, Soft-NMS, get_video_detections with duration clipping :178-215), exclude_overlapping :333-352,
    and the merge of its __main__ :420-441;
  * AFSD/thumos14/threshold.py: thresholding :71-150 for all six OOD scoring rules.
Writes tests/golden/cross_data.npz (expected proposals, thresholds); tests/test_cross_data.py replays the same videos
through opental_amd on the GPU.

    python -m oracle.pin_cross_data
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"       # child processes too
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import numpy as np
import torch

from oracle import fake_heads as FH

IDX_TO_CLASS = {i + 1: f"class_{i:02d}" for i in range(15)}
SCORINGS = ("uncertainty", "confidence", "uncertainty_actionness", "a_by_inv_u", "u_by_inv_a", "half_au")


def rows_of(props):
    names = {v: k for k, v in IDX_TO_CLASS.items()}
    return (np.array([names[p["label"]] for p in props], np.int16),
            np.array([[p["segment"][0], p["segment"][1], p["score"], p["uncertainty"], p["actionness"]] for p in props],
                     np.float64).reshape(-1, 5))


def main():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "AFSD", "thumos14"))       # threshold.py does `from test import ...`
    sys.argv = ["pin", os.path.join(REF, "configs/thumos14_opental_final.yaml"), "--open_set", "--split", "0"]
    fake = types.ModuleType("boundary_max_pooling_cuda")
    fake.forward = fake.backward = None
    sys.modules["boundary_max_pooling_cuda"] = fake
    torch.Tensor.cuda = lambda self, *a, **k: self
    import AFSD.thumos14.test_cross_data as X
    import threshold as TH                                           # AFSD/thumos14/threshold.py
    from AFSD.common.config import config

    tmp = tempfile.mkdtemp(prefix="pin_cross_")
    fx = {}
    # ---- get_offsets / prepare_anet_clip
    for n in (100, 256, 300, 384, 640, 700):
        fx[f"offsets_{n}"] = np.array(X.get_offsets(n, 256, 128), np.int64)
    data = torch.from_numpy(np.transpose(FH.synthetic_video(5, 300), [3, 0, 1, 2]).copy())
    clip = X.prepare_anet_clip(data, 256, 256, 96)
    fx["anet_clip_tail_sum"] = np.float64(clip.double().sum())
    fx["anet_clip_pad_absmax"] = np.float64(clip[0, :, 44:].abs().max())

    # ---- test_anet end to end (stand-in detector), duration clipping, exclude_overlapping, merge
    anet_dir = os.path.join(tmp, "anet")
    os.makedirs(anet_dir)
    infos = {}
    for name, seed, frames, fps, duration, labels in FH.ANET_VIDEOS:
        np.save(os.path.join(anet_dir, name + ".npy"), FH.synthetic_video(seed, frames))
        infos[name] = {"fps": fps, "duration": duration, "frame_num": frames, "annotations": [{"label": l} for l in labels]}

    class thumos_cfg:
        pass
    thumos_cfg.fusion, thumos_cfg.use_edl, thumos_cfg.use_rpl, thumos_cfg.evidence = False, True, False, "exp"
    thumos_cfg.crop_size, thumos_cfg.num_classes, thumos_cfg.os_head = 96, 15, True
    thumos_cfg.clip_length, thumos_cfg.stride, thumos_cfg.conf_thresh = 256, 128, 0.01
    thumos_cfg.top_k, thumos_cfg.nms_sigma, thumos_cfg.idx_to_class = 5000, 0.5, IDX_TO_CLASS

    class anet_cfg:
        pass
    anet_cfg.video_infos, anet_cfg.mp4_data_path = infos, anet_dir
    net = FH.FakeNet()
    X.build_model = lambda **kw: (net, None)
    X.tqdm = lambda it, **kw: it
    anet_out = X.test_anet(thumos_cfg, anet_cfg, os.path.join(tmp, "anet_open_rgb.json"))
    for name, props in anet_out["results"].items():
        cls, rows = rows_of(props)
        fx[f"anet_{name}_class"], fx[f"anet_{name}_rows"] = cls, rows
        print(f"test_anet: {name}: {len(props)} proposals")
    fx["anet_result_keys"] = np.array(sorted(anet_out["results"]))
    cf = os.path.join(tmp, "overlapping.txt")
    with open(cf, "w") as f:
        f.write("\n".join(FH.OVERLAPPING) + "\n")
    X.anet_cfg = anet_cfg                                           # exclude_overlapping reads the module global
    kept = X.exclude_overlapping(anet_out, cf)
    fx["anet_kept_keys"] = np.array(sorted(kept["results"]))
    thumos_res = {"video_test_0000004": [{"label": "class_00"}], "bbb222": [{"label": "old"}]}
    merged = dict(thumos_res)
    merged.update(kept["results"])                                   # the __main__ merge, test_cross_data.py:433-435
    fx["merged_keys"] = np.array(sorted(merged))

    # ---- thresholding over the "training" videos, every scoring rule
    tr_dir = os.path.join(tmp, "thumos_train")
    os.makedirs(tr_dir)
    tinfos = {}
    for name, seed, frames, fps in FH.THUMOS_TRAIN:
        np.save(os.path.join(tr_dir, name + ".npy"), FH.synthetic_video(seed, frames))
        tinfos[name] = {"sample_fps": fps, "sample_count": frames, "fps": fps * 3, "count": frames * 3}
    TH.get_video_info = lambda path: tinfos
    TH.get_class_index_map = lambda path: (None, IDX_TO_CLASS)
    TH.build_model = lambda **kw: (net, None)
    config["dataset"]["training"]["video_data_path"] = tr_dir
    config["dataset"]["testing"]["crop_size"] = 96

    class cfg:
        pass
    cfg.fusion, cfg.use_edl, cfg.use_rpl, cfg.use_gcpl, cfg.evidence, cfg.os_head = False, True, False, False, "exp", True
    cfg.num_classes, cfg.clip_length, cfg.stride, cfg.conf_thresh, cfg.top_k, cfg.nms_sigma = 15, 256, 128, 0.01, 5000, 0.5
    cfg.rgb_data_path = cfg.flow_data_path = tr_dir
    TH.cfg = cfg
    for sc in SCORINGS:
        cfg.scoring = sc
        out_file = os.path.join(tmp, f"thr_{sc}.json")
        thr = TH.thresholding(cfg, out_file)
        fx[f"threshold_{sc}"] = np.float64(thr)
        print(f"thresholding[{sc}] = {thr:.6f}")
    with open(out_file) as f:
        res = json.load(f)
    for name, props in res["results"].items():
        cls, rows = rows_of(props)
        fx[f"train_{name}_class"], fx[f"train_{name}_rows"] = cls, rows
        print(f"thresholding: {name}: {len(props)} proposals")
    np.savez_compressed(os.path.join(GOLD, "cross_data.npz"), **fx)
    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
        f.write("cross-dataset / threshold drivers: reference test_anet, exclude_overlapping, thresholding (6 scoring rules) run "
                "with a stand-in detector on seeded videos -> tests/golden/cross_data.npz ("
                + ", ".join(f"{k[5:-5]}: {len(v)}" for k, v in fx.items() if k.startswith("anet_") and k.endswith("_rows")) + ")\n")
    import shutil
    shutil.rmtree(tmp)
    leftovers = [os.path.join(d, n) for d, _, fs in os.walk(REF) for n in fs if n.endswith(".pyc")]
    assert not leftovers, leftovers


if __name__ == "__main__":
    main()
