"""Writes a small synthetic THUMOS14-layout dataset (no dataset ships with the container): video_info / annotation csv,
class index file, uint8 .npy clips, open-set ground-truth json, and a yaml derived from configs/thumos14_opental_final.yaml
whose paths point at it.  Used by tests/test_drivers_gpu.py and `bench.py` extras.

    python tools/make_synthetic_thumos.py OUT_DIR [--videos 2] [--frames 400] [--size 100]
"""
import json
import os
import sys

import numpy as np
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = [(7, "BaseballPitch"), (9, "BasketballDunk"), (12, "Billiards"), (21, "CleanAndJerk"), (22, "CliffDiving"),
           (23, "CricketBowling"), (24, "CricketShot"), (26, "Diving"), (31, "FrisbeeCatch"), (33, "GolfSwing"),
           (36, "HammerThrow"), (40, "HighJump"), (45, "JavelinThrow"), (51, "LongJump"), (68, "PoleVault")]


def make(out, videos=2, frames=400, size=100, seed=0, test_videos=2, uniform=0):
    """`uniform`: the last so many training videos carry actions of ONE length (~32 frames) at irregular distances: no action is
    twice as long as the shortest, so the ssl splice never applies to their windows (flag False: plain steps)."""
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(out, "train_npy"), exist_ok=True)
    os.makedirs(os.path.join(out, "test_npy"), exist_ok=True)
    with open(os.path.join(out, "classes.txt"), "w") as f:
        f.write("".join(f"{i} {n}\n" for i, n in CLASSES))
    info = ["video,fps,sample_fps,count,sample_count"]
    anno = ["video,type,type_idx,start,end,startFrame,endFrame"]
    for v in range(videos):
        name = f"video_validation_{v:07d}"
        np.save(os.path.join(out, "train_npy", name + ".npy"), rs.randint(0, 256, (frames, size, size, 3)).astype(np.uint8))
        info.append(f"{name},30.0,10.0,{frames * 3},{frames}")
        t, k = 20, 0
        while t + 80 < frames:                              # short (~14) and long (~64 frame) actions between long backgrounds:
            cid, cname = CLASSES[rs.randint(len(CLASSES))]  # the ssl splice needs an action longer than twice the shortest one
            if v >= videos - uniform:
                ln = int(rs.randint(30, 35))
                gap = int(rs.randint(20, 110))
            else:
                ln = int(rs.randint(12, 16)) if k % 2 == 0 else int(rs.randint(58, 70))
                gap = int(rs.randint(60, 90))
            k += 1
            anno.append(f"{name},{cname},{cid},{t / 10:.1f},{(t + ln) / 10:.1f},{t * 3},{(t + ln) * 3}")
            t += ln + gap
    tinfo = ["video,fps,sample_fps,count,sample_count"]
    gt = {"database": {}}
    for v in range(test_videos):
        name = f"video_test_{v:07d}"
        fr = frames - 60 * v
        np.save(os.path.join(out, "test_npy", name + ".npy"), rs.randint(0, 256, (fr, size, size, 3)).astype(np.uint8))
        tinfo.append(f"{name},30.0,10.0,{fr * 3},{fr}")
        gt["database"][name] = {"subset": "test", "duration": fr / 10.0, "annotations": [
            {"label": CLASSES[(3 * v + k) % len(CLASSES)][1], "segment": [2.0 + 9 * k, 7.5 + 9 * k]} for k in range(3)] +
            [{"label": "Shotput", "segment": [29.0, 33.0]}]}           # a class outside the known set: "unknown" in the open-set metrics
    for fn, rows in (("train_info.csv", info), ("train_anno.csv", anno), ("test_info.csv", tinfo)):
        with open(os.path.join(out, fn), "w") as f:
            f.write("\n".join(rows) + "\n")
    with open(os.path.join(out, "gt_open.json"), "w") as f:
        json.dump(gt, f)
    with open(os.path.join(REPO, "configs", "thumos14_opental_final.yaml")) as f:
        cfg = yaml.load(f.read(), Loader=yaml.FullLoader)
    cfg["dataset"]["class_info_path"] = os.path.join(out, "classes.txt")
    cfg["dataset"]["training"].update(video_info_path=os.path.join(out, "train_info.csv"),
                                      video_anno_path=os.path.join(out, "train_anno.csv"),
                                      video_data_path=os.path.join(out, "train_npy"))
    cfg["dataset"]["testing"].update(video_info_path=os.path.join(out, "test_info.csv"),
                                     video_anno_path=os.path.join(out, "train_anno.csv"),
                                     video_data_path=os.path.join(out, "test_npy"))
    cfg["training"]["checkpoint_path"] = os.path.join(out, "models")
    cfg["testing"]["checkpoint_path"] = os.path.join(out, "models", "checkpoint-latest.ckpt")
    cfg["testing"]["output_path"] = os.path.join(out, "output")
    path = os.path.join(out, "synthetic.yaml")
    with open(path, "w") as f:
        yaml.dump(cfg, f)
    return path


if __name__ == "__main__":
    a = sys.argv[1:]
    kw = {a[i][2:]: int(a[i + 1]) for i in range(1, len(a) - 1, 2)}
    print(make(a[0], **kw))
