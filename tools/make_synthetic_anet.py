"""Writes a small synthetic ActivityNet1.3-layout training set (no dataset ships with the container): the video_info json of
AFSD/common/anet_dataset.py:32-40 (subset / frame_num / annotations with start_frame, end_frame, label_id), uint8 .npy
videos, and a yaml derived from configs/anet_opental.yaml whose paths point at it.  Used by tests/test_anet_dataset.py and
`bench.py --recipe anet`.

    python tools/make_synthetic_anet.py OUT_DIR [--videos 4] [--size 100]
"""
import json
import os
import sys

import numpy as np
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(out, videos=4, size=100, seed=0, clip=768):
    rs = np.random.RandomState(seed)
    vdir = os.path.join(out, "train_val_npy")
    os.makedirs(vdir, exist_ok=True)
    info = {}
    for v in range(videos):
        name = f"v_synth{v:05d}"
        frames = int(rs.randint(500, 768)) if v % 2 else 768          # every second video is padded with 127.5
        np.save(os.path.join(vdir, name + ".npy"), rs.randint(0, 256, (frames, size, size, 3)).astype(np.uint8))
        annos, t, k = [], 20, 0
        while t + 150 < frames:                          # short and long actions between long backgrounds (the splice needs both)
            ln = int(rs.randint(24, 32)) if k % 2 == 0 else int(rs.randint(110, 140))
            k += 1
            annos.append({"start_frame": t, "end_frame": t + ln, "label_id": int(rs.randint(1, 151))})
            t += ln + int(rs.randint(80, 120))
        info[name] = {"subset": "training", "frame_num": frames, "duration": frames / 10.0, "annotations": annos}
    info_path = os.path.join(out, "video_info.json")
    with open(info_path, "w") as f:
        json.dump(info, f)
    with open(os.path.join(REPO, "configs", "anet_opental.yaml")) as f:
        cfg = yaml.load(f.read(), Loader=yaml.FullLoader)
    for part in ("training", "testing"):
        cfg["dataset"][part].update(video_mp4_path=vdir, video_info_path=info_path, clip_length=clip, clip_stride=clip)
    cfg["dataset"]["class_info_path"] = os.path.join(out, "action_known.txt")
    with open(cfg["dataset"]["class_info_path"], "w") as f:
        f.write("".join(f"class_{i}\n" for i in range(150)))
    cfg["training"]["checkpoint_path"] = os.path.join(out, "ckpt")
    cfg["testing"].update(checkpoint_path=os.path.join(out, "ckpt", "checkpoint-latest.ckpt"), output_path=os.path.join(out, "output"))
    yaml_path = os.path.join(out, "anet_synthetic.yaml")
    with open(yaml_path, "w") as f:
        yaml.dump(cfg, f)
    return yaml_path


if __name__ == "__main__":
    args = sys.argv[1:]
    kw = {}
    for flag in ("--videos", "--size"):
        if flag in args:
            kw[flag[2:]] = int(args[args.index(flag) + 1])
    print(make(args[0], **kw))
