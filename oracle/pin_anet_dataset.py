"""oracle/pin_anet_dataset.py -- TEST INFRASTRUCTURE, runs in the BUILD container only (needs /root/reference).

Pins opental_amd/common/anet_dataset.py (host-side sampling decisions of the ActivityNet1.3 recipe, BASELINE configs[3])
against the reference's AFSD/common/anet_dataset.py, imported from /root/reference, on a seeded synthetic dataset written
to a temp directory (video_info json, uint8 .npy videos shorter and longer than the clip):
  * get_video_info / split_videos: identical training lists, score rows and minimum-action table;
  * ANET_Dataset.__getitem__ with seeded `random`: the reference's clip and ssl clip (pixels from its numpy / torch code,
    127.5 padding included) equal oracle.input_ref.prepare_clip + the frame map drawn by
    opental_amd.common.anet_dataset (same random numbers in the same order); targets, ssl targets, scores, flags equal.
The reference calls `np.float` (removed from numpy 1.24+): the alias is restored in THIS process before the import --
an environment shim, not a change of the reference.
Writes tests/golden/anet_dataset.npz (the dataset's contents + the expected decisions) for the CPU / GPU tests.

    python -m oracle.pin_anet_dataset
"""
import json
import os
import random
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import numpy as np
import torch

from oracle import input_ref as R

CLIP, CROP, STRIDE, H, W = 96, 12, 96, 14, 14


def dataset_spec(seed=3, n=12):
    """name -> (frames in the file, info record).  Half the videos are shorter than the clip (127.5 padding)."""
    rs = np.random.RandomState(seed)
    spec = {}
    for v in range(n):
        frames = int(rs.randint(40, 96)) if v % 2 else int(rs.randint(96, 140))
        annos, t = [], int(rs.randint(2, 8))
        k = 0
        while t + 10 < min(frames, CLIP):
            ln = int(rs.randint(6, 9)) if k % 2 == 0 else int(rs.randint(20, 34))
            k += 1
            annos.append({"start_frame": t, "end_frame": min(t + ln, CLIP - 1), "label_id": int(rs.randint(1, 150))})
            t += ln + int(rs.randint(9, 22))
        if v in (8, 9):                                 # actions tile the clip: no background longer than th, the splice fails
            annos = [{"start_frame": a, "end_frame": b, "label_id": int(rs.randint(1, 150))}
                     for a, b in ((0, 30), (31, 60), (61, 95))]
        if v == 5:
            annos.append({"start_frame": 30, "end_frame": 30, "label_id": 3})      # dropped: end <= start
        spec[f"v_{v:05d}"] = (frames, {"subset": "training" if v != 7 else "validation", "frame_num": frames,
                                       "duration": frames / 10.0, "annotations": annos})
    return spec


def write_dataset(root, spec):
    os.makedirs(os.path.join(root, "npy"), exist_ok=True)
    with open(os.path.join(root, "info.json"), "w") as f:
        json.dump({k: rec for k, (_, rec) in spec.items()}, f)
    videos = {}
    for k, (name, (frames, _)) in enumerate(spec.items()):
        if name == "v_00003":
            continue                                    # no .npy file: skipped by split_videos
        v = np.random.RandomState(80 + k).randint(0, 256, (frames, H, W, 3)).astype(np.uint8)
        np.save(os.path.join(root, "npy", name + ".npy"), v)
        videos[name] = v
    return videos


def main():
    sys.path.insert(0, REF)
    sys.argv = ["pin", os.path.join(REF, "configs/anet_opental.yaml"), "--open_set", "--split", "0"]
    fake = types.ModuleType("boundary_max_pooling_cuda")
    fake.forward = fake.backward = None
    sys.modules["boundary_max_pooling_cuda"] = fake
    if not hasattr(np, "float"):
        np.float = float                                # anet_dataset.py:223 `.astype(np.float)`
    import AFSD.common.anet_dataset as RD
    from opental_amd.common import anet_dataset as MD

    root = tempfile.mkdtemp(prefix="pin_anet_ds_")
    spec = dataset_spec()
    videos = write_dataset(root, spec)
    info_path, vdir = os.path.join(root, "info.json"), os.path.join(root, "npy")
    assert RD.get_video_info(info_path) == MD.get_video_info(info_path)
    r_ds = RD.ANET_Dataset(info_path, vdir, CLIP, CROP, STRIDE)
    m_ds = MD.ANET_Dataset(info_path, vdir, CLIP, CROP, STRIDE, pin=False)
    assert len(r_ds) == len(m_ds) >= 8 and r_ds.th == m_ds.th
    for a, b in zip(r_ds.training_list, m_ds.training_list):
        assert a['video_name'] == b['video_name'] and a['offset'] == b['offset'] and a['annos'] == b['annos']
        assert a['frame_num'] == b['frame_num']
        for k in ('start', 'end', 'action'):
            assert np.array_equal(a[k], b[k])
    fx = {"n": np.int64(len(m_ds)), "clip": np.int64(CLIP), "crop": np.int64(CROP), "names": np.array([s['video_name'] for s in m_ds.training_list])}
    nflag = npad = 0
    for idx in range(len(m_ds)):
        for rep in range(3):                            # three seeds per sample: both outcomes of the splice occur
            seed = 1000 + 10 * idx + rep
            random.seed(seed)
            x, target, scores, ssl_x, ssl_target, flag = r_ds[idx]
            random.seed(seed)
            d = m_ds.decide(idx)
            i, j, flip = d['crop']
            name = m_ds.training_list[idx]['video_name']
            mine = R.prepare_clip(videos[name], d['offset'], CLIP, CROP, i, j, flip, valid=d['valid'], pad_value=127.5)
            assert np.array_equal(mine, x.numpy()), (idx, rep)
            npad += int(d['valid'] < CLIP)
            assert bool(flag) == bool(d['flag'])
            assert np.allclose(np.asarray(target, np.float32), d['target']) and np.array_equal(scores.numpy(), d['scores'])
            if flag:
                nflag += 1
                assert np.array_equal(mine[:, d['frame_map']], ssl_x.numpy()), (idx, rep)
                assert np.array_equal(np.asarray(ssl_target, np.float32), d['ssl_target'])
            else:
                assert np.array_equal(ssl_x.numpy(), x.numpy()) and d['frame_map'] is None
                assert np.array_equal(np.asarray(ssl_target, np.float32), d['ssl_target'])
            key = f"{idx}_{rep}"
            fx["crop_" + key] = np.array([i, j, int(flip), d['offset'], int(d['flag']), d['valid']], np.int64)
            fx["map_" + key] = d['frame_map'] if d['frame_map'] is not None else np.zeros(0, np.int32)
            fx["ssl_target_" + key] = d['ssl_target']
            if rep == 0:
                fx[f"target_{idx}"] = d['target']
                fx[f"scores_{idx}"] = d['scores']
    total = 3 * len(m_ds)
    assert 0 < nflag < total and npad > 0, (nflag, npad)
    fx["info_json"] = np.array(json.dumps({k: rec for k, (_, rec) in spec.items()}))
    np.savez_compressed(os.path.join(GOLD, "anet_dataset.npz"), **fx)
    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
        f.write(f"anet_dataset: get_video_info / split_videos identical; {total} draws over {len(m_ds)} samples: clips (127.5 padding "
                f"in {npad}), ssl clips (splice succeeded for {nflag}), targets, score rows, flags identical to ANET_Dataset.__getitem__\n")
    print(f"pinned {total} draws over {len(m_ds)} samples, splice succeeded for {nflag}, padded {npad}")
    import shutil
    shutil.rmtree(root)
    leftovers = [os.path.join(d_, n) for d_, _, fs in os.walk(REF) for n in fs if n.endswith(".pyc")]
    assert not leftovers, leftovers


if __name__ == "__main__":
    main()
