"""oracle/pin_thumos_dataset.py -- TEST INFRASTRUCTURE, runs in the BUILD container only (needs /root/reference).

Pins opental_amd/common/thumos_dataset.py (host-side sampling decisions) against the reference's
AFSD/common/thumos_dataset.py, imported from /root/reference, on a seeded synthetic dataset written to a temp directory
(video_info csv, annotation csv, class index file, uint8 .npy videos):
  * get_video_info / get_video_anno / split_videos: identical dictionaries and training lists;
  * THUMOS_Dataset.__getitem__ with seeded `random`: the reference's clip and ssl clip (pixels, computed by the
    reference's numpy / torch code) equal oracle.input_ref.prepare_clip + the frame map drawn by
    opental_amd.common.thumos_dataset (same random numbers in the same order); targets, ssl targets, flags equal.
Writes tests/golden/thumos_dataset.npz: the dataset files' contents + the expected decisions, for the GPU test that runs
ClipStager + otal_prepare_clips_map on the same samples.

    python -m oracle.pin_thumos_dataset
"""
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

CLIP, CROP, STRIDE, H, W = 64, 12, 16, 14, 14
INFO_CSV = ("video,fps,sample_fps,count,sample_count\n"
            "video_validation_0000001,30.0,10.0,900,300\n"
            "video_validation_0000002,25.0,10.0,500,200\n"
            "video_validation_0000003,30.0,10.0,150,50\n")
ANNO_CSV = ("video,type,type_idx,start,end,startFrame,endFrame\n"
            "video_validation_0000001,BaseballPitch,7,1.0,4.0,30,150\n"
            "video_validation_0000001,Billiards,12,10.0,13.0,330,420\n"
            "video_validation_0000001,BaseballPitch,7,20.0,21.5,600,660\n"
            "video_validation_0000002,Billiards,12,2.0,6.0,50,160\n"
            "video_validation_0000002,CleanAndJerk,21,10.0,12.0,250,330\n"
            "video_validation_0000003,CleanAndJerk,21,1.0,3.0,30,90\n")
CLASS_TXT = "7 BaseballPitch\n9 BasketballDunk\n12 Billiards\n21 CleanAndJerk\n"
FRAMES = {"video_validation_0000001": 300, "video_validation_0000002": 200, "video_validation_0000003": 50}


def write_dataset(root):
    os.makedirs(os.path.join(root, "npy"), exist_ok=True)
    for name, text in (("info.csv", INFO_CSV), ("anno.csv", ANNO_CSV), ("classes.txt", CLASS_TXT)):
        with open(os.path.join(root, name), "w") as f:
            f.write(text)
    videos = {}
    for k, (name, t) in enumerate(FRAMES.items()):
        v = np.random.RandomState(50 + k).randint(0, 256, (t, H, W, 3)).astype(np.uint8)
        np.save(os.path.join(root, "npy", name + ".npy"), v)
        videos[name] = v
    return videos


def main():
    sys.path.insert(0, REF)
    sys.argv = ["pin", os.path.join(REF, "configs/thumos14_opental_final.yaml"), "--open_set", "--split", "0"]
    fake = types.ModuleType("boundary_max_pooling_cuda")
    fake.forward = fake.backward = None
    sys.modules["boundary_max_pooling_cuda"] = fake
    import AFSD.common.thumos_dataset as RD
    from opental_amd.common import thumos_dataset as MD

    root = tempfile.mkdtemp(prefix="pin_ds_")
    videos = write_dataset(root)
    paths = [os.path.join(root, n) for n in ("info.csv", "anno.csv", "classes.txt")]
    r_info, m_info = RD.get_video_info(paths[0]), MD.get_video_info(paths[0])
    assert r_info == m_info, (r_info, m_info)
    r_anno, m_anno = RD.get_video_anno(r_info, paths[1], paths[2]), MD.get_video_anno(m_info, paths[1], paths[2])
    assert r_anno == m_anno
    assert RD.get_class_index_map(paths[2]) == MD.get_class_index_map(paths[2])
    r_list, r_th = RD.split_videos(r_info, r_anno, CLIP, STRIDE)
    m_list, m_th = MD.split_videos(m_info, m_anno, CLIP, STRIDE)
    assert r_th == m_th and len(r_list) == len(m_list) > 10
    for a, b in zip(r_list, m_list):
        assert a['video_name'] == b['video_name'] and a['offset'] == b['offset'] and a['annos'] == b['annos']
        assert np.array_equal(a['start'], b['start']) and np.array_equal(a['end'], b['end'])
    r_data = RD.load_video_data(r_info, os.path.join(root, "npy"))
    m_data = MD.load_video_data(m_info, os.path.join(root, "npy"), pin=False)
    r_ds = RD.THUMOS_Dataset(r_data, r_info, r_anno, clip_length=CLIP, crop_size=CROP, stride=STRIDE)
    m_ds = MD.THUMOS_Dataset(m_data, m_info, m_anno, clip_length=CLIP, crop_size=CROP, stride=STRIDE)
    assert len(r_ds) == len(m_ds)
    fx = {"n": np.int64(len(m_ds)), "clip": np.int64(CLIP), "crop": np.int64(CROP), "stride": np.int64(STRIDE)}
    nflag = 0
    for idx in range(len(m_ds)):
        random.seed(1000 + idx)
        x, target, scores, ssl_x, ssl_target, flag = r_ds[idx]
        random.seed(1000 + idx)
        d = m_ds.decide(idx)
        i, j, flip = d['crop']
        mine = R.prepare_clip(videos[m_ds.training_list[idx]['video_name']], d['offset'], CLIP, CROP, i, j, flip)
        assert np.array_equal(mine, x.numpy()), idx
        assert bool(flag) == bool(d['flag'])
        assert np.allclose(np.asarray(target, np.float32), d['target']) and np.array_equal(scores.numpy(), d['scores'])
        if flag:
            nflag += 1
            assert np.array_equal(mine[:, d['frame_map']], ssl_x.numpy()), idx
            assert np.array_equal(np.asarray(ssl_target, np.float32), d['ssl_target'])
        else:
            assert np.array_equal(ssl_x.numpy(), x.numpy()) and d['frame_map'] is None
        fx[f"crop_{idx}"] = np.array([i, j, int(flip), d['offset'], int(d['flag'])], np.int64)
        fx[f"map_{idx}"] = d['frame_map'] if d['frame_map'] is not None else np.zeros(0, np.int32)
        fx[f"ssl_target_{idx}"] = d['ssl_target']
        fx[f"target_{idx}"] = d['target']
    assert 0 < nflag < len(m_ds), nflag            # both outcomes of the splice occur
    np.savez_compressed(os.path.join(GOLD, "thumos_dataset.npz"), **fx)
    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
        f.write(f"thumos_dataset: get_video_info / get_video_anno / split_videos identical; {len(m_ds)} samples: clips, ssl "
                f"clips (splice succeeded for {nflag}), targets, flags identical to THUMOS_Dataset.__getitem__\n")
    print(f"pinned {len(m_ds)} samples, splice succeeded for {nflag}")
    import shutil
    shutil.rmtree(root)
    leftovers = [os.path.join(d_, n) for d_, _, fs in os.walk(REF) for n in fs if n.endswith(".pyc")]
    assert not leftovers, leftovers


if __name__ == "__main__":
    main()
