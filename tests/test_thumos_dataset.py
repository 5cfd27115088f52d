"""Training-side input pipeline (reference AFSD/common/thumos_dataset.py): the host-side sampling
decisions against the fixture pinned to the reference's THUMOS_Dataset (oracle/pin_thumos_dataset.py ->
tests/golden/thumos_dataset.npz), and on the GPU the pinned double-buffered stager + otal_prepare_clips_map against the
oracle's numpy clip preparation, bit for bit, including the self-supervised clip splice."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import input_ref as R
from oracle import pin_thumos_dataset as P


def _dataset(tmp_path, pin):
    from opental_amd.common import thumos_dataset as MD
    root = str(tmp_path)
    videos = P.write_dataset(root)
    info = MD.get_video_info(os.path.join(root, "info.csv"))
    anno = MD.get_video_anno(info, os.path.join(root, "anno.csv"), os.path.join(root, "classes.txt"))
    data = MD.load_video_data(info, os.path.join(root, "npy"), pin=pin)
    return MD.THUMOS_Dataset(data, info, anno, clip_length=P.CLIP, crop_size=P.CROP, stride=P.STRIDE), videos


def test_sampling_decisions_match_the_reference_fixture(tmp_path, golden_dir):
    ds, _ = _dataset(tmp_path, pin=False)
    fx = np.load(os.path.join(golden_dir, "thumos_dataset.npz"))
    assert len(ds) == int(fx["n"])
    flags = 0
    for idx in range(len(ds)):
        random.seed(1000 + idx)
        d = ds.decide(idx)
        i, j, flip = d['crop']
        assert [i, j, int(flip), d['offset'], int(d['flag'])] == fx[f"crop_{idx}"].tolist()
        if d['flag']:
            flags += 1
            assert np.array_equal(d['frame_map'], fx[f"map_{idx}"])
            assert sorted(d['frame_map'].tolist()) != d['frame_map'].tolist()        # really a splice
        else:
            assert d['frame_map'] is None and fx[f"map_{idx}"].size == 0
        assert np.array_equal(d['ssl_target'], fx[f"ssl_target_{idx}"]) and np.array_equal(d['target'], fx[f"target_{idx}"])
    assert 0 < flags < len(ds)


def test_epoch_batches_cover_the_training_list_once(tmp_path):
    from opental_amd.common import thumos_dataset as MD
    ds, _ = _dataset(tmp_path, pin=False)
    g = torch.Generator().manual_seed(3)
    got = [s for batch in MD.batches(ds, 2, generator=g) for s in batch]
    assert len(got) == len(ds) // 2 * 2
    keys = [(id(s['video']), s['offset']) for s in got]
    assert len(set(keys)) == len(keys)


@pytest.mark.gpu
def test_stager_and_device_kernel_match_the_oracle_bit_for_bit(tmp_path):
    """ClipStager: uint8 frames from PINNED videos over the copy stream (double buffered) + otal_prepare_clips_map.  The
    plain batch equals oracle.input_ref.prepare_clip (pinned to the reference's transforms) and the ssl batch equals it
    with the frame map applied, bit for bit; three consecutive batches reuse both staging slots."""
    from opental_amd.common import thumos_dataset as MD
    ds, videos = _dataset(tmp_path, pin=True)
    B = 3
    st = MD.ClipStager(B, P.CLIP, P.H, P.W, P.CROP)
    order = list(range(len(ds)))
    batches = [order[k:k + B] for k in range(0, 9, B)]
    decided = []
    for k, idxs in enumerate(batches):
        samples = []
        for idx in idxs:
            random.seed(1000 + idx)
            samples.append(ds.decide(idx))
        decided.append(samples)
    st.submit(decided[0])
    for k in range(len(batches)):
        clips, ssl = st.collect(want_ssl=True)
        if k + 1 < len(batches):
            st.submit(decided[k + 1])               # batch k+1 travels while batch k is checked
        for b, smp in enumerate(decided[k]):
            name = ds.training_list[batches[k][b]]['video_name']
            i, j, flip = smp['crop']
            want = R.prepare_clip(videos[name], smp['offset'], P.CLIP, P.CROP, i, j, flip)
            assert np.array_equal(clips[b].cpu().numpy(), want), (k, b)
            fm = smp['frame_map'] if smp['frame_map'] is not None else np.arange(P.CLIP)
            assert np.array_equal(ssl[b].cpu().numpy(), want[:, fm]), (k, b)
    # a batch without any splice: no ssl output unless asked for
    plain = [dict(s, frame_map=None) for s in decided[0]]
    st.submit(plain)
    clips, ssl = st.collect()
    assert ssl is None and clips.shape == (B, 3, P.CLIP, P.CROP, P.CROP)
