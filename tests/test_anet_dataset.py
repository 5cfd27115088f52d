"""ActivityNet1.3 training input (BASELINE configs[3]; reference AFSD/common/anet_dataset.py): the host-side sampling
decisions against the fixture pinned to the reference's ANET_Dataset (oracle/pin_anet_dataset.py ->
tests/golden/anet_dataset.npz), and on the GPU the pinned stager + otal_prepare_clips_map against the oracle's numpy clip
preparation, bit for bit -- 127.5 padding of short videos and the self-supervised splice included -- and the recipe's
driver end to end (python -m opental_amd.anet.train) on a small synthetic dataset."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import input_ref as R
from oracle import pin_anet_dataset as P


def _dataset(tmp_path, pin):
    from opental_amd.common import anet_dataset as MD
    root = str(tmp_path)
    videos = P.write_dataset(root, P.dataset_spec())
    return MD.ANET_Dataset(os.path.join(root, "info.json"), os.path.join(root, "npy"), P.CLIP, P.CROP, P.STRIDE, pin=pin), videos


def test_sampling_decisions_match_the_reference_fixture(tmp_path, golden_dir):
    ds, _ = _dataset(tmp_path, pin=False)
    fx = np.load(os.path.join(golden_dir, "anet_dataset.npz"))
    assert len(ds) == int(fx["n"]) and [s['video_name'] for s in ds.training_list] == [str(n) for n in fx["names"]]
    assert json.loads(str(fx["info_json"])) == {k: rec for k, (_, rec) in P.dataset_spec().items()}
    flags = fails = 0
    for idx in range(len(ds)):
        for rep in range(3):
            random.seed(1000 + 10 * idx + rep)
            d = ds.decide(idx)
            i, j, flip = d['crop']
            key = f"{idx}_{rep}"
            assert [i, j, int(flip), d['offset'], int(d['flag']), d['valid']] == fx["crop_" + key].tolist()
            if d['flag']:
                flags += 1
                assert np.array_equal(d['frame_map'], fx["map_" + key])
            else:
                fails += 1
                assert d['frame_map'] is None and fx["map_" + key].size == 0
            assert np.array_equal(d['ssl_target'], fx["ssl_target_" + key])
        assert np.array_equal(d['target'], fx[f"target_{idx}"]) and np.array_equal(d['scores'], fx[f"scores_{idx}"])
        assert d['scores'].shape == (3, P.CLIP) and d['scores'].max() > 1        # [action, start, end] rows carry label ids
    assert flags > 0 and fails > 0


def test_deciding_an_epoch_loads_no_frames_and_the_video_cache_is_bounded(tmp_path):
    """decide() used to load (and pin) every video of the epoch on every rank before the first step.  Decisions
    need the frame size only -- read from the .npy header -- and the pixels are loaded when the stager slices the record's
    lazy handle, through a least-recently-used cache of `cache_videos` entries."""
    from opental_amd.common import anet_dataset as MD
    root = str(tmp_path)
    videos = P.write_dataset(root, P.dataset_spec())
    ds = MD.ANET_Dataset(os.path.join(root, "info.json"), os.path.join(root, "npy"), P.CLIP, P.CROP, P.STRIDE, cache_videos=2)
    recs = [ds.decide(i) for i in range(len(ds))]
    assert len(ds._cache) == 0                                  # no frame was read
    names = []
    for r in recs:
        v = r['video']
        assert isinstance(v, MD.LazyVideo) and v.dtype == torch.uint8 and tuple(v.shape) == tuple(videos[v.name].shape)
        sl = v[r['offset']: r['offset'] + r['valid']]
        assert torch.equal(sl, torch.from_numpy(videos[v.name][r['offset']: r['offset'] + r['valid']]))
        names.append(v.name)
        assert len(ds._cache) <= 2
    assert len(set(names)) > 2                                  # more videos than cache entries were touched
    assert list(ds._cache) == list(dict.fromkeys(reversed(names)))[:2][::-1]      # the two most recently used, oldest first


def test_background_prefetch_feeds_the_cache_and_a_failed_read_surfaces_at_the_consumer(tmp_path):
    """with lazy videos every sample was a synchronous np.load on the training thread.  prefetch(names) reads
    in the background; video(name) takes the finished read (or waits for it), the cache bound still holds, duplicates and
    already cached names are skipped, and an unreadable file raises where the video is USED, not in the reader thread."""
    from opental_amd.common import anet_dataset as MD
    root = str(tmp_path)
    videos = P.write_dataset(root, P.dataset_spec())
    ds = MD.ANET_Dataset(os.path.join(root, "info.json"), os.path.join(root, "npy"), P.CLIP, P.CROP, P.STRIDE, cache_videos=4)
    names = list(videos)[:3]
    ds.prefetch(names + names[:1])
    assert set(ds._pending) <= set(names) and len(ds._pending) <= 2            # at most cache_videos / 2 outstanding
    for n in names:
        assert torch.equal(ds.video(n), torch.from_numpy(videos[n]))
    assert not ds._pending and list(ds._cache) == names
    ds.prefetch(names)                                          # all cached: nothing to do
    assert not ds._pending
    ds.prefetch(["no_such_video"])
    with pytest.raises(FileNotFoundError):
        ds.video("no_such_video")
    assert len(ds._cache) <= 4


def test_prefetched_videos_nobody_asks_for_are_dropped_and_failed_reads_are_reported(tmp_path):
    """a name that was prefetched but never consumed (epoch end, a --max_steps cut) kept its frames for good
    and counted against the outstanding-read limit, so prefetching stopped silently; a failed read was never surfaced.  The
    next prefetch() drops finished reads its request does not name and warns about the failed ones."""
    import time
    import warnings
    from opental_amd.common import anet_dataset as MD
    root = str(tmp_path)
    videos = P.write_dataset(root, P.dataset_spec())
    ds = MD.ANET_Dataset(os.path.join(root, "info.json"), os.path.join(root, "npy"), P.CLIP, P.CROP, P.STRIDE, cache_videos=4)
    names = list(videos)
    ds.prefetch([names[0], "no_such_video"])                    # two outstanding reads = the limit (cache_videos / 2)
    for f in list(ds._pending.values()):
        while not f.done():
            time.sleep(0.01)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        ds.prefetch(names[1:3])                                 # the epoch moved on: the two old reads are nobody's
    assert set(ds._pending) == set(names[1:3])                  # ... and prefetching goes on
    assert any("no_such_video" in str(w.message) for w in caught)
    for n in names[1:3]:
        assert torch.equal(ds.video(n), torch.from_numpy(videos[n]))


def test_slice_assignment_semantics_of_the_splice():
    """`new[:, a:b] = old[:, c:d]` as torch evaluates it: equal lengths copy, a one-frame source broadcasts, anything
    else is the RuntimeError the reference catches (anet_dataset.py:194-207) -> the splice is given up."""
    from opental_amd.common.anet_dataset import _assign
    fm = np.arange(10, dtype=np.int32)
    assert _assign(fm, (2, 5), (6, 9), 10) and fm.tolist() == [0, 1, 6, 7, 8, 5, 6, 7, 8, 9]
    assert _assign(fm, (0, 2), (9, 10), 10) and fm[:2].tolist() == [9, 9]
    assert not _assign(fm, (8, 12), (0, 4), 10)          # destination clipped at the clip's end: shapes differ
    assert _assign(fm, (5, 3), (7, 7), 10)               # both empty


@pytest.mark.gpu
def test_stager_and_device_kernel_match_the_oracle_bit_for_bit(tmp_path):
    from opental_amd.common import anet_dataset as MD
    from opental_amd.common.thumos_dataset import ClipStager
    ds, videos = _dataset(tmp_path, pin=True)
    B = 3
    st = ClipStager(B, P.CLIP, P.H, P.W, P.CROP)
    batches = [list(range(k, k + B)) for k in range(0, 9, B)]
    decided = []
    for idxs in batches:
        samples = []
        for idx in idxs:
            random.seed(1000 + 10 * idx)
            samples.append(ds.decide(idx))
        decided.append(samples)
    st.submit(decided[0])
    padded = 0
    for k in range(len(batches)):
        clips, ssl = st.collect(want_ssl=True)
        if k + 1 < len(batches):
            st.submit(decided[k + 1])
        for b, smp in enumerate(decided[k]):
            name = ds.training_list[batches[k][b]]['video_name']
            i, j, flip = smp['crop']
            want = R.prepare_clip(videos[name], smp['offset'], P.CLIP, P.CROP, i, j, flip, valid=smp['valid'], pad_value=127.5)
            padded += int(smp['valid'] < P.CLIP)
            assert np.array_equal(clips[b].cpu().numpy(), want), (k, b)
            fm = smp['frame_map'] if smp['frame_map'] is not None else np.arange(P.CLIP)
            assert np.array_equal(ssl[b].cpu().numpy(), want[:, fm]), (k, b)
            if smp['valid'] < P.CLIP:
                assert float(clips[b, :, smp['valid']:].abs().max()) == 0.0        # 127.5 -> exactly 0.0
    assert padded > 0


@pytest.mark.gpu
def test_anet_train_driver_end_to_end(tmp_path):
    """configs/anet_opental.yaml -> ANET_Dataset -> pinned staging -> DetectorTrainer (two optimizer groups, ssl branch when
    the splice succeeded) -> save_model -> --resume continues bit-identically.  768-frame clips of 100 x 100 frames, 4 videos."""
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_synthetic_anet import make
    from opental_amd.anet import train as R_
    from opental_amd.common import ops
    old = ops.CONV_PRECISION
    try:
        yaml_path = make(str(tmp_path / "data"), videos=4, size=100)
        common = [yaml_path, '--open_set', '--split', '0', '--lw', '1', '--cw', '1', '--piou', '0.6', '--ssl', '0.1',
                  '--random_init', '--save_after', '0', '--max_steps', '2']
        tr_a, hist_a = R_.main(common + ['--max_epoch', '2', '--checkpoint_path', str(tmp_path / "run_a")])
        assert len(hist_a) == 2 and all(np.isfinite(h).all() for h in hist_a) and tr_a.step_count == 4
        assert [round(g / tr_a.lr, 6) for _, _, g in tr_a._group_ranges] == [0.1, 1.0] or \
               [round(g / tr_a.lr, 6) for _, _, g in tr_a._group_ranges] == [1.0, 0.1]
        R_.main(common + ['--max_epoch', '1', '--checkpoint_path', str(tmp_path / "run_b")])
        tr_b, hist_b = R_.main(common + ['--max_epoch', '2', '--resume', '1', '--checkpoint_path', str(tmp_path / "run_b")])
        assert tr_b.step_count == 4 and torch.equal(tr_a.arena.flat, tr_b.arena.flat) and hist_b[0] == hist_a[1]
    finally:
        ops.CONV_PRECISION = old
