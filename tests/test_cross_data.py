"""Cross-dataset open-set inference driver (AFSD/thumos14/test_cross_data.py): host logic on CPU,
and on the GPU the batched driver against the reference's per-video, per-window order of operations."""
import numpy as np
import pytest
import torch


def test_anet_padding_equals_the_thumos_padding():
    """127.5 padded before normalisation (test_cross_data.py:80-89) is bit-for-bit the 0.0 padded after it
    (test.py:67-76): the batched windows of detect_batch serve both datasets."""
    from opental_amd.thumos14 import test as T, test_cross_data as X
    rs = np.random.RandomState(0)
    data = torch.from_numpy(rs.randint(0, 256, size=(3, 300, 8, 8)).astype(np.uint8))
    for off in (0, 44, 256):
        assert torch.equal(X.prepare_anet_clip(data, off, 256, 8), T.prepare_clip(data, off, 256))
    assert X.prepare_anet_clip(data, 256, 256, 8).shape == (1, 3, 256, 8, 8)
    assert float(X.prepare_anet_clip(data, 256, 256, 8)[0, :, 44:].abs().max()) == 0.0


def test_duration_clipping_and_empty_segments():
    from opental_amd.thumos14 import test as T
    rows = torch.zeros(2, 3, 5)
    rows[0, 0] = torch.tensor([1.0, 9.0, 0.9, 0.2, 0.7])
    rows[0, 1] = torch.tensor([7.5, 12.0, 0.5, 0.3, 0.6])     # clipped to the duration
    rows[1, 0] = torch.tensor([8.5, 11.0, 0.4, 0.1, 0.8])     # starts past the duration: dropped
    rows[1, 1] = torch.tensor([3.0, 3.0, 0.3, 0.1, 0.8])      # empty: dropped only by the cross-dataset variants
    counts = torch.tensor([2, 2])
    names = {1: 'A', 2: 'B'}
    plain = T.get_video_detections(rows, counts, names)
    assert [p['segment'] for p in plain] == [[1.0, 9.0], [7.5, 12.0], [8.5, 11.0], [3.0, 3.0]]
    cross = T.get_video_detections(rows, counts, names, duration=8.0)
    assert [(p['label'], p['segment']) for p in cross] == [('A', [1.0, 8.0]), ('A', [7.5, 8.0])]
    thumos_leg = T.get_video_detections(rows, counts, names, drop_empty=True)
    assert [p['segment'] for p in thumos_leg] == [[1.0, 9.0], [7.5, 12.0], [8.5, 11.0]]


def test_exclude_overlapping_and_merge():
    from opental_amd.thumos14 import test as T, test_cross_data as X
    infos = {'v_a': {'annotations': [{'label': 'Long jump'}, {'label': 'Knitting'}]},
             'v_b': {'annotations': [{'label': 'Knitting'}]},
             'v_c': {'annotations': []}}
    anet = T.results_json({'a': [{'label': 'x'}], 'b': [{'label': 'y'}], 'c': []})
    kept = X.exclude_overlapping(anet, infos, ['Long jump\n', 'Shot put\n'])
    assert sorted(kept['results']) == ['b', 'c'] and kept['version'] == 'THUMOS14' and kept['external_data'] == {}
    thumos = T.results_json({'video_test_0000004': [{'label': 'z'}], 'b': [{'label': 'old'}]})
    merged = X.merge_results(thumos, kept)
    assert sorted(merged['results']) == ['b', 'c', 'video_test_0000004']
    assert merged['results']['b'] == [{'label': 'y'}]          # dict.update semantics of the reference's merge
    assert thumos['results']['b'] == [{'label': 'old'}]        # inputs untouched


@pytest.mark.gpu
def test_test_anet_matches_per_window_order(golden_dir):
    """The batched driver (videos batched, all windows decoded and suppressed in two launches) against the reference's
    order: per video, per window at b=1 with prepare_anet_clip, then decode, Soft-NMS, duration clipping."""
    import os
    from oracle import arch
    from opental_amd.thumos14 import test as T, test_cross_data as X
    from opental_amd.thumos14.BDNet import BDNet
    fx = np.load(os.path.join(golden_dir, "thumos_b1.npz"))
    net = BDNet(training=False, use_edl=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in arch.make_params(int(fx["param_seed"])).items()})
    net = net.cuda().eval()
    rs = np.random.RandomState(3)
    frames = {'v_one': 300, 'v_two': 200, 'v_gone': 256}
    videos = {n: torch.from_numpy(rs.randint(0, 256, size=(3, t, 96, 96)).astype(np.uint8)).cuda() for n, t in frames.items()}
    del videos['v_gone']                                        # listed but not on disk: skipped (test_cross_data.py:268-270)
    infos = {n: {'fps': 10.0, 'duration': 0.08 * t, 'frame_num': t} for n, t in frames.items()}
    out = X.test_anet(net, videos, infos, batch_clips=1, conf_thresh=0.01)
    assert sorted(out['results']) == ['one', 'two'] and out['version'] == 'THUMOS14'
    total = 0
    for name in ('v_one', 'v_two'):
        data, t = videos[name], frames[name]
        offs = T.get_offsets(t, 256, 128)
        assert offs == ([0, 44] if t == 300 else [0])
        with torch.no_grad():
            outs = [net(X.prepare_anet_clip(data, o, 256, 96)) for o in offs]
        merged = {k: (torch.cat([o[k] for o in outs], 0) if k != 'priors' else outs[0][k])
                  for k in ('loc', 'conf', 'prop_loc', 'prop_conf', 'center', 'act', 'prop_act', 'priors')}
        dec = T.decode_clips(merged, [float(o) for o in offs], [10.0] * len(offs), 256, 0.01)
        rows, counts, _ = T.softnms_classes(dec, [0, len(offs)], 5000, 0.5)
        ref = T.get_video_detections(rows[0], counts[0], None, 5000, duration=infos[name]['duration'])
        got = out['results'][name[2:]]
        assert got == ref
        assert all(0.0 <= p['segment'][0] < p['segment'][1] <= infos[name]['duration'] for p in got)
        total += len(got)
    assert total > 0


@pytest.mark.gpu
def test_prepare_windows_is_bit_identical_to_prepare_clip():
    """otal_prepare_windows (one launch per forward pass) against the reference's per-window preparation: full windows,
    a short last window (zero padded after normalisation), a video shorter than the window, both datasets' variants."""
    from opental_amd.thumos14 import test as T, test_cross_data as X
    rs = np.random.RandomState(11)
    videos = [torch.from_numpy(rs.randint(0, 256, size=(3, t, 96, 96)).astype(np.uint8)).cuda() for t in (300, 200, 513)]
    windows = [(v, o) for v, d in enumerate(videos) for o in T.get_offsets(d.shape[1], 256, 128)] + [(2, 400), (0, 299)]
    got = T.prepare_windows(videos, windows, 256)
    ref = torch.cat([T.prepare_clip(videos[v], o, 256) for v, o in windows], 0)
    assert got.shape == ref.shape == (len(windows), 3, 256, 96, 96)
    assert torch.equal(got, ref)
    assert torch.equal(got, torch.cat([X.prepare_anet_clip(videos[v], o, 256, 96) for v, o in windows], 0))
    assert float(got[-1, :, 1:].abs().max()) == 0.0 and float(got[-1, :, 0].abs().max()) > 0
    with pytest.raises(RuntimeError):
        T.prepare_windows(videos, [(0, 300)], 256)               # offset past the end
    with pytest.raises(RuntimeError):
        T.prepare_windows([videos[0].float()], [(0, 0)], 256)    # not uint8


# ----------------------------------------------------------------------------- pinned against the reference's own drivers
# tests/golden/cross_data.npz is written by oracle/pin_cross_data.py, which runs the REFERENCE's test_anet /
# exclude_overlapping / thresholding end to end with a stand-in detector (oracle/fake_heads.FakeNet) on seeded videos.
IDX_TO_CLASS = {i + 1: f"class_{i:02d}" for i in range(15)}
SCORINGS = ("uncertainty", "confidence", "uncertainty_actionness", "a_by_inv_u", "u_by_inv_a", "half_au")


def _props_from_fixture(fx, key):
    cls, rows = fx[key + "_class"], fx[key + "_rows"]
    return [{'label': IDX_TO_CLASS[int(c)], 'segment': [float(r[0]), float(r[1])], 'score': float(r[2]),
             'uncertainty': float(r[3]), 'actionness': float(r[4])} for c, r in zip(cls, rows)]


def test_host_logic_matches_the_reference_fixture(golden_dir):
    """CPU: get_offsets (test_cross_data.py:49-56), the six OOD thresholds of threshold.py:128-150 recomputed from the
    reference's own detections, exclude_overlapping (:333-352) and the merge of its __main__ (:433-435)."""
    import os
    from oracle import fake_heads as FH
    from opental_amd.thumos14 import test as T, test_cross_data as X
    fx = np.load(os.path.join(golden_dir, "cross_data.npz"))
    for n in (100, 256, 300, 384, 640, 700):
        assert T.get_offsets(n, 256, 128) == fx[f"offsets_{n}"].tolist()
    train = {name: _props_from_fixture(fx, "train_" + name) for name, _, _, _ in FH.THUMOS_TRAIN}
    assert sum(len(v) for v in train.values()) > 1000
    for sc in SCORINGS:
        assert abs(T.ood_threshold(train, sc) - float(fx[f"threshold_{sc}"])) < 1e-12, sc
    infos = {name: {'annotations': [{'label': l} for l in labels]} for name, _, _, _, _, labels in FH.ANET_VIDEOS}
    anet = T.results_json({name[2:]: [] for name, *_ in FH.ANET_VIDEOS})
    kept = X.exclude_overlapping(anet, infos, [c + "\n" for c in FH.OVERLAPPING])
    assert sorted(kept['results']) == fx["anet_kept_keys"].tolist()
    merged = X.merge_results(T.results_json({"video_test_0000004": [], "bbb222": [{"label": "old"}]}), kept)
    assert sorted(merged['results']) == fx["merged_keys"].tolist()


def _check_props(got, cls, rows, what):
    assert len(got) == len(cls), (what, len(got), len(cls))
    names = {v: k for k, v in IDX_TO_CLASS.items()}
    assert [names[p['label']] for p in got] == cls.tolist(), what
    mine = np.array([[p['segment'][0], p['segment'][1], p['score'], p['uncertainty'], p['actionness']] for p in got]).reshape(-1, 5)
    np.testing.assert_allclose(mine, rows, rtol=3e-6, atol=2e-6, err_msg=what)


@pytest.mark.gpu
def test_cross_dataset_driver_matches_the_reference_run(golden_dir):
    """GPU: X.test_anet (batched windows -> otal_decode_clips -> otal_softnms_classes -> duration clipping) on the seeded
    videos and stand-in detector of the pin script must reproduce the proposal lists the REFERENCE's test_anet wrote
    (same detections, same order, values to fp32 exp tolerance), video by video, including the videos whose duration
    cuts segments and the one shorter than a window."""
    import os
    from oracle import fake_heads as FH
    from opental_amd.thumos14 import test_cross_data as X
    fx = np.load(os.path.join(golden_dir, "cross_data.npz"))
    videos, infos = {}, {}
    for name, seed, frames, fps, duration, labels in FH.ANET_VIDEOS:
        videos[name] = torch.from_numpy(np.transpose(FH.synthetic_video(seed, frames), [3, 0, 1, 2]).copy()).cuda()
        infos[name] = {'fps': fps, 'duration': duration, 'frame_num': frames, 'annotations': [{'label': l} for l in labels]}
    for batch_clips, batch_videos in ((32, 8), (1, 1), (3, 2)):
        out = X.test_anet(FH.FakeNet(), videos, infos, IDX_TO_CLASS, clip_length=256, stride=128, conf_thresh=0.01, top_k=5000,
                          nms_sigma=0.5, batch_clips=batch_clips, batch_videos=batch_videos)
        assert sorted(out['results']) == fx["anet_result_keys"].tolist()
        for name in out['results']:
            _check_props(out['results'][name], fx[f"anet_{name}_class"], fx[f"anet_{name}_rows"], name)
            assert all(p['segment'][1] <= infos['v_' + name]['duration'] for p in out['results'][name])


@pytest.mark.gpu
def test_threshold_step_matches_the_reference_run(golden_dir):
    """GPU: the threshold step (threshold.py:71-150) = the inference path over the TRAINING videos + ood_threshold: the
    detections equal the reference run's, and so do the six operating points."""
    import os
    from oracle import fake_heads as FH
    from opental_amd.thumos14 import test as T
    fx = np.load(os.path.join(golden_dir, "cross_data.npz"))
    vids = [torch.from_numpy(np.transpose(FH.synthetic_video(seed, frames), [3, 0, 1, 2]).copy()).cuda()
            for _, seed, frames, _ in FH.THUMOS_TRAIN]
    rows, counts, _, _ = T.detect_batch(FH.FakeNet(), vids, [fps for *_, fps in FH.THUMOS_TRAIN], 256, 128, 0.01, 5000, 0.5)
    result = {}
    for v, (name, *_rest) in enumerate(FH.THUMOS_TRAIN):
        result[name] = T.get_video_detections(rows[v], counts[v], IDX_TO_CLASS, 5000)
        _check_props(result[name], fx[f"train_{name}_class"], fx[f"train_{name}_rows"], name)
    for sc in SCORINGS:
        assert abs(T.ood_threshold(result, sc) - float(fx[f"threshold_{sc}"])) < 5e-6, sc
