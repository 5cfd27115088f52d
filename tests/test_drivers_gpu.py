"""GPU: the end-to-end drivers (VERDICT r1 missing #4; reference AFSD/thumos14/train.py:306-363, test.py:203-288) on a
small synthetic THUMOS14-layout dataset: config file -> dataset csv / npy -> pinned staging -> DetectorTrainer epochs
(ssl branch included) -> save_model -> --resume -> test driver -> result JSON -> open-set evaluation."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
FLAGS = ['--open_set', '--split', '0', '--lw', '1', '--cw', '10', '--piou', '0.5', '--ssl', '0.001', '--batch_size', '2']


@pytest.fixture(autouse=True)
def _restore_precision():
    """The drivers select the GEMM operand type for the process (OTAL_DTYPE, default bf16): put it back for the other tests."""
    from opental_amd.common import ops
    old = ops.CONV_PRECISION
    yield
    ops.CONV_PRECISION = old


def test_train_resume_and_test_drivers(tmp_path):
    from make_synthetic_thumos import make
    from opental_amd.thumos14 import train as R, test as T
    yaml_path = make(str(tmp_path / "data"), videos=2, frames=400, size=100)
    common = [yaml_path] + FLAGS + ['--random_init', '--save_after', '0', '--max_steps', '3']
    # ---- three epochs in one go
    tr_a, hist_a = R.main(common + ['--max_epoch', '3', '--checkpoint_path', str(tmp_path / "run_a")])
    assert len(hist_a) == 3 and all(np.isfinite(h).all() for h in hist_a)
    assert tr_a.step_count == 9
    for e in (1, 2, 3):
        assert os.path.exists(tmp_path / "run_a" / f"checkpoint-{e}.ckpt")
        assert os.path.exists(tmp_path / "run_a" / "training" / f"checkpoint_{e}.ckpt")
    # ---- two epochs, then a NEW process-level run resumed from epoch 2: same final weights, bit for bit
    R.main(common + ['--max_epoch', '2', '--checkpoint_path', str(tmp_path / "run_b")])
    tr_b, hist_b = R.main(common + ['--max_epoch', '3', '--resume', '2', '--checkpoint_path', str(tmp_path / "run_b")])
    assert len(hist_b) == 1 and tr_b.step_count == 9
    assert torch.equal(tr_a.arena.flat, tr_b.arena.flat)
    assert hist_b[0] == hist_a[2]
    # the ssl branch really ran in some steps (flags[0]) -- its weight is tiny, so check the sampler instead
    from opental_amd.common import thumos_dataset as D
    info = D.get_video_info(str(tmp_path / "data" / "train_info.csv"))
    anno = D.get_video_anno(info, str(tmp_path / "data" / "train_anno.csv"), str(tmp_path / "data" / "classes.txt"))
    lst, th = D.split_videos(info, anno, 256, 30)
    import random
    random.seed(0)
    assert any(D.ssl_splice(s['annos'], th[s['video_name']])[2] for s in lst)
    # ---- the test driver on the saved checkpoint, then the open-set evaluation of its JSON
    known = tmp_path / "known.txt"
    known.write_text(open(tmp_path / "data" / "classes.txt").read())
    out_file, metrics = T.main([yaml_path, '--open_set', '--split', '0', '--checkpoint_path', str(tmp_path / "run_a" / "checkpoint-3.ckpt"),
                                '--evaluate', str(tmp_path / "data" / "gt_open.json"), str(known)])
    res = json.load(open(out_file))
    assert res['version'] == 'THUMOS14' and sorted(res['results']) == ['video_test_0000000', 'video_test_0000001']
    for props in res['results'].values():
        for p in props[:50]:
            assert set(p) == {'label', 'score', 'segment', 'uncertainty', 'actionness'} and len(p['segment']) == 2
    assert metrics is None or all(np.isfinite(np.asarray(v)).all() for v in metrics.values())
