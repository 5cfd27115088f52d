"""Pins oracle/input_ref.py against the reference's own transform classes (imported from /root/reference) and writes
tests/golden/input_pipeline.npz: videos, the (offset, i, j, flip) decisions and the clips the REFERENCE code produced."""
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"       # child processes too (joblib workers of the evaluation code)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                  # noqa: E402
from AFSD.common import videotransforms as VT               # noqa: E402  (no config import needed)
from oracle import input_ref as R                             # noqa: E402


def reference_clip(video_thwc, offset, clip_length, crop, training):
    """The lines of THUMOS_Dataset.__getitem__ (thumos_dataset.py:246-263) on the reference's transform objects."""
    data = np.transpose(video_thwc, [3, 0, 1, 2])
    input_data = data[:, offset: offset + clip_length]
    c, t, h, w = input_data.shape
    if t < clip_length:
        input_data = np.concatenate([input_data, np.zeros([c, clip_length - t, h, w], input_data.dtype)], 1)
    if training:
        input_data = VT.RandomHorizontalFlip()(VT.RandomCrop(crop)(input_data))
    else:
        input_data = VT.CenterCrop(crop)(input_data)
    input_data = torch.from_numpy(np.ascontiguousarray(input_data)).float()
    return ((input_data / 255.0) * 2.0 - 1.0).numpy()


def main():
    rs = np.random.RandomState(7)
    out = {}
    cases = [(40, 14, 14, 12, 5, 16, True), (10, 14, 14, 12, 3, 16, True), (30, 12, 16, 12, 20, 16, True),
             (20, 14, 14, 12, 0, 16, False), (9, 13, 13, 8, 2, 8, True), (33, 14, 14, 12, 8, 16, True)]
    for n, (T, H, W, crop, offset, L, training) in enumerate(cases):
        video = rs.randint(0, 256, (T, H, W, 3)).astype(np.uint8)
        random.seed(100 + n)
        state = random.getstate()
        ref = reference_clip(video, offset, L, crop, training)
        random.setstate(state)
        i, j, flip = R.sample_params(H, W, crop, training)
        mine = R.prepare_clip(video, offset, L, crop, i, j, flip)
        assert mine.dtype == np.float32 and mine.shape == ref.shape
        assert np.array_equal(mine, ref), n
        out[f"video{n}"] = video
        out[f"params{n}"] = np.array([offset, L, crop, i, j, int(flip), int(training)], np.int64)
        out[f"clip{n}"] = ref
    out["n"] = np.array(len(cases))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "input_pipeline.npz")
    np.savez_compressed(path, **out)
    print("pinned", len(cases), "cases ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
