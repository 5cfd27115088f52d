"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the reference's clip preparation.

Follows AFSD/common/thumos_dataset.py:136-137 (THWC -> CTHW), :246-252 (temporal slice + zero padding to clip_length),
:255-258 (random crop + random flip in training, centre crop otherwise), :261-263 (float, (x/255)*2-1) and
AFSD/common/videotransforms.py:44-72 (RandomCrop), :86-103 (CenterCrop), :106-124 (RandomHorizontalFlip).
Pinned against the reference's own transform classes by oracle/pin_input_pipeline.py -> tests/golden/input_pipeline.npz.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import random

import numpy as np


def sample_params(h, w, crop, training, rng=random):
    """(i, j, flip) with the reference's sequence of `random` calls (RandomCrop.get_params, then RandomHorizontalFlip)."""
    if training:
        if w == crop and h == crop:
            i = j = 0
        else:
            i = rng.randint(0, h - crop) if h != crop else 0
            j = rng.randint(0, w - crop) if w != crop else 0
        flip = rng.random() < 0.5
        return i, j, bool(flip)
    return int(np.round((h - crop) / 2.)), int(np.round((w - crop) / 2.)), False


def prepare_clip(video_thwc, offset, clip_length, crop, i, j, flip, valid=None, pad_value=0.0):
    """video (T,H,W,3) uint8 -> (3, clip_length, crop, crop) float32.  `valid` / `pad_value=127.5`: the ActivityNet loader
    (anet_dataset.py:219-229) keeps min(frame_num, clip_length) frames and pads with 127.5 in floating point."""
    data = np.transpose(video_thwc, [3, 0, 1, 2])                       # thumos_dataset.py:137
    x = data[:, offset: offset + (clip_length if valid is None else valid)]
    c, t, h, w = x.shape
    if t < clip_length:                                                  # :248-252
        if pad_value:
            x = np.concatenate([x.astype(np.float64), np.ones([c, clip_length - t, h, w], np.float64) * pad_value], 1)
        else:
            x = np.concatenate([x, np.zeros([c, clip_length - t, h, w], x.dtype)], 1)
    x = x[:, :, i:i + crop, j:j + crop]                                  # videotransforms.py:67-69
    if flip:
        x = np.flip(x, axis=3).copy()                                    # :119-121
    x = x.astype(np.float32)                                             # torch .float()
    return ((x / np.float32(255.0)) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)   # :262-263
