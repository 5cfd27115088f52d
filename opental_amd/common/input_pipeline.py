"""Clip preparation on the device, host side.

The reference prepares every clip in numpy inside `THUMOS_Dataset.__getitem__` (AFSD/common/thumos_dataset.py:246-263)
with `RandomCrop` / `RandomHorizontalFlip` / `CenterCrop` (AFSD/common/videotransforms.py:44-124) and uploads 28 MB of
fp32 per clip.  Here the host only slices uint8 frames and takes the random decisions -- with the same sequence of
`random` calls as the reference, so a seeded run reproduces it -- and `otal_prepare_clips` writes the normalised
(B,3,T,Ho,Wo) batch on the GPU from the uint8 upload (4x fewer PCIe bytes).
"""
import ctypes
import random

import numpy as np
import torch

from .. import _lib as L


def sample_crop_flip(h, w, crop, training, rng=random):
    """(i, j, flip): RandomCrop.get_params then RandomHorizontalFlip (training) or CenterCrop (videotransforms.py:54-66,
    :98-101,:119)."""
    if training:
        if w == crop and h == crop:
            i = j = 0
        else:
            i = rng.randint(0, h - crop) if h != crop else 0
            j = rng.randint(0, w - crop) if w != crop else 0
        return i, j, bool(rng.random() < 0.5)
    return int(np.round((h - crop) / 2.)), int(np.round((w - crop) / 2.)), False


_PARAM_DTYPE = np.dtype([("frame0", "<i8"), ("valid_t", "<i4"), ("crop_i", "<i4"), ("crop_j", "<i4"), ("flip", "<i4")])


def prepare_clips(videos, offsets, clip_length=256, crop=96, training=True, device="cuda", rng=random, decisions=None):
    """videos: list of uint8 arrays (T_i,H,W,3) (numpy or torch); offsets: first frame of each clip.
    Returns the fp32 batch (B,3,clip_length,crop,crop) on `device` and the list of (i, j, flip) decisions used."""
    B = len(videos)
    H, W = int(videos[0].shape[1]), int(videos[0].shape[2])
    chunks, recs, used = [], np.zeros(B, _PARAM_DTYPE), []
    pos = 0
    for b, (v, off) in enumerate(zip(videos, offsets)):
        v = torch.as_tensor(v)
        if v.dtype != torch.uint8 or v.dim() != 4 or v.shape[3] != 3 or tuple(v.shape[1:3]) != (H, W):
            raise RuntimeError("videos must be uint8 (T,H,W,3) with one frame size per batch")
        sl = v[off: off + clip_length].contiguous()
        i, j, flip = decisions[b] if decisions is not None else sample_crop_flip(H, W, crop, training, rng)
        used.append((i, j, flip))
        recs[b] = (pos, sl.shape[0], i, j, int(flip))
        chunks.append(sl.reshape(-1))
        pos += sl.numel()
    frames = torch.cat(chunks).to(device, non_blocking=True)                 # the only bulk upload: uint8
    params = torch.from_numpy(recs.view(np.uint8).copy()).to(device, non_blocking=True)
    out = torch.empty((B, 3, clip_length, crop, crop), dtype=torch.float32, device=device)
    L.require_device(frames, params, out)
    L.check(L.lib().otal_prepare_clips(L.ptr(frames), L.ptr(params), L.ptr(out), B, clip_length, H, W, crop, crop,
                                       L.stream()), "otal_prepare_clips")
    return out, used
