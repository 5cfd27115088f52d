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


class PaddedTargets:
    """Fixed-shape form of a batch's targets: rows (B,G,3) float32 [start, end, label] zero padded, validity (B,G) uint8.
    MultiSegmentLoss (both recipes) takes it wherever the reference takes the list of ragged (n_i,3) arrays."""
    __slots__ = ('gt', 'valid')

    def __init__(self, gt, valid):
        self.gt, self.valid = gt, valid

    def __iter__(self):
        yield self.gt
        yield self.valid


class LabelRecord:
    """The labels of ONE batch in ONE flat buffer of fixed shape -- what the captured training step replays from.

    The reference hands `run_one_epoch` a list of per-sample (n_i,3) target arrays of any length plus the (B,2,T) boundary
    masks (AFSD/thumos14/train.py:204-252, AFSD/common/thumos_dataset.py:278-300 `detection_collate`).  Here they travel as
    ONE record whose layout does not depend on the n_i:

        gt      float32 (B, G, 3)   [start, end, label] rows, zero padded to G = `max_targets`
        scores  float32 (B, R, T)   boundary masks (R = 2: [start, end]; 3 for ActivityNet: [action, start, end])
        ssl     float32 (B, 3, 2)   anchor / positive / negative segments of the self-supervised branch (frames)
        valid   uint8   (B, G)      1 for the first n_i rows of a sample

    so a batch is one pinned host buffer, one asynchronous H2D copy and -- into the captured step's input record -- one
    device-to-device copy; the loss kernels read `valid` as their row mask (otal_detection_loss, `gvalid`), which makes the
    padded rows inert: results are bit-identical to the ragged form."""

    def __init__(self, batch, max_targets, score_rows, clip_length, device="cpu", pin=False):
        self.B, self.G, self.R, self.T = int(batch), int(max_targets), int(score_rows), int(clip_length)
        n_gt, n_sc, n_ssl = self.B * self.G * 3, self.B * self.R * self.T, self.B * 6
        self._float_count = n_gt + n_sc + n_ssl
        nbytes = 4 * self._float_count + (self.B * self.G + 15) // 16 * 16
        self.flat = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        if pin:
            self.flat = self.flat.pin_memory()
        f = self.flat[:4 * self._float_count].view(torch.float32)
        self.gt = f[:n_gt].view(self.B, self.G, 3)
        self.scores = f[n_gt:n_gt + n_sc].view(self.B, self.R, self.T)
        self.ssl = f[n_gt + n_sc:].view(self.B, 3, 2)
        self.valid = self.flat[4 * self._float_count: 4 * self._float_count + self.B * self.G].view(self.B, self.G)

    @property
    def targets(self):
        """What MultiSegmentLoss takes in place of the list of ragged arrays."""
        return PaddedTargets(self.gt, self.valid)

    def ssl_targets(self):
        return [self.ssl[b] for b in range(self.B)]

    def fill(self, samples):
        """Host side: write the batch's labels (numpy) into this (CPU) record."""
        gt, valid, sc, ssl = self.gt.numpy(), self.valid.numpy(), self.scores.numpy(), self.ssl.numpy()
        gt[:] = 0.0
        valid[:] = 0
        for b, s in enumerate(samples):
            t = np.asarray(s['target'], np.float32).reshape(-1, 3)
            if t.shape[0] > self.G:
                raise RuntimeError(f"LabelRecord: a sample has {t.shape[0]} targets, the record holds {self.G} "
                                   "(max_targets must cover the dataset's longest target list)")
            gt[b, :t.shape[0]] = t
            valid[b, :t.shape[0]] = 1
            sc[b] = s['scores']
            # (anchor / positive / negative rows exist only where the splice succeeded; otherwise the datasets hand the
            # plain annotations through, thumos_dataset.py:264-270)
            st = s.get('ssl_target') if s.get('flag', True) else None
            st = None if st is None else np.asarray(st, np.float32)
            ssl[b] = st[:, :2] if (st is not None and st.shape[0] == 3) else 0.0


def max_target_count(dataset, multiple=4):
    """The longest target list any window of `dataset` carries (its `training_list[i]['annos']`), rounded up: the G of the
    label records of a run."""
    n = max((len(info['annos']) for info in dataset.training_list), default=1)
    return max(multiple, (n + multiple - 1) // multiple * multiple)
