"""ActivityNet1.3 clip sampling for training (reference: AFSD/common/anet_dataset.py), MI355X-first.

Same names and the same host-side arithmetic as the reference -- `load_json` :11-18, `annos_transform` :21-29,
`get_video_info` :32-40, `split_videos` :43-105 (one 768-frame clip per video; the [action, start, end] score rows carry
the label id, as the reference writes them), `ANET_Dataset` :129-257 incl. the self-supervised splice `augment_`
:171-209 -- and, as for THUMOS14 (common/thumos_dataset.py), no pixel is touched on the host: a sample is a list of
DECISIONS taken with the reference's exact sequence of `random` calls (crop corner, flip, the splice's frame map), the
uint8 frames cross PCIe from pinned memory on a copy stream, and `otal_prepare_clips_map` writes the normalised batch
and the spliced ssl batch on the device.  Clips shorter than clip_length are padded with 127.5 BEFORE normalisation
(:226-229), i.e. exactly 0.0 after it: bit 1 of the clip record's `flip` word selects that padding in the kernel.
"""
import json
import concurrent.futures
import math
import threading
import os
import random

import numpy as np
import torch

from .input_pipeline import sample_crop_flip

PAD_HALF = 2        # ClipParams.flip bit 1: pad frames are 127.5 before normalisation (0.0 after), not 0 (-1.0 after)


def load_json(file):
    with open(file) as json_file:
        return json.load(json_file)


def annos_transform(annos, clip_length):
    return [[a[0] * 1.0 / clip_length, a[1] * 1.0 / clip_length, a[2]] for a in annos]


def get_video_info(video_info_path, subset='training'):
    json_data = load_json(video_info_path)
    return {name: tmp for name, tmp in json_data.items() if tmp['subset'] == subset}


def split_videos(video_info, clip_length, video_dir, binary_class=False):
    """anet_dataset.py:43-105: one training sample per video that has an .npy file and at least one valid annotation."""
    training_list, min_anno_dict = [], {}
    for video_name in list(video_info.keys()):
        if not os.path.exists(os.path.join(video_dir, video_name + '.npy')):
            continue
        frame_num = min(video_info[video_name]['frame_num'], clip_length)
        annos = []
        min_anno = clip_length
        for anno in video_info[video_name]['annotations']:
            if binary_class:
                anno['label_id'] = 1 if anno['label_id'] > 0 else 0
            if anno['end_frame'] <= anno['start_frame']:
                continue
            annos.append([anno['start_frame'], anno['end_frame'], anno['label_id']])
        if len(annos) == 0:
            continue
        cur_annos = [[a[0], a[1], a[2]] for a in annos]
        min_anno_len = min(x[1] - x[0] for x in cur_annos)
        if min_anno_len < min_anno:
            min_anno = min_anno_len
        start, end, action = np.zeros([clip_length]), np.zeros([clip_length]), np.zeros([clip_length])
        for s, e, cid in cur_annos:
            d = max((e - s) / 10.0, 2.0)
            action[np.clip(int(round(s)), 0, clip_length - 1): np.clip(int(round(e)), 0, clip_length - 1) + 1] = cid
            start[np.clip(int(round(s - d / 2)), 0, clip_length - 1): np.clip(int(round(s + d / 2)), 0, clip_length - 1) + 1] = cid
            end[np.clip(int(round(e - d / 2)), 0, clip_length - 1): np.clip(int(round(e + d / 2)), 0, clip_length - 1) + 1] = cid
        training_list.append({'video_name': video_name, 'offset': 0, 'annos': cur_annos, 'frame_num': frame_num,
                              'start': start, 'end': end, 'action': action})
        min_anno_dict[video_name] = math.floor(min_anno)
    return training_list, min_anno_dict


def get_bg(annos, min_action, clip_length, rng=random):
    """anet_dataset.py:155-169."""
    annos = [[a[0], a[1]] for a in annos]
    times = []
    for a in annos:
        times.extend(a)
    times.extend([0, clip_length - 1])
    times.sort()
    regions = [[times[i], times[i + 1]] for i in range(len(times) - 1)]
    regions = list(filter(lambda x: x not in annos and math.floor(x[1]) - math.ceil(x[0]) > min_action, regions))
    region = rng.choice(regions)
    return [math.ceil(region[0]), math.floor(region[1])]


def _assign(fmap, dst, src, clip_length):
    """`new[:, dst] = old[:, src]` on frame indices with torch's slice semantics: False where torch raises (the reference
    catches that RuntimeError and gives the splice up, anet_dataset.py:194-207)."""
    d = range(clip_length)[slice(*dst)]
    s = range(clip_length)[slice(*src)]
    if len(s) != len(d) and len(s) != 1:
        return False
    if len(d):
        fmap[d.start:d.stop] = np.arange(s.start, s.stop) if len(s) == len(d) else s.start
    return True


def ssl_splice(annos, th, clip_length=768, rng=random):
    """The decisions of `augment_` (anet_dataset.py:171-209) WITHOUT touching pixels: (frame_map, new_annos, True) with
    new_clip[:, f] = clip[:, frame_map[f]], or (None, annos, False).  Differences from the THUMOS14 splice: an action of
    exactly 2 * th frames qualifies (>=), and slice-shape mismatches give the splice up instead of raising."""
    try:
        gt = rng.choice(list(filter(lambda x: x[1] - x[0] >= 2 * th, annos)))
    except IndexError:
        return None, annos, False
    gt_len = gt[1] - gt[0]
    region = range(math.floor(th), math.ceil(gt_len - th))
    t = rng.choice(region) + math.ceil(gt[0])
    try:
        bg = get_bg(annos, th, clip_length, rng)
    except IndexError:
        return None, annos, False
    start_idx = rng.choice(range(bg[1] - bg[0] - th)) + bg[0]
    end_idx = start_idx + th
    fmap = np.arange(clip_length, dtype=np.int32)
    if gt[1] < start_idx:
        ok = _assign(fmap, (t, t + th), (start_idx, end_idx), clip_length) and \
            _assign(fmap, (t + th, end_idx), (t, start_idx), clip_length)
        new_annos = [[gt[0], t], [t + th, th + gt[1]], [t + 1, t + th - 1]]
    else:
        ok = _assign(fmap, (start_idx, t - th), (end_idx, t), clip_length) and \
            _assign(fmap, (t - th, t), (start_idx, end_idx), clip_length)
        new_annos = [[gt[0] - th, t - th], [t, gt[1]], [t - th + 1, t - 1]]
    if not ok:
        return None, annos, False
    return fmap, new_annos, True


class LazyVideo:
    """What a decision record carries instead of the pixels: the video's name and shape (read from the .npy HEADER only).
    The frames are loaded when the stager slices it -- `v[a:b]` -- through the dataset's bounded cache, so deciding a whole
    epoch up front (thumos14.train.run_one_epoch materialises every rank's batch list) touches no pixel data."""
    dtype = torch.uint8

    def __init__(self, dataset, name, shape):
        self._ds, self.name, self.shape = dataset, name, tuple(shape)

    def dim(self):
        return len(self.shape)

    def __getitem__(self, idx):
        return self._ds.video(self.name)[idx]


# set by DetectorTrainer while it captures a step (thumos14/train.py): background readers do not pin host memory meanwhile
CAPTURE_IN_PROGRESS = [False]


class ANET_Dataset:
    """Same constructor arguments as the reference's Dataset (anet_dataset.py:129-153).  `decide(idx)` is `__getitem__`
    up to the pixels.  Videos are read from <video_dir>/<name>.npy (uint8, (T,H,W,3)) when a batch is SUBMITTED and kept in a
    bounded least-recently-used cache (`cache_videos`, default 64 = ~1.9 GB of 768 x 112 x 112 x 3 videos; the reference
    re-reads the file for every sample, :218; ActivityNet1.3 has ~10 k training videos = ~290 GB, which an unbounded cache
    -- the THUMOS14 pattern of ~200 videos -- would try to hold on every rank)."""

    def __init__(self, video_info_path, video_dir, clip_length, crop_size, stride, channels=3, rgb_norm=True, training=True,
                 binary_class=False, pin=False, cache_videos=64):
        self.training = training
        video_info = get_video_info(video_info_path, 'training' if training else 'validation')
        self.training_list, self.th = split_videos(video_info, clip_length, video_dir, binary_class)
        self.clip_length, self.crop_size, self.rgb_norm = clip_length, crop_size, rgb_norm
        self.video_dir, self.channels = video_dir, channels
        self._pin = pin
        self._cache = {}                # name -> tensor, in least-recently-used order (dicts keep insertion order)
        self._cap = max(1, int(cache_videos))
        self._shapes = {}
        # background reads (prefetch()): the reference hides the ~29 MB np.load per sample in DataLoader workers; here the epoch
        # loop names the NEXT batch's videos while the current step runs and two reader threads bring them into the cache
        self._lock = threading.Lock()
        self._pending = {}              # name -> Future of the tensor
        self._pool = None
        if not rgb_norm:
            raise NotImplementedError("the device kernel normalises (rgb_norm=True, the only setting the reference uses)")

    def __len__(self):
        return len(self.training_list)

    def video_shape(self, name):
        """(T, H, W, 3) from the file header: np.load(mmap_mode='r') maps the file without reading the frames."""
        shp = self._shapes.get(name)
        if shp is None:
            m = np.load(os.path.join(self.video_dir, name + '.npy'), mmap_mode='r')
            if m.dtype != np.uint8 or m.ndim != 4 or m.shape[3] != 3:
                raise RuntimeError(f"{name}.npy: expected uint8 (T,H,W,3)")
            shp = self._shapes[name] = tuple(int(d) for d in m.shape)
            del m
        return shp

    def _read(self, name):
        v = torch.from_numpy(np.load(os.path.join(self.video_dir, name + '.npy')))
        if v.dtype != torch.uint8 or v.dim() != 4 or v.shape[3] != 3:
            raise RuntimeError(f"{name}.npy: expected uint8 (T,H,W,3)")
        # (pinning = hipHostMalloc: done by the thread that reads -- the main thread when it waits in video(), a reader thread
        #  otherwise -- but never while the main thread captures the step: hipHostMalloc is not legal during a stream capture
        #  in relaxed mode on every runtime, so background readers hand over pageable frames then and the stager's own pinned
        #  slots, which every frame is copied into at submit time anyway, carry the upload)
        if self._pin and torch.cuda.is_available() and not (threading.current_thread() is not threading.main_thread()
                                                            and CAPTURE_IN_PROGRESS[0]):
            v = v.pin_memory()
        return v

    def prefetch(self, names):
        """Start reading the named videos in the background (at most cache_videos / 2 outstanding); video(name) then finds
        them in the cache or waits for the read that is already under way.  Called by the epoch loop with the next batch's
        names while the current step runs (thumos14.train.run_one_epoch)."""
        with self._lock:
            if self._pool is None:
                self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=2, thread_name_prefix="anet-video")
            # finished reads nobody came for (the epoch ended, --max_steps cut the loop) and that this request does not name:
            # their pinned frames would be held for good and count against the cap until prefetching stops silently.  A read
            # that FAILED is reported here -- video() would only have raised it had somebody asked for that video
            want = set(names)
            for old_name in [n for n, f in self._pending.items() if n not in want and f.done()]:
                fut = self._pending.pop(old_name)
                if fut.exception() is not None:
                    import warnings
                    warnings.warn(f"background read of {old_name}.npy failed: {fut.exception()!r}")
            for name in dict.fromkeys(names):
                if name in self._cache or name in self._pending or len(self._pending) >= max(1, self._cap // 2):
                    continue
                self._pending[name] = self._pool.submit(self._read, name)

    def video(self, name):
        with self._lock:
            v = self._cache.pop(name, None)
            fut = self._pending.pop(name, None) if v is None else None
        if v is None:
            v = fut.result() if fut is not None else self._read(name)       # (a failed background read raises here)
        with self._lock:
            while len(self._cache) >= self._cap:            # evict the least recently used (the stager copies frames into
                self._cache.pop(next(iter(self._cache)))    # its own pinned slots at submit time: nothing points here later)
            self._cache[name] = v
        return v

    def decide(self, idx, rng=random):
        info = self.training_list[idx]
        video = LazyVideo(self, info['video_name'], self.video_shape(info['video_name']))
        th = int(self.th[info['video_name']] / 4)                   # :216
        offset = info['offset']
        valid = min(offset + self.clip_length, info['frame_num']) - offset      # :219-221
        H, W = int(video.shape[1]), int(video.shape[2])
        i, j, flip = sample_crop_flip(H, W, self.crop_size, self.training, rng)     # :232-235
        fmap, ssl_annos, flag = ssl_splice(info['annos'], th, self.clip_length, rng)   # augment(..., max_iter=1), :240
        return {'video': video, 'offset': offset, 'valid': valid, 'crop': (i, j, flip), 'pad_half': True,
                'frame_map': fmap, 'flag': flag,
                'target': np.stack(annos_transform(info['annos'], self.clip_length), 0).astype(np.float32),
                'ssl_target': np.stack(ssl_annos, 0).astype(np.float32),
                'scores': np.stack([info['action'], info['start'], info['end']], 0).astype(np.float32)}
