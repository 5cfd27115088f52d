"""THUMOS14 clip sampling for training (reference: AFSD/common/thumos_dataset.py), MI355X-first.

Same names and the same host-side arithmetic as the reference -- `get_class_index_map` :13-21, `get_video_info` :24-34,
`get_video_anno` :37-57, `annos_transform` :60-68, `split_videos` :71-131, `load_video_data` :134-141,
`THUMOS_Dataset` :144-275 (incl. the self-supervised clip splice `augment_` :187-228) -- but no pixel is touched on
the host:

  * `load_video_data` keeps the uint8 frames as stored, (T,H,W,3), in PINNED host memory;
  * a sample is a list of DECISIONS -- clip offset, crop corner, flip, and for the ssl branch a 256-entry frame map --
    taken with the reference's exact sequence of `random` calls, so a seeded run draws the same numbers;
  * `ClipStager` slices the clips' uint8 frames straight out of the pinned videos with asynchronous copies on a copy
    stream into one of two device staging buffers (double buffering: batch k+1 travels over PCIe while step k runs), and
    `otal_prepare_clips_map` (csrc/input.hip) writes the normalised (B,3,T,96,96) batch AND the spliced ssl batch from the
    same upload.  The reference builds both clips in numpy / torch-CPU per sample and uploads 2 x 28 MB of fp32.
"""
import ctypes
import math
import os
import random

import numpy as np
import torch

from .. import _lib as L
from .input_pipeline import _PARAM_DTYPE, LabelRecord, max_target_count, sample_crop_flip  # noqa: F401


def get_class_index_map(class_info_path='datasets/thumos14/annotations/Class_Index_Detection.txt'):
    txt = np.loadtxt(class_info_path, dtype=str)
    originidx_to_idx, idx_to_class = {}, {}
    for idx, l in enumerate(np.atleast_2d(txt)):
        originidx_to_idx[int(l[0])] = idx + 1
        idx_to_class[idx + 1] = l[1]
    return originidx_to_idx, idx_to_class


def _read_csv_rows(path):
    import csv
    with open(path, newline='') as f:
        rows = list(csv.reader(f))
    return rows[1:]                      # pandas.read_csv treats the first line as the header


def _num(s):
    try:
        return int(s)
    except ValueError:
        return float(s)


def get_video_info(video_info_path):
    infos = {}
    for r in _read_csv_rows(video_info_path):
        infos[r[0]] = {'fps': _num(r[1]), 'sample_fps': _num(r[2]), 'count': _num(r[3]), 'sample_count': _num(r[4])}
    return infos


def get_video_anno(video_infos, video_anno_path, class_info_path):
    originidx_to_idx, _ = get_class_index_map(class_info_path)
    video_annos = {}
    for r in _read_csv_rows(video_anno_path):
        name, originidx, start_frame, end_frame = r[0], int(r[2]), _num(r[-2]), _num(r[-1])
        ratio = video_infos[name]['sample_count'] * 1.0 / video_infos[name]['count']
        video_annos.setdefault(name, []).append([start_frame * ratio, end_frame * ratio, originidx_to_idx[originidx]])
    return video_annos


def annos_transform(annos, clip_length):
    return [[a[0] * 1.0 / clip_length, a[1] * 1.0 / clip_length, a[2]] for a in annos]


def split_videos(video_infos, video_annos, clip_length=256, stride=30):
    """The sliding-window training list + the per-video minimum action length (thumos_dataset.py:71-131)."""
    training_list, min_anno_dict = [], {}
    for video_name in video_annos.keys():
        min_anno = clip_length
        sample_count = video_infos[video_name]['sample_count']
        annos = video_annos[video_name]
        if sample_count <= clip_length:
            offsetlist = [0]
            min_anno_len = min([x[1] - x[0] for x in annos])
            if min_anno_len < min_anno:
                min_anno = min_anno_len
        else:
            offsetlist = list(range(0, sample_count - clip_length + 1, stride))
            if (sample_count - clip_length) % stride:
                offsetlist += [sample_count - clip_length]
        for offset in offsetlist:
            left, right = offset + 1, offset + clip_length
            cur_annos, save_offset = [], False
            for anno in annos:
                max_l, min_r = max(left, anno[0]), min(right, anno[1])
                ioa = (min_r - max_l) * 1.0 / (anno[1] - anno[0])
                if ioa >= 1.0:
                    save_offset = True
                if ioa >= 0.5:
                    cur_annos.append([max(anno[0] - offset, 1), min(anno[1] - offset, clip_length), anno[2]])
            if len(cur_annos) > 0:
                min_anno_len = min([x[1] - x[0] for x in cur_annos])
                if min_anno_len < min_anno:
                    min_anno = min_anno_len
            if save_offset:
                start, end = np.zeros([clip_length]), np.zeros([clip_length])
                for s, e, _ in cur_annos:
                    d = max((e - s) / 10.0, 2.0)
                    start[np.clip(int(round(s - d / 2.0)), 0, clip_length - 1): np.clip(int(round(s + d / 2.0)), 0, clip_length - 1) + 1] = 1
                    end[np.clip(int(round(e - d / 2.0)), 0, clip_length - 1): np.clip(int(round(e + d / 2.0)), 0, clip_length - 1) + 1] = 1
                training_list.append({'video_name': video_name, 'offset': offset, 'annos': cur_annos, 'start': start, 'end': end})
        min_anno_dict[video_name] = math.ceil(min_anno)
    return training_list, min_anno_dict


def load_video_data(video_infos, npy_data_path, pin=True):
    """uint8 (T,H,W,3) frames per video, as stored on disk, in pinned host memory (asynchronous H2D copies need it;
    the reference transposes to (3,T,H,W) here, thumos_dataset.py:137 -- the device kernel does that instead)."""
    data_dict = {}
    for video_name in video_infos.keys():
        t = torch.from_numpy(np.load(os.path.join(npy_data_path, video_name + '.npy')))
        if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
            raise RuntimeError(f"{video_name}.npy: expected uint8 (T,H,W,3)")
        data_dict[video_name] = t.pin_memory() if (pin and torch.cuda.is_available()) else t
    return data_dict


def get_bg(annos, min_action, clip_length, rng=random):
    """thumos_dataset.py:172-185."""
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


def ssl_splice(annos, th, clip_length=256, rng=random):
    """The decisions of `augment_` (thumos_dataset.py:187-228) WITHOUT touching pixels: (frame_map, new_annos, True), where
    new_clip[:, f] = clip[:, frame_map[f]], or (None, annos, False) when no action / background is long enough.  Draws
    the same `random` numbers in the same order as the reference."""
    try:
        gt = rng.choice(list(filter(lambda x: x[1] - x[0] > 2 * th, annos)))
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
        fmap[t:t + th] = np.arange(start_idx, end_idx)
        fmap[t + th:end_idx] = np.arange(t, start_idx)
        new_annos = [[gt[0], t], [t + th, th + gt[1]], [t + 1, t + th - 1]]
    else:
        fmap[start_idx:t - th] = np.arange(end_idx, t)
        fmap[t - th:t] = np.arange(start_idx, end_idx)
        new_annos = [[gt[0] - th, t - th], [t, gt[1]], [t - th + 1, t - 1]]
    return fmap, new_annos, True


class THUMOS_Dataset:
    """Same constructor arguments as the reference's Dataset (thumos_dataset.py:144-170).  `decide(idx)` is
    `__getitem__` up to the pixels: it returns the decisions of one sample."""

    def __init__(self, data_dict, video_infos, video_annos, clip_length=256, crop_size=96, stride=30, rgb_norm=True,
                 training=True, origin_ratio=0.5):
        self.training_list, self.th = split_videos(video_infos, video_annos, clip_length, stride)
        self.data_dict = data_dict
        self.clip_length, self.crop_size = clip_length, crop_size
        self.rgb_norm, self.training, self.origin_ratio = rgb_norm, training, origin_ratio
        if not rgb_norm:
            raise NotImplementedError("the device kernel normalises (rgb_norm=True, the only setting the reference uses)")

    def __len__(self):
        return len(self.training_list)

    def decide(self, idx, rng=random):
        info = self.training_list[idx]
        video = self.data_dict[info['video_name']]
        H, W = int(video.shape[1]), int(video.shape[2])
        i, j, flip = sample_crop_flip(H, W, self.crop_size, self.training, rng)      # random_flip(random_crop(x)) :255-256
        fmap, ssl_annos, flag = ssl_splice(info['annos'], self.th[info['video_name']], self.clip_length, rng)   # :264
        return {'video': video, 'offset': info['offset'], 'crop': (i, j, flip), 'frame_map': fmap, 'flag': flag,
                'target': np.stack(annos_transform(info['annos'], self.clip_length), 0).astype(np.float32),
                'ssl_target': np.stack(ssl_annos, 0).astype(np.float32),
                'scores': np.stack([info['start'], info['end']], 0).astype(np.float32)}


class ClipStager:
    """Pinned-host -> device staging of uint8 clip frames on a copy stream, double buffered, + the device kernel.

        stager = ClipStager(batch, clip_length, H, W, crop)
        stager.submit(samples)            # decisions of batch k+1: async copies start now, on the copy stream
        clips, ssl_clips = stager.collect()   # on the compute stream: waits for the copies, one prepare launch

    The copies are cudaMemcpyAsync from pinned memory.  `submit` waits on the HOST only for one thing: the H2D copies
    it queued from the same slot two batches ago (`ready[s]`), because it is about to overwrite the pinned records
    (crop / flip / frame maps) those copies read -- a host running more than two batches ahead of the device would
    otherwise pair batch k's frames with batch k+2's decisions.  That wait is over long before it is reached unless
    the host really is that far ahead."""

    def __init__(self, batch, clip_length, H, W, crop, device="cuda", max_targets=None, score_rows=2, copy_stream=None):
        """`max_targets` (input_pipeline.max_target_count(dataset)): the batch's labels -- targets padded to that row count,
        boundary masks, ssl segments -- travel with the frames as ONE fixed-shape record per slot (LabelRecord): one pinned
        buffer, one asynchronous copy; `labels()` hands out the device record of the batch collected last."""
        self.B, self.T, self.H, self.W, self.crop = batch, clip_length, H, W, crop
        self.device = torch.device(device)
        self.labels_host = self.labels_dev = None
        if max_targets is not None:
            self.labels_host = [LabelRecord(batch, max_targets, score_rows, clip_length, pin=True) for _ in range(2)]
            self.labels_dev = [LabelRecord(batch, max_targets, score_rows, clip_length, device=self.device) for _ in range(2)]
        self._last = None
        self.frame_bytes = H * W * 3
        self.stage = [torch.empty(batch * clip_length * self.frame_bytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.params_host = [torch.empty(batch * _PARAM_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.maps_host = [torch.empty(batch * clip_length, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.params_dev = [torch.empty_like(p, device=self.device) for p in self.params_host]
        self.maps_dev = [torch.empty(batch * clip_length, dtype=torch.int32, device=self.device) for _ in range(2)]
        # Which stream carries the copies matters more than it should: HIP hands its few hardware queues (4) to streams in
        # creation order, and a copy stream of its own can land on the queue of the step's main or weight-gradient lane --
        # that lane's launches then queue behind 77 MB of DMA per step (measured by moving the creation order: 948 -> 812
        # clips/s fed, the resident step unchanged; a high-priority stream only moves the collision).  `copy_stream`: the
        # caller's choice -- the training drivers pass the weight-gradient lane's stream, which is idle during the forward
        # pass: the next batch's copies, issued in front of the step's replay, run there and no fifth stream exists.
        self.copy_stream = copy_stream if copy_stream is not None else torch.cuda.Stream(device=self.device)
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self.slot = 0
        self.pending = None
        self._copied = [False, False]       # slot s has H2D copies queued that read params_host[s] / maps_host[s]

    def submit(self, samples):
        if len(samples) != self.B:
            raise RuntimeError("ClipStager: batch size mismatch")
        s = self.slot
        if self._copied[s]:
            self.ready[s].synchronize()     # the queued DMA out of this slot's pinned records has run (see class docstring)
        recs = np.zeros(self.B, _PARAM_DTYPE)
        maps = self.maps_host[s].numpy().reshape(self.B, self.T)
        any_ssl = False
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[s])            # the kernel that read this slot two batches ago is done
            pos = 0
            for b, smp in enumerate(samples):
                v, off = smp['video'], int(smp['offset'])
                if tuple(v.shape[1:]) != (self.H, self.W, 3) or v.dtype != torch.uint8:
                    raise RuntimeError("ClipStager: videos must be uint8 (T,H,W,3) of the configured frame size")
                sl = v[off: off + min(self.T, int(smp.get('valid', self.T)))]     # contiguous frame range of the pinned video
                n = sl.shape[0] * self.frame_bytes
                self.stage[s][pos: pos + n].copy_(sl.reshape(-1), non_blocking=True)
                i, j, flip = smp['crop']
                recs[b] = (pos, sl.shape[0], i, j, int(flip) | (2 if smp.get('pad_half') else 0))    # bit 1: pad with 127.5
                fm = smp.get('frame_map')
                maps[b] = np.arange(self.T, dtype=np.int32) if fm is None else fm
                any_ssl = any_ssl or fm is not None
                pos += self.T * self.frame_bytes
            self.params_host[s].numpy()[:] = recs.view(np.uint8)
            self.params_dev[s].copy_(self.params_host[s], non_blocking=True)
            self.maps_dev[s].copy_(self.maps_host[s], non_blocking=True)     # 1 KB per clip: always (collect may ask for ssl)
            if self.labels_host is not None and 'target' in samples[0]:
                self.labels_host[s].fill(samples)
                self.labels_dev[s].flat.copy_(self.labels_host[s].flat, non_blocking=True)
            self.ready[s].record(self.copy_stream)
        self._copied[s] = True
        self.pending = (s, any_ssl)
        self.slot ^= 1

    def collect(self, want_ssl=None, out=None):
        """-> (clips, ssl_clips or None), fp32 (B,3,T,crop,crop) on the device, produced on the CURRENT stream.
        `out`: write the clips THERE (the input buffer a captured step replays from, DetectorTrainer.static_inputs():
        the batch then needs no device-to-device copy)."""
        if self.pending is None:
            raise RuntimeError("ClipStager.collect without submit")
        s, any_ssl = self.pending
        self.pending = None
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.ready[s])
        shape = (self.B, 3, self.T, self.crop, self.crop)
        if out is not None and (tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous()):
            raise RuntimeError("ClipStager.collect: `out` must be a contiguous fp32 (B,3,T,crop,crop) tensor")
        clips = torch.empty(shape, dtype=torch.float32, device=self.device) if out is None else out
        ssl = torch.empty(shape, dtype=torch.float32, device=self.device) if (any_ssl if want_ssl is None else want_ssl) else None
        L.check(L.lib().otal_prepare_clips_map(L.ptr(self.stage[s]), L.ptr(self.params_dev[s]),
                                               L.ptr(self.maps_dev[s]) if ssl is not None else None, L.ptr(clips),
                                               L.ptr(ssl) if ssl is not None else None, self.B, self.T, self.H, self.W,
                                               self.crop, self.crop, L.stream()), "otal_prepare_clips_map")
        self.consumed[s].record(cur)
        self._last = s
        return clips, ssl

    def labels(self):
        """The device LabelRecord of the batch collected last (None without `max_targets`).  It is a slot of the double
        buffer: call `release()` once the step that reads it has been issued."""
        return None if self.labels_dev is None or self._last is None else self.labels_dev[self._last]

    def release(self):
        """The consumer of the last collected slot (frames AND labels) has been issued on the current stream: the slot may
        be refilled once that work has run.  (collect() marks the frames consumed; a step reads the label record later.)"""
        if self._last is not None:
            self.consumed[self._last].record(torch.cuda.current_stream(self.device))


def batches(dataset, batch_size, shuffle=True, drop_last=True, generator=None, rng=random):
    """Index batches the way DataLoader(shuffle=True, drop_last=True) draws them (train.py:343-346): one torch.randperm
    per epoch; the per-sample decisions come from `rng` in batch order (single stream: the reference's four worker
    processes each own a `random` state seeded GLOBAL_SEED + worker_id, train.py:72-76)."""
    n = len(dataset)
    order = torch.randperm(n, generator=generator).tolist() if shuffle else list(range(n))
    for k in range(0, n - (batch_size - 1 if drop_last else 0), batch_size):
        idx = order[k:k + batch_size]
        if len(idx) < batch_size and drop_last:
            break
        yield [dataset.decide(i, rng) for i in idx]
