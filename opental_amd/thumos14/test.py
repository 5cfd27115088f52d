"""Inference path of OpenTAL/AFSD on MI355X, with the reference's function names
(AFSD/thumos14/test.py): get_offsets (:48-56), prepare_clip (:67-76), parse_output (:79-109),
decode_predictions (:112-140), filtering (:143-162), get_video_detections (:165-200).

`detect_batch` is the MI355X-first entry point: all sliding windows of a batch of videos go through
the network in large batches (the reference runs b=1, test.py:227-235), then TWO launches do the
rest -- otal_decode_clips (decode + per-class threshold for every clip) and otal_softnms_classes
(gather + Soft-NMS for every (video, class)) -- with no host synchronisation until the final copy.
Multi-GPU: shard the video list across ranks (as the reference's unused AFSD/anet/test.py:248-273
sketches); there is no collective on this path.
"""
import ctypes

import torch

from .. import _lib as L


def get_offsets(sample_count, clip_length, stride):
    """test.py:48-56 with the video's sample count passed in."""
    if sample_count < clip_length:
        return [0]
    out = list(range(0, sample_count - clip_length + 1, stride))
    if (sample_count - clip_length) % stride:
        out += [sample_count - clip_length]
    return out


def prepare_clip(data, offset, clip_length):
    """uint8 (C,T,H,W) device tensor -> (1,C,clip_length,H,W) float in [-1,1], zero padded (test.py:67-76)."""
    clip = data[:, offset: offset + clip_length].float()
    clip = (clip / 255.0) * 2.0 - 1.0
    if clip.size(1) < clip_length:
        pad = torch.zeros([clip.size(0), clip_length - clip.size(1), clip.size(2), clip.size(3)], device=clip.device)
        clip = torch.cat([clip, pad], dim=1)
    return clip.unsqueeze(0)


_WINDOW_DTYPE = None


def prepare_windows(videos, windows, clip_length):
    """All windows of one forward pass in ONE launch (otal_prepare_windows): videos = uint8 (C,Tv,H,W) device tensors,
    windows = [(video index, offset)].  Bit-identical to torch.cat([prepare_clip(videos[v], o, clip_length) ...])."""
    import numpy as np
    global _WINDOW_DTYPE
    if _WINDOW_DTYPE is None:
        _WINDOW_DTYPE = np.dtype([("src", "<u8"), ("chan_stride4", "<i4"), ("valid_t", "<i4")])
    C, _, H, W = videos[windows[0][0]].shape
    recs = np.zeros(len(windows), _WINDOW_DTYPE)
    for i, (v, o) in enumerate(windows):
        d = videos[v]
        if d.dtype != torch.uint8 or not d.is_cuda or not d.is_contiguous() or tuple(d.shape[2:]) != (H, W) or d.shape[0] != C:
            raise RuntimeError("videos must be contiguous uint8 (C,T,H,W) device tensors of one frame size")
        if (H * W) % 4 or d.data_ptr() % 4 or not 0 <= o < d.shape[1]:
            raise RuntimeError("otal_prepare_windows needs H*W % 4 == 0, 4-byte aligned videos and offsets inside the video")
        recs[i] = (d.data_ptr() + o * H * W, d.shape[1] * H * W // 4, min(clip_length, d.shape[1] - o))
    dev = videos[windows[0][0]].device
    params = torch.from_numpy(recs.view(np.uint8).copy()).to(dev, non_blocking=True)
    out = torch.empty((len(windows), C, clip_length, H, W), dtype=torch.float32, device=dev)
    L.check(L.lib().otal_prepare_windows(L.ptr(params), L.ptr(out), len(windows), C, clip_length, H, W, L.stream()),
            "otal_prepare_windows")
    return out


def decode_clips(output_dict, offsets, fps, clip_length=256, conf_thresh=0.01):
    """Batched parse_output + decode_predictions + threshold masks.  output_dict: model outputs for
    `n` clips; offsets/fps: per-clip tensors or lists.  Returns dict(seg, score, unct, actn, flag)."""
    loc = output_dict['loc'].contiguous()
    n, A, _ = loc.shape
    K = output_dict['conf'].shape[-1]
    dev = loc.device
    offs = torch.as_tensor(offsets, dtype=torch.float32, device=dev).contiguous()
    fpst = torch.as_tensor(fps, dtype=torch.float32, device=dev).contiguous()
    if fpst.numel() == 1:
        fpst = fpst.expand(n).contiguous()
    seg = torch.empty((n, A, 2), device=dev)
    score = torch.empty((n, K, A), device=dev)
    unct = torch.empty((n, A), device=dev)
    actn = torch.empty((n, A), device=dev)
    flag = torch.empty((n, K, A), dtype=torch.uint8, device=dev)
    t = lambda k: output_dict[k].contiguous()
    L.check(L.lib().otal_decode_clips(L.ptr(loc), L.ptr(t('prop_loc')), L.ptr(output_dict['priors'].contiguous()),
                                      L.ptr(t('conf')), L.ptr(t('prop_conf')), L.ptr(t('center')), L.ptr(t('act')),
                                      L.ptr(t('prop_act')), L.ptr(offs), L.ptr(fpst), L.ptr(seg), L.ptr(score),
                                      L.ptr(unct), L.ptr(actn), L.ptr(flag), n, A, K, ctypes.c_float(clip_length),
                                      ctypes.c_float(conf_thresh), L.stream()), "otal_decode_clips")
    return dict(seg=seg, score=score, unct=unct, actn=actn, flag=flag)


def decode_predictions(output_dict, idx, offset, sample_fps, clip_length=256):
    """Single-clip view with the reference's return values (test.py:112-140):
    decoded_segments (A,2), conf_scores (K,A), uncertainty (A,), actionness (A,)."""
    one = {k: (v[idx:idx + 1] if (v is not None and k != 'priors') else v) for k, v in output_dict.items()}
    d = decode_clips(one, [float(offset)], [float(sample_fps)], clip_length)
    return d['seg'][0], d['score'][0], d['unct'][0], d['actn'][0]


def filtering(decoded_segments, conf_score_cls, uncertainty, actionness, conf_thresh, use_edl=True, os_head=True):
    """test.py:143-162 for one class: (n,5) rows [start,end,score,unct,act] or None."""
    m = (conf_score_cls > conf_thresh) & (actionness > 0.5)
    if int(m.sum()) == 0:
        return None
    return torch.cat([decoded_segments[m], conf_score_cls[m, None], uncertainty[m, None], actionness[m, None]], -1)


def softnms_classes(dec, clip_start, top_k=5000, sigma=0.5, score_threshold=0.001):
    """All (video, class) Soft-NMS problems in one launch.  clip_start: per-video clip ranges (V+1).
    Returns rows (V,K,top_k,5), counts (V,K), index (V,K,top_k)."""
    n, K, A = dec['score'].shape
    dev = dec['score'].device
    cs = torch.as_tensor(clip_start, dtype=torch.int32, device=dev).contiguous()
    V = cs.numel() - 1
    starts = [int(v) for v in clip_start]
    max_clips = max(b - a for a, b in zip(starts[:-1], starts[1:]))
    tk = min(int(top_k), max_clips * A)
    out = torch.zeros((V, K, tk, 5), device=dev)
    counts = torch.zeros((V, K), dtype=torch.int32, device=dev)
    index = torch.zeros((V, K, tk), dtype=torch.int32, device=dev)
    lib = L.lib()
    lib.otal_softnms_scratch_bytes.restype = ctypes.c_size_t
    nbytes = int(lib.otal_softnms_scratch_bytes(int(n), int(max_clips), int(A), int(K)))   # > 0: a video exceeds the LDS working set
    scratch = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    L.check(lib.otal_softnms_classes_ws(L.ptr(dec['seg']), L.ptr(dec['score']), L.ptr(dec['unct']),
                                        L.ptr(dec['actn']), L.ptr(dec['flag']), L.ptr(cs), V, max_clips, A, K,
                                        ctypes.c_float(sigma), tk, ctypes.c_float(score_threshold), L.ptr(out),
                                        L.ptr(counts), L.ptr(index), 5, L.ptr(scratch), ctypes.c_size_t(nbytes), int(n),
                                        L.stream()), "otal_softnms_classes_ws")
    return out, counts, index


def get_video_detections(rows, counts, idx_to_class=None, top_k=5000, duration=None, drop_empty=False):
    """test.py:165-200: per-video proposal list from the suppressed rows of one video (K,top_k,5).
    With `duration` (seconds) the cross-dataset variant, test_cross_data.py:178-215: segments are clipped to
    [0, duration] and the ones left empty are dropped (`drop_empty` alone: that script's THUMOS14 leg, which passes
    no duration but still drops empty segments)."""
    rows, counts = rows.cpu().numpy(), counts.cpu().numpy()
    proposal_list = []
    for cl in range(rows.shape[0]):
        name = idx_to_class[cl + 1] if idx_to_class is not None else cl + 1
        for i in range(int(counts[cl])):
            r = rows[cl, i]
            if r[2] > 0:
                start, end = float(r[0]), float(r[1])
                if duration is not None or drop_empty:
                    start, end = max(0, start), (min(duration, end) if duration is not None else end)
                    if end <= start:
                        continue
                proposal_list.append({'label': name, 'score': float(r[2]), 'segment': [start, end],
                                      'uncertainty': float(r[3]), 'actionness': float(r[4])})
    return proposal_list


@torch.no_grad()
def detect_batch(net, videos, sample_fps, clip_length=256, stride=128, conf_thresh=0.01, top_k=5000, nms_sigma=0.5,
                 batch_clips=32, flow_net=None, flow_videos=None):
    """videos: list of uint8 (C,T,96,96) device tensors (already centre-cropped).  Returns the
    per-video rows/counts of Soft-NMS.  test.py:203-252 without the JSON dump.
    Two-stream runs (`--fusion`, test.py:227-240 + parse_output :90-108): `flow_net` sees the same windows of
    `flow_videos` (2-channel optical flow) and the two networks' RAW outputs are averaged before decoding."""
    if (flow_net is None) != (flow_videos is None):
        raise RuntimeError("detect_batch: flow_net and flow_videos go together")
    clips, offsets, fps, clip_start = [], [], [], [0]
    for v, data in enumerate(videos):
        offs = get_offsets(data.shape[1], clip_length, stride)
        for o in offs:
            clips.append((v, o))
        offsets += [float(o) for o in offs]
        fps += [float(sample_fps[v] if hasattr(sample_fps, '__len__') else sample_fps)] * len(offs)
        clip_start.append(clip_start[-1] + len(offs))
    outs = []
    # 32 windows per forward pass: past that the first feature maps leave the 32-bit buffer offsets of the vector-gather
    # kernels (64 windows: Conv3d_1a's output is 4.8 GB) and the generic kernels take over at half the speed
    batch_clips = max(1, min(int(batch_clips), 32))
    for i in range(0, len(clips), batch_clips):
        out = net(prepare_windows(videos, clips[i:i + batch_clips], clip_length))
        if flow_net is not None:
            out = fuse_outputs(out, flow_net(prepare_windows(flow_videos, clips[i:i + batch_clips], clip_length)))
        outs.append(out)
    if flow_net is not None and 'unct' not in outs[0]:
        # fuse_outputs only carries the uncertainty maps when BOTH networks produced them (use_edl + os_head); the reference's
        # two-stream fusion of a closed-set model (thumos14.yaml) decodes softmax scores, which this decode kernel does not
        raise NotImplementedError("two-stream fusion needs evidential (use_edl, os_head) networks on both streams")
    keys = ('loc', 'conf', 'prop_loc', 'prop_conf', 'center', 'act', 'prop_act', 'priors') + \
        (('unct', 'prop_unct') if flow_net is not None else ())
    merged = {k: (torch.cat([o[k] for o in outs], 0) if k != 'priors' else outs[0][k]) for k in keys}
    dec = decode_clips(merged, offsets, fps, clip_length, conf_thresh)
    if flow_net is not None:
        # the decode kernel derives the uncertainty from the (fused) logits; the reference averages the two networks'
        # OWN uncertainties instead (parse_output :105-108, decode_predictions :122) -- not the same number
        dec['unct'] = ((merged['unct'] + merged['prop_unct']) / 2.0).contiguous()
    return softnms_classes(dec, clip_start, top_k, nms_sigma) + (dec,)


# ----------------------------------------------------------------------------- result files (test.py:246-252, threshold.py:128-150)
OOD_SCORES = {
    'uncertainty': lambda p: p['uncertainty'],
    'confidence': lambda p: 1 - p['score'],
    'uncertainty_actionness': lambda p: p['uncertainty'] * p['actionness'],
    'a_by_inv_u': lambda p: p['actionness'] / (1 - p['uncertainty'] + 1e-6),
    'u_by_inv_a': lambda p: p['uncertainty'] / (1 - p['actionness'] + 1e-6),
    'half_au': lambda p: 0.5 * (p['actionness'] + 1) * p['uncertainty'],
}


def results_json(result_dict, threshold=None, version="THUMOS14"):
    """The result-file layout AFSD/evaluation reads: {'version', 'results': {video: [proposal, ...]}, 'external_data'}."""
    ext = {} if threshold is None else {'threshold': float(threshold)}
    return {"version": version, "results": dict(result_dict), "external_data": ext}


def ood_threshold(result_dict, scoring='uncertainty'):
    """The known/unknown operating point of AFSD/thumos14/threshold.py:128-150: run the detector over the TRAINING
    videos, turn every detection into a known-ness score (1 - its OOD score) and take the value that 95 % of the
    detections exceed."""
    import numpy as np
    score = OOD_SCORES[scoring]
    all_scores = [1 - score(p) for props in result_dict.values() for p in props]
    n = len(all_scores)
    if n == 0:
        raise ValueError("no detections to threshold")
    return float(np.sort(all_scores)[n - int(n * 0.95) - 1])


# ----------------------------------------------------------------------------- the inference driver (test.py:203-288)
def prepare_data(data_path, video_name, crop_size, device='cuda'):
    """test.py:59-64: <video>.npy uint8 (T,H,W,3) -> centre-cropped planar (3,T,crop,crop) uint8 on the device."""
    import os
    import numpy as np
    data = np.load(os.path.join(data_path, video_name + '.npy'))
    data = np.transpose(data, [3, 0, 1, 2])
    h, w = data.shape[2:]
    i, j = int(np.round((h - crop_size) / 2.)), int(np.round((w - crop_size) / 2.))
    return torch.from_numpy(np.ascontiguousarray(data[:, :, i:i + crop_size, j:j + crop_size])).to(device)


def fuse_outputs(rgb_out, flow_out):
    """Two-stream fusion by averaging the two networks' RAW outputs before decoding (parse_output, test.py:90-108): loc,
    conf, prop_loc, prop_conf, center, the actionness logits and -- with use_edl -- the two uncertainty maps.
    Reference hazard (H11): with os_head its parse_output squeezes the rgb actionness to (126,) but not the flow one
    ((126,1)), so `act + flow_act` broadcasts to (126,126) and decode_predictions fails on the OpenTAL configuration; the
    elementwise average it evidently means is what is computed here."""
    keys = ('loc', 'conf', 'prop_loc', 'prop_conf', 'center', 'act', 'prop_act', 'unct', 'prop_unct')
    fused = {k: (rgb_out[k] + flow_out[k]) / 2.0 for k in keys if rgb_out.get(k) is not None and flow_out.get(k) is not None}
    fused['priors'] = rgb_out['priors']
    return fused


def test(net, video_infos, npy_data_path, idx_to_class=None, clip_length=256, stride=128, crop_size=96, conf_thresh=0.01,
         top_k=5000, nms_sigma=0.5, batch_clips=32, batch_videos=8, rank=0, world=1, device='cuda', flow_net=None,
         flow_data_path=None):
    """The loop of test.py:203-252 over a video list, batched: `batch_videos` videos' windows go through the network
    together and ONE decode + ONE Soft-NMS launch serve all of them.  Ranks take every world-th video (no collective).
    `flow_net` + `flow_data_path`: the two-stream (fusion) run of test.py:213-240."""
    names = list(video_infos.keys())[rank::world]
    result_dict = {}
    for i in range(0, len(names), batch_videos):
        part = names[i:i + batch_videos]
        vids = [prepare_data(npy_data_path, n, crop_size, device) for n in part]
        flows = [prepare_data(flow_data_path, n, crop_size, device) for n in part] if flow_net is not None else None
        rows, counts, _, _ = detect_batch(net, vids, [float(video_infos[n]['sample_fps']) for n in part], clip_length, stride,
                                          conf_thresh, top_k, nms_sigma, batch_clips, flow_net=flow_net, flow_videos=flows)
        for v, n in enumerate(part):
            result_dict[n] = get_video_detections(rows[v], counts[v], idx_to_class, top_k)
    return result_dict


def gather_results(result_dict, names, rank, world, device=None):
    """Several ranks (video list sharded): every rank's result dict travels to rank 0, which returns the merged
    dict in the video list's order (the pattern sketched in AFSD/anet/test.py:248-273, with a collective instead of
    multiprocessing queues); the other ranks return None.  One rank: the dict itself."""
    if world == 1:
        return result_dict
    import torch.distributed as dist
    if not dist.is_initialized():
        import os
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29534')
        if device is not None and torch.device(device).type == 'cuda':
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(result_dict, parts, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in parts:
        merged.update(part)
    return {n: merged[n] for n in names if n in merged}


def main(argv=None):
    """python -m opental_amd.thumos14.test <yaml> --open_set --split 0 [--random_init] [--evaluate GT.json KNOWN.txt]

    The reference's test driver (AFSD/thumos14/test.py:203-288): config -> model + checkpoint -> sliding windows over
    every test video -> result JSON at <output_path>/<output_json>; `--evaluate` then runs the open-set evaluation of
    opental_amd.thumos14.eval_open on it."""
    import json
    import os
    import sys
    from ..common import config as C
    from ..common import ops
    from ..common.thumos_dataset import get_class_index_map, get_video_info
    from .BDNet import BDNet, model_cfg_from
    argv = list(sys.argv[1:] if argv is None else argv)
    random_init, evaluate, rest, i = False, None, [], 0
    while i < len(argv):
        if argv[i] == '--random_init':
            random_init = True
        elif argv[i] == '--evaluate':
            evaluate = (argv[i + 1], argv[i + 2]); i += 2
        else:
            rest.append(argv[i])
        i += 1
    config = C.set_config(C.get_config(rest))
    te, md, ds = config['testing'], config['model'], config['dataset']['testing']
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    ops.CONV_PRECISION = 1 if os.environ.get('OTAL_DTYPE', 'bf16') == 'bf16' else 0
    flow_net, data_path, flow_path = None, ds['video_data_path'], None
    if te.get('fusion', False):             # build_model(fusion=True), test.py:24-40: rgb + flow networks, their own checkpoints
        net = BDNet(in_channels=3, training=False, use_edl=md.get('use_edl', False), cfg=model_cfg_from(config))
        flow_net = BDNet(in_channels=2, training=False, use_edl=md.get('use_edl', False), cfg=model_cfg_from(config))
        if not random_init:
            net.load_state_dict(torch.load(te.get('rgb_checkpoint_path', './models/thumos14/checkpoint-15.ckpt'), map_location='cpu'))
            flow_net.load_state_dict(torch.load(te.get('flow_checkpoint_path', './models/thumos14_flow/checkpoint-16.ckpt'), map_location='cpu'))
        flow_net = flow_net.to(dev).eval()
        data_path = te.get('rgb_data_path', './datasets/thumos14/test_npy/')
        flow_path = te.get('flow_data_path', './datasets/thumos14/test_flow_npy/')
    else:
        net = BDNet(in_channels=md['in_channels'], training=False, use_edl=md.get('use_edl', False), cfg=model_cfg_from(config))
        if not random_init:
            net.load_state_dict(torch.load(te['checkpoint_path'], map_location='cpu'))
    net = net.to(dev).eval()
    video_infos = get_video_info(config['dataset']['testing']['video_info_path'])
    _, idx_to_class = get_class_index_map(config['dataset']['class_info_path'])
    results = test(net, video_infos, data_path, idx_to_class, ds['clip_length'], ds['clip_stride'], ds['crop_size'],
                   te['conf_thresh'], te['top_k'], te['nms_sigma'], rank=rank, world=world, device=dev, flow_net=flow_net,
                   flow_data_path=flow_path)
    results = gather_results(results, list(video_infos.keys()), rank, world, dev)
    if results is None:
        return None, None           # ranks > 0: their detections went to rank 0
    os.makedirs(te['output_path'], exist_ok=True)
    out_file = os.path.join(te['output_path'], te['output_json'])
    with open(out_file, 'w') as f:
        json.dump(results_json(results), f)
    print(f"{len(results)} videos, {sum(len(v) for v in results.values())} detections -> {out_file}")
    if evaluate is not None:
        from .eval_open import evaluate_split
        return out_file, evaluate_split(out_file, evaluate[0], evaluate[1], [0.3, 0.4, 0.5, 0.6, 0.7], ['test'], True,
                                        te.get('ood_scoring', 'confidence'))
    return out_file, None


if __name__ == '__main__':
    main()
