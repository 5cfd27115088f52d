"""Inference path of the ActivityNet1.3 recipe on MI355X, with the reference's function names (AFSD/anet/test.py):
prepare_clip (:83-92), decode_prediction (:95-132), filtering (:135-156), get_video_prediction (:159-200).

A video is ONE 768-frame clip here (the data is resampled to the clip length, anet/test.py:71-80), so a batch of videos
is a batch of clips: the network runs on them together and the same two launches as THUMOS14 finish the job --
otal_decode_clips (refine + decode + Dirichlet scores + thresholds, for every clip) and otal_softnms_classes (one
workgroup per (video, class)).  Differences from the THUMOS14 file that are kept: short videos are padded with 127.5
(mid-grey), the confidence threshold is 0.001, proposals are clipped to [0, duration] and empty ones dropped.
"""
import torch

from ..thumos14 import test as _t

CLIP_LENGTH = 768


def prepare_clip(data, offset, clip_length=CLIP_LENGTH, crop_size=96):
    """uint8 (C,T,H,W) device tensor -> (1,C,clip_length,H,W) float in [-1,1]; the tail is padded with 127.5 -> 0.0."""
    clip = data[:, offset: offset + clip_length].float()
    if clip.size(1) < clip_length:
        pad = torch.full([clip.size(0), clip_length - clip.size(1), crop_size, crop_size], 127.5, device=clip.device)
        clip = torch.cat([clip, pad], dim=1)
    return ((clip / 255.0) * 2.0 - 1.0).unsqueeze(0)


def _heads(output_dict):
    """The ActivityNet priors carry the level id in a second column (anet/BDNet.py:262-269); decoding uses the centres."""
    d = dict(output_dict)
    if d['priors'].dim() == 2 and d['priors'].shape[1] > 1:
        d['priors'] = d['priors'][:, 0].contiguous()
    return d


def decode_clips(output_dict, fps, clip_length=CLIP_LENGTH, conf_thresh=0.001):
    """Batched decode_prediction + the threshold masks of filtering for n videos (offset 0)."""
    n = output_dict['loc'].shape[0]
    return _t.decode_clips(_heads(output_dict), [0.0] * n, fps, clip_length, conf_thresh)


def decode_prediction(output_dict, idx=0, sample_fps=1.0, clip_length=CLIP_LENGTH):
    """Single-clip view: decoded_segments (A,2) in SECONDS (the reference divides by fps in filtering), conf_scores
    (K,A), uncertainty (A,), actionness (A,)."""
    return _t.decode_predictions(_heads(output_dict), idx, 0.0, sample_fps, clip_length)


def filtering(decoded_segments, conf_score_cls, uncertainty, actionness, conf_thresh=0.001):
    return _t.filtering(decoded_segments, conf_score_cls, uncertainty, actionness, conf_thresh)


def get_video_prediction(rows, counts, duration, idx_to_class=None):
    """anet/test.py:159-200: the proposal list of one video from its suppressed rows (K,top_k,5), clipped to
    [0, duration]; proposals that end before they start are dropped."""
    rows, counts = rows.cpu().numpy(), counts.cpu().numpy()
    proposal_list = []
    for cl in range(rows.shape[0]):
        name = idx_to_class[cl + 1] if idx_to_class is not None else cl + 1
        for i in range(int(counts[cl])):
            r = rows[cl, i]
            if not r[2] > 0:
                continue
            start, end = max(0, float(r[0])), min(duration, float(r[1]))
            if end <= start:
                continue
            proposal_list.append({'label': name, 'score': float(r[2]), 'segment': [start, end],
                                  'uncertainty': float(r[3]), 'actionness': float(r[4])})
    return proposal_list


@torch.no_grad()
def detect_batch(net, videos, sample_fps, durations, idx_to_class=None, clip_length=CLIP_LENGTH, conf_thresh=0.001,
                 top_k=5000, nms_sigma=0.85, batch_clips=4):
    """videos: list of uint8 (C,T,96,96) device tensors (centre-cropped).  Returns {index: proposal list}."""
    outs = []
    for i in range(0, len(videos), batch_clips):
        # one launch per forward pass; same values as prepare_clip per video (127.5 padded before the normalisation IS 0.0)
        batch = _t.prepare_windows(videos, [(j, 0) for j in range(i, min(i + batch_clips, len(videos)))], clip_length)
        outs.append(net(batch))
    merged = {k: (torch.cat([o[k] for o in outs], 0) if k != 'priors' else outs[0][k])
              for k in ('loc', 'conf', 'prop_loc', 'prop_conf', 'center', 'act', 'prop_act', 'priors')}
    fps = [float(f) for f in sample_fps]
    dec = decode_clips(merged, fps, clip_length, conf_thresh)
    rows, counts, _ = _t.softnms_classes(dec, list(range(len(videos) + 1)), top_k, nms_sigma)
    return {v: get_video_prediction(rows[v], counts[v], durations[v], idx_to_class) for v in range(len(videos))}


# ----------------------------------------------------------------------------- the driver (anet/test.py:203-348)
def prepare_data(npy_path, video_name, crop_size=96, device='cuda'):
    """anet/test.py:71-80: <npy dir>/<name>.npy uint8 (T,H,W,3) -> centre-cropped planar (3,T,crop,crop) uint8 on the device."""
    return _t.prepare_data(npy_path, video_name, crop_size, device)


def get_class_names(class_info_path):
    """anet/test.py:54-59: one class name per line -> {1..K: name}."""
    with open(class_info_path) as f:
        return {i + 1: line.strip() for i, line in enumerate(f.readlines())}


def testing(net, video_list, video_infos, npy_path, idx_to_class=None, clip_length=CLIP_LENGTH, crop_size=96, conf_thresh=0.001,
            top_k=5000, nms_sigma=0.85, batch_clips=4, rank=0, world=1, device='cuda'):
    """anet/test.py:294-331 over this rank's share of the video list (every world-th video: the reference's
    testing_multithread :248-273 splits the list over mp.Process workers; here the workers are the ranks of a torchrun
    launch, one per GPU, and the per-rank dicts are gathered on rank 0).  Returns {name without "v_": proposal list}."""
    mine = list(video_list)[rank::world]
    out = {}
    for i in range(0, len(mine), batch_clips):
        part = mine[i:i + batch_clips]
        vids = [prepare_data(npy_path, n, crop_size, device) for n in part]
        fps = [float(video_infos[n]['fps']) for n in part]                # sample_fps = video_infos[name]['fps'] (:74)
        res = detect_batch(net, vids, fps, [float(video_infos[n]['duration']) for n in part], idx_to_class, clip_length,
                           conf_thresh, top_k, nms_sigma, batch_clips)
        for v, n in enumerate(part):
            out[n[2:]] = res[v]
    return out


def main(argv=None):
    """python -m opental_amd.anet.test configs/anet_opental.yaml --open_set --split 0 [--random_init]

    anet/test.py:334-348: the validation videos that exist on disk, a complete result file re-used, else the run; the file is
    {"version": "ActivityNet-v1.3", "results": {...}, "external_data": {}} at <output_path>/<output_json>."""
    import json
    import os
    import sys
    from ..common import config as C
    from ..common import ops
    from .BDNet import BDNet, model_cfg_from
    argv = list(sys.argv[1:] if argv is None else argv)
    random_init = '--random_init' in argv
    argv = [a for a in argv if a != '--random_init']
    config = C.set_config(C.get_config(argv))
    te, md, ds = config['testing'], config['model'], config['dataset']
    t = ds['testing']
    with open(t['video_info_path']) as f:
        infos = {k: v for k, v in json.load(f).items() if v.get('subset', 'validation') == 'validation'}
    on_disk = {f[:-4] for f in os.listdir(t['video_mp4_path']) if f.endswith('.npy')}
    video_list = [n for n in infos if n in on_disk]
    out_file = os.path.join(te['output_path'], te['output_json'])
    if os.path.exists(out_file):
        with open(out_file) as f:
            if len(json.load(f)['results']) == len(video_list):
                print(f'Result file exist and it is complete! \n{out_file}')
                return out_file
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    ops.CONV_PRECISION = 1 if os.environ.get('OTAL_DTYPE', 'bf16') == 'bf16' else 0
    net = BDNet(in_channels=md['in_channels'], training=False, frame_num=t['clip_length'], use_edl=md.get('use_edl', False),
                cfg=model_cfg_from(config))
    if not random_init:
        net.load_state_dict(torch.load(te['checkpoint_path'], map_location='cpu'))
    net = net.to(dev).eval()
    idx_to_class = None
    if ds.get('class_info_path') and os.path.exists(ds['class_info_path']):
        idx_to_class = get_class_names(ds['class_info_path'])
    res = testing(net, video_list, infos, t['video_mp4_path'], idx_to_class, t['clip_length'], t['crop_size'], te['conf_thresh'],
                  te['top_k'], te['nms_sigma'], rank=rank, world=world, device=dev)
    res = _t.gather_results(res, [n[2:] for n in video_list], rank, world, dev)
    if res is None:
        return None
    assert len(res) == len(video_list), "Incomplete testing results!"
    os.makedirs(te['output_path'], exist_ok=True)
    with open(out_file, 'w') as f:
        json.dump(_t.results_json(res, version="ActivityNet-v1.3"), f)
    print(f"{len(res)} videos, {sum(len(v) for v in res.values())} detections -> {out_file}")
    return out_file


if __name__ == '__main__':
    main()
