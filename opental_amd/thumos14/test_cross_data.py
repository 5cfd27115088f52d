"""Cross-dataset open-set inference: the THUMOS14-trained detector run over ActivityNet1.3 videos, whose detections
all count as "unknown" in the open-set evaluation (AFSD/thumos14/test_cross_data.py).  Same names as the reference:
prepare_anet_clip (:80-89), test_anet (:265-312), exclude_overlapping (:315-334), and the merge of its __main__
(:420-441).

The device work is the config-5 path unchanged (`test.detect_batch`: batched windows -> otal_decode_clips ->
otal_softnms_classes); what differs is host-side: ActivityNet arrays are windowed with the THUMOS clip length, a short
last window is padded with mid-grey 127.5 BEFORE normalisation (which is exactly the 0.0 `prepare_clip` pads with
AFTER it, so the batched windows are shared), segments are clipped to the video duration, result keys drop the "v_"
prefix, and videos annotated with a class THUMOS14 also has are left out before the two result files are merged.
"""
import torch

from . import test as T


def prepare_anet_clip(data, offset, clip_length, crop_size):
    """uint8 (C,T,H,W) device tensor -> (1,C,clip_length,H,W) float in [-1,1]; test_cross_data.py:80-89."""
    clip = data[:, offset: offset + clip_length].float()
    if clip.size(1) < clip_length:
        pad = torch.ones([clip.size(0), clip_length - clip.size(1), crop_size, crop_size], device=clip.device) * 127.5
        clip = torch.cat([clip, pad], dim=1)
    return (clip.unsqueeze(0) / 255.0) * 2.0 - 1.0


@torch.no_grad()
def test_anet(net, videos, video_infos, idx_to_class=None, clip_length=256, stride=128, conf_thresh=0.01, top_k=5000,
              nms_sigma=0.5, batch_clips=32, batch_videos=8):
    """videos: {name: uint8 (C,T,96,96) device tensor, centre-cropped}; video_infos[name] holds 'fps', 'duration'
    and 'frame_num' (the ActivityNet video_info_train_val.json rows).  Returns the result dict of
    test_cross_data.py:265-312 (keys without the "v_" prefix)."""
    names = [n for n in video_infos if n in videos]
    result_dict = {}
    for i in range(0, len(names), batch_videos):
        part = names[i:i + batch_videos]
        rows, counts, _, _ = T.detect_batch(net, [videos[n] for n in part], [float(video_infos[n]['fps']) for n in part],
                                            clip_length, stride, conf_thresh, top_k, nms_sigma, batch_clips)
        for v, n in enumerate(part):
            result_dict[n[2:]] = T.get_video_detections(rows[v], counts[v], idx_to_class, top_k,
                                                        duration=video_infos[n]['duration'])
    return T.results_json(result_dict)


def exclude_overlapping(anet_out, video_infos, excluded_classes):
    """Drop the videos annotated with any class that THUMOS14 has as well (test_cross_data.py:315-334);
    `excluded_classes`: the lines of overlapping_classes_in_thumos.txt."""
    excluded = set(c.strip() for c in excluded_classes)
    kept = {}
    for name, preds in anet_out['results'].items():
        if not any(ann['label'] in excluded for ann in video_infos['v_' + name]['annotations']):
            kept[name] = preds
    return T.results_json(kept)


def merge_results(thumos_out, anet_out):
    """One result file for the open-set evaluation: THUMOS14 detections plus the remaining ActivityNet ones
    (test_cross_data.py:433-441; an ActivityNet key equal to a THUMOS one replaces it, as dict.update does)."""
    merged = dict(thumos_out['results'])
    merged.update(anet_out['results'])
    return T.results_json(merged)


# ----------------------------------------------------------------------------- the driver (test_cross_data.py:219-262, :336-447)
def load_anet_video(npy_path, name, crop_size=96, device='cuda'):
    """<npy dir>/<name>.npy uint8 (T,H,W,3) -> centre-cropped planar (3,T,crop,crop) uint8 on the device (:244-250)."""
    return T.prepare_data(npy_path, name, crop_size, device)


def get_anet_video_info(video_info_path, subset='validation'):
    """test_cross_data.py:381-391: the ActivityNet video_info_train_val.json rows of one subset."""
    import json
    with open(video_info_path) as f:
        data = json.load(f)
    return {k: v for k, v in data.items() if v['subset'] == subset}


def main(argv=None):
    """python -m opental_amd.thumos14.test_cross_data <yaml> --open_set --split 0 [--random_init]
           [--anet_info datasets/activitynet/annotations/video_info_train_val.json]
           [--anet_npy datasets/activitynet/train_val_npy_112]
           [--anet_overlap datasets/activitynet/overlapping_classes_in_thumos.txt]

    The reference's __main__ (test_cross_data.py:420-447): the THUMOS14 test detections (<output_path>/thumos14_open_rgb.json,
    re-used when it exists), the detections of the same network on the ActivityNet1.3 validation videos
    (<output_path>/anet_open_rgb.json, re-used likewise), the ActivityNet videos annotated with a class THUMOS14 has as well
    dropped, and the two merged into <output_path>/<output_json>.  Under torchrun both video lists are sharded over the ranks
    (no collective in the data path; results gathered on rank 0, which writes the files)."""
    import json
    import os
    import sys
    from ..common import config as C
    from ..common import ops
    from ..common.thumos_dataset import get_class_index_map, get_video_info
    from .BDNet import BDNet, model_cfg_from
    argv = list(sys.argv[1:] if argv is None else argv)
    opts = {'--anet_info': 'datasets/activitynet/annotations/video_info_train_val.json',
            '--anet_npy': 'datasets/activitynet/train_val_npy_112',
            '--anet_overlap': 'datasets/activitynet/overlapping_classes_in_thumos.txt'}
    random_init, rest, i = False, [], 0
    while i < len(argv):
        if argv[i] == '--random_init':
            random_init = True
        elif argv[i] in opts:
            opts[argv[i]] = argv[i + 1]; i += 1
        else:
            rest.append(argv[i])
        i += 1
    config = C.set_config(C.get_config(rest))
    te, md, ds = config['testing'], config['model'], config['dataset']
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    ops.CONV_PRECISION = 1 if os.environ.get('OTAL_DTYPE', 'bf16') == 'bf16' else 0
    net = BDNet(in_channels=md['in_channels'], training=False, use_edl=md.get('use_edl', False), cfg=model_cfg_from(config))
    if not random_init:
        net.load_state_dict(torch.load(te['checkpoint_path'], map_location='cpu'))
    net = net.to(dev).eval()
    _, idx_to_class = get_class_index_map(ds['class_info_path'])
    t = ds['testing']
    os.makedirs(te['output_path'], exist_ok=True)

    def cached(path, run, expected):
        """A result file of an earlier run is re-used only when it parses AND holds exactly the videos this run would
        produce (a killed run leaves truncated or partial files; anet/test.py checks its file the same way); new files are
        written next to their place and moved in (os.replace is atomic)."""
        if os.path.exists(path):
            try:
                with open(path) as f:
                    old = json.load(f)
                if sorted(old.get('results', {})) == sorted(expected):
                    return old
                print(f"{path}: holds {len(old.get('results', {}))} of {len(expected)} videos -- recomputed")
            except (ValueError, OSError):
                print(f"{path}: unreadable -- recomputed")
        out = run()
        if out is not None:
            tmp = path + '.tmp'
            with open(tmp, 'w') as f:
                json.dump(out, f)
            os.replace(tmp, path)
        return out

    def run_thumos():
        infos = get_video_info(t['video_info_path'])
        res = T.test(net, infos, t['video_data_path'], idx_to_class, t['clip_length'], t['clip_stride'], t['crop_size'],
                     te['conf_thresh'], te['top_k'], te['nms_sigma'], rank=rank, world=world, device=dev)
        res = T.gather_results(res, list(infos.keys()), rank, world, dev)
        return None if res is None else T.results_json(res)

    anet_infos = get_anet_video_info(opts['--anet_info'], 'validation')
    on_disk = {f[:-4] for f in os.listdir(opts['--anet_npy']) if f.endswith('.npy')}
    anet_names = [n for n in anet_infos if n in on_disk]

    def run_anet():
        mine = anet_names[rank::world]
        res = {}
        for j in range(0, len(mine), 8):            # eight videos' windows per forward batch, loaded as they are needed
            part = mine[j:j + 8]
            vids = {n: load_anet_video(opts['--anet_npy'], n, t['crop_size'], dev) for n in part}
            out = test_anet(net, vids, {n: anet_infos[n] for n in part}, idx_to_class, t['clip_length'], t['clip_stride'],
                            te['conf_thresh'], te['top_k'], te['nms_sigma'])
            res.update(out['results'])
        res = T.gather_results(res, [n[2:] for n in anet_names], rank, world, dev)
        return None if res is None else T.results_json(res)

    thumos_out = cached(os.path.join(te['output_path'], 'thumos14_open_rgb.json'), run_thumos,
                        list(get_video_info(t['video_info_path']).keys()))
    anet_out = cached(os.path.join(te['output_path'], 'anet_open_rgb.json'), run_anet, [n[2:] for n in anet_names])
    if thumos_out is None or anet_out is None:
        return None                                  # ranks > 0
    print(f"Number of thumos videos: {len(thumos_out['results'])}; anet videos (before filtering): {len(anet_out['results'])}")
    with open(opts['--anet_overlap']) as f:
        anet_out = exclude_overlapping(anet_out, anet_infos, f.readlines())
    print(f"Number of anet videos (after filtering): {len(anet_out['results'])}")
    merged = merge_results(thumos_out, anet_out)
    out_file = os.path.join(te['output_path'], te['output_json'])
    with open(out_file, 'w') as f:
        json.dump(merged, f)
    print(f"Number of all merged videos: {len(merged['results'])} -> {out_file}")
    return out_file


if __name__ == '__main__':
    main()
