"""python -m opental_amd.thumos14.threshold <yaml> --open_set --split 0 [--ood_scoring uncertainty] [--random_init]

The known / unknown operating point of the open-set evaluation (AFSD/thumos14/threshold.py:71-157): the detector runs over
the TRAINING videos, every detection becomes a known-ness score (1 - its OOD score under `--ood_scoring`, :129-143) and the
threshold is the value 95 % of them exceed (:144-147); the result file carries the detections and
external_data.threshold, and an existing complete file is re-used (:158-166).

Device work = the config-5 path (`test.test`: batched sliding windows -> otal_decode_clips -> otal_softnms_classes); under
torchrun the video list is sharded over the ranks and gathered on rank 0, which writes the file."""
import json
import os
import sys

import torch

from . import test as T


def thresholding(net, video_infos, npy_data_path, output_file, idx_to_class=None, scoring='uncertainty', clip_length=256, stride=128,
                 crop_size=96, conf_thresh=0.01, top_k=5000, nms_sigma=0.5, rank=0, world=1, device='cuda', flow_net=None,
                 flow_data_path=None):
    """threshold.py:71-150.  Returns the threshold on rank 0 (None on the other ranks)."""
    results = T.test(net, video_infos, npy_data_path, idx_to_class, clip_length, stride, crop_size, conf_thresh, top_k, nms_sigma,
                     rank=rank, world=world, device=device, flow_net=flow_net, flow_data_path=flow_data_path)
    results = T.gather_results(results, list(video_infos.keys()), rank, world, device)
    if results is None:
        return None
    thr = T.ood_threshold(results, scoring)
    os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
    with open(output_file, 'w') as f:
        json.dump(T.results_json(results, threshold=thr), f)
    return thr


def main(argv=None):
    from ..common import config as C
    from ..common import ops
    from ..common.thumos_dataset import get_class_index_map, get_video_info
    from .BDNet import BDNet, model_cfg_from
    argv = list(sys.argv[1:] if argv is None else argv)
    random_init = '--random_init' in argv
    argv = [a for a in argv if a != '--random_init']
    args = C.build_parser().parse_args(argv)
    config = C.set_config(C.get_config(argv))
    te, md, ds = config['testing'], config['model'], config['dataset']
    output_file = os.path.join(te['output_path'], te['output_json'])
    if os.path.exists(output_file):                                  # threshold.py:158-166
        with open(output_file) as f:
            data = json.load(f)
        thr = data.get('external_data', {}).get('threshold')
        if thr is not None:
            print(f'Thresholding result file already exist at {output_file}!')
            print(f'The threshold is: {thr:.12f}')
            return output_file, thr
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    ops.CONV_PRECISION = 1 if os.environ.get('OTAL_DTYPE', 'bf16') == 'bf16' else 0
    net = BDNet(in_channels=md['in_channels'], training=False, use_edl=md.get('use_edl', False), cfg=model_cfg_from(config))
    if not random_init:
        net.load_state_dict(torch.load(te['checkpoint_path'], map_location='cpu'))
    net = net.to(dev).eval()
    video_infos = get_video_info(ds['training']['video_info_path'])          # the TRAINING list (:73)
    _, idx_to_class = get_class_index_map(ds['class_info_path'])
    t = ds['testing']
    thr = thresholding(net, video_infos, ds['training']['video_data_path'], output_file, idx_to_class, args.ood_scoring,
                       t['clip_length'], t['clip_stride'], t['crop_size'], te['conf_thresh'], te['top_k'], te['nms_sigma'],
                       rank=rank, world=world, device=dev)
    if thr is not None:
        print(f'The threshold is: {thr:.12f}')
    return output_file, thr


if __name__ == '__main__':
    main()
