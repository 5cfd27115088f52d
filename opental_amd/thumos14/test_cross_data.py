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
