"""MultiSegmentLoss of OpenTAL/AFSD (reference AFSD/thumos14/multisegment_loss.py:70-259) with the
same constructor and the same 7-tuple result, vectorised over the batch and free of host
synchronisation: anchor<->GT matching runs for all clips at once on padded targets, and every
"select the positives, then reduce" of the reference is a masked reduction over all anchors.

Reference quirks kept on purpose (documented in DESIGN.md):
  * the IoU-calibration term pairs iou_pred (stored prior-major, (126,B)) with logits flattened
    batch-major (multisegment_loss.py:116,:234-236) -- identical for batch 1, the yaml's batch size;
  * the tIoU target of the quality head carries gradient into loc / prop_loc (it is not detached,
    multisegment_loss.py:176-187).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..common.input_pipeline import PaddedTargets
from .cls_loss import ActionnessLoss, EvidenceLoss, FocalLoss_Ori

_EPS = torch.finfo(torch.float32).eps
FUSED = True      # single-launch HIP loss for the final recipe (set False to force the torch formulation)


def _tiou(pred, target):
    inter = torch.min(pred[..., 0], target[..., 0]) + torch.min(pred[..., 1], target[..., 1])
    union = (target[..., 0] + target[..., 1]) + (pred[..., 0] + pred[..., 1]) - inter
    return inter / union.clamp(min=_EPS), inter, union


def iou_loss(pred, target, weight=None, loss_type='giou', reduction='none'):
    """(multisegment_loss.py:20-53); any leading shape, last dim = (left, right) extents."""
    ious, _, union = _tiou(pred, target)
    if loss_type == 'linear_iou':
        loss = 1.0 - ious
    elif loss_type == 'giou':
        hull = torch.max(pred[..., 0], target[..., 0]) + torch.max(pred[..., 1], target[..., 1])
        loss = 1.0 - (ious - (hull - union) / hull.clamp(min=_EPS))
    else:
        loss = ious
    if weight is not None:
        loss = loss * weight.view(loss.size())
    if reduction == 'sum':
        return loss.sum()
    if reduction == 'mean':
        return loss.mean()
    return loss


_PAD_INDEX = {}     # (target counts, device) -> (row index of every target in the padded (B*G) table, validity mask); device tensors


def pad_targets(targets, device):
    """list of (n_i,3) [start,end,label] -> (B,G,3) padded + (B,G) validity, built from host shapes.
    Three launches whatever the batch (concatenate, zero, scatter the rows): the scatter index and the mask depend only on
    the per-sample target counts and are cached on the device.  A first-time count pattern met while a HIP graph is being
    captured takes the per-sample form (no host -> device copy may be recorded into the graph)."""
    lens = tuple(int(t.shape[0]) for t in targets)
    G, B = max(lens), len(targets)
    device = torch.device(device)
    key = (lens, device)
    hit = _PAD_INDEX.get(key)
    if hit is None and not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        if len(_PAD_INDEX) > 256:
            _PAD_INDEX.clear()
        idx = torch.tensor([i * G + j for i, n in enumerate(lens) for j in range(n)], dtype=torch.long).to(device)
        valid = (torch.arange(G).view(1, G) < torch.tensor(lens).view(B, 1)).to(device)
        hit = _PAD_INDEX[key] = (idx, valid)
    if hit is None:
        out = torch.zeros(B, G, 3, device=device)
        valid = torch.zeros(B, G, dtype=torch.bool, device=device)
        for i, t in enumerate(targets):
            out[i, :lens[i]] = t.to(device)
            valid[i, :lens[i]] = True
        return out, valid
    idx, valid = hit
    rows = torch.cat([t.to(device=device, dtype=torch.float32).reshape(-1, 3) for t in targets], 0)     # (sum n_i, 3)
    out = torch.zeros(B * G, 3, device=device).index_copy_(0, idx, rows).view(B, G, 3)
    return out, valid


def as_padded(targets, device):
    """(rows (B,G,3), validity (B,G)) of either form of a batch's targets: the reference's list of ragged (n_i,3) arrays
    (padded here to the batch's longest list) or an input_pipeline.PaddedTargets (fixed G, used as it is)."""
    if isinstance(targets, PaddedTargets):
        return targets.gt, targets.valid
    if isinstance(targets, (list, tuple)):
        return pad_targets(targets, device)
    return targets


class MultiSegmentLoss(nn.Module):
    def __init__(self, num_classes, overlap_thresh, negpos_ratio, use_gpu=True, cls_loss_type='focal',
                 edl_config=None, rpl_config=None, os_head=False, act_config=None, size_average=False,
                 clip_length=256):
        super(MultiSegmentLoss, self).__init__()
        self.num_classes = num_classes
        self.overlap_thresh = overlap_thresh
        self.negpos_ratio = negpos_ratio
        self.use_gpu = use_gpu
        self.cls_loss_type = cls_loss_type
        self.clip_length = clip_length      # config['dataset']['training']['clip_length'] (:110)
        self._focal_alpha0 = None
        if cls_loss_type == 'focal':
            self.cls_loss = FocalLoss_Ori(num_classes, balance_index=0, size_average=size_average, alpha=0.25)
            al = self.cls_loss.alpha            # still on the host here: decide once whether the HIP kernel's form applies
            if (self.cls_loss.gamma == 2 and al.numel() >= 2 and bool((al[1:] == al[1]).all())
                    and abs(float(al[0]) + float(al[1]) - 1.0) < 1e-6):
                self._focal_alpha0 = float(al[0])
        elif cls_loss_type == 'edl':
            self.cls_loss = EvidenceLoss(num_classes, edl_config, size_average=size_average)
        else:
            raise NotImplementedError("rpl: baseline outside the OpenTAL hot path")
        self.iou_aware = cls_loss_type == 'edl' and self.cls_loss.iou_aware
        self.os_head = os_head
        if not os_head:
            raise NotImplementedError("closed-set (background-class) variant; OpenTAL uses os_head")
        self.act_loss = ActionnessLoss(size_average=size_average, cfg=act_config)
        self.size_average = size_average

    @torch.no_grad()
    def match(self, loc, priors, targets):
        """Anchor <-> GT assignment for the whole batch (multisegment_loss.py:120-153)."""
        clip = float(self.clip_length)
        gt, valid = as_padded(targets, loc.device)
        valid = valid.bool()
        c = priors[:, 0].view(1, -1, 1)                                     # (1,K,1)
        left = (c - gt[:, None, :, 0]) * clip                               # (B,K,G)
        right = (gt[:, None, :, 1] - c) * clip
        big = clip * 2
        area = left + right
        area = torch.where((left < 0) | (right < 0) | ~valid[:, None, :], torch.full_like(area, big), area)
        best_area, best = area.min(-1)                                      # first minimum, like torch.min
        g0 = torch.gather(gt[:, :, 0], 1, best)
        g1 = torch.gather(gt[:, :, 1], 1, best)
        lab = torch.gather(gt[:, :, 2], 1, best)
        p = priors[:, 0].view(1, -1)
        loc_t = torch.stack([(p - g0) * clip, (g1 - p) * clip], -1)
        conf_t = torch.where(best_area >= big, torch.zeros_like(lab), lab).long()
        iou = _tiou(loc, loc_t)[0]
        prop_conf_t = torch.where(iou < self.overlap_thresh, torch.zeros_like(conf_t), conf_t)
        w = (loc[..., 0] + loc[..., 1]).unsqueeze(-1)
        prop_loc_t = (loc_t - loc) / (0.5 * w)
        return loc_t, conf_t, prop_loc_t, prop_conf_t, iou

    def _fused_ok(self, loc):
        """The single-launch HIP loss (csrc/loss.hip) covers the final recipe; other settings use the torch formulation."""
        cl = self.cls_loss
        common = (FUSED and loc.is_cuda and loc.dtype == torch.float32 and loc.shape[0] * loc.shape[1] <= 2048
                  and self.os_head and not self.size_average and self.act_loss.weight == 0
                  and not self.act_loss.size_average and not cl.size_average)
        if self.cls_loss_type == 'focal':       # the as-shipped THUMOS14 dispatch (train.py:27-31)
            return common and self._focal_alpha0 is not None
        return (common and self.cls_loss_type == 'edl' and cl.loss_type == 'log' and cl.evidence == 'exp' and cl.num_bins <= 64)

    def forward(self, output_dict, targets, pre_locs=None):
        loc, conf = output_dict['loc'], output_dict['conf']
        prop_loc, prop_conf = output_dict['prop_loc'], output_dict['prop_conf']
        center, priors = output_dict['center'], output_dict['priors']
        act, prop_act = output_dict['act'], output_dict['prop_act']
        B, K = loc.shape[0], priors.shape[0]
        C = self.num_classes
        if self._fused_ok(loc):
            from ..common.ops import DetectionLossFunction
            gt, valid = as_padded(targets, loc.device)
            cl = self.cls_loss
            if self.cls_loss_type == 'focal':
                if getattr(self, '_no_ibm', None) is None or self._no_ibm.device != loc.device:
                    self._no_ibm = torch.ones(1, dtype=torch.float32, device=loc.device)    # unused EMA slot of the ABI
                return DetectionLossFunction.apply(
                    loc, conf, prop_loc, prop_conf, center.reshape(B, K), act.reshape(B, K), prop_act.reshape(B, K),
                    priors[:, 0], gt, valid, self._no_ibm, float(self.clip_length), float(self.overlap_thresh),
                    False, 1, 0.0, False, 1, self._focal_alpha0)
            return DetectionLossFunction.apply(
                loc, conf, prop_loc, prop_conf, center.reshape(B, K), act.reshape(B, K), prop_act.reshape(B, K),
                priors[:, 0], gt, valid, cl.weight_accum, float(self.clip_length), float(self.overlap_thresh),
                bool(cl.with_ibm and cl.epoch >= cl.ibm_start), cl.num_bins, float(cl.momentum), bool(self.iou_aware))
        loc_t, conf_t, prop_loc_t, prop_conf_t, iou_pred = self.match(loc.detach(), priors, targets)
        pos, prop_pos = conf_t > 0, prop_conf_t > 0
        zero = loc.new_zeros(())
        # coarse localisation: GIoU over positives
        loss_l = torch.where(pos, iou_loss(loc, loc_t, loss_type='giou'), zero).sum()
        # refined localisation: L1 over refined positives
        loss_prop_l = torch.where(prop_pos.unsqueeze(-1), (prop_loc - prop_loc_t).abs(), zero).sum()
        # quality head: BCE(center, tIoU of the refined segment); the target is NOT detached
        w = (loc[..., 0] + loc[..., 1]).unsqueeze(-1)
        cur = 0.5 * w * prop_loc + loc
        q = _tiou(cur, loc_t)[0].clamp(min=0)
        x = center.view(B, K)
        bce = torch.clamp(x, min=0) - x * q + torch.log1p(torch.exp(-x.abs()))
        loss_ct = torch.where(pos, bce, zero).sum()

        def classify(logits, tgt):
            lg = logits.reshape(-1, C)
            t = tgt.reshape(-1)
            keep = t > 0
            cls_id = (t - 1).clamp(min=0)
            if self.cls_loss_type == 'focal':
                return self.cls_loss(F.softmax(lg, dim=1), cls_id, keep), keep
            return self.cls_loss(lg, cls_id, keep), keep

        loss_c, keep = classify(conf, conf_t)
        loss_act, AN = self.act_loss(act.reshape(-1, 1), keep.to(act.dtype))
        loss_prop_c, pkeep = classify(prop_conf, prop_conf_t)
        loss_prop_act, PAN = self.act_loss(prop_act.reshape(-1, 1), pkeep.to(act.dtype))
        N = pos.sum().clamp(min=1)
        PN = prop_pos.sum().clamp(min=1)
        if not self.size_average:
            loss_l, loss_c, loss_ct = loss_l / N, loss_c / N, loss_ct / N
            loss_prop_l, loss_prop_c = loss_prop_l / PN, loss_prop_c / PN
            loss_act, loss_prop_act = loss_act / AN, loss_prop_act / PAN
        if self.iou_aware:
            # reference pairing: iou_pred is (K,B) flattened prior-major against batch-major logits
            ious = iou_pred.transpose(0, 1).reshape(-1)
            loss_prop_c = loss_prop_c + self.cls_loss.iou_calib(prop_conf.reshape(-1, C), ious, mean=True)
        return loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act
