"""MultiSegmentLoss of the ActivityNet1.3 recipe (reference AFSD/anet/multisegment_loss.py:87-301): same constructor,
same `predictions` list and the same 7-tuple.  The reference loops over the samples of the batch in Python and
normalises every term per sample before averaging; here all samples are matched and reduced at once on padded
targets with masked per-sample reductions -- no host synchronisation.

What differs from the THUMOS14 loss and is reproduced: per-level regression bounds on max(left, right) (:69-84,
:156-166); refined-stage positives use min(overlap_thresh, best IoU among the sample's positives) (:178-184);
smooth-L1 for the refinement (:206); IoU calibration pairs each sample's logits with its own IoUs (:259-261).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..thumos14.multisegment_loss import _tiou, as_padded, iou_loss, pad_targets  # noqa: F401  (iou_loss: reference name)
from .cls_loss import ActionnessLoss, EvidenceLoss, FocalLoss_Ori

bounds = [[0, 30], [15, 60], [30, 120], [60, 240], [96, 768], [256, 768]]
FUSED = True      # one workgroup per sample (csrc/loss.hip) for the recipe's settings; False forces the torch formulation


class MultiSegmentLoss(nn.Module):
    def __init__(self, num_classes, overlap_thresh, negpos_ratio, use_gpu=True, cls_loss_type='focal', edl_config=None,
                 os_head=False, size_average=False, clip_length=768):
        super(MultiSegmentLoss, self).__init__()
        self.num_classes = num_classes
        self.overlap_thresh = overlap_thresh
        self.negpos_ratio = negpos_ratio
        self.use_gpu = use_gpu
        self.cls_loss_type = cls_loss_type
        self.clip_length = clip_length          # config['dataset']['training']['clip_length'] (:118)
        if size_average:
            raise NotImplementedError("size_average: the reference's `x /= N if not size_average else x` divides a "
                                      "loss by itself in that mode; the recipe never sets it")
        if cls_loss_type == 'focal':
            self.cls_loss = FocalLoss_Ori(num_classes, balance_index=0, size_average=False, alpha=0.25)
        elif cls_loss_type == 'edl':
            self.cls_loss = EvidenceLoss(num_classes, edl_config, size_average=False)
        else:
            raise NotImplementedError(cls_loss_type)
        self.iou_aware = cls_loss_type == 'edl' and self.cls_loss.iou_aware
        self.os_head = os_head
        if not os_head:
            raise NotImplementedError("closed-set (background-class) variant; OpenTAL uses os_head")
        self.act_loss = ActionnessLoss(size_average=False, weight=0.1)
        self.size_average = size_average
        self.register_buffer('level_bounds', torch.tensor(bounds, dtype=torch.float32), persistent=False)

    @torch.no_grad()
    def match(self, loc, priors, targets):
        """Anchor <-> GT assignment for the whole batch (anet/multisegment_loss.py:144-188)."""
        clip = float(self.clip_length)
        gt, valid = as_padded(targets, loc.device)
        valid = valid.bool()
        lvl = priors[:, 1].long()
        lb = self.level_bounds.to(loc.device)[lvl, 0].view(1, -1, 1)
        rb = self.level_bounds.to(loc.device)[lvl, 1].view(1, -1, 1)
        c = priors[:, 0].view(1, -1, 1)                                     # (1,K,1)
        left = (c - gt[:, None, :, 0]) * clip                               # (B,K,G)
        right = (gt[:, None, :, 1] - c) * clip
        far = torch.max(left, right)
        big = clip * 2
        area = left + right
        out = (left < 0) | (right < 0) | (far <= lb) | (far > rb) | ~valid[:, None, :]
        area = torch.where(out, torch.full_like(area, big), area)
        best_area, best = area.min(-1)                                      # first minimum, like torch.min
        g0 = torch.gather(gt[:, :, 0], 1, best)
        g1 = torch.gather(gt[:, :, 1], 1, best)
        lab = torch.gather(gt[:, :, 2], 1, best)
        p = priors[:, 0].view(1, -1)
        loc_t = torch.stack([(p - g0) * clip, (g1 - p) * clip], -1)
        conf_t = torch.where(best_area >= big, torch.zeros_like(lab), lab).long()
        iou = _tiou(loc, loc_t)[0]
        pos = conf_t > 0
        best_iou = iou.masked_fill(~pos, -float('inf')).max(-1)[0]
        thr = torch.where(pos.any(-1), best_iou.clamp(max=float(self.overlap_thresh)),
                          torch.full_like(best_iou, float(self.overlap_thresh)))
        prop_conf_t = torch.where(iou < thr.unsqueeze(-1), torch.zeros_like(conf_t), conf_t)
        w = (loc[..., 0] + loc[..., 1]).unsqueeze(-1)
        prop_loc_t = (loc_t - loc) / (0.5 * w)
        return loc_t, conf_t, prop_loc_t, prop_conf_t, iou

    def _bounds_list(self):
        """The level table of the `level_bounds` BUFFER (the single source of truth of both paths), as host floats."""
        lb = self.level_bounds
        key = (lb.data_ptr(), lb._version)
        if getattr(self, '_bounds_key', None) != key:
            self._bounds_host, self._bounds_key = lb.detach().cpu().tolist(), key
        return self._bounds_host

    def _fused_ok(self, loc, conf, priors):
        """The HIP loss of this recipe (csrc/loss.hip, otal_detection_loss_anet) covers what configs/anet_opental.yaml trains
        with; other settings stay on the torch formulation below."""
        cl = self.cls_loss
        if not (FUSED and loc.is_cuda and loc.dtype == torch.float32 and priors.shape[0] <= 1024 and priors.shape[1] == 2
                and self.cls_loss_type == 'edl' and cl.loss_type == 'log' and cl.evidence == 'exp' and not cl.size_average
                and cl.num_cls == conf.shape[-1] and cl.num_cls < 32768 and not self.act_loss.size_average
                and self.level_bounds.shape[0] <= 8):        # (the kernel keeps labels in 16 bits and <= MAX_LEVELS_A = 8 levels)
            return False
        # the kernel clamps a prior's level id into the table; the torch path would raise on an out-of-range one: keep the
        # torch path's behaviour for such priors (checked once per priors tensor -- one host read, not one per step)
        key = (priors.data_ptr(), priors._version, tuple(priors.shape))
        if getattr(self, '_lvl_checked', None) != key:
            self._lvl_ok = bool(0 <= int(priors[:, 1].min()) and int(priors[:, 1].max()) < self.level_bounds.shape[0])
            self._lvl_checked = key
        return self._lvl_ok

    def forward(self, predictions, targets, pre_locs=None):
        loc, conf, prop_loc, prop_conf, center, priors, act, prop_act = predictions
        B, K = loc.shape[0], priors.shape[0]
        if self._fused_ok(loc, conf, priors):
            from ..common.ops import AnetDetectionLossFunction
            gt, valid = as_padded(targets, loc.device)
            cl = self.cls_loss
            return AnetDetectionLossFunction.apply(
                loc, conf, prop_loc, prop_conf, center.reshape(B, K), act.reshape(B, K), prop_act.reshape(B, K), priors, gt, valid,
                self._bounds_list(), float(self.clip_length), float(self.overlap_thresh), bool(cl.with_ibm and cl.epoch >= cl.ibm_start),
                float(cl.coeff), bool(self.iou_aware), float(self.act_loss.weight), float(self.act_loss.margin))
        loc_t, conf_t, prop_loc_t, prop_conf_t, iou_pred = self.match(loc.detach(), priors, targets)
        pos, prop_pos = conf_t > 0, prop_conf_t > 0
        zero = loc.new_zeros(())
        loss_l = torch.where(pos, iou_loss(loc, loc_t, loss_type='giou'), zero).sum(-1)
        d = (prop_loc - prop_loc_t).abs()
        sl1 = torch.where(d < 1.0, 0.5 * d * d, d - 0.5)
        loss_prop_l = torch.where(prop_pos.unsqueeze(-1), sl1, zero).sum((-1, -2))
        # quality head: BCE(center, tIoU of the refined segment); the target is NOT detached (:212-222)
        w = (loc[..., 0] + loc[..., 1]).unsqueeze(-1)
        cur = 0.5 * w * prop_loc + loc
        q = _tiou(cur, loc_t)[0].clamp(min=0)
        x = center.view(B, K)
        bce = torch.clamp(x, min=0) - x * q + torch.log1p(torch.exp(-x.abs()))
        loss_ct = torch.where(pos, bce, zero).sum(-1)

        def classify(logits, tgt):
            keep = tgt > 0
            cls_id = (tgt - 1).clamp(min=0)
            if self.cls_loss_type == 'focal':
                return self.cls_loss(F.softmax(logits, dim=-1), cls_id, keep), keep
            return self.cls_loss(logits, cls_id, keep), keep

        loss_c, keep = classify(conf, conf_t)
        loss_act, AN = self.act_loss(act.view(B, K), keep.to(act.dtype))
        loss_prop_c, pkeep = classify(prop_conf, prop_conf_t)
        loss_prop_act, PAN = self.act_loss(prop_act.view(B, K), pkeep.to(act.dtype))
        N = pos.sum(-1).clamp(min=1)
        PN = prop_pos.sum(-1).clamp(min=1)
        loss_l, loss_c, loss_ct = loss_l / N, loss_c / N, loss_ct / N
        loss_prop_l, loss_prop_c = loss_prop_l / PN, loss_prop_c / PN
        loss_act, loss_prop_act = loss_act / AN, loss_prop_act / PAN
        if self.iou_aware:
            loss_prop_c = loss_prop_c + self.cls_loss.iou_calib(prop_conf, iou_pred, mean=True)
        return tuple(v.sum() / B for v in (loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act))
