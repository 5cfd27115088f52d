"""Classification / actionness losses of the ActivityNet1.3 recipe with the reference's class names and constructor
arguments (AFSD/anet/cls_loss.py: FocalLoss_Ori :6-75, EvidenceLoss :78-246, ActionnessLoss :249-296).

The ActivityNet MultiSegmentLoss evaluates every term PER SAMPLE (anet/multisegment_loss.py:129-287), so these
modules take the anchors of a whole batch as (B,K,...) plus a boolean mask and reduce over the anchor axis only:
one set of launches for the batch, no boolean-mask gathers, no `.item()`, no host synchronisation.

Differences from the THUMOS14 file that matter numerically and are kept:
  * the influence-balanced weight is the closed form 1 / (|z|_1 exp(coeff g) + 1e-10) -- no EMA bins (:225-232);
  * |z|_1 inside that weight is NOT detached (:137), so the weight carries gradient;
  * ActionnessLoss's rank hinge is active (weight 0.1, anet/multisegment_loss.py:102).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..thumos14.cls_loss import FocalLoss_Ori as _FocalLoss, _evidence


class FocalLoss_Ori(_FocalLoss):
    """Per-sample sums: prob (B,K,C), target (B,K), mask (B,K) -> (B,)."""

    def __init__(self, num_class, alpha=(0.25, 0.75), gamma=2, balance_index=-1, size_average=True):
        super(FocalLoss_Ori, self).__init__(num_class, alpha=alpha, gamma=gamma, balance_index=balance_index,
                                            size_average=size_average)

    def forward(self, prob, target, mask):
        pt = prob.gather(-1, target.unsqueeze(-1)).squeeze(-1) + self.eps
        loss = -torch.pow(1.0 - pt, self.gamma) * (self.alpha.to(prob.device)[target] * pt.log())
        loss = torch.where(mask, loss, torch.zeros_like(loss))
        return loss.sum(-1) / mask.sum(-1).clamp(min=1) if self.size_average else loss.sum(-1)


class EvidenceLoss(nn.Module):
    def __init__(self, num_cls, cfg, size_average=False):
        super(EvidenceLoss, self).__init__()
        self.num_cls = num_cls
        self.loss_type = cfg['loss_type']
        self.evidence = cfg['evidence']
        if cfg.get('with_ghm', False):
            raise NotImplementedError("with_ghm: ablation variant outside the opental recipe")
        if self.loss_type not in ('log', 'digamma'):
            raise NotImplementedError(self.loss_type)
        self.iou_aware = cfg.get('iou_aware', False)
        self.with_ibm = cfg.get('with_ibm', False)
        self.eps = 1e-10
        self.ibm_start = cfg.get('ibm_start', 0)
        self.coeff = cfg.get('ibm_coeff', 10)
        self.epoch, self.total_epoch = 0, 25
        self.size_average = size_average

    def evidence_func(self, logit):
        return _evidence(logit, self.evidence)

    def iou_calib(self, logits, ious, mean=False):
        """logits (...,K,C), ious (...,K) -> reduced over the anchor axis."""
        ious = torch.where(ious < 0, torch.full_like(ious, 1e-3), ious)
        u = self.num_cls / (self.evidence_func(logits) + 1).sum(dim=-1)
        reg = -ious * torch.log(1 - u) - (1 - ious) * torch.log(u)
        return reg.mean(-1) if mean else reg.sum(-1)

    def forward(self, logit, target, mask):
        """logit (B,K,C), target (B,K) class ids (any valid id where mask is False), mask (B,K) -> (B,)."""
        func = torch.log if self.loss_type == 'log' else torch.digamma
        alpha = self.evidence_func(logit) + 1
        S = alpha.sum(dim=-1)
        a_y = alpha.gather(-1, target.unsqueeze(-1)).squeeze(-1)
        per = func(S) - func(a_y)                      # sum_k y_k (f(S) - f(alpha_k)) with one-hot y
        if self.with_ibm and self.epoch >= self.ibm_start:
            with torch.no_grad():
                gnorm = torch.abs(1 / a_y - self.num_cls / S)
            feat_norm = logit.abs().sum(-1)             # carries gradient, as in the reference
            per = per / (feat_norm * torch.exp(self.coeff * gnorm) + self.eps)
        per = torch.where(mask, per, torch.zeros_like(per))
        return per.sum(-1) / mask.sum(-1).clamp(min=1) if self.size_average else per.sum(-1)


class ActionnessLoss(nn.Module):
    """Positive-unlabelled BCE per sample: positives + the top-M lowest-scoring negatives, M = min(P, N) - 1, plus
    weight * max(0, margin - max(neg) + max(pos).detach()) when M > 0.  logit (B,K), target (B,K) ->
    (loss (B,), samples used (B,))."""

    def __init__(self, size_average=False, weight=0.1, margin=1.0):
        super(ActionnessLoss, self).__init__()
        self.size_average, self.weight, self.margin = size_average, weight, margin

    def forward(self, logit, target):
        pred = logit
        pos = target > 0
        neg = ~pos
        npos, nneg = pos.sum(-1, keepdim=True), neg.sum(-1, keepdim=True)
        top_m = torch.minimum(npos, nneg) - 1
        big = torch.finfo(pred.dtype).max
        order = torch.argsort(torch.where(neg, pred.detach(), torch.full_like(pred, big)), dim=-1)
        rank = torch.empty_like(order).scatter_(-1, order, torch.arange(order.size(-1), device=order.device).expand_as(order))
        use_neg = torch.where(top_m > 0, neg & (rank < top_m), neg)
        used = pos | use_neg
        bce = F.binary_cross_entropy_with_logits(pred, pos.to(pred.dtype), reduction='none')
        bce = torch.where(used, bce, torch.zeros_like(bce))
        count = used.sum(-1)
        loss = bce.sum(-1) / count.clamp(min=1) if self.size_average else bce.sum(-1)
        if self.weight != 0:
            neg_max = torch.where(neg, pred, torch.full_like(pred, -big)).max(-1)[0]
            pos_max = torch.where(pos, pred, torch.full_like(pred, -big)).max(-1)[0].detach()
            rank_loss = torch.clamp(self.margin - neg_max + pos_max, min=0.0)
            loss = loss + self.weight * torch.where(top_m.squeeze(-1) > 0, rank_loss, torch.zeros_like(rank_loss))
        return loss, count
