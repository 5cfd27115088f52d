"""Classification / actionness losses of OpenTAL with the reference's class names and constructor
arguments (AFSD/thumos14/cls_loss.py: FocalLoss_Ori :6-78, EvidenceLoss :81-285, ActionnessLoss
:288-339), re-expressed for the GPU: every function takes ALL anchors plus a boolean mask and
uses masked sums, scatter-adds and rank masks, so there is no boolean-mask gather, no `.item()`
and no Python loop over bins -- i.e. no host synchronisation inside the training step
(the reference syncs ~60 times per step in these losses).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _evidence(logit, kind='exp'):
    if kind == 'exp':
        return torch.exp(torch.clamp(logit, -10, 10))
    if kind == 'relu':
        return F.relu(logit)
    if kind == 'softplus':
        return F.softplus(logit)
    raise NotImplementedError(kind)


class FocalLoss_Ori(nn.Module):
    """-alpha_c (1 - p)^gamma log(p + 1e-6) on softmax probabilities (cls_loss.py:6-78)."""

    def __init__(self, num_class, alpha=None, gamma=2, balance_index=-1, size_average=True):
        super(FocalLoss_Ori, self).__init__()
        self.num_class, self.gamma, self.size_average, self.eps = num_class, gamma, size_average, 1e-6
        if alpha is None:
            alpha = [0.25, 0.75]
        if isinstance(alpha, (list, tuple)):
            assert len(alpha) == num_class
            a = torch.Tensor(list(alpha))
        elif isinstance(alpha, (float, int)):
            assert 0 < alpha < 1.0 and balance_index > -1
            a = torch.ones(num_class) * (1 - alpha)
            a[balance_index] = alpha
        else:
            a = alpha
        self.register_buffer('alpha', a, persistent=False)

    def forward(self, prob, target, mask=None):
        """prob (N,K) softmax scores, target (N,) class ids, mask (N,) rows that count."""
        target = target.view(-1)
        pt = prob.gather(1, target.view(-1, 1)).view(-1) + self.eps
        loss = -torch.pow(1.0 - pt, self.gamma) * (self.alpha.to(prob.device)[target] * pt.log())
        if mask is not None:
            loss = torch.where(mask, loss, torch.zeros_like(loss))
            return loss.sum() / mask.sum().clamp(min=1) if self.size_average else loss.sum()
        return loss.mean() if self.size_average else loss.sum()


class EvidenceLoss(nn.Module):
    """EDL loss ('log' / 'digamma' / 'mse', evidence exp / relu / softplus) with influence-balanced (IBM) re-weighting from
    a 50-bin EMA (cls_loss.py:186-285) and the IoU-calibration term (cls_loss.py:120-129).  The final recipe's
    combination (exp / log / IBM) runs inside the single-launch HIP loss (csrc/loss.hip); the other kinds use this masked
    torch formulation on the device."""

    def __init__(self, num_cls, cfg, size_average=False):
        super(EvidenceLoss, self).__init__()
        self.num_cls = num_cls
        self.loss_type = cfg['loss_type']
        self.evidence = cfg['evidence']
        for flag in ('with_focal', 'with_ghm', 'with_ibloss'):
            if cfg.get(flag, False):
                raise NotImplementedError(f"{flag}: ablation variant outside the opental_final recipe")
        if self.loss_type not in ('log', 'digamma', 'mse'):
            raise NotImplementedError(self.loss_type)
        if cfg.get('soft_label', 0.0):
            raise NotImplementedError("soft_label")
        self.iou_aware = cfg.get('iou_aware', False)
        self.with_ibm = cfg.get('with_ibm', False)
        self.ibm_start = cfg.get('ibm_start', 0)
        self.num_bins = cfg.get('num_bins', 50)
        self.momentum = cfg.get('momentum', 0.99)
        # checkpointed here (the reference forgets to save it)
        self.register_buffer('weight_accum', torch.ones(self.num_bins))
        self.epoch, self.total_epoch = 0, 25
        self.size_average = size_average

    def evidence_func(self, logit):
        return _evidence(logit, self.evidence)

    def iou_calib(self, logits, ious, mean=False):
        ious = torch.where(ious < 0, torch.full_like(ious, 1e-3), ious)
        u = self.num_cls / (self.evidence_func(logits) + 1).sum(dim=-1)
        reg = -ious * torch.log(1 - u) - (1 - ious) * torch.log(u)
        return reg.mean() if mean else reg.sum()

    def forward(self, logit, target, mask=None):
        """logit (N,K), target (N,) in [0,K) (any valid id where mask is False), mask (N,) bool."""
        target = target.view(-1)
        if mask is None:
            mask = torch.ones_like(target, dtype=torch.bool)
        alpha = self.evidence_func(logit) + 1
        S = alpha.sum(dim=1, keepdim=True)
        if self.loss_type == 'mse':
            # mse_loss + loglikelihood_loss (cls_loss.py:186-201,:280-285): squared error of the Dirichlet mean plus its
            # variance, summed over the positives; the IBM / focal re-weightings do not apply to this loss type there either
            y = F.one_hot(target, self.num_cls).to(alpha.dtype)
            per = ((y - alpha / S) ** 2).sum(1) + (alpha * (S - alpha) / (S * S * (S + 1))).sum(1)
            per = torch.where(mask, per, torch.zeros_like(per))
            return per.sum() / mask.sum().clamp(min=1) if self.size_average else per.sum()
        func = torch.log if self.loss_type == 'log' else torch.digamma
        a_y = alpha.gather(1, target.view(-1, 1))
        per = (func(S) - func(a_y)).view(-1)          # sum_k y_k (f(S) - f(alpha_k)) with one-hot y
        if self.with_ibm and self.epoch >= self.ibm_start:
            with torch.no_grad():
                u = self.num_cls / S.view(-1)
                gnorm = torch.abs(1 / a_y.view(-1) - u)
                ghat = gnorm * logit.abs().sum(1)
                bins = torch.ceil(gnorm * self.num_bins).long()                 # 1..num_bins (0 if gnorm == 0)
                slot = torch.remainder(bins - 1, self.num_bins)                 # python-style [-1] of the reference
                m = mask.to(ghat.dtype)
                tot = torch.zeros(self.num_bins, device=logit.device).index_add_(0, slot, ghat * m)
                cnt = torch.zeros(self.num_bins, device=logit.device).index_add_(0, slot, m)
                # the reference only updates bins 1..num_bins; bin 0 reads slot -1 without updating it
                upd_cnt = torch.zeros(self.num_bins, device=logit.device).index_add_(0, slot, m * (bins > 0).to(m.dtype))
                upd_tot = torch.zeros(self.num_bins, device=logit.device).index_add_(0, slot, ghat * m * (bins > 0).to(m.dtype))
                new = self.momentum * self.weight_accum + (1 - self.momentum) * upd_tot / upd_cnt.clamp(min=1)
                self.weight_accum.copy_(torch.where(upd_cnt > 0, new, self.weight_accum))
                w = self.weight_accum[slot]
            per = w * per
        per = torch.where(mask, per, torch.zeros_like(per))
        return per.sum() / mask.sum().clamp(min=1) if self.size_average else per.sum()


class ActionnessLoss(nn.Module):
    """Positive-unlabelled BCE: positives + the top-M lowest-scoring negatives, M = min(P, N) - 1
    (cls_loss.py:288-339).  Returns (loss, number of samples used) -- both tensors, no sync."""

    def __init__(self, size_average=False, cfg=None):
        super(ActionnessLoss, self).__init__()
        self.size_average = size_average
        self.weight = cfg.get('weight', 0.1) if cfg is not None else 0.1
        self.margin = cfg.get('margin', 1.0) if cfg is not None else 1.0

    def forward(self, logit, target):
        pred = logit.reshape(-1)
        pos = target.reshape(-1) > 0
        neg = ~pos
        npos, nneg = pos.sum(), neg.sum()
        top_m = torch.minimum(npos, nneg) - 1
        big = torch.finfo(pred.dtype).max
        order = torch.argsort(torch.where(neg, pred.detach(), torch.full_like(pred, big)))
        rank = torch.empty_like(order).scatter_(0, order, torch.arange(order.numel(), device=order.device))
        use_neg = torch.where(top_m > 0, neg & (rank < top_m), neg)
        used = pos | use_neg
        bce = F.binary_cross_entropy_with_logits(pred, pos.to(pred.dtype), reduction='none')
        bce = torch.where(used, bce, torch.zeros_like(bce))
        count = used.sum()
        loss = bce.sum() / count.clamp(min=1) if self.size_average else bce.sum()
        if self.weight != 0:
            neg_max = torch.where(neg, pred, torch.full_like(pred, -big)).max()
            pos_max = torch.where(pos, pred, torch.full_like(pred, -big)).max().detach()
            rank_loss = torch.clamp(self.margin - neg_max + pos_max, min=0.0)
            loss = loss + self.weight * torch.where(top_m > 0, rank_loss, torch.zeros_like(rank_loss))
        return loss, count
