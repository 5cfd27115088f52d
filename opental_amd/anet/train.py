"""Training step of the ActivityNet1.3 recipe on MI355X (reference: AFSD/anet/train.py; BASELINE config 4).

`calc_bce_loss` / `forward_one_epoch` keep the reference's names and arithmetic (anet/train.py:136-186); the step
itself is thumos14.train.DetectorTrainer (flat arenas, bucketed RCCL all-reduce overlapped with backward, one-launch
Adam per optimizer group, optional HIP-graph replay) with the recipe's two optimizer groups: backbone at lr/10,
detection pyramid at lr (anet/train.py:304-312).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..thumos14.train import DetectorTrainer, total_cost  # noqa: F401


def calc_bce_loss(start, end, scores):
    """anet/train.py:136-144.  start/end (B,T,C) features; scores (B,3,T) = [action, start, end] masks."""
    start = torch.tanh(start).mean(-1)
    end = torch.tanh(end).mean(-1)
    loss_start = F.binary_cross_entropy(start.view(-1), scores[:, 1].contiguous().view(-1), reduction='mean')
    loss_end = F.binary_cross_entropy(end.view(-1), scores[:, 2].contiguous().view(-1), reduction='mean')
    return loss_start, loss_end


def forward_one_epoch(net, criterion, clips, targets, scores=None, training=True, ssl=True):
    """anet/train.py:147-186 with the criterion passed in (the reference uses a global CPD_Loss)."""
    if training:
        output_dict = net(clips, proposals=targets, ssl=ssl) if ssl else net(clips, ssl=False)
    else:
        with torch.no_grad():
            output_dict = net(clips)
    if ssl:
        anchor, positive, negative = output_dict
        weights = [1, 0.1, 0.1]
        loss_ = [nn.TripletMarginLoss()(anchor[i], positive[i], negative[i]) * weights[i] for i in range(3)]
        return torch.stack(loss_).sum(0)
    loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act = criterion(
        [output_dict['loc'], output_dict['conf'], output_dict['prop_loc'], output_dict['prop_conf'],
         output_dict['center'], output_dict['priors'], output_dict['act'], output_dict['prop_act']], targets)
    src = getattr(output_dict, 'boundary_maps', None)
    if src is not None and src[0].is_cuda and src[0].dtype == torch.float32:
        # one launch per map: tanh, channel mean, BCE and the gradient, on the channel-major maps in place (csrc/bce.hip)
        from ..common.ops import BoundaryBCEFunction
        loss_start, loss_end = BoundaryBCEFunction.apply(src[0], scores, 1, 1)
        a, b = BoundaryBCEFunction.apply(src[1], scores, 1, 8)       # F.interpolate(scale_factor=1/8), nearest
        c, d = BoundaryBCEFunction.apply(src[2], scores, 1, 8)
    else:
        loss_start, loss_end = calc_bce_loss(output_dict['start'], output_dict['end'], scores)
        scores_ = scores[:, :, ::8]     # F.interpolate(scale_factor=1/8), nearest
        a, b = calc_bce_loss(output_dict['start_loc_prop'], output_dict['end_loc_prop'], scores_)
        c, d = calc_bce_loss(output_dict['start_conf_prop'], output_dict['end_conf_prop'], scores_)
    loss_start = loss_start + 0.1 * (a + c)
    loss_end = loss_end + 0.1 * (b + d)
    return loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_start, loss_end, loss_act, loss_prop_act


def optimizer_groups(net, learning_rate):
    """The two torch.optim.Adam groups of anet/train.py:304-312, in the reference's order."""
    return [(list(net.backbone.parameters()), learning_rate * 0.1),
            (list(net.coarse_pyramid_detection.parameters()), learning_rate)]


def make_trainer(net, criterion, loss_weights, learning_rate=1e-4, weight_decay=1e-4, **kw):
    return DetectorTrainer(net, criterion, loss_weights, learning_rate, weight_decay,
                           param_groups=optimizer_groups(net, learning_rate), forward_fn=forward_one_epoch, **kw)
