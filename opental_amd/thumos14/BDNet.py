"""OpenTAL / AFSD THUMOS14 detector on MI355X.

Same constructor, forward signature, output dict and state-dict keys as the reference's
AFSD/thumos14/BDNet.py (I3D_BackBone :25-52, ScaleExp :55-61, ProposalBranch :64-113,
CoarsePyramid :116-432, BDNet :435-535, DirichletLayer :538-561).  What differs is HOW it runs:

  * level batching -- towers, heads and both ProposalBranches share their weights across the six
    pyramid levels (BDNet.py:333-412), so the levels are packed side by side into one
    (B,512,126) buffer and every shared layer runs ONCE with a level table (taps, GroupNorm
    statistics and pooling windows never cross a level boundary) instead of six times;
  * every Conv + GroupNorm + ReLU block is one fused autograd node (layers.ConvGNReLU);
  * the ~150 tiny elementwise launches of the proposal index math (BDNet.py:355-384) are one
    bit-exact kernel; the 24 BoundaryMaxPooling calls become 4 (2 level-batched, 2 frame-level
    with all 126 proposals).

Known reference hazards handled explicitly: the ssl/triplet branch only works for
batch 1 in the reference (H3) -- sample 0 is used; BatchNorm must be frozen.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..common import ops
from ..common.i3d_backbone import InceptionI3d
from ..common.layers import ConvGNReLU, Unit1D, Unit3D, conv_gn_relu_pair
from ..prop_pooling.boundary_pooling_op import (BoundaryMaxPooling, BoundaryMaxPoolingFunction,
                                                BoundaryMaxPoolingLevelsFunction)

layer_num = 6
conv_channels = 512
feat_t = 256 // 4

DEFAULT_MODEL_CFG = dict(num_classes=16, freeze_bn=True, freeze_bn_affine=True, evidence='exp', dropout=0.0,
                         os_head=True)


def model_cfg_from(config=None):
    """The keys BDNet.py:12-18 reads from the global config at import time."""
    cfg = dict(DEFAULT_MODEL_CFG)
    if config is not None:
        cfg['num_classes'] = config['dataset']['num_classes']
        for k in ('freeze_bn', 'freeze_bn_affine', 'evidence', 'dropout', 'os_head'):
            if k in config['model']:
                cfg[k] = config['model'][k]
        if config['model'].get('transformer', False) or config['model'].get('use_rpl', False):
            raise NotImplementedError("TransformerHead / RPLHead baselines are outside the OpenTAL hot path")
    return cfg


class I3D_BackBone(nn.Module):
    def __init__(self, final_endpoint='Mixed_5c', name='inception_i3d', in_channels=3, freeze_bn=True,
                 freeze_bn_affine=True):
        super(I3D_BackBone, self).__init__()
        self._model = InceptionI3d(final_endpoint=final_endpoint, name=name, in_channels=in_channels)
        self._model.build()
        self._freeze_bn = freeze_bn
        self._freeze_bn_affine = freeze_bn_affine

    def load_pretrained_weight(self, model_path='models/i3d_models/rgb_imagenet.pt'):
        self._model.load_state_dict(torch.load(model_path), strict=False)

    def train(self, mode=True):
        super(I3D_BackBone, self).train(mode)
        if self._freeze_bn and mode:
            for m in self._model.modules():
                if isinstance(m, nn.BatchNorm3d):
                    m.eval()
                    if self._freeze_bn_affine:
                        m.weight.requires_grad_(False)
                        m.bias.requires_grad_(False)
        return self

    def forward(self, x):
        return self._model.extract_features(x)


class ScaleExp(nn.Module):
    def __init__(self, init_value=1.0):
        super(ScaleExp, self).__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))

    def forward(self, input):
        return torch.exp(input * self.scale)


def _block(cin, cout, k, stride=1):
    return ConvGNReLU(Unit1D(cin, cout, k, stride=stride, use_bias=True, activation_fn=None), cout)


class ProposalBranch(nn.Module):
    def __init__(self, in_channels, proposal_channels):
        super(ProposalBranch, self).__init__()
        self.cur_point_conv = _block(in_channels, proposal_channels, 1)
        self.lr_conv = _block(in_channels, proposal_channels * 2, 1)
        self.boundary_max_pooling = BoundaryMaxPooling()
        self.roi_conv = _block(proposal_channels, proposal_channels, 1)
        self.proposal_conv = _block(proposal_channels * 4, in_channels, 1)

    def pool_frame_level(self, frame_level_feature, frame_segments, levels=None):
        """BoundaryMaxPooling of the frame-level map (BDNet.py:109).  Its inputs are the same for the loc and the conf
        branch, so CoarsePyramid pools ONCE and hands the result to both (`roi_pooled=`): same values, one forward and
        one backward launch instead of two (autograd adds the two branches' gradients before the pooling backward)."""
        from ..prop_pooling import boundary_pooling_op as _bp
        if levels is not None and _bp.COMPAT_REFERENCE_BWD:
            # gradient-parity mode: the reference's backward addresses rows with stride N = t_l of each
            # per-level call (boundary_max_pooling_kernel.cu:121), so pool level by level here
            return torch.cat([
                self.boundary_max_pooling(frame_level_feature, frame_segments[:, levels[i]:levels[i + 1]].contiguous())
                for i in range(len(levels) - 1)], dim=2)
        return self.boundary_max_pooling(frame_level_feature, frame_segments)

    def forward(self, feature, frame_level_feature, segments, frame_segments, levels=None, roi_pooled=None):
        """`levels` None: one pyramid level, as the reference calls it (BDNet.py:105-113);
        a level table: all levels packed along the last axis."""
        fm_short = self.cur_point_conv(feature, levels)
        feature = self.lr_conv(feature, levels)
        if levels is None:
            prop_feature = self.boundary_max_pooling(feature, segments)
        else:
            prop_feature = BoundaryMaxPoolingLevelsFunction.apply(feature, segments, levels, levels)
        prop_roi_feature = roi_pooled if roi_pooled is not None else self.pool_frame_level(frame_level_feature, frame_segments, levels)
        prop_roi_feature = self.roi_conv(prop_roi_feature, levels)
        prop_feature = torch.cat([prop_roi_feature, prop_feature, fm_short], dim=1)
        prop_feature = self.proposal_conv(prop_feature, levels)
        return prop_feature, feature


class CoarsePyramid(nn.Module):
    def __init__(self, feat_channels, num_cls, frame_num=256, use_rpl=False, dropout=0.0, os_head=True,
                 projections=(('Mixed_4f', [1, 6, 6]), ('Mixed_5c', [1, 3, 3])), first_level_t=feat_t, fpn_strides=None):
        """`projections`, `first_level_t`, `fpn_strides`: what the ActivityNet variant changes (anet/BDNet.py:120-269,
        see opental_amd/anet/BDNet.py); the defaults are the THUMOS14 model."""
        super(CoarsePyramid, self).__init__()
        if use_rpl:
            raise NotImplementedError("RPL baseline head is outside the OpenTAL hot path")
        C = conv_channels
        self.frame_num = frame_num
        self.layer_num = layer_num
        self.dropout = dropout
        self.num_classes = num_cls
        self.os_head = os_head
        self.fpn_strides = fpn_strides
        self.dirichlet_exp = False      # set by BDNet: uncertainty maps come out of the head-tail launch (evidence 'exp')
        self.projection_inputs = tuple(ep for ep, _ in projections)
        self.pyramids = nn.ModuleList()
        self.loc_heads = nn.ModuleList()
        for fc, (_, kk) in zip(feat_channels, projections):
            self.pyramids.append(ConvGNReLU(Unit3D(fc, C, kernel_shape=kk, padding='spatial_valid',
                                                   use_batch_norm=False, use_bias=True, activation_fn=None), C))
        for _ in range(len(projections), layer_num):
            self.pyramids.append(_block(C, C, 3, stride=2))
        self.loc_tower = nn.Sequential(_block(C, C, 3), _block(C, C, 3))
        self.conf_tower = nn.Sequential(_block(C, C, 3), _block(C, C, 3))
        head = lambda co, k: Unit1D(C, co, kernel_shape=k, stride=1, use_bias=True, activation_fn=None)
        self.loc_head = head(2, 3)
        self.conf_head = head(num_cls, 3)
        if self.os_head:
            self.actionness_head = head(1, 3)
        self.loc_proposal_branch = ProposalBranch(C, 512)
        self.conf_proposal_branch = ProposalBranch(C, 512)
        self.prop_loc_head = head(2, 1)
        self.prop_conf_head = head(num_cls, 1)
        if self.os_head:
            self.prop_actionness_head = head(1, 1)
        self.center_head = head(1, 3)
        dec = []
        for k in (3, 3, 1):
            dec += [Unit1D(C, C, k, activation_fn=None), nn.GroupNorm(32, C), nn.ReLU(inplace=True)]
        self.deconv = nn.Sequential(*dec)
        self.priors = []
        self.level_lengths = []
        t = first_level_t
        for i in range(layer_num):
            self.loc_heads.append(ScaleExp())
            if fpn_strides is None:
                self.priors.append(torch.Tensor([[(c + 0.5) / t] for c in range(t)]).view(-1, 1))
            else:       # the level id rides along for the per-level regression bounds (anet/BDNet.py:262-269)
                self.priors.append(torch.Tensor([[(c + 0.5) / t, i] for c in range(t)]).view(-1, 2))
            self.level_lengths.append(t)
            t = t // 2
        self.levels = tuple(int(v) for v in np.concatenate([[0], np.cumsum(self.level_lengths)]))

    def _priors_on(self, device):
        """The concatenated prior centres on `device`, uploaded once (a host->device copy cannot be captured in a graph)."""
        cached = getattr(self, '_priors_dev', None)
        if cached is None or cached.device != device:
            cached = torch.cat(self.priors, 0).to(device)
            self._priors_dev = cached
        return cached

    def _stride_cols(self, device):
        cached = getattr(self, '_strides_dev', None)
        if cached is None or cached.device != device:
            cached = torch.cat([torch.full((t,), float(s)) for s, t in zip(self.fpn_strides, self.level_lengths)]).to(device)
            self._strides_dev = cached
        return cached

    # ------------------------------------------------------------------ pieces
    def _deconv(self, x):
        for i in (0, 3, 6):
            unit, gn = self.deconv[i], self.deconv[i + 1]
            from ..common.layers import ConvGNReLUFunction
            x = ConvGNReLUFunction.apply(x, unit.conv1d.weight, unit.conv1d.bias, gn.weight, gn.bias,
                                         unit._kernel_shape, unit._stride, False, None, gn.num_groups, gn.eps)
        return x

    def _pyramid(self, feat_dict):
        if len(self.projection_inputs) == 2:
            x1, x2 = (feat_dict[ep] for ep in self.projection_inputs)
            p0 = self.pyramids[0](x1)
            p1 = self.pyramids[1](x2)
            p0 = p0 + F.interpolate(p1, p0.size()[2:], mode='nearest')          # BDNet.py:316-319
            feats = [p0, p1]
        else:                                                                   # anet/BDNet.py:284-290
            p0 = self.pyramids[0](feat_dict[self.projection_inputs[0]])
            feats = [p0]
        x = feats[-1]
        for i in range(len(feats), self.layer_num):
            x = self.pyramids[i](x)
            feats.append(x)
        frame = F.interpolate(p0.unsqueeze(-1), [self.frame_num, 1]).squeeze(-1)   # BDNet.py:324-325
        return feats, self._deconv(frame)

    def _proposal_branches(self, loc_feat, conf_feat, frame_level_feat, segments, frame_segments, lev, roi):
        """ProposalBranch.forward (BDNet.py:105-113) of the loc and the conf branch, stage by stage: the two branches have
        the same shapes, so each of their four conv blocks runs as ONE set of launches for both (conv_gn_relu_pair)."""
        lb, cb = self.loc_proposal_branch, self.conf_proposal_branch
        short_l, short_c = conv_gn_relu_pair(lb.cur_point_conv, cb.cur_point_conv, loc_feat, conf_feat, lev)
        lr_l, lr_c = conv_gn_relu_pair(lb.lr_conv, cb.lr_conv, loc_feat, conf_feat, lev)
        pool_l = BoundaryMaxPoolingLevelsFunction.apply(lr_l, segments, lev, lev)
        pool_c = BoundaryMaxPoolingLevelsFunction.apply(lr_c, segments, lev, lev)
        roi_l = roi if roi is not None else lb.pool_frame_level(frame_level_feat, frame_segments, lev)
        roi_c = roi if roi is not None else cb.pool_frame_level(frame_level_feat, frame_segments, lev)
        roi_l, roi_c = conv_gn_relu_pair(lb.roi_conv, cb.roi_conv, roi_l, roi_c, lev)
        prop_l, prop_c = conv_gn_relu_pair(lb.proposal_conv, cb.proposal_conv, torch.cat([roi_l, pool_l, short_l], dim=1),
                                           torch.cat([roi_c, pool_c, short_c], dim=1), lev)
        return (prop_l, lr_l), (prop_c, lr_c)

    def _drop(self, x):
        return F.dropout(x, p=self.dropout) if self.dropout > 0 else x

    def forward(self, feat_dict, ssl=False, get_feat=False):
        tr = lambda y: y.permute(0, 2, 1).contiguous()
        from . import pyramid_fused as PF
        # the pyramid as two hand-scheduled autograd nodes (thumos14/pyramid_fused.py) where the layout is the THUMOS14 one;
        # the module-by-module composition below otherwise (ActivityNet, the ssl branch, the compat-gradient mode)
        fused = ops.FUSED_PYRAMID and not ssl and PF.eligible(self, feat_dict)
        lev = self.levels
        if fused:
            loc_feat, conf_feat, frame_level_feat = PF.trunk(self, feat_dict)
            batch_num = loc_feat.size(0)
        else:
            feats, frame_level_feat = self._pyramid(feat_dict)
            batch_num = feats[0].size(0)
            if ssl:   # triplet branch: level 0 only (BDNet.py:392-398)
                loc_feat = self.loc_tower[1](self.loc_tower[0](feats[0]))
                conf_feat = self.conf_tower[1](self.conf_tower[0](feats[0]))
                return [frame_level_feat, self.loc_proposal_branch.lr_conv(loc_feat),
                        self.conf_proposal_branch.lr_conv(conf_feat)]
            packed = torch.cat(feats, dim=2)                                    # (B,512,126)
            # the loc and the conf tower are siblings of one shape: each of their two stages is ONE set of launches for both
            # (common/layers.py conv_gn_relu_pair; values as of the blocks on their own)
            l0, c0 = conv_gn_relu_pair(self.loc_tower[0], self.conf_tower[0], packed, packed, lev)
            loc_feat, conf_feat = conv_gn_relu_pair(self.loc_tower[1], self.conf_tower[1], l0, c0, lev)
        # Head convolutions are level-batched GEMM launches; their tails -- ScaleExp (x fpn stride in the ActivityNet model),
        # the permute(0,2,1).contiguous() of every map and the Dirichlet uncertainty -- are one launch per stage
        # (csrc/heads.hip): the coarse stage here, the refined stage after the proposal branches.
        scales = [h.scale for h in self.loc_heads]
        um = 2 if self.dirichlet_exp else 0
        # the skinny heads of a stage: one fused launch (csrc/headconv.hip) where it applies, else head by head
        stage = [(loc_feat, self.loc_head), (self._drop(conf_feat), self.conf_head)]
        if self.os_head:
            stage.append((conf_feat, self.actionness_head))
        raws = ops.head_convs(lev, stage)
        if raws is None:            # wide heads (150 classes): the skinny ones still share the fused launches
            raws = ops.head_convs_mixed(lev, stage, lambda x, head: head(x, lev))
        raws = list(raws) if raws is not None else [head(x, lev) for x, head in stage]
        res = ops.HeadOutputsFunction.apply(tuple(lev), self.fpn_strides, (1, um, 0)[:len(raws)], *scales, *raws)
        loc, conf = res[0], res[1]
        act = res[2] if self.os_head else None
        unct = res[len(raws)] if self.dirichlet_exp else None
        with torch.no_grad():
            segments, frame_segments = ops.proposal_windows(loc.detach(), lev, float(self.frame_num))
        from ..prop_pooling import boundary_pooling_op as _bp
        # compat-gradient mode keeps one pooling per branch: the reference's (buggy) backward runs once per branch, and
        # bwd(g1) + bwd(g2) is only bit-identical to bwd(g1 + g2) for the correct gradient up to fp32 rounding anyway
        t0 = self.level_lengths[0]
        if fused:
            loc_prop_feat, conf_prop_feat, loc_lr0, conf_lr0 = PF.branches(self, loc_feat, conf_feat, frame_level_feat,
                                                                           segments, frame_segments)
        else:
            roi = None if _bp.COMPAT_REFERENCE_BWD else self.loc_proposal_branch.pool_frame_level(frame_level_feat, frame_segments, lev)
            (loc_prop_feat, loc_lr), (conf_prop_feat, conf_lr) = self._proposal_branches(
                loc_feat, conf_feat, frame_level_feat, segments, frame_segments, lev, roi)
            loc_lr0, conf_lr0 = loc_lr[:, :, :t0], conf_lr[:, :, :t0]
        # The six boundary maps of the output dict are (B,T,C) VIEWS of the channel-major maps (the reference returns
        # permuted copies, BDNet.py:328-331,:392-396; same values): their only consumer, the start / end losses of the
        # training step, reads the channel-major maps in place (ops.BoundaryBCEFunction via OutputDict.boundary_maps).
        pv = lambda y: y.permute(0, 2, 1)
        half = frame_level_feat.size(1) // 2
        start, end = pv(frame_level_feat[:, :half]), pv(frame_level_feat[:, half:])
        ndim = loc_lr0.size(1) // 2
        start_loc_prop, end_loc_prop = pv(loc_lr0[:, :ndim]), pv(loc_lr0[:, ndim:])
        start_conf_prop, end_conf_prop = pv(conf_lr0[:, :ndim]), pv(conf_lr0[:, ndim:])
        boundary_maps = (frame_level_feat, loc_lr0, conf_lr0)
        stage = [(loc_prop_feat, self.prop_loc_head), (self._drop(conf_prop_feat), self.prop_conf_head), (loc_prop_feat, self.center_head)]
        if self.os_head:
            stage.append((conf_prop_feat, self.prop_actionness_head))
        one = lambda x, head: head(x, lev) if head._kernel_shape != 1 else head(x)
        raws = ops.head_convs(lev, stage)
        if raws is None:
            raws = ops.head_convs_mixed(lev, stage, one)
        raws = list(raws) if raws is not None else [one(x, head) for x, head in stage]
        res = ops.HeadOutputsFunction.apply(tuple(lev), None, (0, um, 0, 0)[:len(raws)], *[h.detach() for h in scales], *raws)
        prop_loc, prop_conf, center = res[0], res[1], res[2]
        prop_act = res[3] if self.os_head else None
        fused_unct = (unct, res[len(raws)]) if self.dirichlet_exp else None
        priors = self._priors_on(loc.device)
        outs = (loc, conf, prop_loc, prop_conf, center, priors, start, end,
                start_loc_prop, end_loc_prop, start_conf_prop, end_conf_prop, act, prop_act)
        ctr_feat = prop_ctr_feat = None
        if get_feat:
            ctr_feat, prop_ctr_feat = tr(conf_feat), tr(conf_prop_feat)
        self._last_windows = (segments, frame_segments)       # no-grad index tensors (tests / debugging)
        # Graph-attached by-products travel with the return value, never on the module: a tensor with a grad_fn parked on
        # `self` would keep the step's autograd graph (and its AccumulateGrad nodes, bound to the stream they were created
        # on) alive into the next step -- which breaks a later HIP-graph capture of the step.
        extras = {'unct': fused_unct, 'boundary_maps': boundary_maps}
        return outs + (ctr_feat, prop_ctr_feat, extras)


class OutputDict(dict):
    """The reference's output dict (same keys, tensors only) + `boundary_maps`: the channel-major maps behind the six
    boundary entries, for the fused start / end loss.  An attribute, not a key, and it dies with the dict."""
    boundary_maps = None


class BDNet(nn.Module):
    def __init__(self, in_channels=3, backbone_model=None, training=True, use_edl=False, use_rpl=False, cfg=None):
        super(BDNet, self).__init__()
        if use_rpl:
            raise NotImplementedError("RPL baseline is outside the OpenTAL hot path")
        if cfg is None:
            try:
                from ..common import config as _c
                cfg = model_cfg_from(_c._config) if _c._config is not None else dict(DEFAULT_MODEL_CFG)
            except Exception:
                cfg = dict(DEFAULT_MODEL_CFG)
        self.cfg = cfg
        self.os_head = cfg['os_head']
        self.num_classes = cfg['num_classes'] - 1 if self.os_head else cfg['num_classes']
        self.coarse_pyramid_detection = CoarsePyramid([832, 1024], self.num_classes, use_rpl=use_rpl,
                                                      dropout=cfg['dropout'], os_head=self.os_head)
        self.reset_params()
        self.backbone = I3D_BackBone(in_channels=in_channels, freeze_bn=cfg['freeze_bn'],
                                     freeze_bn_affine=cfg['freeze_bn_affine'])
        self.boundary_max_pooling = BoundaryMaxPooling()
        self._training = training
        if self._training:
            if backbone_model is None:
                self.backbone.load_pretrained_weight()
            else:
                self.backbone.load_pretrained_weight(backbone_model)
        self.scales = [1, 4, 4]
        self.use_edl = use_edl
        self.evidence = cfg['evidence']
        if self.use_edl:
            self.out_layer = DirichletLayer(self.evidence, dim=-1)
        self.coarse_pyramid_detection.dirichlet_exp = bool(self.use_edl and self.evidence == 'exp')
        self.use_rpl = use_rpl

    @staticmethod
    def weight_init(m):
        """glorot-uniform of BDNet.py:460-473: limit = sqrt(3 / max(1, (fan_in + fan_out) / 2))."""
        if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose3d)):
            fan_in, fan_out = nn.init._calculate_fan_in_and_fan_out(m.weight)
            limit = float(np.sqrt(3.0 / max(1.0, (fan_in + fan_out) / 2.0)))
            with torch.no_grad():
                m.weight.uniform_(-limit, limit)
                if m.bias is not None:
                    m.bias.zero_()

    def reset_params(self):
        for m in self.modules():
            self.weight_init(m)

    def forward(self, x, proposals=None, ssl=False, get_feat=False):
        if not ssl and x.is_cuda:
            from . import pyramid_fused as PF
            ops.ENDPOINT_HOOKS = PF.early_projection_hooks(self.coarse_pyramid_detection)
        try:
            feat_dict = self.backbone(x)
        finally:
            ops.ENDPOINT_HOOKS = None
        if ssl:
            top_feat = self.coarse_pyramid_detection(feat_dict, ssl)
            d = proposals[0].unsqueeze(0)
            plen = d[:, :, 1:] - d[:, :, :1] + 1.0
            in_plen = torch.clamp(plen / 4.0, min=1.0)
            out_plen = torch.clamp(plen / 10.0, min=1.0)
            frame_segments = torch.cat([torch.round(d[:, :, :1] - out_plen), torch.round(d[:, :, :1] + in_plen),
                                        torch.round(d[:, :, 1:] - in_plen), torch.round(d[:, :, 1:] + out_plen)], -1)
            anchor, positive, negative = [], [], []
            for i in range(3):
                # the reference passes batch-1 segments with batch-b features (out of bounds for b > 1,
                # SURVEY H3); sample 0 is the defined behaviour here
                seg = (frame_segments / self.scales[i]).contiguous()
                bound_feat = self.boundary_max_pooling(top_feat[i][:1].contiguous(), seg)
                ndim = bound_feat.size(1) // 2
                anchor.append(bound_feat[:, ndim:, 0])
                positive.append(bound_feat[:, :ndim, 1])
                negative.append(bound_feat[:, :ndim, 2])
            return anchor, positive, negative
        outs = self.coarse_pyramid_detection(feat_dict, get_feat=True) if get_feat else self.coarse_pyramid_detection(feat_dict)
        loc, conf, prop_loc, prop_conf, center, priors, start, end, start_loc_prop, end_loc_prop, \
            start_conf_prop, end_conf_prop, act, prop_act = outs[:14]
        extras = outs[-1] if isinstance(outs[-1], dict) else {}
        ctr_feat, prop_ctr_feat = outs[14:16] if len(outs) > 15 else (None, None)
        out_dict = OutputDict({'loc': loc, 'conf': conf, 'priors': priors, 'prop_loc': prop_loc, 'prop_conf': prop_conf,
                    'center': center, 'start': start, 'end': end, 'start_loc_prop': start_loc_prop,
                    'end_loc_prop': end_loc_prop, 'start_conf_prop': start_conf_prop,
                    'end_conf_prop': end_conf_prop, 'act': act, 'prop_act': prop_act})
        if self.use_edl:
            fused = extras.get('unct')
            if fused is not None:
                out_dict.update({'unct': fused[0], 'prop_unct': fused[1]})
            else:
                out_dict.update({'unct': self.out_layer.compute_uncertainty(conf),
                                 'prop_unct': self.out_layer.compute_uncertainty(prop_conf)})
        if get_feat and not self.training:
            out_dict.update({'conf_feat': ctr_feat, 'prop_conf_feat': prop_ctr_feat})
        out_dict.boundary_maps = extras.get('boundary_maps')    # channel-major sources of the six boundary maps (loss input)
        return out_dict


class DirichletLayer(nn.Module):
    def __init__(self, evidence='exp', dim=-1):
        super(DirichletLayer, self).__init__()
        self.evidence = evidence
        self.dim = dim

    def evidence_func(self, logit):
        if self.evidence == 'relu':
            return F.relu(logit)
        if self.evidence == 'exp':
            return torch.exp(torch.clamp(logit, -10, 10))
        if self.evidence == 'softplus':
            return F.softplus(logit)
        raise NotImplementedError(self.evidence)

    def compute_uncertainty(self, logit):
        alpha = self.evidence_func(logit) + 1
        return logit.size(-1) / alpha.sum(-1)

    def forward(self, logit):
        alpha = self.evidence_func(logit) + 1
        return alpha / alpha.sum(dim=self.dim, keepdim=True)
