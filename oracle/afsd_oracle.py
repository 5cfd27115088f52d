"""oracle/afsd_oracle.py -- TEST INFRASTRUCTURE (CPU checker), never shipped or measured.

Functional CPU restatement (PyTorch fp32, flat parameter dict keyed like the reference's
state_dict) of the OpenTAL/AFSD THUMOS14 detection path.  Every function cites the reference
file:line it follows.  Third-party arithmetic: the dense math is torch.nn.functional
(conv1d/conv3d/batch_norm/group_norm/max_pool3d/pad); the reference pins torch==1.9.0
(requirements.txt:1) and the container has 2.10 -- semantics of these ops are unchanged.

Checked against the imported reference by oracle/pin_against_reference.py (build container
only; the reference never travels).  BoundaryMaxPooling is oracle/bmp_ref.c.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

from . import arch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force=False):
    """gcc -O2 the C restatement into oracle/_build/libotal_oracle.so (git-ignored)."""
    out = os.path.join(_HERE, "_build", "libotal_oracle.so")
    src = os.path.join(_HERE, "bmp_ref.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-std=c99", "-o", out, src, "-lm"])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c())
        _LIB.otal_oracle_softnms.restype = ctypes.c_int
    return _LIB


def _fp(t):
    return ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------- pooling (a1, a2)
def bmp_forward(x, seg):
    """boundary_max_pooling_kernel.cu:17-46 via oracle/bmp_ref.c.  x (B,C,T), seg (B,N,4)."""
    x = x.detach().contiguous().float()
    seg = seg.detach().contiguous().float()
    B, C, T = x.shape
    N = seg.shape[1]
    assert seg.shape[0] == B and seg.shape[2] == 4 and C % 2 == 0
    out = torch.empty(B, C, N)
    _lib().otal_oracle_bmp_fwd(_fp(x), _fp(seg), _fp(out), B, C, T, N)
    return out


def bmp_backward(gout, x, seg, compat_reference_bwd=False):
    """boundary_max_pooling_kernel.cu:48-82; compat -> tscale = N as at :121."""
    gout = gout.detach().contiguous().float()
    x = x.detach().contiguous().float()
    seg = seg.detach().contiguous().float()
    B, C, T = x.shape
    N = seg.shape[1]
    gin = torch.empty(B, C, T)
    _lib().otal_oracle_bmp_bwd(_fp(gout), _fp(x), _fp(seg), _fp(gin), B, C, T, N,
                                N if compat_reference_bwd else T)
    return gin


def bmp_forward_py(x, seg):
    """Brute-force pure-Python loop of the same definition (tiny cases; pins bmp_ref.c)."""
    B, C, T = x.shape
    N = seg.shape[1]
    out = torch.empty(B, C, N)
    for n in range(B):
        for c in range(C):
            w = 0 if c < C // 2 else 2
            for k in range(N):
                l = min(max(0, int(float(seg[n, k, w]))), T - 1)
                r = min(max(0, int(float(seg[n, k, w + 1]))), T - 1)
                best = float(x[n, c, l])
                for i in range(l + 1, r + 1):
                    if float(x[n, c, i]) > best:
                        best = float(x[n, c, i])
                out[n, c, k] = best
    return out


class _BMPFn(torch.autograd.Function):
    """AFSD/prop_pooling/boundary_pooling_op.py:7-24."""

    @staticmethod
    def forward(ctx, x, seg, compat):
        ctx.save_for_backward(x, seg)
        ctx.compat = compat
        return bmp_forward(x, seg).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        x, seg = ctx.saved_tensors
        return bmp_backward(g, x, seg, ctx.compat).to(x.dtype), None, None


def boundary_max_pool(x, seg, compat_reference_bwd=False):
    return _BMPFn.apply(x, seg, compat_reference_bwd)


# --------------------------------------------------------------------------- layers (a3, a8)
def same_pad(size, k, s):
    """TF-'SAME' split used by every unit (layers.py:198-210, i3d_backbone.py:46-72):
    total = max(k - s, 0) if size % s == 0 else max(k - size % s, 0); front = total // 2."""
    total = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
    return total // 2, total - total // 2


# Operand rounding of the throughput path (BASELINE configs[1] names bf16): the HIP library's bf16 mode rounds BOTH operands
# of every convolution that runs on the bf16 MFMA to bfloat16 (round to nearest even) while it stages them and accumulates
# in fp32; the skinny detection heads stay exact fp32 (csrc/headconv.hip).  `with operand_rounding("bf16")` makes this
# restatement do the same on the CPU -- fp32 convolutions of bf16-rounded operands -- so the bf16 path can be pinned end to
# end and layer by layer instead of against a loose bf16-vs-fp32 tolerance.  Not part of the reference (which is fp32).
# grads=True extends it to the BACKWARD pass of the same convolutions: the data- and weight-gradient GEMMs of the HIP path
# round their gradient operand dy (the gradient w.r.t. the convolution's own output, i.e. behind the frozen-BN scale and
# the ReLU mask) to bf16 as well -- on the CPU a pass-through node behind each such convolution that rounds the incoming
# gradient.  stored_until="Mixed_4f" models the bf16 STORAGE of the backbone (ops.HALF_CHAIN): activations between
# Conv3d_1a and that endpoint exist as bf16 values only, so the max-pools up to it choose their winners among ROUNDED
# values (forward values unchanged -- max commutes with the rounding -- but ties, and with them the routing of the
# gradient, are those of the rounded window).
_ROUND = None
_ROUND_GRADS = False
_STORED_UNTIL = None
_STORED_NOW = False


class operand_rounding:
    def __init__(self, mode, grads=False, stored_until=None):
        self.mode, self.grads, self.stored_until = mode, grads, stored_until

    def __enter__(self):
        global _ROUND, _ROUND_GRADS, _STORED_UNTIL
        self.old = (_ROUND, _ROUND_GRADS, _STORED_UNTIL)
        _ROUND, _ROUND_GRADS, _STORED_UNTIL = self.mode, self.grads, self.stored_until
        return self

    def __exit__(self, *a):
        global _ROUND, _ROUND_GRADS, _STORED_UNTIL
        _ROUND, _ROUND_GRADS, _STORED_UNTIL = self.old


class _RoundOperand(torch.autograd.Function):
    """Round-to-nearest-even to bfloat16 and back, with a pass-through gradient.  (A plain `.to(bfloat16).to(float32)`
    makes autograd cast the GRADIENT through bfloat16 as well: every weight gradient and every convolution's data-gradient
    contribution would come out rounded to 8 bits -- an artefact of the restatement, not something the HIP path does: its
    weight gradients are fp32 sums and a stored data gradient is rounded once, where `_rg` rounds it here.)"""
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _r(t):
    if _ROUND == "bf16" and t.dtype == torch.float32:
        return _RoundOperand.apply(t) if t.requires_grad else t.to(torch.bfloat16).to(t.dtype)
    return t


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _rg(y):
    """Behind a bf16-MFMA convolution: its gradient operand dy is rounded in the backward pass (grads=True)."""
    return _RoundGrad.apply(y) if (_ROUND == "bf16" and _ROUND_GRADS and y.requires_grad) else y


def unit1d(x, w, b, stride=1, exact=False):
    """Unit1D with activation_fn=None (layers.py:178-214)."""
    f, bk = same_pad(x.shape[2], w.shape[2], stride)
    if not exact:
        x, w = _r(x), _r(w)
        return _rg(F.conv1d(F.pad(x, [f, bk]), w, b, stride=stride))
    return F.conv1d(F.pad(x, [f, bk]), w, b, stride=stride)


def gn_relu(x, gamma, beta):
    """nn.GroupNorm(32, C) (eps 1e-5) + ReLU (BDNet.py:67-103)."""
    return F.relu(F.group_norm(x, 32, gamma, beta, 1e-5))


def block1d(P, prefix, x, stride=1, conv_idx=0, gn_idx=1):
    x = unit1d(x, P[f"{prefix}.{conv_idx}.conv1d.weight"], P[f"{prefix}.{conv_idx}.conv1d.bias"], stride)
    return gn_relu(x, P[f"{prefix}.{gn_idx}.weight"], P[f"{prefix}.{gn_idx}.bias"])


def _pad3(x, k, s, spatial=True):
    t, h, w = x.shape[2:]
    pt = same_pad(t, k[0], s[0])
    ph = same_pad(h, k[1], s[1]) if spatial else (0, 0)
    pw = same_pad(w, k[2], s[2]) if spatial else (0, 0)
    return F.pad(x, [pw[0], pw[1], ph[0], ph[1], pt[0], pt[1]])


def conv3d_bn_relu(P, prefix, x, k, s):
    """backbone Unit3D (i3d_backbone.py:46-87): SAME pad -> Conv3d(no bias) -> frozen BN
    (eps 1e-3, running stats; BDNet.py:39-49 keeps BN in eval) -> ReLU."""
    x = _rg(F.conv3d(_pad3(_r(x), k, s), _r(P[f"{prefix}.conv3d.weight"]), None, stride=s))
    x = F.batch_norm(x, P[f"{prefix}.bn.running_mean"], P[f"{prefix}.bn.running_var"],
                     P[f"{prefix}.bn.weight"], P[f"{prefix}.bn.bias"], False, 0.0, 1e-3)
    return F.relu(x)


def maxpool3d_same(x, k, s):
    """MaxPool3dSamePadding (layers.py:9-35): ZERO padding (not -inf), then plain max-pool."""
    if _STORED_NOW:
        x = _r(x)               # bf16-stored pool input: the winner is the first maximum of the ROUNDED window
    return F.max_pool3d(_pad3(x, k, s), k, s)


def mixed(P, prefix, x):
    """InceptionModule (i3d_backbone.py:90-121)."""
    one, three = (1, 1, 1), (3, 3, 3)
    b0 = conv3d_bn_relu(P, f"{prefix}.b0", x, one, one)
    b1 = conv3d_bn_relu(P, f"{prefix}.b1b", conv3d_bn_relu(P, f"{prefix}.b1a", x, one, one), three, one)
    b2 = conv3d_bn_relu(P, f"{prefix}.b2b", conv3d_bn_relu(P, f"{prefix}.b2a", x, one, one), three, one)
    b3 = conv3d_bn_relu(P, f"{prefix}.b3b", maxpool3d_same(x, three, one), one, one)
    return torch.cat([b0, b1, b2, b3], 1)


def i3d_features(P, x, endpoints=None):
    """InceptionI3d.extract_features up to Mixed_5c (i3d_backbone.py:335-342)."""
    global _STORED_NOW
    feats = {}
    _STORED_NOW = _ROUND == "bf16" and _STORED_UNTIL is not None
    for name, kind, args in arch.I3D_ENDPOINTS:
        if kind == "conv":
            _cin, _cout, k, s = args
            x = conv3d_bn_relu(P, f"backbone._model.{name}", x, k, s)
        elif kind == "pool":
            x = maxpool3d_same(x, *args)
        else:
            x = mixed(P, f"backbone._model.{name}", x)
        if endpoints is None or name in endpoints:
            feats[name] = x
        if name == _STORED_UNTIL:
            _STORED_NOW = False
    _STORED_NOW = False
    return feats


# --------------------------------------------------------------------------- index math (a5)
def proposal_windows(loc, t, frame_num):
    """BDNet.py:355-384 (no_grad).  loc (b,t,2) -> level-space and frame-space windows (b,t,4).
    Operation order is kept exactly: results are compared bit for bit."""
    with torch.no_grad():
        b = loc.shape[0]
        prior = torch.Tensor([[(c + 0.5) / t] for c in range(t)]).view(-1, 1)
        pri = prior.expand(b, t, 1)
        seg = loc / frame_num * t
        centre = torch.round(pri * t - 0.5)
        plen = seg[:, :, :1] + seg[:, :, 1:]
        inner = torch.clamp(plen / 4.0, min=1.0)
        outer = torch.clamp(plen / 10.0, min=1.0)
        left = centre - seg[:, :, :1]
        right = centre + seg[:, :, 1:]
        level_win = torch.cat([torch.round(left - outer), torch.round(left + inner),
                               torch.round(right - inner), torch.round(right + outer)], -1)
        d0 = pri * frame_num - loc[:, :, :1]
        d1 = pri * frame_num + loc[:, :, 1:]
        plen = d1 - d0 + 1.0
        inner = torch.clamp(plen / 4.0, min=1.0)
        outer = torch.clamp(plen / 10.0, min=1.0)
        frame_win = torch.cat([torch.round(d0 - outer), torch.round(d0 + inner),
                               torch.round(d1 - inner), torch.round(d1 + outer)], -1)
    return level_win, frame_win


def priors_all(cfg=arch.THUMOS):
    """CoarsePyramid.priors concatenated (BDNet.py:285-293, :415); the ActivityNet variant carries the level id
    in a second column (anet/BDNet.py:262-269)."""
    if cfg.get("fpn_strides"):
        return torch.cat([torch.Tensor([[(c + 0.5) / t, i] for c in range(t)]).view(-1, 2)
                          for i, t in enumerate(arch.level_lengths(cfg))], 0)
    return torch.cat([torch.Tensor([[(c + 0.5) / t] for c in range(t)]).view(-1, 1)
                      for t in arch.level_lengths(cfg)], 0)


# --------------------------------------------------------------------------- model (a4, a6, a7)
def proposal_branch(P, prefix, feat, frame_feat, seg, fseg, compat):
    """ProposalBranch.forward (BDNet.py:105-113)."""
    short = block1d(P, f"{prefix}.cur_point_conv", feat)
    lr = block1d(P, f"{prefix}.lr_conv", feat)
    pooled = boundary_max_pool(lr, seg, compat)
    roi = block1d(P, f"{prefix}.roi_conv", boundary_max_pool(frame_feat, fseg, compat))
    fused = block1d(P, f"{prefix}.proposal_conv", torch.cat([roi, pooled, short], 1))
    return fused, lr


def _head(P, name, x, cfg=None):
    # the THUMOS14 heads (1 / 2 / 15 rows) are exact fp32 FMA kernels on the device; the 150-class ActivityNet heads take
    # the bf16 GEMM path like every other layer
    exact = cfg is None or cfg.get("num_classes", 16) <= 32
    return unit1d(x, P[f"{name}.conv1d.weight"], P[f"{name}.conv1d.bias"], exact=exact)


def coarse_pyramid(P, f4, f5, cfg=arch.THUMOS, compat=False, keep=None):
    """CoarsePyramid.forward, os_head=True: THUMOS14 variant (BDNet.py:295-432) or, with cfg=arch.ANET, the
    ActivityNet variant (anet/BDNet.py:271-391; f4 is unused there)."""
    Q = "coarse_pyramid_detection"
    frame_num = cfg["frame_num"]
    strides = cfg.get("fpn_strides")
    if cfg.get("two_projections", True):
        # pyramids[0], [1]: Unit3D 'spatial_valid' (temporal k=1 -> no pad) + GN + ReLU (BDNet.py:129-155)
        x0 = _rg(F.conv3d(_r(f4), _r(P[f"{Q}.pyramids.0.0.conv3d.weight"]), P[f"{Q}.pyramids.0.0.conv3d.bias"]))
        x0 = gn_relu(x0.squeeze(-1).squeeze(-1), P[f"{Q}.pyramids.0.1.weight"], P[f"{Q}.pyramids.0.1.bias"])
        x1 = _rg(F.conv3d(_r(f5), _r(P[f"{Q}.pyramids.1.0.conv3d.weight"]), P[f"{Q}.pyramids.1.0.conv3d.bias"]))
        x1 = gn_relu(x1.squeeze(-1).squeeze(-1), P[f"{Q}.pyramids.1.1.weight"], P[f"{Q}.pyramids.1.1.bias"])
        x0 = x0 + F.interpolate(x1, x0.shape[2:], mode="nearest")   # BDNet.py:316-319
        feats = [x0, x1]
        x = x1
    else:
        # anet/BDNet.py:130-142,:284-290: one projection of Mixed_5c
        x = _rg(F.conv3d(_r(f5), _r(P[f"{Q}.pyramids.0.0.conv3d.weight"]), P[f"{Q}.pyramids.0.0.conv3d.bias"]))
        x = gn_relu(x.squeeze(-1).squeeze(-1), P[f"{Q}.pyramids.0.1.weight"], P[f"{Q}.pyramids.0.1.bias"])
        feats = [x]
    for i in range(len(feats), cfg["layer_num"]):
        x = block1d(P, f"{Q}.pyramids.{i}", x, stride=2)
        feats.append(x)
    # frame-level branch (BDNet.py:324-331)
    ff = F.interpolate(feats[0].unsqueeze(-1), [frame_num, 1]).squeeze(-1)
    ff = block1d(P, f"{Q}.deconv", ff, 1, 0, 1)
    ff = block1d(P, f"{Q}.deconv", ff, 1, 3, 4)
    ff = block1d(P, f"{Q}.deconv", ff, 1, 6, 7)
    half = ff.shape[1] // 2
    out = {"start": ff[:, :half].permute(0, 2, 1).contiguous(),
           "end": ff[:, half:].permute(0, 2, 1).contiguous()}
    if keep is not None:
        keep["pyramid_feats"] = feats
        keep["frame_level_feat"] = ff
        keep["segments"], keep["frame_segments"] = [], []
        keep["loc_feat"], keep["conf_feat"] = [], []
    locs, confs, acts, plocs, pconfs, pacts, ctrs = [], [], [], [], [], [], []
    tr = lambda y: y.permute(0, 2, 1).contiguous()
    for i, feat in enumerate(feats):
        lf, cf = feat, feat
        for j in range(2):
            lf = block1d(P, f"{Q}.loc_tower.{j}", lf)
            cf = block1d(P, f"{Q}.conf_tower.{j}", cf)
        loc = tr(torch.exp(_head(P, f"{Q}.loc_head", lf) * P[f"{Q}.loc_heads.{i}.scale"]))  # ScaleExp :55-61
        if strides:
            loc = loc * strides[i]      # anet/BDNet.py:307-311
        locs.append(loc)
        confs.append(tr(_head(P, f"{Q}.conf_head", cf)))
        acts.append(tr(_head(P, f"{Q}.actionness_head", cf)))
        t = feat.shape[2]
        seg, fseg = proposal_windows(loc.detach(), t, frame_num)
        lp, lp_lr = proposal_branch(P, f"{Q}.loc_proposal_branch", lf, ff, seg, fseg, compat)
        cp, cp_lr = proposal_branch(P, f"{Q}.conf_proposal_branch", cf, ff, seg, fseg, compat)
        if keep is not None:
            keep["segments"].append(seg)
            keep["frame_segments"].append(fseg)
            keep["loc_feat"].append(lf)
            keep["conf_feat"].append(cf)
        if i == 0:
            nd = lp_lr.shape[1] // 2
            out["start_loc_prop"] = tr(lp_lr[:, :nd])
            out["end_loc_prop"] = tr(lp_lr[:, nd:])
            out["start_conf_prop"] = tr(cp_lr[:, :nd])
            out["end_conf_prop"] = tr(cp_lr[:, nd:])
            if keep is not None:
                keep["trip"] = [ff, lp_lr, cp_lr]
        plocs.append(tr(_head(P, f"{Q}.prop_loc_head", lp)))
        pconfs.append(tr(_head(P, f"{Q}.prop_conf_head", cp)))
        pacts.append(tr(_head(P, f"{Q}.prop_actionness_head", cp)))
        ctrs.append(tr(_head(P, f"{Q}.center_head", lp)))
    out.update(loc=torch.cat(locs, 1), conf=torch.cat(confs, 1), act=torch.cat(acts, 1),
               prop_loc=torch.cat(plocs, 1), prop_conf=torch.cat(pconfs, 1),
               prop_act=torch.cat(pacts, 1), center=torch.cat(ctrs, 1), priors=priors_all(cfg))
    return out


def dirichlet_uncertainty(logit):
    """DirichletLayer.compute_uncertainty, evidence='exp' (BDNet.py:538-556)."""
    alpha = torch.exp(torch.clamp(logit, -10, 10)) + 1
    return logit.shape[-1] / alpha.sum(-1)


def dirichlet_mean(logit):
    """DirichletLayer.forward (BDNet.py:558-561)."""
    alpha = torch.exp(torch.clamp(logit, -10, 10)) + 1
    return alpha / alpha.sum(-1, keepdim=True)


def bdnet_forward(P, x, cfg=arch.THUMOS, compat_reference_bwd=False, keep=None):
    """BDNet.forward(x, ssl=False) with use_edl=True (BDNet.py:479-535)."""
    feats = i3d_features(P, x, endpoints=("Mixed_4f", "Mixed_5c") if keep is None else None)
    if keep is not None:
        keep["endpoints"] = feats
    out = coarse_pyramid(P, feats["Mixed_4f"], feats["Mixed_5c"], cfg, compat_reference_bwd, keep)
    out["unct"] = dirichlet_uncertainty(out["conf"])
    out["prop_unct"] = dirichlet_uncertainty(out["prop_conf"])
    return out


def ssl_triplets(P, x, proposals, cfg=arch.THUMOS, compat_reference_bwd=False):
    """BDNet.forward(ssl=True) (BDNet.py:482-503).  The reference only works for b == 1
    (segments get batch 1); the restatement uses sample 0's features."""
    keep = {}
    feats = i3d_features(P, x, endpoints=("Mixed_4f", "Mixed_5c"))
    coarse_pyramid(P, feats["Mixed_4f"], feats["Mixed_5c"], cfg, compat_reference_bwd, keep)
    d = proposals[0].unsqueeze(0)
    plen = d[:, :, 1:] - d[:, :, :1] + 1.0
    inner = torch.clamp(plen / 4.0, min=1.0)
    outer = torch.clamp(plen / 10.0, min=1.0)
    fs = torch.cat([torch.round(d[:, :, :1] - outer), torch.round(d[:, :, :1] + inner),
                    torch.round(d[:, :, 1:] - inner), torch.round(d[:, :, 1:] + outer)], -1)
    anchor, positive, negative = [], [], []
    for feat, scale in zip(keep["trip"], (1, 4, 4)):
        bf = boundary_max_pool(feat[:1], fs / scale, compat_reference_bwd)
        nd = bf.shape[1] // 2
        anchor.append(bf[:, nd:, 0])
        positive.append(bf[:, :nd, 1])
        negative.append(bf[:, :nd, 2])
    return anchor, positive, negative


# --------------------------------------------------------------------------- losses (a9-a13)
def tiou(pred, target):
    """iou_loss(..., loss_type='calc iou') (multisegment_loss.py:20-53): plain tIoU."""
    eps = torch.finfo(torch.float32).eps
    inter = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 1], target[:, 1])
    union = (target[:, 0] + target[:, 1]) + (pred[:, 0] + pred[:, 1]) - inter
    return inter / union.clamp(min=eps)


def giou_loss_sum(pred, target):
    """iou_loss(loss_type='giou', reduction='sum') (multisegment_loss.py:20-53)."""
    eps = torch.finfo(torch.float32).eps
    inter = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 1], target[:, 1])
    union = (target[:, 0] + target[:, 1]) + (pred[:, 0] + pred[:, 1]) - inter
    iou = inter / union.clamp(min=eps)
    hull = torch.max(pred[:, 0], target[:, 0]) + torch.max(pred[:, 1], target[:, 1])
    return (1.0 - (iou - (hull - union) / hull.clamp(min=eps))).sum()


def match_anchors(loc, priors, targets, clip_length, overlap_thresh):
    """Per-sample anchor<->GT assignment (multisegment_loss.py:120-153), no grad."""
    b, k = loc.shape[0], priors.shape[0]
    loc_t = torch.zeros(b, k, 2)
    conf_t = torch.zeros(b, k, dtype=torch.long)
    prop_loc_t = torch.zeros(b, k, 2)
    prop_conf_t = torch.zeros(b, k, dtype=torch.long)
    iou_pred = torch.zeros(k, b)
    big = clip_length * 2
    with torch.no_grad():
        for i in range(b):
            gt, lab = targets[i][:, :-1], targets[i][:, -1]
            c = priors[:, 0:1]
            left = (c - gt[:, 0][None, :]) * clip_length
            right = (gt[:, 1][None, :] - c) * clip_length
            area = left + right
            area[left < 0] = big
            area[right < 0] = big
            best_area, best = area.min(1)
            loc_t[i, :, 0] = (priors[:, 0] - gt[best, 0]) * clip_length
            loc_t[i, :, 1] = (gt[best, 1] - priors[:, 0]) * clip_length
            conf = lab[best].clone()
            conf[best_area >= big] = 0
            conf_t[i] = conf
            iou = tiou(loc[i], loc_t[i])
            iou_pred[:, i] = iou
            pc = conf.clone()
            pc[iou < overlap_thresh] = 0
            prop_conf_t[i] = pc
            w = loc[i, :, 0] + loc[i, :, 1]
            prop_loc_t[i, :, 0] = (loc_t[i, :, 0] - loc[i, :, 0]) / (0.5 * w)
            prop_loc_t[i, :, 1] = (loc_t[i, :, 1] - loc[i, :, 1]) / (0.5 * w)
    return loc_t, conf_t, prop_loc_t, prop_conf_t, iou_pred


class EvidenceState:
    """The mutable bits of EvidenceLoss (cls_loss.py:81-118): epoch counter + IBM EMA bins."""

    def __init__(self, num_bins=50, momentum=0.99, ibm_start=10):
        self.num_bins, self.momentum, self.ibm_start = num_bins, momentum, ibm_start
        self.epoch = 0
        self.weight_accum = torch.ones(num_bins)


def evidence_loss_sum(logit, target, state, num_cls=15):
    """EvidenceLoss.forward -> edl_loss, loss_type='log', evidence='exp', with_ibm
    (cls_loss.py:132-168, :212-278).  logit (M,K) of positives, target (M,) in [0,K)."""
    y = torch.eye(num_cls)[target]
    alpha = torch.exp(torch.clamp(logit, -10, 10)) + 1
    S = alpha.sum(1, keepdim=True)
    per = (y * (torch.log(S) - torch.log(alpha))).sum(1)
    if state is not None and state.epoch >= state.ibm_start:
        a = alpha.detach()
        u = num_cls / a.sum(-1, keepdim=True)
        gnorm = (torch.abs(1 / a - u) * y).sum(1)
        ghat = gnorm * logit.detach().abs().sum(1)
        bins = torch.ceil(gnorm * state.num_bins).long()
        for i in range(state.num_bins):
            sel = bins == i + 1
            if int(sel.sum()) > 0:
                state.weight_accum[i] = state.momentum * state.weight_accum[i] + \
                    (1 - state.momentum) * ghat[sel].mean()
        per = state.weight_accum[bins - 1] * per
    return per.sum()


def iou_calibration_mean(logit, ious, num_cls=15):
    """EvidenceLoss.iou_calib(mean=True) (cls_loss.py:120-129)."""
    ious = ious.clone()
    ious[ious < 0] = 1e-3
    u = num_cls / (torch.exp(torch.clamp(logit, -10, 10)) + 1).sum(-1)
    return (-ious * torch.log(1 - u) - (1 - ious) * torch.log(u)).mean()


def focal_loss_sum(prob, target, num_cls=15, alpha=0.25, gamma=2):
    """FocalLoss_Ori(balance_index=0, alpha=0.25, size_average=False) (cls_loss.py:6-78)."""
    a = torch.ones(num_cls) * (1 - alpha)
    a[0] = alpha
    pt = prob.gather(1, target.view(-1, 1)).view(-1) + 1e-6
    return (-torch.pow(1.0 - pt, gamma) * (a[target] * pt.log())).sum()


def actionness_loss(logit, label, weight=0.0, margin=1.0):
    """ActionnessLoss.forward, size_average=False (cls_loss.py:288-339) -> (loss, count)."""
    pred = logit.view(-1)
    label = label.view(-1)
    pos, neg = pred[label > 0], pred[label == 0]
    npos, nneg = pos.numel(), neg.numel()
    top_m = min(npos, nneg) - 1
    use_pred, use_label = pred, label
    if top_m > 0:
        order = neg.sort()[1][:top_m]
        use_pred = torch.cat([pos, neg[order]])
        use_label = torch.cat([torch.ones(npos), torch.zeros(top_m)])
        nneg = top_m
    loss = F.binary_cross_entropy_with_logits(use_pred, use_label, reduction="sum")
    if top_m > 0:
        rank = torch.clamp(margin - neg.max() + pos.max().detach(), min=0.0)
        loss = loss + weight * rank
    return loss, npos + nneg


def multisegment_loss(out, targets, cfg=arch.THUMOS, piou=0.5, cls_loss_type="edl",
                      state=None, act_weight=0.0, act_margin=1.0):
    """MultiSegmentLoss.forward, os_head=True (multisegment_loss.py:92-259) -> the 7-tuple
    (loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act)."""
    K = cfg["num_classes"]
    clip = cfg["frame_num"]
    loc, conf, ploc, pconf = out["loc"], out["conf"], out["prop_loc"], out["prop_conf"]
    center, priors, act, pact = out["center"], out["priors"], out["act"], out["prop_act"]
    loc_t, conf_t, ploc_t, pconf_t, iou_pred = match_anchors(loc, priors, targets, clip, piou)
    pos = conf_t > 0
    ppos = pconf_t > 0
    lp, lt = loc[pos], loc_t[pos]
    loss_l = giou_loss_sum(lp, lt) if lp.numel() > 0 else lp.sum()
    pp, pt = ploc[ppos], ploc_t[ppos]
    loss_pl = F.l1_loss(pp, pt, reduction="sum") if pp.numel() > 0 else pp.sum()
    if lp.numel() > 0:
        w = (lp[:, 0] + lp[:, 1]).unsqueeze(-1)
        cur = 0.5 * w * ploc[pos] + lp
        q = tiou(cur, lt).clamp(min=0)
        loss_ct = F.binary_cross_entropy_with_logits(center[pos].view(-1), q, reduction="sum")
    else:
        loss_ct = lp.sum()

    def cls(logits, tgt):
        p = logits.reshape(-1, K)
        t = tgt.view(-1)
        if cls_loss_type == "focal":
            p = F.softmax(p, dim=1)
        keep = t > 0
        if int(keep.sum()) == 0:
            return torch.tensor(0.0), keep
        if cls_loss_type == "focal":
            return focal_loss_sum(p[keep], t[keep] - 1, K), keep
        return evidence_loss_sum(p[keep], t[keep] - 1, state, K), keep

    loss_c, keep = cls(conf, conf_t)
    loss_act, an = actionness_loss(act.reshape(-1, 1), keep.float(), act_weight, act_margin)
    loss_pc, pkeep = cls(pconf, pconf_t)
    loss_pact, pan = actionness_loss(pact.reshape(-1, 1), pkeep.float(), act_weight, act_margin)
    n = max(int(pos.sum()), 1)
    pn = max(int(ppos.sum()), 1)
    loss_pc = loss_pc / pn
    if cls_loss_type == "edl":  # iou_aware (multisegment_loss.py:234-236, :249-250)
        loss_pc = loss_pc + iou_calibration_mean(pconf.reshape(-1, K), iou_pred.view(-1), K)
    return (loss_l / n, loss_c / n, loss_pl / pn, loss_pc, loss_ct / n, loss_act / an, loss_pact / pan)


# --------------------------------------------------------------------------- ActivityNet losses (config 4)
def evidence_loss_sum_anet(logit, target, epoch, ibm_start=10, coeff=10.0, num_cls=150):
    """anet EvidenceLoss.forward -> edl_loss, loss_type='log', evidence='exp', with_ibm (anet/cls_loss.py:115-148,
    :190-239): the influence-balanced weight is the closed form 1 / (|z|_1 * exp(coeff * g) + 1e-10) (no EMA bins),
    and -- unlike the THUMOS14 file -- |z|_1 is NOT detached (:137, :229)."""
    y = torch.eye(num_cls)[target]
    alpha = torch.exp(torch.clamp(logit, -10, 10)) + 1
    S = alpha.sum(1, keepdim=True)
    per = (y * (torch.log(S) - torch.log(alpha))).sum(1)
    if epoch >= ibm_start:
        a = alpha.detach().clone()
        u = num_cls / a.sum(-1, keepdim=True)
        gnorm = (torch.abs(1 / a - u) * y).sum(1)
        w = 1.0 / (logit.abs().sum(1).reshape(-1) * torch.exp(coeff * gnorm) + 1e-10)
        per = w * per
    return per.sum()


def multisegment_loss_anet(out, targets, cfg=arch.ANET, piou=0.5, epoch=0, ibm_start=10, act_weight=0.1):
    """anet MultiSegmentLoss.forward, edl + os_head (anet/multisegment_loss.py:106-301): every term is computed PER
    SAMPLE with its own normalisers and the batch mean is returned; anchors only regress ground truths whose
    larger extent lies inside their level's bounds (:69-84, :156-166); the refined-stage IoU threshold drops to the
    best IoU among the positives when none reaches piou (:178-184); smooth-L1 for the refinement (:206)."""
    K = cfg["num_classes"]
    clip = cfg["frame_num"]
    priors = out["priors"]
    lb = torch.tensor([cfg["bounds"][int(l)][0] for l in priors[:, 1]], dtype=torch.float32).unsqueeze(1)
    rb = torch.tensor([cfg["bounds"][int(l)][1] for l in priors[:, 1]], dtype=torch.float32).unsqueeze(1)
    acc = [[] for _ in range(7)]
    big = clip * 2
    for i in range(out["loc"].shape[0]):
        loc, conf, ploc, pconf = out["loc"][i], out["conf"][i], out["prop_loc"][i], out["prop_conf"][i]
        center, act, pact = out["center"][i], out["act"][i], out["prop_act"][i]
        with torch.no_grad():
            gt, lab = targets[i][:, :-1], targets[i][:, -1]
            c = priors[:, 0:1]
            left = (c - gt[:, 0][None, :]) * clip
            right = (gt[:, 1][None, :] - c) * clip
            far = torch.max(left, right)
            area = left + right
            area[left < 0] = big
            area[right < 0] = big
            area[far <= lb.expand_as(far)] = big
            area[far > rb.expand_as(far)] = big
            best_area, best = area.min(1)
            loc_t = torch.stack([(priors[:, 0] - gt[best, 0]) * clip, (gt[best, 1] - priors[:, 0]) * clip], 1)
            conf_t = lab[best].clone()
            conf_t[best_area >= big] = 0
            conf_t = conf_t.long()
            iou = tiou(loc, loc_t)
            iou_pred = iou.clone()
            thr = piou
            if int((conf_t > 0).sum()) > 0:
                thr = min(piou, float(iou[conf_t > 0].max()))
            pconf_t = conf_t.clone()
            pconf_t[iou < thr] = 0
            w = loc[:, 0] + loc[:, 1]
            ploc_t = torch.stack([(loc_t[:, 0] - loc[:, 0]) / (0.5 * w), (loc_t[:, 1] - loc[:, 1]) / (0.5 * w)], 1)
        pos, ppos = conf_t > 0, pconf_t > 0
        lp, lt = loc[pos], loc_t[pos]
        loss_l = giou_loss_sum(lp, lt) if lp.numel() > 0 else lp.sum()
        pp, pt = ploc[ppos], ploc_t[ppos]
        loss_pl = F.smooth_l1_loss(pp, pt, reduction="sum") if pp.numel() > 0 else pp.sum()
        if lp.numel() > 0:
            ww = (lp[:, 0] + lp[:, 1]).unsqueeze(-1)
            cur = 0.5 * ww * ploc[pos] + lp
            q = tiou(cur, lt).clamp(min=0)
            loss_ct = F.binary_cross_entropy_with_logits(center[pos].view(-1), q, reduction="sum")
        else:
            loss_ct = lp.sum()
        loss_c = evidence_loss_sum_anet(conf[pos], conf_t[pos] - 1, epoch, ibm_start, num_cls=K) \
            if int(pos.sum()) > 0 else torch.tensor(0.0)
        loss_act, an = actionness_loss(act.reshape(-1, 1), pos.float(), act_weight)
        loss_pc = evidence_loss_sum_anet(pconf[ppos], pconf_t[ppos] - 1, epoch, ibm_start, num_cls=K) \
            if int(ppos.sum()) > 0 else torch.tensor(0.0)
        loss_pact, pan = actionness_loss(pact.reshape(-1, 1), ppos.float(), act_weight)
        n = max(int(pos.sum()), 1)
        pn = max(int(ppos.sum()), 1)
        loss_pc = loss_pc / pn + iou_calibration_mean(pconf.reshape(-1, K), iou_pred, K)
        for lst, v in zip(acc, (loss_l / n, loss_c / n, loss_pl / pn, loss_pc, loss_ct / n,
                                loss_act / an, loss_pact / pan)):
            lst.append(v)
    b = out["loc"].shape[0]
    return tuple(sum(lst) / b for lst in acc)


def train_cost_anet(out, targets, scores, cfg=arch.ANET, lw=1.0, cw=1.0, ctw=1.0, actw=1.0, piou=0.5, epoch=0):
    """anet forward_one_epoch(ssl=False) + the weighted sum of run_one_epoch (anet/train.py:136-224): the boundary
    masks are rows 1 / 2 of the (b,3,T) scores, the level-0 masks are every 8th frame."""
    l, c, pl, pc, ct, la, pla = multisegment_loss_anet(out, targets, cfg, piou, epoch)
    se = scores[:, 1:3]
    ls, le = boundary_bce(out["start"], out["end"], se)
    se8 = se[:, :, ::8]
    a, b = boundary_bce(out["start_loc_prop"], out["end_loc_prop"], se8)
    c2, d = boundary_bce(out["start_conf_prop"], out["end_conf_prop"], se8)
    ls = ls + 0.1 * (a + c2)
    le = le + 0.1 * (b + d)
    parts = dict(loss_l=l * lw, loss_c=c * cw, loss_prop_l=pl * lw, loss_prop_c=pc * cw,
                 loss_ct=ct * ctw, loss_start=ls, loss_end=le, loss_act=la * actw,
                 loss_prop_act=pla * actw)
    return sum(parts.values()), parts


def boundary_bce(start, end, scores):
    """calc_bce_loss (train.py:152-161)."""
    s = torch.tanh(start).mean(-1)
    e = torch.tanh(end).mean(-1)
    return (F.binary_cross_entropy(s.view(-1), scores[:, 0].contiguous().view(-1)),
            F.binary_cross_entropy(e.view(-1), scores[:, 1].contiguous().view(-1)))


def train_cost(out, targets, scores, cfg=arch.THUMOS, lw=1.0, cw=10.0, ctw=1.0, actw=1.0,
               piou=0.5, cls_loss_type="edl", state=None, act_weight=0.0):
    """forward_one_epoch(ssl=False) + the weighted sum of run_one_epoch (train.py:164-235)."""
    l, c, pl, pc, ct, la, pla = multisegment_loss(out, targets, cfg, piou, cls_loss_type, state, act_weight)
    ls, le = boundary_bce(out["start"], out["end"], scores)
    sc4 = scores[:, :, ::4]  # F.interpolate(scale_factor=1/4), nearest (train.py:188-192)
    a, b = boundary_bce(out["start_loc_prop"], out["end_loc_prop"], sc4)
    c2, d = boundary_bce(out["start_conf_prop"], out["end_conf_prop"], sc4)
    ls = ls + 0.1 * (a + c2)
    le = le + 0.1 * (b + d)
    parts = dict(loss_l=l * lw, loss_c=c * cw, loss_prop_l=pl * lw, loss_prop_c=pc * cw,
                 loss_ct=ct * ctw, loss_start=ls, loss_end=le, loss_act=la * actw,
                 loss_prop_act=pla * actw)
    cost = sum(parts.values())
    return cost, parts


def triplet_cost(anchor, positive, negative, ssl_weight):
    """forward_one_epoch(ssl=True) (train.py:177-184) times --ssl."""
    ws = (1, 0.1, 0.1)
    return sum(F.triplet_margin_loss(a, p, n) * w for a, p, n, w in zip(anchor, positive, negative, ws)) * ssl_weight


def adam_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam with L2 weight decay folded into the gradient (train.py:321-323)."""
    g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = v.sqrt() / (1 - b2 ** step) ** 0.5 + eps
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


# --------------------------------------------------------------------------- inference (a16, a17)
def decode_predictions(out, idx, offset, fps, cfg=arch.THUMOS):
    """parse_output + decode_predictions for sample idx, use_edl + os_head (test.py:79-140)."""
    clip = cfg["frame_num"]
    loc, ploc, pri = out["loc"][idx], out["prop_loc"][idx], out["priors"]
    w = loc[:, :1] + loc[:, 1:]
    loc = 0.5 * w * ploc + loc
    seg = torch.cat([pri[:, :1] * clip - loc[:, :1], pri[:, :1] * clip + loc[:, 1:]], -1)
    seg = seg.clamp(min=0, max=clip)
    seg = (seg + offset) / fps
    unct = (out["unct"][idx] + out["prop_unct"][idx]) / 2.0
    actn = (out["act"][idx].squeeze(-1).sigmoid() + out["prop_act"][idx].squeeze(-1).sigmoid()) / 2.0
    score = (dirichlet_mean(out["conf"][idx]) + dirichlet_mean(out["prop_conf"][idx])) / 2.0
    score = score * out["center"][idx].sigmoid() * actn.unsqueeze(-1)
    return seg, score.transpose(1, 0).contiguous(), unct, actn


def fuse_outputs(rgb, flow):
    """Two-stream fusion of parse_output(fusion=True) (test.py:90-108): the two networks' raw outputs -- and their
    uncertainty maps -- are averaged before decoding.  (With os_head the reference adds a squeezed (126,) and an
    unsqueezed (126,1) actionness and breaks; the elementwise average it means is restated here -- pinned with the flow
    actionness handed to the reference already squeezed.)"""
    keys = ("loc", "conf", "prop_loc", "prop_conf", "center", "act", "prop_act", "unct", "prop_unct")
    fused = {k: (rgb[k] + flow[k]) / 2.0 for k in keys}
    fused["priors"] = rgb["priors"]
    return fused


def filtering(seg, score_cls, unct, actn, conf_thresh=0.01):
    """filtering (test.py:143-162) -> (n,5) rows [start,end,score,unct,act] or None."""
    m = (score_cls > conf_thresh) & (actn > 0.5)
    if int(m.sum()) == 0:
        return None
    return torch.cat([seg[m], score_cls[m, None], unct[m, None], actn[m, None]], -1)


def softnms_v2(segments, sigma=0.5, top_k=5000, score_threshold=0.001):
    """softnms_v2 (segment_utils.py:128-162) in torch ops; returns (rows, count, kept_mask)."""
    segments = segments.clone()
    ts, te, sc = segments[:, 0], segments[:, 1], segments[:, 2]
    done = sc < -1
    undone = sc >= score_threshold
    while int(undone.sum()) > 1 and int(done.sum()) < top_k:
        j = int(undone.nonzero()[sc[undone].argmax()])
        undone[j] = False
        done[j] = True
        a, b = ts[undone], te[undone]
        inter = torch.clamp(b.clamp(max=te[j]) - a.clamp(min=ts[j]), min=0)
        width = torch.clamp(te[j] - ts[j], min=1e-5)
        iou = inter / (width + (b - a) - inter)
        sc[undone] *= torch.exp(-iou ** 2 / sigma)
        undone[sc < score_threshold] = False
    rows = torch.cat([torch.stack([ts[done], te[done], sc[done]], -1), segments[done, 3:]], -1)
    return rows, int(done.sum()), done


def softnms_v2_c(segments, sigma=0.5, top_k=5000, score_threshold=0.001):
    """Same definition through oracle/bmp_ref.c (fast path for the CPU baseline)."""
    s3 = segments[:, :3].detach().contiguous().float().clone()
    n = s3.shape[0]
    done = torch.zeros(n, dtype=torch.uint8)
    cnt = _lib().otal_oracle_softnms(_fp(s3), n, ctypes.c_float(sigma), top_k,
                                     ctypes.c_float(score_threshold), _fp(done))
    keep = done.bool()
    rows = torch.cat([s3[keep], segments[keep, 3:].float()], -1)
    return rows, cnt, keep


def to_torch(params_np, requires_grad=False):
    """numpy param dict -> torch dict; trainables get requires_grad (BN affine+stats frozen,
    BDNet.py:39-49)."""
    out = {}
    for k, v in params_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        if requires_grad and t.is_floating_point() and ".bn." not in k:
            t.requires_grad_(True)
        out[k] = t
    return out
