"""CoarsePyramid (AFSD/thumos14/BDNet.py:116-432) as TWO autograd nodes with hand-written tapes, the way the I3D backbone is
one (common/i3d_backbone.py):

  TrunkFunction      Mixed_4f, Mixed_5c -> level-packed pyramid (BDNet.py:310-326), frame-level deconv (:324-326), loc / conf
                     towers (:333-336)
  BranchesFunction   both ProposalBranches (BDNet.py:105-113, :386-391) on the level-packed maps

Between them sit the fused head launches and the no-grad proposal windows (their inputs are the towers' outputs).  The
values are those of the module-by-module composition in BDNet.py (same kernels: `ops.conv_forward`, `ops.gn_relu_*`,
BoundaryMaxPooling); what the explicit tapes buy:
  * no tensor-library glue: upsample + add + cat + upsample become ONE launch (`ops.pyramid_merge_forward`), stride-2 blocks
    write their level of the packed buffer in place, the ProposalBranch concatenation is written in place by its three
    producers, gradients that meet in one tensor are added inside the GroupNorm-backward launch that consumes them
    (`otal_gn_relu_bwd_sum`) instead of by autograd's accumulation kernels;
  * independent chains run side by side on the branch lane (`ops.branch_lane`): the frame-level deconv beside the stride-2
    pyramid + towers, the roi path beside the level path -- forward AND backward (autograd's engine serialises nodes; inside
    one node the order and the streams are ours);
  * weight gradients go to the side lane in the order the data-gradient chain frees them (`ops.side_wgrads`).
Applies to the THUMOS14 layout (two projections, six levels); everything else runs the module-by-module path.
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib as L
from ..common import ops
from ..prop_pooling import boundary_pooling_op as bp

ONE = (1, 1, 1)
K3, K1, S2 = (3, 1, 1), (1, 1, 1), (2, 1, 1)


def _bs_cs(t):
    """(batch stride, channel stride) of a (B,C,T) tensor whose positions are dense (a level / channel slice included)."""
    B, C, T = t.shape
    if T > 1 and t.stride(2) != 1:
        raise RuntimeError("pyramid_fused: positions must be dense")
    cs = t.stride(1) if C > 1 else T
    return (t.stride(0) if B > 1 else cs * C), cs


def _gn_bwd_raw(dy, c, gamma, beta, stats, groups, levels):
    """(dc, partial) of otal_gn_relu_bwd: no tensor-library call (safe on a lane)."""
    B, C, T = c.shape
    if dy.dim() != 3 or tuple(dy.shape) != (B, C, T) or dy.stride(2) != 1 or dy.stride(1) != T or dy.stride(0) < C * T:
        raise RuntimeError("pyramid_fused: gradient layout")
    nlev, lev = ops._lev_arg(levels)
    dc = torch.empty_like(c)
    partial = torch.empty((B, 3, C), dtype=torch.float32, device=c.device)
    L.check(L.lib().otal_gn_relu_bwd(L.ptr(dy), ctypes.c_int64(dy.stride(0)), L.ptr(c), L.ptr(gamma), L.ptr(beta), L.ptr(stats),
                                     L.ptr(dc), L.ptr(partial), B, C, T, groups, 1, nlev, lev, L.stream()), "otal_gn_relu_bwd")
    return dc, partial


def _gn_bwd_adds(adds, c, gamma, beta, stats, groups, levels):
    """GroupNorm + ReLU backward of a block whose output gradient is the SUM of `adds` [(tensor, positions or None)]: the
    sum happens while the launch stages the map (otal_gn_relu_bwd_sum).  -> (dc, partial)."""
    B, C, T = c.shape
    n = len(adds)
    dc = torch.empty_like(c)
    partial = torch.empty((B, 3, C), dtype=torch.float32, device=c.device)
    nlev, lev = ops._lev_arg(levels)
    strides = [_bs_cs(t) for t, _ in adds]
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in adds])
    I64 = ctypes.c_int64 * n
    L.check(L.lib().otal_gn_relu_bwd_sum(n, ptrs, I64(*[s[0] for s in strides]), I64(*[s[1] for s in strides]),
                                         (ctypes.c_int * n)(*[t.shape[2] if Ta is None else Ta for t, Ta in adds]),
                                         L.ptr(c), L.ptr(gamma), L.ptr(beta), L.ptr(stats), L.ptr(dc), L.ptr(partial),
                                         B, C, T, groups, 1, nlev, lev, L.stream()), "otal_gn_relu_bwd_sum")
    return dc, partial


class _Wgrads:
    """The node's weight gradients: recorded on the side lane, handed back in parameter order."""

    def __init__(self, device):
        self.side = ops.side_wgrads(device)
        self.in_slots = True

    def one(self, x, dc, w, k, s, sv=False, levels=None):
        slot = ops.grad_slot(w)
        self.in_slots = self.in_slots and slot is not None
        return self.side.wgrad(x, dc, w.shape, k, s, spatial_valid=sv, levels=levels, out=slot)

    def pair(self, xs, dcs, ws, k, s, levels=None):
        slots = (ops.grad_slot(ws[0]), ops.grad_slot(ws[1]))
        self.in_slots = self.in_slots and slots[0] is not None and slots[1] is not None
        return self.side.wgrad_pair(xs, dcs, ws[0].shape, k, s, levels, slots)


def _conv_gn_pair(xs, ps, k, levels, groups, eps, outs=None):
    """Two sibling blocks: (c, y, stats) per problem; pair launches where the library has them."""
    ws, bs = (ps[0][0], ps[1][0]), (ps[0][1], ps[1][1])
    cs = ops.conv_forward_pair(xs, ws, k, ONE, bs, levels)
    if cs is None:
        cs = [ops.conv_forward(x, w, k, ONE, shift=b, levels=levels) for x, w, b in zip(xs, ws, bs)]
    r = ops.gn_relu_forward_pair(cs, (ps[0][2], ps[1][2]), (ps[0][3], ps[1][3]), groups, eps, True, levels, outs=outs)
    if r is None:
        r = [ops.gn_relu_forward(c, p[2], p[3], groups, eps, True, levels, out=None if outs is None else outs[i])
             for i, (c, p) in enumerate(zip(cs, ps))]
    return [(cs[i], r[i][0], r[i][1]) for i in range(2)]


def _gn_bwd_pair(dys, cs, ps, stats, groups, levels):
    """[(dc, partial)] * 2 of two sibling blocks (no tensor-library call)."""
    B, C, T = cs[0].shape
    nlev, lev = ops._lev_arg(levels)
    ok = ops.PAIR_LAUNCHES and all(d.dim() == 3 and d.stride(2) == 1 and d.stride(1) == T and d.stride(0) >= C * T for d in dys)
    if ok:
        dcs = [torch.empty_like(c) for c in cs]
        parts = [torch.empty((B, 3, C), dtype=torch.float32, device=c.device) for c in cs]
        pp = ops._pp
        rc = L.lib().otal_gn_relu_bwd_pair(pp(*dys), (ctypes.c_int64 * 2)(dys[0].stride(0), dys[1].stride(0)), pp(*cs),
                                           pp(ps[0][2], ps[1][2]), pp(ps[0][3], ps[1][3]), pp(*stats), pp(*dcs), pp(*parts),
                                           B, C, T, groups, 1, nlev, lev, L.stream())
        if rc == 0:
            return list(zip(dcs, parts))
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "otal_gn_relu_bwd_pair")
    return [_gn_bwd_raw(dys[i], cs[i], ps[i][2], ps[i][3], stats[i], groups, levels) for i in range(2)]


def _dgrad_pair(dcs, ws, x_shape, k, levels):
    r = ops.conv_dgrad_pair(dcs, ws, x_shape, k, ONE, levels)
    if r is None:
        r = [ops.conv_dgrad(dc, w, x_shape, k, ONE, levels=levels) for dc, w in zip(dcs, ws)]
    return r


class TrunkFunction(Function):
    """apply(cfg, x1, x2, *params) -> (loc_feat, conf_feat, frame_level_feat).
    params: (weight, bias, gamma, beta) of pyramids[0..5], deconv[0..2], loc_tower[0..1], conf_tower[0..1].
    cfg: dict(levels, up, groups, eps, k0, k1)."""

    @staticmethod
    def forward(ctx, cfg, x1, x2, *params):
        lev, up, G, eps = cfg["levels"], cfg["up"], cfg["groups"], cfg["eps"]
        P = [params[4 * i:4 * i + 4] for i in range(13)]
        pyr, dec, lt, ct = P[0:6], P[6:9], P[9:11], P[11:13]
        B = x1.shape[0]
        C = pyr[0][0].shape[0]
        t0, total = lev[1], lev[-1]
        tape = {}
        # projections (BDNet.py:310-319); the Mixed_4f one may already be running on the early lane (ops.early_lane)
        early = ops.EARLY_RESULTS.pop(x1.data_ptr(), None)
        if early is None:
            c0 = ops.conv_forward(x1, pyr[0][0], cfg["k0"], ONE, shift=pyr[0][1], spatial_valid=True).view(B, C, t0)
            p0, st0 = ops.gn_relu_forward(c0, pyr[0][2], pyr[0][3], G, eps, True, None)
        c1 = ops.conv_forward(x2, pyr[1][0], cfg["k1"], ONE, shift=pyr[1][1], spatial_valid=True).view(B, C, t0 // 2)
        p1, st1 = ops.gn_relu_forward(c1, pyr[1][2], pyr[1][3], G, eps, True, None)
        if early is not None:
            c0, p0, st0, elane = early
            elane.join()
        packed, frame_in = ops.pyramid_merge_forward(p0, p1, total, up)
        tape["proj"] = (c0, st0, c1, st1)
        # the frame-level deconv (BDNet.py:324-326) on the branch lane, beside the stride-2 levels and the towers
        lane = ops.branch_lane(x1.device)
        use_lane = lane.on and ops.PYRAMID_LANE
        dtape = []

        def deconv():
            x = frame_in
            for i, k in enumerate((K3, K3, K1)):
                c = ops.conv_forward(x, dec[i][0], k, ONE, shift=dec[i][1])
                y, st = ops.gn_relu_forward(c, dec[i][2], dec[i][3], G, eps, True, None)
                dtape.append((x, c, st, k))
                x = y
            return x
        if use_lane:
            lane.fork()
            with lane:
                frame = deconv()
        # stride-2 levels (BDNet.py:320-321), each written into its slice of the packed buffer
        ptape = []
        for l in range(2, len(lev) - 1):
            x = packed[:, :, lev[l - 1]:lev[l]]
            c = ops.conv_forward(x, pyr[l][0], K3, S2, shift=pyr[l][1])
            _, st = ops.gn_relu_forward(c, pyr[l][2], pyr[l][3], G, eps, True, None, out=packed[:, :, lev[l]:lev[l + 1]])
            ptape.append((c, st))
        # towers (BDNet.py:333-336): loc / conf siblings stage by stage
        s1 = _conv_gn_pair((packed, packed), (lt[0], ct[0]), K3, lev, G, eps)
        s2 = _conv_gn_pair((s1[0][1], s1[1][1]), (lt[1], ct[1]), K3, lev, G, eps)
        if use_lane:
            lane.join()
        else:
            frame = deconv()
        loc_feat, conf_feat = s2[0][1], s2[1][1]
        ctx.cfg = cfg
        ctx.tape = (tape["proj"], packed, frame_in, dtape, ptape,
                    ((s1[0][0], s1[0][2]), (s1[1][0], s1[1][2]), (s2[0][0], s2[0][2]), (s2[1][0], s2[1][2])),
                    (s1[0][1], s1[1][1]))
        ctx.save_for_backward(x1, x2, *params)
        return loc_feat, conf_feat, frame

    @staticmethod
    def backward(ctx, d_loc, d_conf, d_frame):
        cfg = ctx.cfg
        lev, up, G = cfg["levels"], cfg["up"], cfg["groups"]
        x1, x2 = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        P = [params[4 * i:4 * i + 4] for i in range(13)]
        pyr, dec, lt, ct = P[0:6], P[6:9], P[9:11], P[11:13]
        (c0, st0, c1, st1), packed, frame_in, dtape, ptape, tw, (l0, cf0) = ctx.tape
        B, C, t0 = c0.shape
        dev = x1.device
        grads = [None] * len(params)
        WG = _Wgrads(dev)
        sums = []                           # (block index, partial, batches): the batch sums are taken after the lanes have joined

        def put(i, dw):
            grads[4 * i] = dw
        lane = ops.branch_lane(dev)
        use_lane = lane.on and ops.PYRAMID_LANE and d_frame is not None
        d_loc, d_conf = d_loc.contiguous(), d_conf.contiguous()
        if d_frame is not None:
            d_frame = d_frame.contiguous()
        d_frame_in = None

        lane_wgrads = []                    # recorded on the lane, handed to the side lane only after the join: the side lane
                                            # waits for the MAIN lane's position, which does not cover the branch lane's work

        hold = []                           # every tensor a lane kernel reads stays alive until the join: the caching allocator
                                            # believes the main stream owns it and would hand a freed block to the next request

        def deconv_bwd():
            dy = d_frame
            for i in (2, 1, 0):
                x, c, st, k = dtape[i]
                dc, part = _gn_bwd_raw(dy, c, dec[i][2], dec[i][3], st, G, None)
                sums.append((6 + i, part, B))
                lane_wgrads.append((6 + i, x, dc, dec[i][0], k))
                dy = ops.conv_dgrad(dc, dec[i][0], x.shape, k, ONE)
                hold.append(dy)
            return dy
        if use_lane:
            lane.fork()
            with lane:
                d_frame_in = deconv_bwd()
        # towers, stage 2 then stage 1
        r = _gn_bwd_pair((d_loc, d_conf), (tw[2][0], tw[3][0]), (lt[1], ct[1]), (tw[2][1], tw[3][1]), G, lev)
        sums += [(10, r[0][1], B), (12, r[1][1], B)]
        dws = WG.pair((l0, cf0), (r[0][0], r[1][0]), (lt[1][0], ct[1][0]), K3, ONE, lev)
        put(10, dws[0]); put(12, dws[1])
        dmid = _dgrad_pair((r[0][0], r[1][0]), (lt[1][0], ct[1][0]), l0.shape, K3, lev)
        r = _gn_bwd_pair(dmid, (tw[0][0], tw[1][0]), (lt[0], ct[0]), (tw[0][1], tw[1][1]), G, lev)
        sums += [(9, r[0][1], B), (11, r[1][1], B)]
        dws = WG.pair((packed, packed), (r[0][0], r[1][0]), (lt[0][0], ct[0][0]), K3, ONE, lev)
        put(9, dws[0]); put(11, dws[1])
        da, db = _dgrad_pair((r[0][0], r[1][0]), (lt[0][0], ct[0][0]), packed.shape, K3, lev)
        if not use_lane:                    # (never while the branch lane is forked: a flush may CUT the lane-graph capture, and a
            WG.side.flush()                 #  capture cannot end before a forked stream has rejoined it -- found with OTAL_WGRAD_CHUNK=5)
        # stride-2 levels from the top: the level's gradient = the towers' two + the data gradient of the level above
        dnext = None
        for l in range(len(lev) - 2, 1, -1):
            c, st = ptape[l - 2]
            adds = [(da[:, :, lev[l]:lev[l + 1]], None), (db[:, :, lev[l]:lev[l + 1]], None)]
            if dnext is not None:
                adds.append((dnext, None))
            dc, part = _gn_bwd_adds(adds, c, pyr[l][2], pyr[l][3], st, G, None)
            sums.append((l, part, B))
            x = packed[:, :, lev[l - 1]:lev[l]]
            put(l, WG.one(x, dc, pyr[l][0], K3, S2))
            dnext = ops.conv_dgrad(dc, pyr[l][0], x.shape, K3, S2)
        if use_lane:
            lane.join()
        elif d_frame is not None:
            d_frame_in = deconv_bwd()
        else:
            d_frame_in = torch.zeros_like(frame_in)
        for i, x, dc, w, k in lane_wgrads:
            put(i, WG.one(x, dc, w, k, ONE))
        dp0, dp1 = ops.pyramid_merge_backward(da, db, d_frame_in, dnext, t0, up)
        WG.side.flush()
        # projections: the small one first (its data gradient is what the backbone's backward starts from)
        dxs = [None, None]
        elane = ops.early_lane(dev)
        for i, (x, c, st, k, dp) in ((1, (x2, c1, st1, cfg["k1"], dp1)), (0, (x1, c0, st0, cfg["k0"], dp0))):
            dc, part = _gn_bwd_raw(dp, c, pyr[i][2], pyr[i][3], st, G, None)
            sums.append((i, part, B))
            dc5 = dc.view(B, C, dc.shape[2], 1, 1)
            put(i, WG.one(x, dc5, pyr[i][0], k, ONE, sv=True))
            if ctx.needs_input_grad[1 + i]:
                dgrad = lambda: (ops.conv_dgrad_collapse(dc5, pyr[i][0], x.shape) if ops.is_full_collapse(x.shape, k, ONE, True)
                                 else ops.conv_dgrad(dc5, pyr[i][0], x.shape, k, ONE, spatial_valid=True))
                if i == 0 and ops.EARLY_PROJ and elane.on and ops.PYRAMID_LANE and ctx.needs_input_grad[2] and ops.LANES is None:
                    # the gradient of Mixed_4f is needed only after Mixed_5c / 5b / MaxPool3d_5a have been walked back:
                    # its GEMM runs on the early lane beside them, the backbone joins where it picks the gradient up.
                    # (Eager launches only: a lane-graph capture is CUT where the weight-gradient lane takes a chunk, and a
                    #  capture cannot end while a forked stream has not rejoined it.)
                    elane.fork()
                    with elane:
                        dxs[0] = dgrad()
                    ops.PENDING_JOINS[dxs[0].data_ptr()] = elane
                    ctx.early_keep = (dc5, dxs[0])
                else:
                    dxs[i] = dgrad()
        for i, part, nb in sums:            # d_gamma, d_beta, d_bias: deferred into the arena, or summed now
            dg, dbe, dbi = ops._gn_sums(part, P[i][2], P[i][3], P[i][1], C, nb)
            grads[4 * i + 1], grads[4 * i + 2], grads[4 * i + 3] = dbi, dg, dbe
        WG.side.node_end(WG.in_slots)
        ctx.tape = None
        return (None, dxs[0], dxs[1]) + tuple(grads)


class BranchesFunction(Function):
    """apply(cfg, loc_feat, conf_feat, frame_level_feat, segments, frame_segments, *params)
    -> (loc_prop_feat, conf_prop_feat, loc_lr[:, :, :t0], conf_lr[:, :, :t0]).
    params: (weight, bias, gamma, beta) of cur_point_conv, lr_conv, roi_conv, proposal_conv of the loc branch, then of the
    conf branch.  The level-0 slices of the lr maps are outputs of their own: the boundary losses read only those, and a
    slice taken outside would come back as a zero-filled full-size gradient."""

    @staticmethod
    def forward(ctx, cfg, loc_feat, conf_feat, frame, segments, frame_segments, *params):
        lev, G, eps = cfg["levels"], cfg["groups"], cfg["eps"]
        P = [params[4 * i:4 * i + 4] for i in range(8)]
        cur, lr, roi, prop = (P[0], P[4]), (P[1], P[5]), (P[2], P[6]), (P[3], P[7])
        B, C, T = loc_feat.shape
        Cp = cur[0][0].shape[0]
        dev = loc_feat.device
        cat = [torch.empty((B, 4 * Cp, T), dtype=torch.float32, device=dev) for _ in range(2)]      # [roi | pooled | short]
        lane = ops.branch_lane(dev)
        use_lane = lane.on and ops.PYRAMID_LANE

        def roi_path():
            pooled = bp.bmp_forward(frame, frame_segments)                                          # BDNet.py:109 (shared)
            return pooled, _conv_gn_pair((pooled, pooled), roi, K1, lev, G, eps, outs=(cat[0][:, :Cp], cat[1][:, :Cp]))
        if use_lane:
            lane.fork()
            with lane:
                pooled_roi, r_roi = roi_path()
        r_cur = _conv_gn_pair((loc_feat, conf_feat), cur, K1, lev, G, eps, outs=(cat[0][:, 3 * Cp:], cat[1][:, 3 * Cp:]))
        r_lr = _conv_gn_pair((loc_feat, conf_feat), lr, K1, lev, G, eps)
        for i in range(2):                                  # pooled rows straight into their slice of the concatenation
            bp.bmp_forward_levels_to(r_lr[i][1], segments, lev, lev, cat[i][:, Cp:3 * Cp])
        if use_lane:
            lane.join()
        else:
            pooled_roi, r_roi = roi_path()
        r_prop = _conv_gn_pair(cat, prop, K1, lev, G, eps)
        t0 = lev[1]
        lr_l, lr_c = r_lr[0][1], r_lr[1][1]
        ctx.cfg = cfg
        ctx.tape = (cat, pooled_roi, [(r[i][0], r[i][2]) for r in (r_cur, r_lr, r_roi, r_prop) for i in range(2)], (lr_l, lr_c))
        ctx.save_for_backward(loc_feat, conf_feat, frame, segments, frame_segments, *params)
        return r_prop[0][1], r_prop[1][1], lr_l[:, :, :t0], lr_c[:, :, :t0]

    @staticmethod
    def backward(ctx, d_prop_l, d_prop_c, d_lr_l0, d_lr_c0):
        cfg = ctx.cfg
        lev, G = cfg["levels"], cfg["groups"]
        loc_feat, conf_feat, frame, segments, frame_segments = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        P = [params[4 * i:4 * i + 4] for i in range(8)]
        cur, lr, roi, prop = (P[0], P[4]), (P[1], P[5]), (P[2], P[6]), (P[3], P[7])
        cat, pooled_roi, cst, (lr_l, lr_c) = ctx.tape
        (c_cur, c_lr, c_roi, c_prop) = [(cst[2 * j], cst[2 * j + 1]) for j in range(4)]
        B, C, T = loc_feat.shape
        Cp = cur[0][0].shape[0]
        dev = loc_feat.device
        t0 = lev[1]
        grads = [None] * len(params)
        WG = _Wgrads(dev)
        sums = []

        def put(j, dws):                    # block j of the loc branch, j + 4 of the conf branch
            grads[4 * j], grads[4 * (j + 4)] = dws[0], dws[1]
        zero = lambda like: torch.zeros_like(like)
        d_prop = (d_prop_l.contiguous() if d_prop_l is not None else zero(c_prop[0][0]),
                  d_prop_c.contiguous() if d_prop_c is not None else zero(c_prop[1][0]))
        # proposal_conv
        r = _gn_bwd_pair(d_prop, (c_prop[0][0], c_prop[1][0]), prop, (c_prop[0][1], c_prop[1][1]), G, lev)
        sums += [(3, r[0][1]), (7, r[1][1])]
        put(3, WG.pair(cat, (r[0][0], r[1][0]), (prop[0][0], prop[1][0]), K1, ONE, lev))
        dcat = _dgrad_pair((r[0][0], r[1][0]), (prop[0][0], prop[1][0]), cat[0].shape, K1, lev)
        lane = ops.branch_lane(dev)
        use_lane = lane.on and ops.PYRAMID_LANE
        d_frame = None
        hold = []

        def roi_bwd():
            rr = _gn_bwd_pair((dcat[0][:, :Cp], dcat[1][:, :Cp]), (c_roi[0][0], c_roi[1][0]), roi, (c_roi[0][1], c_roi[1][1]), G, lev)
            # the two branches' gradients w.r.t. the shared pooled map: the second data gradient adds to the first
            dpool = ops.conv_dgrad(rr[0][0], roi[0][0], pooled_roi.shape, K1, ONE, levels=lev)
            ops.conv_dgrad(rr[1][0], roi[1][0], pooled_roi.shape, K1, ONE, levels=lev, out=dpool, accumulate=True)
            hold.append(dpool)              # (read by a lane kernel: alive until the join, see TrunkFunction.backward)
            return rr, bp.bmp_backward(dpool, frame, frame_segments)
        if use_lane:
            lane.fork()
            with lane:
                rr, d_frame = roi_bwd()
        # pooled rows -> the lr maps; their level-0 columns also receive the boundary losses' gradients
        dlr = []
        for i, d_l0 in enumerate((d_lr_l0, d_lr_c0)):
            dpool = bp.bmp_backward_levels_from(dcat[i][:, Cp:3 * Cp], (lr_l, lr_c)[i], segments, lev, lev)
            adds = [(dpool, None)]
            if d_l0 is not None:
                adds.append((d_l0.contiguous(), t0))
            dlr.append(_gn_bwd_adds(adds, c_lr[i][0], lr[i][2], lr[i][3], c_lr[i][1], G, lev))
        sums += [(1, dlr[0][1]), (5, dlr[1][1])]
        put(1, WG.pair((loc_feat, conf_feat), (dlr[0][0], dlr[1][0]), (lr[0][0], lr[1][0]), K1, ONE, lev))
        dfeat = _dgrad_pair((dlr[0][0], dlr[1][0]), (lr[0][0], lr[1][0]), loc_feat.shape, K1, lev)
        # cur_point_conv: its data gradient adds to the lr path's
        rc = _gn_bwd_pair((dcat[0][:, 3 * Cp:], dcat[1][:, 3 * Cp:]), (c_cur[0][0], c_cur[1][0]), cur, (c_cur[0][1], c_cur[1][1]), G, lev)
        sums += [(0, rc[0][1]), (4, rc[1][1])]
        put(0, WG.pair((loc_feat, conf_feat), (rc[0][0], rc[1][0]), (cur[0][0], cur[1][0]), K1, ONE, lev))
        for i in range(2):
            ops.conv_dgrad(rc[i][0], cur[i][0], loc_feat.shape, K1, ONE, levels=lev, out=dfeat[i], accumulate=True)
        if use_lane:
            lane.join()
        else:
            rr, d_frame = roi_bwd()
        # (recorded after the join: the side lane waits for the main lane's position only)
        put(2, WG.pair((pooled_roi, pooled_roi), (rr[0][0], rr[1][0]), (roi[0][0], roi[1][0]), K1, ONE, lev))
        sums += [(2, rr[0][1]), (6, rr[1][1])]
        for j, part in sums:
            dg, dbe, dbi = ops._gn_sums(part, P[j][2], P[j][3], P[j][1], part.shape[2], part.shape[0])
            grads[4 * j + 1], grads[4 * j + 2], grads[4 * j + 3] = dbi, dg, dbe
        WG.side.node_end(WG.in_slots)
        ctx.tape = None
        return (None, dfeat[0], dfeat[1], d_frame, None, None) + tuple(grads)


def _block_params(block):
    unit, gn = block[0], block[1]
    conv = unit.conv1d if hasattr(unit, "conv1d") else unit.conv3d
    return (conv.weight, conv.bias, gn.weight, gn.bias)


def eligible(pyramid, feat_dict):
    """The THUMOS14 layout the two nodes are written for (BDNet.py:116-203): two projections, halving levels, GroupNorm(32)
    everywhere, bias everywhere, CUDA tensors."""
    from ..prop_pooling import boundary_pooling_op as _bp
    if len(pyramid.projection_inputs) != 2 or pyramid.fpn_strides is not None or _bp.COMPAT_REFERENCE_BWD:
        return False
    lv = pyramid.level_lengths
    if any(lv[i] != 2 * lv[i + 1] for i in range(len(lv) - 1)) or len(lv) < 3 or lv[0] > 256 or pyramid.frame_num % lv[0]:
        return False
    x1 = feat_dict[pyramid.projection_inputs[0]]
    if not x1.is_cuda or x1.dtype != torch.float32 or pyramid.frame_num > 256:
        return False
    blocks = list(pyramid.pyramids) + [pyramid.loc_tower[0], pyramid.loc_tower[1], pyramid.conf_tower[0], pyramid.conf_tower[1]]
    for b in blocks:
        if b[1].num_groups != 32 or _block_params(b)[1] is None or b[1].weight.shape[0] // 32 not in (16, 32):
            return False
    return True


def trunk(pyramid, feat_dict):
    p = pyramid
    x1, x2 = (feat_dict[ep] for ep in p.projection_inputs)
    params = []
    for b in p.pyramids:
        params += _block_params(b)
    for i in (0, 3, 6):
        unit, gn = p.deconv[i], p.deconv[i + 1]
        params += [unit.conv1d.weight, unit.conv1d.bias, gn.weight, gn.bias]
    for b in (p.loc_tower[0], p.loc_tower[1], p.conf_tower[0], p.conf_tower[1]):
        params += _block_params(b)
    cfg = dict(levels=tuple(p.levels), up=p.frame_num // p.level_lengths[0], groups=32, eps=p.pyramids[0][1].eps,
               k0=tuple(p.pyramids[0][0]._kernel_shape), k1=tuple(p.pyramids[1][0]._kernel_shape))
    return TrunkFunction.apply(cfg, x1, x2, *params)


def branches(pyramid, loc_feat, conf_feat, frame, segments, frame_segments):
    p = pyramid
    params = []
    for br in (p.loc_proposal_branch, p.conf_proposal_branch):
        for b in (br.cur_point_conv, br.lr_conv, br.roi_conv, br.proposal_conv):
            params += _block_params(b)
    cfg = dict(levels=tuple(p.levels), groups=32, eps=p.pyramids[0][1].eps)
    return BranchesFunction.apply(cfg, loc_feat, conf_feat, frame, segments, frame_segments, *params)


def eligible_static(pyramid):
    """eligible() for what is known before the backbone runs."""
    lv = pyramid.level_lengths
    return (len(pyramid.projection_inputs) == 2 and pyramid.fpn_strides is None and not bp.COMPAT_REFERENCE_BWD
            and all(lv[i] == 2 * lv[i + 1] for i in range(len(lv) - 1)) and len(lv) >= 3 and lv[0] <= 256
            and pyramid.frame_num % lv[0] == 0 and pyramid.frame_num <= 256)


def early_projection_hooks(pyramid):
    """{endpoint: hook} for ops.ENDPOINT_HOOKS: the Mixed_4f projection starts on the early lane the moment the backbone has
    the endpoint (TrunkFunction.forward picks the result up)."""
    p = pyramid
    if not (ops.FUSED_PYRAMID and ops.EARLY_PROJ and ops.PYRAMID_LANE and eligible_static(p)):
        return None
    w, b, gamma, beta = _block_params(p.pyramids[0])
    k0, eps = tuple(p.pyramids[0][0]._kernel_shape), p.pyramids[0][1].eps
    t0 = p.level_lengths[0]

    def hook(x1):
        lane = ops.early_lane(x1.device)
        if not lane.on or not x1.is_cuda or x1.dtype != torch.float32:
            return
        ops.EARLY_RESULTS.clear()                       # (a forward pass whose pyramid never ran leaves nothing behind)
        lane.fork()
        with lane, torch.no_grad():
            B = x1.shape[0]
            c0 = ops.conv_forward(x1, w, k0, ONE, shift=b, spatial_valid=True).view(B, w.shape[0], t0)
            p0, st0 = ops.gn_relu_forward(c0, gamma, beta, 32, eps, True, None)
        ops.EARLY_RESULTS[x1.data_ptr()] = (c0, p0, st0, lane)
    return {p.projection_inputs[0]: hook}
