"""Training step of OpenTAL/AFSD on MI355X (reference: AFSD/thumos14/train.py).

`calc_bce_loss` / `forward_one_epoch` keep the reference's names and arithmetic
(train.py:152-201); `DetectorTrainer` replaces run_one_epoch's inner loop (train.py:221-252):

  * one process per GPU; gradients are all-reduced over RCCL (torch.distributed backend "nccl")
    in a few large contiguous buckets of a FLAT gradient arena, launched from autograd hooks as
    soon as a bucket is complete, so the pyramid/head buckets travel over xGMI while the I3D
    backward (97 % of the FLOPs) is still running;
  * parameters, gradients and Adam moments live in flat fp32 arenas: the optimizer is ONE
    kernel launch (otal_adam_flat) instead of ~160 per-tensor updates, and the all-reduce needs
    no pack/unpack copies;
  * nothing in the step synchronises with the host (the reference does ~20 .item()/.cpu() per
    step for logging): losses are returned as device tensors.

Loss-state policy under data parallelism: EvidenceLoss.weight_accum (50 floats) is averaged
across ranks after each step that updates it; the per-rank normalisers N / PN / AN stay
per-rank, i.e. the optimised objective is the mean over ranks of per-rank losses.
"""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from ..common import ops
from ..common.input_pipeline import PaddedTargets


def calc_bce_loss(start, end, scores):
    """train.py:152-161.  start/end (B,T,C) features, scores (B,2,T) boundary masks."""
    start = torch.tanh(start).mean(-1)
    end = torch.tanh(end).mean(-1)
    loss_start = F.binary_cross_entropy(start.view(-1), scores[:, 0].contiguous().view(-1), reduction='mean')
    loss_end = F.binary_cross_entropy(end.view(-1), scores[:, 1].contiguous().view(-1), reduction='mean')
    return loss_start, loss_end


def forward_one_epoch(net, criterion, clips, targets, scores=None, training=True, ssl=True):
    """train.py:164-201 with the criterion passed in (the reference uses a global CPD_Loss)."""
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
    loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act = criterion(output_dict, targets)
    src = getattr(output_dict, 'boundary_maps', None)
    if src is not None and src[0].is_cuda and src[0].dtype == torch.float32:
        # one launch per map (tanh, channel mean, BCE and the gradient, on the channel-major maps in place: csrc/bce.hip),
        # one for the six means and the two weighted sums below, one for the backward of all three maps
        from ..common.ops import BoundaryLossesFunction
        loss_start, loss_end = BoundaryLossesFunction.apply(scores, (1, 4, 4), (1.0, 0.1, 0.1), *src)    # 4: F.interpolate(1/4)
        return loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_start, loss_end, loss_act, loss_prop_act
    else:
        loss_start, loss_end = calc_bce_loss(output_dict['start'], output_dict['end'], scores)
        scores_ = scores[:, :, ::4]     # F.interpolate(scale_factor=1/4), nearest
        a, b = calc_bce_loss(output_dict['start_loc_prop'], output_dict['end_loc_prop'], scores_)
        c, d = calc_bce_loss(output_dict['start_conf_prop'], output_dict['end_conf_prop'], scores_)
    loss_start = loss_start + 0.1 * (a + c)
    loss_end = loss_end + 0.1 * (b + d)
    return loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_start, loss_end, loss_act, loss_prop_act


_COST_WEIGHTS = {}


def total_cost(losses, w):
    """The weighted sum of run_one_epoch (train.py:226-235) as one stack + one dot product (the reference's nine scalar
    multiplies and eight adds are ~20 one-element kernels, forward and backward)."""
    loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_start, loss_end, loss_act, loss_prop_act = losses
    parts = [loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_start, loss_end]
    weights = [w['lw'], w['cw'], w['lw'], w['cw'], w['ctw'], 1.0, 1.0]
    if loss_act is not None:
        parts += [loss_act, loss_prop_act]
        weights += [w['actw'], w['actw']]
    key = (tuple(weights), parts[0].device)
    wv = _COST_WEIGHTS.get(key)
    if wv is None:
        wv = _COST_WEIGHTS[key] = torch.tensor(weights, dtype=torch.float32, device=parts[0].device)
    return torch.dot(torch.stack([p.reshape(()) for p in parts]), wv)


def _detached(losses):
    """The loss terms a step hands back, without their autograd graph.  A caller that keeps the tuple (an epoch loop's
    `cost, losses = trainer.step(...)`) would otherwise keep the step's graph alive, and with it the parameters'
    AccumulateGrad nodes -- which remember the stream they were created on.  A later step captured on another stream then
    re-uses them, autograd makes THAT stream (the default stream of an eager step) wait for events recorded inside the
    capture, the default stream joins the capture and hipStreamEndCapture dies (found with AMD_LOG_LEVEL=3, round 5)."""
    if isinstance(losses, (list, tuple)):
        return tuple(None if l is None else l.detach() for l in losses)
    return None if losses is None else losses.detach()


def _flatten_inputs(clips, targets, scores):
    """The device tensors of a step's inputs, in a fixed order (targets: a list of ragged arrays, or the padded
    (rows, validity) pair of an input_pipeline.LabelRecord)."""
    out = [clips]
    if isinstance(targets, (list, tuple, PaddedTargets)):
        out += list(targets)
    elif targets is not None:
        out.append(targets)
    if scores is not None:
        out.append(scores)
    return out


def _bytes_of(t):
    """uint8 view of the whole storage `t` lives in."""
    return torch.empty(0, dtype=torch.uint8, device=t.device).set_(t.untyped_storage())


def _clone_inputs(clips, targets, scores):
    """Clones of a step's inputs for a captured step to replay from.  Tensors that are views of ONE storage (the fields of
    a LabelRecord) stay views of one cloned storage, so a replay refreshes them with a single copy."""
    leaves = _flatten_inputs(clips, targets, scores)
    by_storage = {}
    for t in leaves:
        by_storage.setdefault(t.untyped_storage().data_ptr(), []).append(t)
    fresh = {}
    for ptr, ts in by_storage.items():
        if len(ts) > 1:
            fresh[ptr] = _bytes_of(ts[0]).clone()

    def one(t):
        flat = fresh.get(t.untyped_storage().data_ptr())
        if flat is None:
            return t.clone()
        return torch.empty(0, dtype=t.dtype, device=t.device).set_(flat.untyped_storage(), t.storage_offset(), t.shape, t.stride())
    c = one(clips)
    if isinstance(targets, PaddedTargets):
        tg = PaddedTargets(one(targets.gt), one(targets.valid))
    elif isinstance(targets, (list, tuple)):
        tg = [one(u) for u in targets]
    else:
        tg = None if targets is None else one(targets)
    return c, tg, (None if scores is None else one(scores))


class FlatArena:
    """Trainable parameters re-homed into one contiguous fp32 buffer (+ parallel grad / m / v)."""

    def __init__(self, params, bucket_bytes, adjacent=(), split_key=None):
        """`adjacent`: groups of parameters that must sit back to back, in the given order (e.g. the three 1x1 weights of
        an Inception module that one fused launch reads as a single matrix, InceptionModule.fused_1x1_weights).
        `split_key`: function parameter -> hashable; a new bucket starts wherever the key changes along the arena (the
        trainer separates the backbone's buckets from the pyramid's: they complete at different times of backward)."""
        self.params = [p for p in params if p.requires_grad]
        # arena order = expected gradient-completion order: autograd finishes the heads / pyramid
        # first and the (single-node) backbone last, i.e. reverse registration order
        self.params = list(reversed(self.params))
        for group in adjacent:
            ids = {id(p) for p in group}
            if not all(any(q is p for q in self.params) for p in group):
                continue
            first = min(i for i, q in enumerate(self.params) if id(q) in ids)
            rest = [q for q in self.params if id(q) not in ids]
            first -= sum(1 for q in self.params[:first] if id(q) in ids)
            self.params = rest[:first] + list(group) + rest[first:]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets, self.grad_views = [], []
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            self.grad_views.append(self.grad[off:off + k].view(p.shape))
            p.grad = self.grad_views[-1]
            self.offsets.append(off)
            off += k
        self.numel = n
        # buckets: contiguous [lo,hi) ranges of about bucket_bytes
        self.buckets, self.bucket_of = [], []
        lo, cur = 0, 0
        key = split_key if split_key is not None else (lambda q: 0)
        for i, p in enumerate(self.params):
            self.bucket_of.append(len(self.buckets))
            cur += p.numel() * 4
            last = i == len(self.params) - 1
            if cur >= bucket_bytes or last or key(self.params[i + 1]) != key(p):
                hi = self.offsets[i] + p.numel()
                self.buckets.append((lo, hi))
                lo, cur = hi, 0
        self.bucket_size = [0] * len(self.buckets)
        self.bucket_members = [[] for _ in self.buckets]
        for i, b in enumerate(self.bucket_of):
            self.bucket_size[b] += 1
            self.bucket_members[b].append(i)


class DetectorTrainer:
    def __init__(self, net, criterion, loss_weights, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8,
                 process_group=None, bucket_mb=16, distributed=None, force_collectives=False, param_groups=None,
                 forward_fn=None):
        """`param_groups`: [(parameters, lr), ...] in the order the reference hands them to torch.optim.Adam (the
        ActivityNet recipe trains the backbone at lr/10, anet/train.py:304-312); default one group = net.parameters().
        `forward_fn`: the recipe's forward_one_epoch (default: this module's, the THUMOS14 one)."""
        self.net, self.criterion, self.w = net, criterion, dict(loss_weights)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self._base_lr = lr          # the groups' rates follow `self.lr` proportionally (a scheduler only touches self.lr)
        self.distributed = dist.is_available() and dist.is_initialized() if distributed is None else distributed
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.forward_fn = forward_one_epoch if forward_fn is None else forward_fn
        adjacent = [m.fused_1x1_weights() for m in net.modules() if hasattr(m, 'fused_1x1_weights')]
        # The backbone is ONE autograd node that finishes last; it announces its weight gradients layer by layer while it
        # runs (ops.GRAD_READY), so its buckets are kept apart from the pyramid's and go to RCCL from inside its backward.
        backbone = getattr(net, 'backbone', None)
        late = {id(p) for p in backbone.parameters()} if isinstance(backbone, nn.Module) else set()
        # ... and the weight whose gradient is computed LAST (the first trainable backbone parameter: Conv3d_1a) gets a
        # bucket of its own: the only all-reduce that cannot hide under later backward work is then a 260 KB one
        last = next((id(p) for p in (backbone.parameters() if late else ()) if p.requires_grad), None)
        # ... and the stem (the layers in front of MaxPool3d_4a: a third of the backward's GPU time, 3 % of the parameters)
        # is kept apart from the trunk: the two-phase captured step all-reduces everything else while the stem's backward
        # graph replays (capture_step(split=True))
        model = getattr(backbone, '_model', None)
        stem = {id(p) for p in model.stem_parameters()} if hasattr(model, 'stem_parameters') else set()
        self._stem_ids = stem
        self.arena = FlatArena(list(net.parameters()), bucket_mb << 20, adjacent,
                               split_key=lambda p: 3 if id(p) == last else (2 if id(p) in stem else int(id(p) in late)))
        a = self.arena
        self.stem_buckets = [b for b in range(len(a.buckets)) if id(a.params[a.bucket_members[b][0]]) in stem]
        self._capturing = False     # inside a stream capture: gradient copies are recorded, collectives are not issued
        # Collectives are ISSUED in one fixed order on every rank -- the pyramid / head buckets (complete first), then the
        # backbone's -- whatever order the gradients happen to arrive in: a rank whose batch leaves a parameter unused
        # would otherwise issue its all-reduces in a different order than its peers (mismatched collectives hang).
        is_late = [id(a.params[a.bucket_members[b][0]]) in late for b in range(len(a.buckets))]
        self.late_buckets = [b for b in range(len(a.buckets)) if is_late[b]]
        self._flush_order = sorted(range(len(a.buckets)), key=lambda b: (is_late[b], b))
        self._index_of = {p.data_ptr(): i for i, p in enumerate(a.params)}
        self.param_groups = [(list(net.parameters()), lr)] if param_groups is None else [(list(ps), g_lr) for ps, g_lr in param_groups]
        self._group_ranges = self._arena_ranges()
        self.step_count = 0
        self._pending, self._works = None, []
        lo = self.arena.flat.data_ptr()
        self._prologues = ops.PrologueCache((lo, lo + 4 * self.arena.numel))   # per-layer tables + bf16 weights, refreshed once per step
        self._graph = None          # (CUDAGraph, static inputs, static outputs) once capture_step() succeeded
        self._graph_key = None      # the host scalars baked into the capture
        self._graph_keepalive = None
        self._copy_plans = {}       # (static, given) addresses -> the copies a replay needs
        self.launch = 'eager'       # 'lanes': step() captures fixed-shape plain steps as lane graphs by itself (the drivers)
        self._eager_shapes = None   # input shapes of the last plain eager step
        self.replayed_steps = 0     # steps that ran as replayed lane graphs
        self._bias_corr = None      # 2-float device tensor: Adam bias corrections of the step being run
        self.collectives = self.distributed and (self.world > 1 or force_collectives)   # force: 1-rank RCCL smoke test
        self._flushed = None
        self._skipped = []          # arena indices of parameters that received no gradient in the current step
        self._used_ones = None      # data-parallel: per-parameter "used on this rank" flags, summed over the ranks
        self._used_work, self._used_global = None, None
        self._early = True          # the backbone may hand finished weight gradients over from inside its backward
        self.measure_exposed = False    # bench.py: record HIP events around the wait for the all-reduces
        self.bucket_trace = None        # bench.py: a list -> (kind, bucket, bytes, host clock, HIP event) per all-reduce issue / wait
        self.exposed_events = []
        self._ibm_work = None
        self._slots = ops.GradSlots(self.arena.flat, self.arena.grad, self.arena.offsets, [p.numel() for p in self.arena.params])
        for i, p in enumerate(self.arena.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _arena_ranges(self):
        """[(lo, hi, lr)]: each optimizer group must own ONE contiguous slice of the flat arena (module-aligned groups
        do: the arena is net.parameters() reversed), so that Adam stays one launch per group."""
        where = {id(p): (off, p.numel()) for p, off in zip(self.arena.params, self.arena.offsets)}
        out, covered = [], 0
        for ps, g_lr in self.param_groups:
            spans = sorted(where[id(p)] for p in ps if id(p) in where)
            if not spans:
                continue
            lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
            if sum(n for _, n in spans) != hi - lo:
                raise ValueError("an optimizer group is not contiguous in the flat arena")
            out.append((lo, hi, g_lr))
            covered += hi - lo
        if covered != self.arena.numel:
            raise ValueError("optimizer groups must cover every trainable parameter exactly once")
        return out

    # ---- gradients -> flat arena (+ all-reduce), bucket by bucket, overlapped with backward
    # Parameters enter backward with .grad = None, so autograd ADOPTS each incoming gradient tensor instead of
    # launching `arena_view += grad` per parameter (~190 tiny kernels per step + a 179 MB zero fill).  A gradient is
    # "done" either when autograd has accumulated it (post-accumulate hook) or -- backbone weights, whose gradients the
    # weight-gradient launches write straight into the arena -- when the backbone's backward announces it
    # (ops.GRAD_READY), long before that node returns.  When the last gradient of a bucket is done, one multi-tensor copy
    # moves the stragglers into the arena and, in data-parallel runs, the bucket's all-reduce is issued (in
    # self._flush_order).
    def _make_hook(self, i):
        def hook(_param):
            if self._pending is None or self._done[i]:
                return
            self._mark(i)
        return hook

    def _mark(self, i):
        self._done[i] = True
        b = self.arena.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._ready[b] = True
            self._drain()

    def _grads_ready(self, pairs):
        """Called from inside the backbone's backward (common/i3d_backbone.py) with (weight, gradient) pairs whose
        gradient is final.  Only gradients that already sit in their arena slice count (ops.grad_slot)."""
        if self._pending is None or not self._early:
            return
        a = self.arena
        for w, g in pairs:
            i = self._index_of.get(w.data_ptr())
            if i is None or g is None or self._done[i] or g.data_ptr() != a.grad_views[i].data_ptr():
                continue
            self._in_arena[i] = True
            self._mark(i)

    def _drain(self, force=False):
        order = self._flush_order
        while self._cursor < len(order):
            b = order[self._cursor]
            if not (self._ready[b] or force):
                break
            self._flush_bucket(b)
            self._cursor += 1

    def _flush_bucket(self, b):
        if self._flushed[b]:
            return
        self._flushed[b] = True
        a = self.arena
        # The bucket's weight gradients run on the side stream (ops.SideWgrads), some still waiting to be issued.  Nothing
        # on the main stream reads them before the optimizer step: a data-parallel run issues what is waiting and hands the
        # bucket to RCCL FROM the side stream (below), so the main stream's data-gradient chain never waits for them.
        if self.collectives:
            ops.side_issue()
        ops.flush_reduces(wait=False)                   # reductions recorded on the main stream (side stream off): one launch
        ops.flush_pending_sums()                        # deferred GroupNorm batch sums land in their arena slices: one launch
        dst, src = [], []
        for i in a.bucket_members[b]:
            if self._in_arena[i]:
                continue                                # written in place and announced by the backbone
            g = a.params[i].grad
            if g is None:
                a.grad_views[i].zero_()                 # parameter unused this step: no update (as torch.optim.Adam)
                self._skipped.append(i)
            elif g.data_ptr() != a.grad_views[i].data_ptr():
                dst.append(a.grad_views[i]); src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        if self.collectives and not self._capturing:
            self._issue_allreduce(b, side=ops.side_busy())
        elif self.collectives and ops.LANES is not None:
            # lane capture: the bucket's weight gradients became a side graph above, and the replay issues the all-reduce
            # right behind it, from the side stream
            ops.LANES.cut(("call", lambda b=b: self._issue_allreduce(b, side=True)))

    def _trace(self, kind, b=None, stream=None):
        """bench.py --gpus N: where on the lane timeline a bucket's all-reduce is issued and where the step waits for them --
        a HIP event on the issuing stream + the host clock, one record per call (`bucket_trace` is a list while tracing)."""
        if self.bucket_trace is None or self._capturing:
            return
        import time
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        nbytes = None if b is None else 4 * (self.arena.buckets[b][1] - self.arena.buckets[b][0])
        self.bucket_trace.append((kind, b, nbytes, time.perf_counter(), ev))

    def _issue_allreduce(self, b, side=False):
        lo, hi = self.arena.buckets[b]
        if side:        # behind everything both lanes have been given: RCCL's stream then waits for the side stream only
            sd = ops.side_wgrads(self.arena.grad.device)
            ops.L.check(ops.L.lib().otal_stream_wait(sd._raw, ops.L.stream()), "otal_stream_wait")
            with torch.cuda.stream(sd.side):
                self._trace("issue(side lane)", b)
                self._works.append(dist.all_reduce(self.arena.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        self._trace("issue(main lane)", b)
        self._works.append(dist.all_reduce(self.arena.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin_backward(self, early=True):
        """Call before cost.backward(): gradients start undefined, buckets open.  `early=False` (a step that runs the
        backbone twice, i.e. the ssl branch): gradients are only final when autograd has accumulated them."""
        a = self.arena
        for p in a.params:
            p.grad = None
        self._pending = list(a.bucket_size)
        self._flushed = [False] * len(a.buckets)
        self._ready = [False] * len(a.buckets)
        self._done = [False] * len(a.params)
        self._in_arena = [False] * len(a.params)
        self._skipped = []
        self._cursor = 0
        self._early = early
        self._slots.reset()
        ops.GRAD_SLOTS = self._slots                # weight gradients are written straight into the arena
        ops.GRAD_READY = self._grads_ready
        ops.SIDE_DEFER_JOIN = early                 # ... on a second stream that end_backward / the bucket flushes join
        # GroupNorm's batch sums are deferred to the bucket flushes -- unless a parameter may be used twice in this backward
        # (ssl step): a second gradient would be accumulated into the arena slice before the deferred sum is written there
        ops.PENDING_SUMS = [] if early else None
        ops.defer_reduces(early and ops.CONV_PROFILE is None)     # split-K reduces of weight gradients: batched (per-op timing: at once)
        if self.collectives and self._ibm_state() is not None:
            # the loss kernel updated the IBM EMA in the forward pass: its 50-float average travels under the backward
            if not self._capturing:
                self._issue_ibm()
            elif ops.LANES is not None:
                ops.LANES.cut(("call", self._issue_ibm))

    def _issue_ibm(self):
        self._ibm_work = dist.all_reduce(self._ibm_state(), op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _ibm_state(self):
        if getattr(self.criterion, 'cls_loss_type', None) == 'edl' and getattr(self.criterion.cls_loss, 'with_ibm', False):
            # (the ActivityNet EvidenceLoss uses the closed-form IBM weight: no cross-step state to average -> None)
            return getattr(self.criterion.cls_loss, 'weight_accum', None)
        return None

    def end_backward(self):
        """Call after cost.backward(): flush the buckets that did not complete (unused parameters), wait for the
        all-reduces, and leave every .grad aliasing its arena slice."""
        self._drain(force=True)
        self._adam_left = None
        late = ops.take_late_weights()
        if late and ops.LANES is not None and self._capturing and not self.collectives and not self._skipped:
            # lane capture, one process: Adam for everything but the stem tail's weights runs NOW on the main lane -- behind the
            # mark the backbone left on the side lane (ops.late_mark), beside the tail's weight gradients -- and only the
            # tail's parameters wait for the final join (_graph_body)
            span = self._late_range(late)
            if span is not None:
                ops.side_issue()                        # the tail's chunk (its fork sits in front of the optimizer launches)
                ops.flush_pending_sums()                # GroupNorm batch sums: main-lane launches into the arena
                ops.LANES.cut(("wait_mark",))
                self._adam_captured(0, span[0])
                self._adam_captured(span[1], self.arena.numel)
                self._adam_left = span
        ops.side_join()                                 # the weight gradients' stream: everything after this reads them
        ops.SIDE_DEFER_JOIN = False
        ops.defer_reduces(False)                        # flushes what is still recorded
        ops.flush_pending_sums()
        ops.PENDING_SUMS = None
        ops.GRAD_SLOTS = None
        ops.GRAD_READY = None
        self._pending = None
        for p, v in zip(self.arena.params, self.arena.grad_views):
            p.grad = v
        if self.collectives and self._capturing and ops.LANES is not None:
            ops.LANES.cut(("call", self._finish_allreduce))     # lane capture: the waits sit between the last two graphs
        self._finish_allreduce()

    def _issue_used_mask(self):
        """Data parallel: whether a parameter is left alone by Adam must be decided GLOBALLY.  A parameter unused on this
        rank has its slice zeroed and then all-reduced, i.e. it holds the peers' gradient; restoring it here while the
        peers apply the update would let the replicas diverge for good.  Every rank therefore contributes a 0/1 flag per
        parameter (one tiny all-reduce per step, issued by every kind of step -- eager, ssl, two-graph -- at the same
        place of the collective order), and a locally unused parameter is put back only where the sum is zero."""
        if not self.collectives or self._capturing:
            return
        a = self.arena
        if self._used_ones is None:
            self._used_ones = torch.ones(len(a.params), dtype=torch.float32, device=a.flat.device)
        used = self._used_ones.clone()
        if self._skipped:
            used[torch.tensor(self._skipped, dtype=torch.long, device=used.device)] = 0.0     # rare: the only H2D copy
        self._used_global = used
        self._used_work = dist.all_reduce(used, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _finish_allreduce(self):
        if not self.collectives or self._capturing:
            return
        self._issue_used_mask()
        self._trace("wait begin")
        ev = None
        if self.measure_exposed:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        for w in self._works:
            w.wait()
        self._works = []
        if self._used_work is not None:
            self._used_work.wait()
            self._used_work = None
        if self._ibm_work is not None:
            self._ibm_work.wait()
            self._ibm_work = None
            self._ibm_state().div_(self.world)
        self._trace("wait end")
        if ev is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            self.exposed_events.append((ev, end))

    # ---- one optimisation step
    def compute_cost(self, clips, targets, scores, ssl_clips=None, ssl_targets=None):
        losses = self.forward_fn(self.net, self.criterion, clips, targets, scores, training=True, ssl=False)
        cost = total_cost(losses, self.w)
        if ssl_clips is not None:
            trip = self.forward_fn(self.net, self.criterion, ssl_clips, ssl_targets, training=True, ssl=True)
            cost = cost + trip * self.w['ssl']
        return cost, losses

    def step(self, clips, targets, scores, ssl_clips=None, ssl_targets=None):
        self._trace("step begin")
        stale = False
        if self._graph is not None and ssl_clips is None:
            if self._graph_key == self._capture_key():
                return self._replay(clips, targets, scores)
            stale = True            # a baked-in host scalar changed (learning rate, IBM switch): eager now, capture again
            was_split, was_lanes = self._graph[0] == "split", self._graph[0] == "lanes"
            self._graph = None
        elif self.launch == 'lanes' and ssl_clips is None and isinstance(targets, PaddedTargets):
            # The drivers' launch mode (run_one_epoch): fixed-shape inputs (a LabelRecord's padded targets) replay as lane
            # graphs.  The first plain step of a shape runs eagerly -- a real step, after which every region, workspace and
            # launch plan exists -- the second one is captured (a capture executes nothing) and replayed.
            shapes = tuple(tuple(t.shape) for t in _flatten_inputs(clips, targets, scores))
            if self._eager_shapes == shapes:
                if self._try_capture_lanes(clips, targets, scores):
                    return self._replay(clips, targets, scores)
            else:
                self._eager_shapes = shapes
        # an eager step next to a captured graph (ssl branch) must not touch the descriptor buffers the graph replays
        # from: it works on its own prologue cache
        cache = self._prologues if self._graph is None else self._eager_prologues()
        ops.activate_prologues(cache)                  # ONE launch re-packs the weights of every known conv layer
        try:
            cost, losses = self.compute_cost(clips, targets, scores, ssl_clips, ssl_targets)
            self.begin_backward(early=ssl_clips is None)
            cost.backward()
            self.end_backward()
        finally:
            ops.deactivate_prologues()
            ops.GRAD_SLOTS = None
            ops.GRAD_READY = None
            ops.PENDING_SUMS = None
            ops.SIDE_DEFER_JOIN = False
            ops.defer_reduces(False)
        self.step_count += 1
        self.optimizer_update()
        if stale:
            self.capture_step(clips, targets, scores, warmup=0, split=was_split, lanes=was_lanes)
        return cost.detach(), _detached(losses)

    def _try_capture_lanes(self, clips, targets, scores):
        """The drivers' in-step capture.  A capture that cannot be had -- a parameter without a gradient (an unused head:
        torch.optim.Adam leaves it alone, a captured Adam cannot), a HIP capture error -- must not end the training run at
        its second step: everything the aborted capture touched is put back, the trainer stays on eager launches for good
        and says so once."""
        try:
            self.capture_step(clips, targets, scores, warmup=0, lanes=True)
            return True
        except RuntimeError as err:
            import warnings
            dev = clips.device
            ops.LANES = None
            self._capturing = False
            self._graph = None
            self._graph_key = None
            self._pending = None
            self._adam_left = None
            ops.take_late_weights()
            ops.deactivate_prologues()
            ops.GRAD_SLOTS = ops.GRAD_READY = ops.PENDING_SUMS = None
            ops.SIDE_DEFER_JOIN = False
            try:
                ops.defer_reduces(False)
            except RuntimeError:
                pass
            for sd in ops._SIDES.values():          # launches the aborted capture recorded for the side lane: never issued
                sd.pending.clear()
                sd.keep.clear()
            torch.cuda.synchronize(dev)
            self.launch = 'eager'
            self._eager_shapes = None
            warnings.warn(f"lane-graph capture of the training step failed ({err}); continuing with eager launches")
            return False

    def _eager_prologues(self):
        c = getattr(self, '_prologues_eager', None)
        if c is None:
            lo = self.arena.flat.data_ptr()
            c = self._prologues_eager = ops.PrologueCache((lo, lo + 4 * self.arena.numel))
        return c

    def _capture_key(self):
        """Every host scalar a captured step bakes into its launches."""
        cl = getattr(self.criterion, 'cls_loss', None)
        ibm = None
        if cl is not None and hasattr(cl, 'epoch'):
            ibm = (bool(getattr(cl, 'with_ibm', False)), int(cl.epoch) >= int(getattr(cl, 'ibm_start', 0)),
                   float(getattr(cl, 'annealing_coef', 0.0)) if hasattr(cl, 'annealing_coef') else None)
        return (float(self.lr), float(self._base_lr), tuple(float(g) for _, _, g in self._group_ranges), float(self.wd),
                tuple(self.betas), float(self.eps), self.world, bool(self.collectives), ibm,
                tuple(sorted((k, float(v)) for k, v in self.w.items())))

    def optimizer_update(self):
        """Adam (L2 weight decay in the gradient, train.py:321-323) on the flat arena: one launch."""
        a = self.arena
        keep = self._stash_skipped()
        for lo, hi, g_lr in self._group_ranges:
            g_lr = g_lr * (self.lr / self._base_lr) if self._base_lr else g_lr
            ops.adam_flat(a.flat[lo:hi], a.grad[lo:hi], a.m[lo:hi], a.v[lo:hi], self.step_count, g_lr,
                          self.betas[0], self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world)
        self._restore_skipped(keep)

    # torch.optim.Adam leaves a parameter whose .grad is None alone (no weight decay, moments untouched); the flat launch
    # covers the whole arena, so the (rare) parameters without a gradient are put back afterwards.
    def _stash_skipped(self):
        a = self.arena
        keep = []
        for i in self._skipped:
            sl = slice(a.offsets[i], a.offsets[i] + a.params[i].numel())
            keep.append((sl, a.flat[sl].clone(), a.m[sl].clone(), a.v[sl].clone()))
        return keep

    def _restore_skipped(self, keep):
        a = self.arena
        g = self._used_global if self.collectives else None
        for (sl, p, m, v), i in zip(keep, self._skipped):
            if g is None:
                a.flat[sl].copy_(p); a.m[sl].copy_(m); a.v[sl].copy_(v)
            else:       # device-side decision (no host sync): keep the update where any peer used the parameter
                unused = g[i] == 0
                a.flat[sl].copy_(torch.where(unused, p, a.flat[sl]))
                a.m[sl].copy_(torch.where(unused, m, a.m[sl]))
                a.v[sl].copy_(torch.where(unused, v, a.v[sl]))

    # ---- the same step as ONE HIP graph: ~1500 launches per step are replayed without host involvement
    def _graph_body(self, clips, targets, scores):
        ops.activate_prologues(self._prologues)
        try:
            cost, losses = self.compute_cost(clips, targets, scores)
            self.begin_backward()
            cost.backward()
            self.end_backward()
        finally:
            ops.deactivate_prologues()
            ops.GRAD_SLOTS = None
            ops.GRAD_READY = None
            ops.PENDING_SUMS = None
            ops.SIDE_DEFER_JOIN = False
            ops.defer_reduces(False)
        keep = self._stash_skipped()
        span = getattr(self, "_adam_left", None) or (0, self.arena.numel)       # (what end_backward's early launches left)
        self._adam_captured(span[0], span[1])
        self._adam_left = None
        self._restore_skipped(keep)
        return cost.detach(), _detached(losses)

    def _adam_captured(self, start, stop):
        """Adam over the arena range [start, stop) (graph-replayable: bias corrections in device memory), group by group."""
        a = self.arena
        for lo, hi, g_lr in self._group_ranges:
            lo, hi = max(lo, start), min(hi, stop)
            if lo >= hi:
                continue
            g_lr = g_lr * (self.lr / self._base_lr) if self._base_lr else g_lr
            ops.adam_flat_dev(a.flat[lo:hi], a.grad[lo:hi], a.m[lo:hi], a.v[lo:hi], self._bias_corr, g_lr,
                              self.betas[0], self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world)

    def _late_range(self, weights):
        """[lo, hi) of the arena when the stem tail's weights are neighbours there (the backbone's parameters sit in reverse
        registration order: its first layers end its block), else None."""
        a = self.arena
        idx = sorted(self._index_of.get(w.data_ptr(), -1) for w in weights)
        if not idx or idx[0] < 0 or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        return a.offsets[idx[0]], a.offsets[idx[-1]] + a.params[idx[-1]].numel()

    def capture_step(self, clips, targets, scores, warmup=2, split=False, lanes=False):
        """Capture forward + losses + backward (+ gradient all-reduce) + Adam for inputs of these shapes.

        The captured launches bake in every host-side scalar of the step; the only one that changes per step --
        Adam's bias correction -- lives in device memory and is refreshed before each replay.  Quantities that
        change per EPOCH (the evidential loss's annealing coefficient) need a re-capture at the epoch boundary.
        `warmup` eager steps run first on a side stream (they are real optimisation steps)."""
        dev = clips.device
        self._graph = None
        if lanes:
            return self._capture_lanes(clips, targets, scores, warmup)
        if split:
            return self._capture_split(clips, targets, scores, warmup)
        if self.collectives:
            # ProcessGroupNCCL's watchdog thread queries the events of the collectives it is handed; for an event recorded
            # in a capturing stream that query fails (hipErrorCapturedEvent) and the watchdog terminates the process
            raise RuntimeError("capture_step: a step with RCCL collectives cannot be captured as ONE graph (watchdog event "
                               "query inside stream capture); use split=True (two graphs, collectives between them) or "
                               "eager launches")
        self._bias_corr = torch.zeros(2, dtype=torch.float32, device=dev)
        static = _clone_inputs(clips, targets, scores)
        # (the warm-up steps run on a side stream on purpose: the AccumulateGrad stream-mismatch warning is noise here)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = ops.capture_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._set_bias(self.step_count + 1)
                self._graph_body(*static)
                self.step_count += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        ops.activate_prologues(self._prologues)     # upload the descriptors of regions created by the last warm-up step
        ops.deactivate_prologues()                  # (a host->device copy cannot be captured)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        self._set_bias(self.step_count + 1)
        # thread_local: RCCL's watchdog thread polls events concurrently; only this thread's calls must be capture-safe
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = self._graph_body(*static)
        self.step_count += 0        # capture does not execute
        self._graph = (graph, static, out)
        self._graph_key = self._capture_key()
        # the captured prologue launch reads these descriptor buffers by raw pointer: they live as long as the graph
        self._graph_keepalive = (self._prologues.dev_descs, self._prologues.dev_starts)
        return self

    # ---- the step as a SEQUENCE of HIP graphs on two lanes (ops.LanePlan): the main lane (forward, losses, the data-gradient
    # chain, Adam) and the weight-gradient lane.  A replayed hipGraph runs its branches one after the other, so the one-graph
    # step loses the overlap of the two families that eager launches get from two streams; eager launches in turn leave the
    # GPU waiting for the host in the pyramid / head region (~240 launches of 3-30 us).  Here the host issues ~40 graph
    # launches per step, the weight gradients of chunk k run beside the main lane's chunk k+1, and a data-parallel run
    # issues each bucket's all-reduce between two graphs, behind the side graph that completes the bucket.
    def _capture_lanes(self, clips, targets, scores, warmup):
        import gc
        dev = clips.device
        self._bias_corr = torch.zeros(2, dtype=torch.float32, device=dev)
        static = _clone_inputs(clips, targets, scores)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        cap = ops.capture_stream(dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            for _ in range(warmup):                 # eager steps (real ones): regions, workspaces and plans exist afterwards
                self.step(*static)
        torch.cuda.current_stream(dev).wait_stream(cap)
        ops.activate_prologues(self._prologues)     # upload the descriptors of regions created by the warm-up
        ops.deactivate_prologues()
        torch.cuda.synchronize(dev)
        gc.collect()
        torch.cuda.empty_cache()
        plan = ops.LanePlan(ops.side_wgrads(dev).side)
        self._set_bias(self.step_count + 1)
        self._capturing = True
        ops.LANES = plan
        from ..common import anet_dataset as _ad
        _ad.CAPTURE_IN_PROGRESS[0] = True           # background video readers must not call hipHostMalloc meanwhile
        try:
            # the capture of a main graph ends and the next begins INSIDE the backward pass: autograd has to run it on this
            # thread (a stream capture is ended by the thread that began it)
            with torch.cuda.stream(cap), torch.autograd.set_multithreading_enabled(False):
                plan.begin_main()
                try:
                    out = self._graph_body(*static)
                finally:
                    plan.end_main()
        finally:
            ops.LANES = None
            self._capturing = False
            self._pending = None
            _ad.CAPTURE_IN_PROGRESS[0] = False
        if self._skipped:
            raise RuntimeError("capture_step(lanes=True): a parameter received no gradient; use eager launches")
        self._graph = ("lanes", plan, static, out)
        self._graph_key = self._capture_key()
        self._graph_keepalive = (self._prologues.dev_descs, self._prologues.dev_starts)
        return self

    def _replay_lanes(self, clips, targets, scores):
        _, plan, static, out = self._graph
        self._copy_inputs(static, (clips, targets, scores))
        self._skipped = []
        self.step_count += 1
        self.replayed_steps += 1
        self._set_bias(self.step_count)
        plan.replay()
        return out

    # ---- the data-parallel step as TWO HIP graphs with the collectives between them
    # graph 1: forward, losses, backward down to the cut behind MaxPool3d_4a (97 % of the parameters' gradients final);
    # eager : their bucket all-reduces go to RCCL's stream (fixed order);
    # graph 2: the stem's backward (Conv3d_1a .. Mixed_3c, ~4 ms of GPU time at b = 8) replays WHILE those all-reduces run;
    # eager : the stem's two small buckets, the wait, Adam (one launch).
    # Nothing of RCCL is captured (its watchdog cannot cope with events recorded during capture), the host issues two graph
    # launches and ~10 collectives per step instead of ~480 kernel launches, and the only exposed communication is the
    # stem's 5.6 MB.  Needs the backbone's two-node form (InceptionI3d.split_backward).
    def _capture_split(self, clips, targets, scores, warmup):
        model = getattr(getattr(self.net, 'backbone', None), '_model', None)
        if model is None or not self._stem_ids or not self.stem_buckets:
            raise RuntimeError("capture_step(split=True): the model has no stem / trunk cut")
        dev = clips.device
        a = self.arena
        # a stale re-capture (step() just ran this batch eagerly in the two-node form) needs no further warm-up step:
        # forcing one would apply the same batch twice and advance Adam's step count past the reference schedule
        already_split = bool(getattr(model, 'split_backward', False))
        model.split_backward = True
        self._bias_corr = torch.zeros(2, dtype=torch.float32, device=dev)
        static = _clone_inputs(clips, targets, scores)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = ops.capture_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 0 if already_split else 1)):     # eager data-parallel steps in the two-node form (real steps)
                self.step(*static)
        torch.cuda.current_stream(dev).wait_stream(side)
        ops.activate_prologues(self._prologues)     # upload the descriptors of regions created by the warm-up
        ops.deactivate_prologues()
        torch.cuda.synchronize(dev)
        stem_ids = self._stem_ids
        phase1 = [p for p in a.params if id(p) not in stem_ids]
        phase2 = [p for p in a.params if id(p) in stem_ids]
        n1 = len(self._flush_order) - len(self.stem_buckets)        # the stem's buckets are issued last
        if sorted(self._flush_order[n1:]) != sorted(self.stem_buckets):
            raise RuntimeError("capture_step(split=True): the stem's buckets are not the last in the issue order")
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self._set_bias(self.step_count + 1)
        self._capturing = True
        model.detach_cut = True                     # the autograd graph of the captured pass is cut into its two phases
        try:
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                ops.activate_prologues(self._prologues)
                cost, losses = self.compute_cost(*static)
                self.begin_backward()
                stem_out, cut_leaf = model.stem_out, model.cut_leaf
                cut_leaf.grad = None
                torch.autograd.backward(cost, inputs=[cut_leaf] + phase1)
                while self._cursor < n1:
                    self._flush_bucket(self._flush_order[self._cursor])      # stragglers of phase 1 -> arena (copies only)
                    self._cursor += 1
                gcut = cut_leaf.grad
                ops.side_join()                     # a capture ends with every forked stream joined
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                torch.autograd.backward(stem_out, grad_tensors=gcut, inputs=phase2)
                self.end_backward()
        finally:
            self._capturing = False
            model.detach_cut = False
            model.cut_leaf = None
            ops.deactivate_prologues()
            ops.GRAD_SLOTS = None
            ops.GRAD_READY = None
            ops.PENDING_SUMS = None
            ops.SIDE_DEFER_JOIN = False
            ops.defer_reduces(False)
            self._pending = None
        if self._skipped:
            raise RuntimeError("capture_step(split=True): a parameter received no gradient; use eager launches")
        out = (cost.detach(), _detached(losses))
        self._graph = ("split", g1, g2, static, out)
        self._graph_key = self._capture_key()
        self._graph_keepalive = (self._prologues.dev_descs, self._prologues.dev_starts, stem_out, gcut)
        return self

    def _replay_split(self, clips, targets, scores):
        _, g1, g2, static, out = self._graph
        self._copy_inputs(static, (clips, targets, scores))
        self._skipped = []                          # (a captured step has none: _capture_split refuses otherwise)
        self.step_count += 1
        self._set_bias(self.step_count)
        a = self.arena
        g1.replay()
        if self.collectives:
            if self._ibm_state() is not None:
                self._ibm_work = dist.all_reduce(self._ibm_state(), op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for b in self._flush_order:
                if b not in self.stem_buckets:
                    self._issue_allreduce(b)
        g2.replay()
        if self.collectives:
            for b in self._flush_order:
                if b in self.stem_buckets:
                    self._issue_allreduce(b)
        self._finish_allreduce()
        for lo, hi, g_lr in self._group_ranges:
            g_lr = g_lr * (self.lr / self._base_lr) if self._base_lr else g_lr
            ops.adam_flat_dev(a.flat[lo:hi], a.grad[lo:hi], a.m[lo:hi], a.v[lo:hi], self._bias_corr, g_lr,
                              self.betas[0], self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world)
        return out

    def _copy_inputs(self, static, given):
        """Refresh the captured step's input buffers from `given`.  What already lives there is left alone; tensors that are
        views of one storage on both sides with the same layout (a LabelRecord) travel as ONE copy of that storage."""
        dsts, srcs = _flatten_inputs(*static), _flatten_inputs(*given)
        if len(dsts) != len(srcs):
            raise RuntimeError("captured step replayed with a different batch; call capture_step again")
        key = tuple((d.data_ptr(), g.data_ptr()) for d, g in zip(dsts, srcs))
        plan = self._copy_plans.get(key)
        if plan is None:
            todo = []
            for dst, src in zip(dsts, srcs):
                if dst.data_ptr() == src.data_ptr():
                    continue
                if dst.shape != src.shape or dst.dtype != src.dtype:
                    # ragged targets: the per-sample row counts are baked into the capture (padded records never get here)
                    raise RuntimeError("captured step replayed with different input shapes; call capture_step again")
                todo.append((dst, src))
            groups = {}
            for dst, src in todo:
                groups.setdefault((dst.untyped_storage().data_ptr(), src.untyped_storage().data_ptr()), []).append((dst, src))
            plan = []
            for pairs in groups.values():
                d0, s0 = pairs[0]
                whole = (len(pairs) > 1 and d0.untyped_storage().nbytes() == s0.untyped_storage().nbytes()
                         and all(d.storage_offset() == g.storage_offset() and d.stride() == g.stride() for d, g in pairs))
                plan += [(_bytes_of(d0), _bytes_of(s0))] if whole else pairs
            # (a cached plan keeps its source tensors alive, which is what makes the address key safe -- so only plans over
            # small, persistent sources are kept: label records; a freshly allocated 226 MB clip batch is not)
            if all(src.numel() * src.element_size() <= (1 << 20) for _, src in plan):
                if len(self._copy_plans) > 64:
                    self._copy_plans.clear()
                self._copy_plans[key] = plan
        for dst, src in plan:
            dst.copy_(src, non_blocking=True)

    def static_inputs(self):
        """The (clips, targets, scores) buffers a captured step replays from, or None.  A producer that writes its batch
        straight into them (the clip kernel's destination, `common.input_pipeline`) and passes THEM to step() skips the
        device-to-device copy of the batch (226 MB at b = 8: ~0.1 ms per step): _copy_inputs only copies what lives elsewhere."""
        g = self._graph
        if g is None:
            return None
        return g[3] if g[0] == "split" else (g[2] if g[0] == "lanes" else g[1])

    def drop_graph(self):
        """Back to eager launches: forget the captured step (and the two-node form of the backbone it needed)."""
        self._graph = None
        self._graph_keepalive = None
        model = getattr(getattr(self.net, 'backbone', None), '_model', None)
        if model is not None and hasattr(model, 'split_backward'):
            model.split_backward = False
            model.stem_out = None

    def _set_bias(self, step):
        bc = ops.adam_bias_corrections(step, self.betas[0], self.betas[1])
        self._bias_corr.copy_(torch.tensor(bc, dtype=torch.float32), non_blocking=False)

    def _replay(self, clips, targets, scores):
        if self._graph[0] == "split":
            return self._replay_split(clips, targets, scores)
        if self._graph[0] == "lanes":
            return self._replay_lanes(clips, targets, scores)
        graph, static, out = self._graph
        self._copy_inputs(static, (clips, targets, scores))
        self.step_count += 1
        self._set_bias(self.step_count)
        graph.replay()
        return out

    # ---- checkpoints in the reference's file formats (train.py:106-131), SURVEY 8f rank 2
    def optimizer_state_dict(self):
        """The Adam state as `torch.optim.Adam(net.parameters(), ...).state_dict()` would hold it (the reference's
        optimizer, train.py:321-323): per-parameter `step` / `exp_avg` / `exp_avg_sq`, indices in net.parameters() order
        (frozen parameters are listed in the group but carry no state)."""
        allp = [p for ps, _ in self.param_groups for p in ps]
        index = {id(p): i for i, p in enumerate(allp)}
        state = {}
        if self.step_count > 0:
            for p, off in zip(self.arena.params, self.arena.offsets):
                k = p.numel()
                state[index[id(p)]] = {'step': torch.tensor(float(self.step_count)),
                                       'exp_avg': self.arena.m[off:off + k].view(p.shape).clone(),
                                       'exp_avg_sq': self.arena.v[off:off + k].view(p.shape).clone()}
        groups, first = [], 0
        rate = (self.lr / self._base_lr) if self._base_lr else 1.0      # a scheduler moves self.lr; the groups follow
        for ps, g_lr in self.param_groups:
            groups.append({'lr': g_lr * rate, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.wd,
                           'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False,
                           'differentiable': False, 'fused': None, 'params': list(range(first, first + len(ps)))})
            first += len(ps)
        return {'state': state, 'param_groups': groups}

    def load_optimizer_state_dict(self, sd):
        allp = [p for ps, _ in self.param_groups for p in ps]
        where = {id(p): (off, p) for p, off in zip(self.arena.params, self.arena.offsets)}
        steps = set()
        self.arena.m.zero_(); self.arena.v.zero_()
        for i, st in sd['state'].items():
            hit = where.get(id(allp[int(i)]))
            if hit is None:
                continue
            off, p = hit
            k = p.numel()
            self.arena.m[off:off + k].copy_(st['exp_avg'].reshape(-1))
            self.arena.v[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise RuntimeError("per-parameter step counts differ; the flat Adam keeps one")
        self.step_count = steps.pop() if steps else 0
        if len(sd['param_groups']) != len(self.param_groups):
            raise RuntimeError("optimizer state has a different number of parameter groups")
        g = sd['param_groups'][-1]          # the detection-head group carries the base learning rate
        self.lr, self.betas, self.eps, self.wd = g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']
        self.param_groups = [(ps, sg['lr']) for (ps, _), sg in zip(self.param_groups, sd['param_groups'])]
        self._base_lr = self.lr
        self._group_ranges = self._arena_ranges()
        self._graph = None

    def save_model(self, epoch, checkpoint_path, train_state_path):
        """`save_model` of the reference (train.py:106-118): weights as checkpoint-{epoch}.ckpt (the same 446 keys),
        optimizer + RNG states as checkpoint_{epoch}.ckpt, plus the '-latest' links.  Extra key 'weight_accum': the IBM
        EMA state, which the reference forgets to checkpoint (it is a registered buffer here, so it is in the weights
        file as well when present)."""
        import os, random
        import numpy as np
        os.makedirs(checkpoint_path, exist_ok=True); os.makedirs(train_state_path, exist_ok=True)
        model_file = os.path.join(checkpoint_path, 'checkpoint-{}.ckpt'.format(epoch))
        torch.save({k: v.detach().clone() for k, v in self.net.state_dict().items()}, model_file)   # views of the arena: clone
        states = [random.getstate(), np.random.get_state(), torch.get_rng_state()]
        if torch.cuda.is_available():
            states.append(torch.cuda.get_rng_state())
        state_file = os.path.join(train_state_path, 'checkpoint_{}.ckpt'.format(epoch))
        extra = {}
        wa = getattr(getattr(self.criterion, 'cls_loss', None), 'weight_accum', None)
        if wa is not None:
            extra['weight_accum'] = wa.detach().clone()
        torch.save(dict({'optimizer': self.optimizer_state_dict(), 'state': states}, **extra), state_file)
        for src, dst in ((model_file, os.path.join(checkpoint_path, 'checkpoint-latest.ckpt')),
                         (state_file, os.path.join(train_state_path, 'checkpoint_latest.ckpt'))):
            if os.path.lexists(dst):
                os.remove(dst)
            os.symlink(os.path.abspath(src), dst)
        return model_file, state_file

    def resume_training(self, resume, checkpoint_path, train_state_path, restore_rng=True):
        """`resume_training` of the reference (train.py:121-131); accepts the reference's own files."""
        import os, random
        import numpy as np
        start_epoch = 1
        if resume > 0:
            start_epoch += resume
            sd = torch.load(os.path.join(checkpoint_path, 'checkpoint-{}.ckpt'.format(resume)), map_location='cpu')
            own = self.net.state_dict()
            missing = [k for k in own if k not in sd]
            if missing:
                raise RuntimeError(f"checkpoint lacks {len(missing)} keys, e.g. {missing[:3]}")
            with torch.no_grad():
                for k, v in own.items():
                    v.copy_(sd[k])                      # in place: parameters stay views of the flat arena
            st = torch.load(os.path.join(train_state_path, 'checkpoint_{}.ckpt'.format(resume)), map_location='cpu',
                            weights_only=False)
            self.load_optimizer_state_dict(st['optimizer'])
            if 'weight_accum' in st and hasattr(getattr(self.criterion, 'cls_loss', None), 'weight_accum'):
                self.criterion.cls_loss.weight_accum.copy_(st['weight_accum'])
            if restore_rng:
                states = st['state']
                random.setstate(states[0]); np.random.set_state(states[1]); torch.set_rng_state(states[2])
                if torch.cuda.is_available() and len(states) > 3:
                    torch.cuda.set_rng_state(states[3])
        return start_epoch

    def grad_norm(self):
        """get_grad_norm (train.py:133-140), as a device tensor."""
        return self.arena.grad.norm(2)


# ----------------------------------------------------------------------------- the training driver (train.py:204-363)
def set_seed(seed):
    """train.py:59-66 (the cudnn switches have no counterpart: every kernel here is deterministic by construction)."""
    import random
    import numpy as np
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def loss_dispatch(config, as_shipped=False):
    """cls_loss_type of the run.  The reference derives it TWICE (train.py:27 then :31): 'edl' if edl_loss, and then
    overwrites it with 'rpl' if rpl_loss else 'focal' -- so its shipped THUMOS14 recipe trains FocalLoss_Ori (SURVEY
    H2).  Default here: what the config says (edl_loss -> 'edl', the OpenTAL method); `as_shipped=True` reproduces
    the overwrite."""
    tr = config['training']
    if tr.get('rpl_loss', False):
        raise NotImplementedError("RPL baseline is outside the OpenTAL hot path")
    if as_shipped:
        return 'focal'
    return 'edl' if tr.get('edl_loss', False) else 'focal'


def build_training(config, device, as_shipped=False, random_init=False, dist_group=None):
    """Model, criterion and trainer from a parsed config (train.py:306-331)."""
    from .BDNet import BDNet, model_cfg_from
    from .multisegment_loss import MultiSegmentLoss
    tr, md = config['training'], config['model']
    use_edl = md.get('use_edl', False)
    net = BDNet(in_channels=md['in_channels'], backbone_model=md.get('backbone_model'), training=not random_init,
                use_edl=use_edl, use_rpl=md.get('use_rpl', False), cfg=model_cfg_from(config))
    if random_init:                 # no pretrained I3D file (synthetic runs): the architecture's glorot initialisation
        net.backbone._model.apply(BDNet.weight_init)
    net = net.to(device).train()
    os_head = md.get('os_head', False)
    num_cls = config['dataset']['num_classes'] - 1 if os_head else config['dataset']['num_classes']
    crit = MultiSegmentLoss(num_cls, tr['piou'], 1.0, cls_loss_type=loss_dispatch(config, as_shipped),
                            edl_config=tr.get('edl_config'), os_head=os_head, act_config=tr.get('act_config'),
                            clip_length=config['dataset']['training']['clip_length']).to(device)
    weights = dict(lw=tr['lw'], cw=tr['cw'], ctw=tr['ctw'], actw=tr.get('actw', 1.0), ssl=tr['ssl'])
    trainer = DetectorTrainer(net, crit, weights, lr=tr['learning_rate'], weight_decay=tr['weight_decay'],
                              process_group=dist_group)
    return net, crit, trainer


def rank_batches(every, rank, world):
    """The batches of one epoch that rank `rank` of `world` runs: every world-th one of a list truncated to a multiple of
    `world` -- drop_last across ranks.  Every rank must run the SAME number of steps (each step's bucket / IBM / used-mask
    all-reduces need all peers; a rank with one batch more would wait for collectives nobody else issues until the RCCL
    watchdog aborts the job at the end of the first epoch).  The truncation comes BEFORE a caller's max_steps cut, and every
    rank materialises the whole list (the sampling decisions draw from one shared random stream, so all ranks must draw all of
    them to stay aligned; decisions are host-only records -- no pixels, see anet_dataset.LazyVideo).  The per-epoch means a
    driver prints are averaged over the ranks (one 8-float all-reduce per epoch, run_one_epoch)."""
    every = list(every)
    every = every[:len(every) // world * world]
    return every[rank::world]


class StepLog:
    """Per-step scalars without a host synchronisation (the reference calls .item() ~20 times per step for TensorBoard and
    the progress bar, train.py:255-283): every `every` steps the step's [cost, 7 losses] vector is copied
    to a pinned host slot with ONE asynchronous D2H copy + an event; a slot is read (and handed to `sink`) only once its
    event has completed, i.e. a few steps later, while the GPU keeps running."""
    NAMES = ('total', 'loc', 'conf', 'prop_loc', 'prop_conf', 'IoU', 'start', 'end')

    def __init__(self, every=20, sink=None, slots=4):
        self.every, self.sink = int(every), sink
        self._free = [(torch.empty(8, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.empty(8),
                       torch.cuda.Event() if torch.cuda.is_available() else None) for _ in range(slots)]
        self._inflight = []
        self.records = []

    def push(self, step, vec):
        self.poll()
        if self.every <= 0 or step % self.every or not self._free:
            return
        buf, ev = self._free.pop()
        buf.copy_(vec.detach(), non_blocking=True)
        if ev is not None:
            ev.record()
        self._inflight.append((step, buf, ev))

    def poll(self, wait=False):
        while self._inflight and (wait or self._inflight[0][2] is None or self._inflight[0][2].query()):
            step, buf, ev = self._inflight.pop(0)
            if wait and ev is not None:
                ev.synchronize()
            rec = (step, buf.tolist())
            self.records.append(rec)
            if self.sink is not None:
                self.sink(*rec)
            self._free.append((buf, ev))


class JsonlSink:
    """A StepLog sink that appends one JSON object per logged step to <dir>/scalars.jsonl -- the scalars the reference hands
    to tensorboardX (Total / loc / conf / prop_loc / prop_conf / IoU / start / end, AFSD/thumos14/train.py:249-268; the
    package is not in this image, and a line-per-step file is what `pandas.read_json(..., lines=True)` or a TensorBoard
    importer reads).  Call it as sink(step, values); the file is flushed per record (records arrive every N steps)."""
    TAGS = ('Total', 'loc', 'conf', 'prop_loc', 'prop_conf', 'IoU', 'start', 'end')

    def __init__(self, directory, filename='scalars.jsonl', echo=None):
        import os
        os.makedirs(directory, exist_ok=True)
        self.path = os.path.join(directory, filename)
        self._f = open(self.path, 'a')
        self.echo = echo

    def __call__(self, step, values):
        import json
        rec = {'step': int(step)}
        rec.update({'Train/' + t: float(v) for t, v in zip(self.TAGS, values)})
        self._f.write(json.dumps(rec) + '\n')
        self._f.flush()
        if self.echo is not None:
            self.echo(step, values)

    def close(self):
        self._f.close()


def run_one_epoch(epoch, trainer, dataset, stager, batch_size, rank=0, world=1, max_steps=None, log=print, step_log=None):
    """One pass over the shuffled sliding-window list (train.py:204-303).  Every rank draws the same permutation and
    the same per-sample decisions (shared seeds) and takes every world-th batch.  Nothing in the loop synchronises with
    the host: the loss sums stay on the device until the epoch's log line; `step_log` (StepLog) receives the per-step
    scalars through asynchronous copies.

    Labels: with a stager built with `max_targets` the batch's targets (any number per sample, train.py:221-224), boundary
    masks and ssl segments cross PCIe as ONE fixed-shape pinned record next to the frames (input_pipeline.LabelRecord), so
    every plain step has the same input shapes and -- `trainer.launch == 'lanes'` -- replays the captured lane graphs, the
    clip kernel writing straight into the captured step's clip buffer; ssl steps (`flags[0]`) run eagerly beside it."""
    from ..common import thumos_dataset as D
    dev = trainer.arena.flat.device
    sums, n_iter = None, 0
    mine = rank_batches(D.batches(dataset, batch_size), rank, world)
    if max_steps is not None:
        mine = mine[:max_steps]
    if not mine:
        return None
    stager.submit(mine[0])
    for k, samples in enumerate(mine):
        use_ssl = bool(samples[0]['flag'])                 # `if flags[0]` (train.py:237)
        static = trainer.static_inputs()
        clips, ssl_clips = stager.collect(want_ssl=use_ssl, out=None if static is None else static[0])
        rec = stager.labels()
        if k + 1 < len(mine):
            stager.submit(mine[k + 1])                     # next batch crosses PCIe while this step runs
        if k + 2 < len(mine) and hasattr(dataset, 'prefetch'):      # lazily loaded videos (ActivityNet): the batch after that is read
            dataset.prefetch([s_['video'].name for s_ in mine[k + 2] if hasattr(s_['video'], 'name')])    # from disk in the background
        if rec is not None:
            targets, scores, ssl_targets = rec.targets, rec.scores, rec.ssl_targets() if use_ssl else None
        else:                                              # a stager without label records: ragged lists, eager steps
            targets = [torch.from_numpy(s['target']).to(dev, non_blocking=True) for s in samples]
            scores = torch.from_numpy(np.stack([s['scores'] for s in samples], 0)).to(dev, non_blocking=True)
            ssl_targets = [torch.from_numpy(s['ssl_target'][:, :2].copy()).to(dev) for s in samples] if use_ssl else None
        cost, losses = trainer.step(clips, targets, scores, ssl_clips if use_ssl else None, ssl_targets)
        stager.release()                                   # the slot's labels have been read by everything issued so far
        vec = torch.stack([cost.reshape(())] + [l.detach().reshape(()) for l in losses[:7]])
        sums = vec if sums is None else sums + vec
        n_iter += 1
        if step_log is not None:
            step_log.push(trainer.step_count, vec)
    if step_log is not None:
        step_log.poll(wait=True)
    mean = sums / n_iter
    if world > 1 and dist.is_available() and dist.is_initialized():
        dist.all_reduce(mean, op=dist.ReduceOp.SUM, group=trainer.group)   # every rank ran n_iter steps (rank_batches):
        mean = mean / world                                                # the line below is the epoch's GLOBAL mean
    v = mean.tolist()                                      # the epoch's only host synchronisation
    log('Epoch-{} Train Loss: Total - {:.5f}, loc - {:.5f}, conf - {:.5f}, prop_loc - {:.5f}, prop_conf - {:.5f}, '
        'IoU - {:.5f}, start - {:.5f}, end - {:.5f}'.format(epoch, *v))
    return v


def main(argv=None):
    """python -m opental_amd.thumos14.train <yaml> --lw 1 --cw 10 --piou 0.5 --ssl 0.001 --open_set --split 0 [--resume N]

    The reference's command line (experiments/opental/train_opental_final.sh; AFSD/common/config.py:10-37) plus:
      --as_shipped_dispatch   train FocalLoss_Ori as the reference's train.py:27-31 really does
      --random_init           no pretrained I3D file (synthetic data)
      --save_after N          save checkpoints for epochs > N (reference: 10, train.py:289)
      --max_steps N           cap the steps per epoch (smoke runs)
      --log_every N           per-step loss line every N steps through asynchronous D2H copies (StepLog; 0 = epoch lines only)
      --launch lanes|eager    lanes (default): plain steps replay captured HIP graphs on two streams (DetectorTrainer.
                              capture_step(lanes=True)) from fixed-shape label records; eager: one launch at a time
    One process per GPU; under torchrun the ranks all-reduce gradients over RCCL (DetectorTrainer)."""
    import os
    import sys
    from ..common import config as C
    from ..common import thumos_dataset as D
    argv = list(sys.argv[1:] if argv is None else argv)
    extra = {'as_shipped_dispatch': False, 'random_init': False, 'save_after': 10, 'max_steps': None, 'log_every': 0,
             'launch': 'lanes'}
    rest, i = [], 0
    while i < len(argv):
        a = argv[i]
        if a in ('--as_shipped_dispatch', '--random_init'):
            extra[a[2:]] = True
        elif a in ('--save_after', '--max_steps', '--log_every'):
            extra[a[2:]] = int(argv[i + 1]); i += 1
        elif a == '--launch':
            extra['launch'] = argv[i + 1]; i += 1
        else:
            rest.append(a)
        i += 1
    if extra['launch'] not in ('lanes', 'eager'):
        raise SystemExit("--launch takes lanes or eager")
    config = C.set_config(C.get_config(rest))
    tr = config['training']
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit("opental_amd.thumos14.train needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=dev)
    ops.CONV_PRECISION = 1 if os.environ.get('OTAL_DTYPE', 'bf16') == 'bf16' else 0
    set_seed(tr['random_seed'])
    net, crit, trainer = build_training(config, dev, extra['as_shipped_dispatch'], extra['random_init'])
    ds_cfg = config['dataset']['training']
    infos = D.get_video_info(ds_cfg['video_info_path'])
    annos = D.get_video_anno(infos, ds_cfg['video_anno_path'], config['dataset']['class_info_path'])
    data = D.load_video_data(infos, ds_cfg['video_data_path'])
    dataset = D.THUMOS_Dataset(data, infos, annos, clip_length=ds_cfg['clip_length'], crop_size=ds_cfg['crop_size'],
                               stride=ds_cfg['clip_stride'])
    any_video = next(iter(data.values()))
    stager = D.ClipStager(tr['batch_size'], ds_cfg['clip_length'], int(any_video.shape[1]), int(any_video.shape[2]),
                          ds_cfg['crop_size'], device=dev, max_targets=D.max_target_count(dataset), score_rows=2,
                          copy_stream=ops.side_wgrads(dev).side)       # the next batch crosses PCIe on the weight-gradient
                                                                        # lane's stream, idle during the forward pass (ClipStager)
    trainer.launch = extra['launch']
    checkpoint_path = tr['checkpoint_path']
    train_state_path = os.path.join(checkpoint_path, 'training')
    start_epoch = trainer.resume_training(tr['resume'], checkpoint_path, train_state_path)
    if rank == 0:
        print(f"batch size: {tr['batch_size']}  learning rate: {tr['learning_rate']}  weight decay: {tr['weight_decay']}  "
              f"max epoch: {tr['max_epoch']}  cls loss: {crit.cls_loss_type}  clips: {len(dataset)}  ranks: {world}  resume: {tr['resume']}")
    history = []
    step_log = None
    if extra['log_every'] > 0 and rank == 0:
        # per-step scalars: a console line AND <checkpoint_path>/training/scalars.jsonl (the reference's TensorBoard scalars)
        step_log = StepLog(extra['log_every'], sink=JsonlSink(train_state_path, echo=lambda step, v: print(
            'step {} '.format(step) + ', '.join('{} {:.5f}'.format(n, x) for n, x in zip(StepLog.NAMES, v)))))
    trainer.step_log = step_log
    for epoch in range(start_epoch, tr['max_epoch'] + 1):
        if crit.cls_loss_type == 'edl':
            crit.cls_loss.epoch = epoch
            crit.cls_loss.total_epoch = tr['max_epoch']
        v = run_one_epoch(epoch, trainer, dataset, stager, tr['batch_size'], rank, world, extra['max_steps'],
                          log=print if rank == 0 else (lambda *a: None), step_log=step_log)
        history.append(v)
        if epoch > extra['save_after'] and rank == 0:
            trainer.save_model(epoch, checkpoint_path, train_state_path)
    if world > 1:
        dist.barrier()
    return trainer, history


if __name__ == '__main__':
    main()
