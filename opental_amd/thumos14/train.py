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
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from ..common import ops


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
        # one launch per map: tanh, channel mean, BCE and the gradient, on the channel-major maps in place (csrc/bce.hip)
        from ..common.ops import BoundaryBCEFunction
        loss_start, loss_end = BoundaryBCEFunction.apply(src[0], scores, 0, 1)
        a, b = BoundaryBCEFunction.apply(src[1], scores, 0, 4)       # F.interpolate(scale_factor=1/4), nearest
        c, d = BoundaryBCEFunction.apply(src[2], scores, 0, 4)
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


class FlatArena:
    """Trainable parameters re-homed into one contiguous fp32 buffer (+ parallel grad / m / v)."""

    def __init__(self, params, bucket_bytes, adjacent=()):
        """`adjacent`: groups of parameters that must sit back to back, in the given order (e.g. the three 1x1 weights of
        an Inception module that one fused launch reads as a single matrix, InceptionModule.fused_1x1_weights)."""
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
        for i, p in enumerate(self.params):
            self.bucket_of.append(len(self.buckets))
            cur += p.numel() * 4
            last = i == len(self.params) - 1
            if cur >= bucket_bytes or last:
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
                 process_group=None, bucket_mb=48, distributed=None, force_collectives=False, param_groups=None,
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
        self.arena = FlatArena(list(net.parameters()), bucket_mb << 20, adjacent)
        self.param_groups = [(list(net.parameters()), lr)] if param_groups is None else [(list(ps), g_lr) for ps, g_lr in param_groups]
        self._group_ranges = self._arena_ranges()
        self.step_count = 0
        self._pending, self._works = None, []
        lo = self.arena.flat.data_ptr()
        self._prologues = ops.PrologueCache((lo, lo + 4 * self.arena.numel))   # per-layer tables + bf16 weights, refreshed once per step
        self._graph = None          # (CUDAGraph, static inputs, static outputs) once capture_step() succeeded
        self._bias_corr = None      # 2-float device tensor: Adam bias corrections of the step being run
        self.collectives = self.distributed and (self.world > 1 or force_collectives)   # force: 1-rank RCCL smoke test
        self._flushed = None
        self._slots = ops.GradSlots(self.arena.flat, self.arena.grad, self.arena.offsets, [p.numel() for p in self.arena.params])
        for i, p in enumerate(self.arena.params):
            p.register_post_accumulate_grad_hook(self._make_hook(self.arena.bucket_of[i]))

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
    # launching `arena_view += grad` per parameter (~190 tiny kernels per step + a 179 MB zero fill); when the last
    # gradient of a bucket has arrived, one multi-tensor copy moves the bucket into the arena and, in data-parallel
    # runs, its all-reduce starts.
    def _make_hook(self, b):
        def hook(_param):
            if self._pending is None:
                return
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._flush_bucket(b)
        return hook

    def _flush_bucket(self, b):
        if self._flushed[b]:
            return
        self._flushed[b] = True
        a = self.arena
        dst, src = [], []
        for i in a.bucket_members[b]:
            g = a.params[i].grad
            if g is None:
                a.grad_views[i].zero_()                 # parameter unused this step
            elif g.data_ptr() != a.grad_views[i].data_ptr():
                dst.append(a.grad_views[i]); src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        if self.collectives:
            lo, hi = a.buckets[b]
            self._works.append(dist.all_reduce(a.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin_backward(self):
        """Call before cost.backward(): gradients start undefined, buckets open."""
        for p in self.arena.params:
            p.grad = None
        self._pending = list(self.arena.bucket_size)
        self._flushed = [False] * len(self.arena.buckets)
        self._slots.reset()
        ops.GRAD_SLOTS = self._slots                # weight gradients are written straight into the arena

    def end_backward(self):
        """Call after cost.backward(): flush the buckets that did not complete (unused parameters), wait for the
        all-reduces, and leave every .grad aliasing its arena slice."""
        ops.GRAD_SLOTS = None
        for b in range(len(self.arena.buckets)):
            self._flush_bucket(b)
        self._pending = None
        for p, v in zip(self.arena.params, self.arena.grad_views):
            p.grad = v
        self._finish_allreduce()

    def _finish_allreduce(self):
        if not self.collectives:
            return
        for w in self._works:
            w.wait()
        self._works = []
        if getattr(self.criterion, 'cls_loss_type', None) == 'edl' and self.criterion.cls_loss.with_ibm:
            acc = self.criterion.cls_loss.weight_accum
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
            acc.div_(self.world)

    # ---- one optimisation step
    def compute_cost(self, clips, targets, scores, ssl_clips=None, ssl_targets=None):
        losses = self.forward_fn(self.net, self.criterion, clips, targets, scores, training=True, ssl=False)
        cost = total_cost(losses, self.w)
        if ssl_clips is not None:
            trip = self.forward_fn(self.net, self.criterion, ssl_clips, ssl_targets, training=True, ssl=True)
            cost = cost + trip * self.w['ssl']
        return cost, losses

    def step(self, clips, targets, scores, ssl_clips=None, ssl_targets=None):
        if self._graph is not None and ssl_clips is None:
            return self._replay(clips, targets, scores)
        ops.activate_prologues(self._prologues)        # ONE launch re-packs the weights of every known conv layer
        try:
            cost, losses = self.compute_cost(clips, targets, scores, ssl_clips, ssl_targets)
            self.begin_backward()
            cost.backward()
            self.end_backward()
        finally:
            ops.deactivate_prologues()
            ops.GRAD_SLOTS = None
        self.step_count += 1
        self.optimizer_update()
        return cost.detach(), losses

    def optimizer_update(self):
        """Adam (L2 weight decay in the gradient, train.py:321-323) on the flat arena: one launch."""
        a = self.arena
        for lo, hi, g_lr in self._group_ranges:
            g_lr = g_lr * (self.lr / self._base_lr) if self._base_lr else g_lr
            ops.adam_flat(a.flat[lo:hi], a.grad[lo:hi], a.m[lo:hi], a.v[lo:hi], self.step_count, g_lr,
                          self.betas[0], self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world)

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
        a = self.arena
        for lo, hi, g_lr in self._group_ranges:
            g_lr = g_lr * (self.lr / self._base_lr) if self._base_lr else g_lr
            ops.adam_flat_dev(a.flat[lo:hi], a.grad[lo:hi], a.m[lo:hi], a.v[lo:hi], self._bias_corr, g_lr,
                              self.betas[0], self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world)
        return cost.detach(), losses

    def capture_step(self, clips, targets, scores, warmup=2):
        """Capture forward + losses + backward (+ gradient all-reduce) + Adam for inputs of these shapes.

        The captured launches bake in every host-side scalar of the step; the only one that changes per step --
        Adam's bias correction -- lives in device memory and is refreshed before each replay.  Quantities that
        change per EPOCH (the evidential loss's annealing coefficient) need a re-capture at the epoch boundary.
        `warmup` eager steps run first on a side stream (they are real optimisation steps)."""
        dev = clips.device
        self._graph = None
        self._bias_corr = torch.zeros(2, dtype=torch.float32, device=dev)
        clone = lambda t: None if t is None else ([u.clone() for u in t] if isinstance(t, (list, tuple)) else t.clone())
        static = tuple(clone(t) for t in (clips, targets, scores))
        # (the warm-up steps run on a side stream on purpose: the AccumulateGrad stream-mismatch warning is noise here)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = torch.cuda.Stream(device=dev)
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
        return self

    def _set_bias(self, step):
        bc = ops.adam_bias_corrections(step, self.betas[0], self.betas[1])
        self._bias_corr.copy_(torch.tensor(bc, dtype=torch.float32), non_blocking=False)

    def _replay(self, clips, targets, scores):
        graph, static, out = self._graph
        pairs = []
        for dst, src in zip(static, (clips, targets, scores)):
            if isinstance(dst, list):
                if len(dst) != len(src):
                    raise RuntimeError("captured step replayed with a different batch; call capture_step again")
                pairs += list(zip(dst, src))
            elif dst is not None:
                pairs.append((dst, src))
        for dst, src in pairs:
            if dst.data_ptr() != src.data_ptr():
                if dst.shape != src.shape:      # ragged targets: the per-sample row counts are baked into the capture
                    raise RuntimeError("captured step replayed with different input shapes; call capture_step again")
                dst.copy_(src, non_blocking=True)
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
        for ps, g_lr in self.param_groups:
            groups.append({'lr': g_lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.wd,
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
