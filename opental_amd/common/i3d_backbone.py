"""Inception-v1 I3D feature extractor on MI355X.

Same module tree / state-dict keys / `extract_features` contract as the reference
(AFSD/common/i3d_backbone.py: Unit3D :7-87, InceptionModule :90-121, InceptionI3d :124-342), but the
execution is MI355X-first:

  * the whole backbone is ONE autograd node (`I3DFeaturesFunction`) with an explicit tape, instead
    of ~190 ATen nodes (50+ conv3d, BN, ReLU, pad, pool, cat);
  * every Unit3D = SAME-pad + Conv3d + frozen BatchNorm3d(eps 1e-3) + ReLU is one implicit-GEMM
    launch on the matrix cores: padding is virtual, BN is a per-channel scale/shift epilogue;
  * Inception concatenation is written in place (each branch's GEMM stores straight into its
    channel slice of the module output) -- no torch.cat copies;
  * backward folds the ReLU mask and the BN scale into the dy loaders of the dgrad/wgrad GEMMs and
    accumulates the four branch gradients into one buffer -- no elementwise passes over the
    (up to 600 MB) activation maps;
  * max-pools keep a uint8 winner tap; their backward is a gather (no atomics).

BatchNorm is supported in the configuration the reference trains with: frozen statistics and frozen
affine (configs/thumos14_opental_final.yaml: freeze_bn / freeze_bn_affine; BDNet.py:39-49).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops
from .layers import MaxPool3dSamePadding

ONE, THREE = (1, 1, 1), (3, 3, 3)
BN_EPS = 1e-3


class Unit3D(nn.Module):
    """Parameter container with the reference's signature (i3d_backbone.py:9-44)."""

    def __init__(self, in_channels, output_channels, kernel_shape=(1, 1, 1), stride=(1, 1, 1), padding=0,
                 activation_fn=F.relu, use_batch_norm=True, use_bias=False, padding_valid_spatial=False,
                 name='unit_3d'):
        super(Unit3D, self).__init__()
        if not use_batch_norm or use_bias or activation_fn is not F.relu or padding_valid_spatial or padding == -1:
            raise NotImplementedError("backbone Unit3D is conv(no bias) + BN + ReLU with SAME padding")
        self._output_channels = output_channels
        self._kernel_shape = tuple(kernel_shape)
        self._stride = tuple(stride)
        self.name = name
        self.conv3d = nn.Conv3d(in_channels, output_channels, self._kernel_shape, self._stride, padding=0, bias=False)
        self.bn = nn.BatchNorm3d(output_channels, eps=BN_EPS, momentum=0.01)

    def folded_bn(self):
        """frozen BN as y = scale * conv + shift."""
        bn = self.bn
        scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        return scale, bn.bias.detach() - bn.running_mean * scale

    def forward(self, x):
        scale, shift = self.folded_bn()
        return I3DFeaturesFunction.apply(x, [("conv", 0, self._kernel_shape, self._stride)], ("out",), scale, shift,
                                         [0, scale.numel()], self.conv3d.weight)[0]


class InceptionModule(nn.Module):
    def __init__(self, in_channels, out_channels, name):
        super(InceptionModule, self).__init__()
        oc = out_channels
        self.b0 = Unit3D(in_channels, oc[0], ONE, name=name + '/Branch_0/Conv3d_0a_1x1')
        self.b1a = Unit3D(in_channels, oc[1], ONE, name=name + '/Branch_1/Conv3d_0a_1x1')
        self.b1b = Unit3D(oc[1], oc[2], THREE, name=name + '/Branch_1/Conv3d_0b_3x3')
        self.b2a = Unit3D(in_channels, oc[3], ONE, name=name + '/Branch_2/Conv3d_0a_1x1')
        self.b2b = Unit3D(oc[3], oc[4], THREE, name=name + '/Branch_2/Conv3d_0b_3x3')
        self.b3a = MaxPool3dSamePadding(kernel_size=[3, 3, 3], stride=(1, 1, 1), padding=0)
        self.b3b = Unit3D(in_channels, oc[5], ONE, name=name + '/Branch_3/Conv3d_0b_1x1')
        self.name = name
        self.out_channels = tuple(oc)

    def units(self):
        return [self.b0, self.b1a, self.b1b, self.b2a, self.b2b, self.b3b]

    def fused_1x1_weights(self):
        """The three 1x1 weights that read the module input, in the channel order of the fused launch [b1a | b2a | b0]:
        a trainer that homes parameters in a flat arena keeps them ADJACENT in this order (FlatArena), so that the fused
        weight is a zero-copy view and its packed form can live in a persistent prologue region."""
        return [self.b1a.conv3d.weight, self.b2a.conv3d.weight, self.b0.conv3d.weight]


# endpoint table (i3d_backbone.py:194-296)
_ENDPOINTS = (
    ('Conv3d_1a_7x7', 'conv', (None, 64, (7, 7, 7), (2, 2, 2))),
    ('MaxPool3d_2a_3x3', 'pool', ((1, 3, 3), (1, 2, 2))),
    ('Conv3d_2b_1x1', 'conv', (64, 64, ONE, ONE)),
    ('Conv3d_2c_3x3', 'conv', (64, 192, THREE, ONE)),
    ('MaxPool3d_3a_3x3', 'pool', ((1, 3, 3), (1, 2, 2))),
    ('Mixed_3b', 'mixed', (192, (64, 96, 128, 16, 32, 32))),
    ('Mixed_3c', 'mixed', (256, (128, 128, 192, 32, 96, 64))),
    ('MaxPool3d_4a_3x3', 'pool', (THREE, (2, 2, 2))),
    ('Mixed_4b', 'mixed', (480, (192, 96, 208, 16, 48, 64))),
    ('Mixed_4c', 'mixed', (512, (160, 112, 224, 24, 64, 64))),
    ('Mixed_4d', 'mixed', (512, (128, 128, 256, 24, 64, 64))),
    ('Mixed_4e', 'mixed', (512, (112, 144, 288, 32, 64, 64))),
    ('Mixed_4f', 'mixed', (528, (256, 160, 320, 32, 128, 128))),
    ('MaxPool3d_5a_2x2', 'pool', ((2, 2, 2), (2, 2, 2))),
    ('Mixed_5b', 'mixed', (832, (256, 160, 320, 32, 128, 128))),
    ('Mixed_5c', 'mixed', (832, (384, 192, 384, 48, 128, 128))),
)


class InceptionI3d(nn.Module):
    VALID_ENDPOINTS = tuple(n for n, _, _ in _ENDPOINTS) + ('Logits', 'Predictions')

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint='Mixed_5c', name='inception_i3d',
                 in_channels=3, dropout_keep_prob=0.5):
        super(InceptionI3d, self).__init__()
        if final_endpoint not in self.VALID_ENDPOINTS:
            raise ValueError('Unknown final endpoint %s' % final_endpoint)
        if final_endpoint in ('Logits', 'Predictions'):
            raise NotImplementedError("the detection path stops at Mixed_5c (BDNet.py:27)")
        self._final_endpoint = final_endpoint
        self.end_points = {}
        for ep, kind, args in _ENDPOINTS:
            if kind == 'conv':
                cin, cout, k, s = args
                self.end_points[ep] = Unit3D(in_channels if cin is None else cin, cout, k, s, name=name + ep)
            elif kind == 'pool':
                self.end_points[ep] = MaxPool3dSamePadding(kernel_size=list(args[0]), stride=args[1], padding=0)
            else:
                self.end_points[ep] = InceptionModule(args[0], list(args[1]), name + ep)
            if ep == final_endpoint:
                break
        self._plan = None

    def build(self):
        for k in self.end_points.keys():
            self.add_module(k, self.end_points[k])

    def _make_plan(self):
        """Flatten the module tree into the launch plan of I3DFeaturesFunction."""
        plan, units = [], []
        for ep, kind, args in _ENDPOINTS:
            if ep not in self.end_points:
                break
            mod = self.end_points[ep]
            if kind == 'conv':
                plan.append(("conv", len(units), mod._kernel_shape, mod._stride))
                units.append(mod)
            elif kind == 'pool':
                plan.append(("pool", tuple(mod.kernel_size), tuple(mod.stride)))
            else:
                plan.append(("mixed", len(units), mod.out_channels))
                units += mod.units()
            plan[-1] = plan[-1] + (ep,)
        return plan, units

    def extract_features(self, x, endpoints=('Mixed_4f', 'Mixed_5c')):
        """dict endpoint -> tensor.  The reference returns all 16 endpoints (i3d_backbone.py:335-342)
        but only Mixed_4f / Mixed_5c are consumed (BDNet.py:307-308); pass `endpoints` for others."""
        if self._plan is None:
            self._plan = self._make_plan()
        plan, units = self._plan
        for u in units:
            if u.bn.training or (torch.is_grad_enabled() and u.bn.weight.requires_grad):
                raise NotImplementedError("I3D BatchNorm must be frozen (freeze_bn + freeze_bn_affine, BDNet.py:39-49)")
        # frozen BN folds to a constant per-channel (scale, shift): rebuild only when a BN tensor changed
        # (load_state_dict / .to() bump the version counters or replace the storages)
        key = tuple((u.bn.weight._version, u.bn.bias._version, u.bn.running_mean._version, u.bn.running_var._version,
                     u.bn.weight.data_ptr(), u.bn.running_var.data_ptr()) for u in units)
        if getattr(self, "_fold_key", None) != key:
            folded = [u.folded_bn() for u in units]
            offs = [0]
            for sc, _ in folded:
                offs.append(offs[-1] + sc.numel())
            self._fold = (torch.cat([sc for sc, _ in folded]), torch.cat([sh for _, sh in folded]), offs)
            self._fold_key = key
        scale, shift, offs = self._fold
        weights = [u.conv3d.weight for u in units]
        cut = self._stem_cut(plan, endpoints) if self.split_backward else 0
        if cut:
            # Two autograd nodes instead of one, cut behind MaxPool3d_4a: a data-parallel trainer runs the backward pass
            # in two phases -- everything down to the cut, then the stem (Conv3d_1a .. Mixed_3c: a third of the backward's
            # GPU time, 3 % of the parameters) -- and all-reduces the first phase's gradients while the second runs
            # (thumos14/train.py, capture_step(split=True)).  The cut tensor is a pool output: its gradient is raw.
            # ("=": the cut tensor is handed over as the stem stores it -- bf16 when ops.HALF_STORAGE is on)
            (stem_out,) = I3DFeaturesFunction.apply(x, plan[:cut], ("=" + plan[cut - 1][-1],), scale, shift, offs, *weights)
            self.stem_out = cut_in = stem_out
            if self.detach_cut and torch.is_grad_enabled() and stem_out.requires_grad:
                # the trunk reads a LEAF copy of the cut tensor (same storage): backward(cost) then stops at the cut and
                # leaves its gradient in cut_leaf.grad; the caller runs the stem with backward(stem_out, cut_leaf.grad).
                # (Naming the non-leaf stem_out in backward(inputs=...) would not do: autograd executes the node that
                # owns a requested non-leaf tensor.)
                self.cut_leaf = cut_in = stem_out.detach().requires_grad_(True)
            outs = I3DFeaturesFunction.apply(cut_in, plan[cut:], tuple(endpoints), scale, shift, offs, *weights)
        else:
            self.stem_out = None
            outs = I3DFeaturesFunction.apply(x, plan, tuple(endpoints), scale, shift, offs, *weights)
        return dict(zip(endpoints, outs))

    split_backward = False      # set by a trainer that wants the two-phase backward (see extract_features)
    detach_cut = False          # ... and, while it captures the two phases, the autograd graph broken at the cut
    stem_out = None             # the cut tensor of the last forward pass when split_backward is on
    cut_leaf = None             # its leaf twin when detach_cut is on
    STEM_CUT_AFTER = 'MaxPool3d_4a_3x3'

    def _stem_cut(self, plan, endpoints):
        """Index of the first trunk step, or 0 when the plan cannot be cut (an endpoint inside the stem, no such pool)."""
        names = [st[-1] for st in plan]
        if self.STEM_CUT_AFTER not in names:
            return 0
        cut = names.index(self.STEM_CUT_AFTER) + 1
        if cut >= len(plan) or any(e in names[:cut] for e in endpoints):
            return 0
        return cut

    def stem_parameters(self):
        """The convolution weights in front of the cut (Conv3d_1a .. Mixed_3c)."""
        if self._plan is None:
            self._plan = self._make_plan()
        plan, units = self._plan
        cut = self._stem_cut(plan, ())
        idx = set()
        for st in plan[:cut]:
            if st[0] == "conv":
                idx.add(st[1])
            elif st[0] == "mixed":
                idx.update(range(st[1], st[1] + 6))
        return [units[i].conv3d.weight for i in sorted(idx)]

    def forward(self, x):
        raise NotImplementedError("classification logits are not part of the detection path; use extract_features")


_FUSED_AFFINE = {}


def _fused_weight(w_b1a, w_b2a, w_b0):
    """[b1a | b2a | b0] along the output channels: a view when the three live back to back in one storage, else a copy."""
    n1, n2 = w_b1a.numel(), w_b2a.numel()
    es = w_b1a.element_size()
    if (w_b1a.is_contiguous() and w_b2a.is_contiguous() and w_b0.is_contiguous()
            and w_b2a.data_ptr() == w_b1a.data_ptr() + n1 * es and w_b0.data_ptr() == w_b2a.data_ptr() + n2 * es
            and w_b1a.untyped_storage().data_ptr() == w_b0.untyped_storage().data_ptr()):
        rows = w_b1a.shape[0] + w_b2a.shape[0] + w_b0.shape[0]
        return w_b1a.as_strided((rows,) + tuple(w_b1a.shape[1:]), w_b1a.stride(), w_b1a.storage_offset())
    return torch.cat([w_b1a, w_b2a, w_b0], 0)


def _cat_cached(cache, owner, tag, parts):
    """torch.cat(parts), cached per (owner storage, tag); the entry keeps `owner` alive, so its address cannot be handed to
    another tensor while the key exists (a refold after load_state_dict / a second model gets its own entries)."""
    key = (owner.data_ptr(), owner._version, tag)
    hit = cache.get(key)
    if hit is None:
        hit = (torch.cat(parts), owner)
        if len(cache) > 256:
            cache.clear()
        cache[key] = hit
    return hit[0]


_OUT_SCALE = {}


def _fused_affine(scale, shift, offs, w0):
    """Folded-BN scale / shift of the fused 1x1 launch of a module, in its channel order [b1a | b2a | b0] (constants)."""
    # the entry holds the folded tensors, so their addresses cannot be recycled under the key: a refold after
    # load_state_dict, or a second model in the process, gets a new entry
    key = (scale.data_ptr(), shift.data_ptr(), scale._version, w0)
    hit = _FUSED_AFFINE.get(key)
    if hit is None:
        pick = lambda t, i: t[offs[i]:offs[i + 1]]
        hit = (torch.cat([pick(scale, w0 + 1), pick(scale, w0 + 3), pick(scale, w0)]),
               torch.cat([pick(shift, w0 + 1), pick(shift, w0 + 3), pick(shift, w0)]), scale, shift)
        if len(_FUSED_AFFINE) > 256:
            _FUSED_AFFINE.clear()
        _FUSED_AFFINE[key] = hit
    return hit[0], hit[1]


_HALF_OK = {}


def _pool_half_ok(shape, k, s):
    """A max-pool that has bf16-tensor kernels (csrc/pool3d.hip): the strided 3x3 pools and the 12x12 / 6x6 branch pools."""
    _, _, T, H, W = shape
    k, s = tuple(k), tuple(s)
    # ... whose OUTPUT map the next bf16-tensor kernel can take: the convolutions and otal_convert_storage move eight
    # positions per lane, so To * Ho * Wo must be a multiple of eight (SAME padding: ceil(in / stride))
    out_positions = -(-T // s[0]) * -(-H // s[1]) * -(-W // s[2])
    if out_positions % 8:
        return False
    if k[1:] == (3, 3) and s[1:] == (2, 2) and H % 2 == 0 and W % 4 == 0:
        return (k[0], s[0]) == (1, 1) or ((k[0], s[0]) == (3, 2) and T % 2 == 0)
    return k == THREE and s == ONE and H == W and H in (12, 6)


def _conv_half_ok(shape, cout, k, first=False):
    """A stride-1 convolution whose forward, data gradient (unless it is the network's first layer) and weight gradient all
    have bf16-tensor kernels for this geometry."""
    modes = (0, 2) if first else (0, 1, 2)
    return all(ops.half_storage_ok(m, tuple(shape), cout, k, ONE, both=True) for m in modes)


def _step_half_ok(step, shape, weights):
    """May this plan step run on bf16-STORED tensors (input of `shape`, its own intermediates, its output)?"""
    key = (step[0], tuple(shape), step[1:-1] if step[0] != "conv" else (tuple(weights[step[1]].shape), step[2], step[3]), ops.HALF_CHAIN,
           int(ops.CONV_PRECISION), ops.HALF_STORAGE)
    hit = _HALF_OK.get(key)
    if hit is None:
        B, C, T, H, W = shape
        if not (ops.HALF_CHAIN and ops.HALF_STORAGE and int(ops.CONV_PRECISION) & 1):
            hit = False
        elif step[0] == "conv":
            hit = tuple(step[3]) == ONE and _conv_half_ok(shape, weights[step[1]].shape[0], tuple(step[2]))
        elif step[0] == "pool":
            hit = _pool_half_ok(shape, step[1], step[2])
        else:
            oc = step[2]
            hit = (_conv_half_ok(shape, oc[1] + oc[3] + oc[0], ONE) and _conv_half_ok((B, oc[1], T, H, W), oc[2], THREE)
                   and _conv_half_ok((B, oc[3], T, H, W), oc[4], THREE) and _pool_half_ok(shape, THREE, ONE)
                   and _conv_half_ok(shape, oc[5], ONE))
        if len(_HALF_OK) > 512:
            _HALF_OK.clear()
        _HALF_OK[key] = hit
    return hit


class I3DFeaturesFunction(Function):
    """Forward + hand-written backward tape of the whole backbone.

    Gradient convention inside the tape: the gradient held for the output Z of a conv / Inception
    step is already multiplied by (Z > 0) * bn_scale[channel] ("dz"), because every kernel that
    contributes to it (the consumers' dgrad GEMMs, max-pool backward) applies that factor in its
    store epilogue.  The producer's wgrad / dgrad GEMMs therefore read ONE tensor per operand
    instead of gradient + activation.  Gradients of pool outputs are raw.

    bf16 STORAGE (ops.HALF_STORAGE + ops.HALF_CHAIN, bf16-operand mode).  Every consumer of a backbone activation rounds it
    to bf16 while staging it (convolutions) or commutes with that rounding (max-pools), and every consumer of a data
    gradient is a convolution's bf16 operand loader: between Conv3d_1a and the last module whose planes the bf16-tensor
    kernels cover (Mixed_4f: 6 x 6), activations AND data gradients are therefore STORED as bf16 -- half the HBM bytes of
    the 1x1x1 layers and the pools, which stream at the memory rate, with forward values that do not change by a bit.
    Backward: a bf16-stored gradient holds exactly what its consumers would have rounded the fp32 tensor to, except where a
    tensor has two producers (a module's input gradient: the fused 1x1 data gradient stores, the branch pool's backward
    adds -- read bf16, add in fp32, round once).  Where the region ends (an endpoint handed to the pyramid, a step without
    bf16-tensor kernels) a ("cvt") tape entry converts: bf16 -> fp32 forward, fp32 -> bf16 backward -- so gradients from
    outside always meet on the fp32 side."""

    @staticmethod
    def forward(ctx, x, plan, endpoints, scale, shift, offs, *weights):
        x = x.contiguous()
        sc = lambda i: scale[offs[i]:offs[i + 1]]
        sh = lambda i: shift[offs[i]:offs[i + 1]]
        tape, found, half_grads = [], {}, {}
        cur = x
        cur_scale = None            # bn scale vector of the tensor `cur` when it is a conv/mixed output
        BF = torch.bfloat16
        need_grad = any(ctx.needs_input_grad)      # (grad mode is off inside forward(): ask what autograd will request)
        raw_out = {e[1:] for e in endpoints if e.startswith("=")}       # "=name": handed over as stored (a bf16 hand-over
        endpoints = tuple(e.lstrip("=") for e in endpoints)             #  between the two nodes of a split backward)

        def leave_half():
            """bf16-stored `cur` -> an fp32 copy; the tape entry converts the gradient back."""
            nonlocal cur
            cur32 = ops.convert_storage(cur, torch.float32)
            tape.append(("cvt", cur, cur32, cur_scale))
            cur = cur32

        for si, step in enumerate(plan):
            kind, name = step[0], step[-1]
            chain = cur.dtype == BF and _step_half_ok(step, cur.shape, weights)
            # a bf16-stored conv output straight in front of its (1,3,3)/(1,2,2) pool WITHOUT the chain (ops.HALF_CHAIN off,
            # or a geometry the chain's kernels do not cover): the pool's bf16-in / fp32-out kernel reads it as it is
            # (ops.maxpool3d_forward on a bfloat16 input) -- no conversion pass, no "cvt" seam, the gradient convention of
            # the tape entry below (half_grads) decides how the pool's backward stores dx
            # (... where that kernel exists: the sign-bit form needs even H and W % 4 == 0, as _pool_half_ok states it -- any other
            # plane takes the conversion seam like every geometry without bf16-tensor kernels)
            pool_io1 = (cur.dtype == BF and not chain and kind == "pool" and cur_scale is not None
                        and tuple(step[1]) == (1, 3, 3) and tuple(step[2]) == (1, 2, 2)
                        and cur.shape[3] % 2 == 0 and cur.shape[4] % 4 == 0)
            if cur.dtype == BF and not chain and not pool_io1:
                leave_half()
            if kind == "conv" and chain:
                _, wi, k, s, _ = step
                y = ops.conv_forward(cur, weights[wi], k, s, scale=sc(wi), shift=sh(wi), relu=True)
                tape.append(("conv", wi, k, s, cur, y, cur_scale, sc(wi)))
                cur, cur_scale = y, sc(wi)
            elif kind == "conv":
                _, wi, k, s, _ = step
                # a convolution whose output only feeds a strided (1,3,3)/(1,2,2) pool -- Conv3d_1a -> MaxPool3d_2a, Conv3d_2c ->
                # MaxPool3d_3a -- may store it as bf16: the pool commutes with the rounding the next convolution applies to its
                # operand anyway, and the pool keeps the layer's ReLU mask as sign bits.  For the FIRST layer (no data gradient)
                # the gradient of that tensor is stored as bf16 too (its only consumer, the weight gradient, rounds it anyway);
                # deeper layers keep fp32 gradients.
                pooled_next = (name not in endpoints and si + 1 < len(plan) and plan[si + 1][0] == "pool"
                               and tuple(plan[si + 1][1]) == (1, 3, 3) and tuple(plan[si + 1][2]) == (1, 2, 2)
                               and ops.half_storage_ok(0, tuple(cur.shape), weights[wi].shape[0], k, s))
                grad_half = (pooled_next and si == 0 and not x.requires_grad
                             and ops.half_storage_ok(2, tuple(cur.shape), weights[wi].shape[0], k, s))
                half = pooled_next and (grad_half or (si > 0 and ops.HALF_ACT_DIRECT))
                # ... and with the bf16 chain on, the first layer's bf16 output starts it whatever follows
                if (not half and si == 0 and not x.requires_grad and ops.HALF_CHAIN and si + 1 < len(plan) and name not in endpoints
                        and ops.half_storage_ok(0, tuple(cur.shape), weights[wi].shape[0], k, s)
                        and ops.half_storage_ok(2, tuple(cur.shape), weights[wi].shape[0], k, s)):
                    half = True
                half_grads[si + 1] = grad_half
                y = ops.conv_forward(cur, weights[wi], k, s, scale=sc(wi), shift=sh(wi), relu=True, half_out=half)
                tape.append(("conv", wi, k, s, cur, y, cur_scale, sc(wi)))
                cur, cur_scale = y, sc(wi)
            elif kind == "pool":
                _, k, s, _ = step
                # a pool behind a conv + ReLU: let the forward kernel keep that layer's ReLU mask as sign bits, so that the
                # backward pass does not re-read the 4-byte activations only for their sign
                if chain:
                    # (cur_scale is not None: `cur` is a conv / Inception output behind its ReLU, so >= +0 everywhere)
                    y, arg, bits = ops.maxpool3d_forward(cur, k, s, signbits=True, half_out=True, nonneg=ops.POOL_KEYS and cur_scale is not None)
                elif cur_scale is not None:
                    y, arg, bits = ops.maxpool3d_forward(cur, k, s, signbits=True)
                else:                                       # (a pool that is not behind a conv + ReLU: nothing to mask)
                    (y, arg), bits = ops.maxpool3d_forward(cur, k, s), None
                tape.append(("pool", k, s, cur, (arg, bits, half_grads.get(si, False)), cur_scale, None))
                cur, cur_scale = y, None
            else:
                _, w0, oc, _ = step
                B, _, T, H, W = cur.shape
                ctot = oc[0] + oc[2] + oc[4] + oc[5]
                # One buffer Z = [h1 | h2 | Y]: the three 1x1 convolutions that read the module input (b1a, b2a, b0) run as
                # ONE GEMM whose output range ends exactly at Y's first slice, so the input is read once instead of three
                # times (forward, weight gradient) and its gradient is written once instead of accumulated three times.
                o1, o13 = oc[1], oc[1] + oc[3]
                Z = torch.empty((B, o13 + ctot, T, H, W), dtype=cur.dtype, device=cur.device)
                h1, h2, Y = Z[:, :o1], Z[:, o1:o13], Z[:, o13:]
                c1, c2, c3 = oc[0], oc[0] + oc[2], oc[0] + oc[2] + oc[4]
                wf = _fused_weight(weights[w0 + 1].detach(), weights[w0 + 3].detach(), weights[w0].detach())
                scf, shf = _fused_affine(scale, shift, offs, w0)
                ops.conv_forward(cur, wf, ONE, ONE, scale=scf, shift=shf, relu=True, out=Z[:, :o13 + c1])
                # the two small branches (3x3x3 on h2; pool + 1x1) run on the branch lane beside the large 3x3x3 on h1: on
                # the 6x6 / 3x3 planes none of them fills the chip alone
                lane = ops.branch_lane(cur.device)
                if lane.on:
                    lane.fork()
                    with lane:
                        ops.conv_forward(h2, weights[w0 + 4], THREE, ONE, scale=sc(w0 + 4), shift=sh(w0 + 4), relu=True,
                                         out=Y[:, c2:c3])
                        pm, argm = ops.maxpool3d_forward(cur, THREE, ONE, half_out=chain)
                        ops.conv_forward(pm, weights[w0 + 5], ONE, ONE, scale=sc(w0 + 5), shift=sh(w0 + 5), relu=True,
                                         out=Y[:, c3:])
                    ops.conv_forward(h1, weights[w0 + 2], THREE, ONE, scale=sc(w0 + 2), shift=sh(w0 + 2), relu=True,
                                     out=Y[:, c1:c2])
                    lane.join()
                else:
                    ops.conv_forward(h1, weights[w0 + 2], THREE, ONE, scale=sc(w0 + 2), shift=sh(w0 + 2), relu=True,
                                     out=Y[:, c1:c2])
                    ops.conv_forward(h2, weights[w0 + 4], THREE, ONE, scale=sc(w0 + 4), shift=sh(w0 + 4), relu=True,
                                     out=Y[:, c2:c3])
                    pm, argm = ops.maxpool3d_forward(cur, THREE, ONE, half_out=chain)
                    ops.conv_forward(pm, weights[w0 + 5], ONE, ONE, scale=sc(w0 + 5), shift=sh(w0 + 5), relu=True,
                                     out=Y[:, c3:])
                out_scale = _cat_cached(_OUT_SCALE, scale, w0, [sc(w0), sc(w0 + 2), sc(w0 + 4), sc(w0 + 5)])
                tape.append(("mixed", w0, (c1, c2, c3), cur, h1, h2, pm, argm, Y, cur_scale, (wf, o1, o13), out_scale))
                cur, cur_scale = Y, out_scale
            if name in endpoints:
                if cur.dtype == BF and name not in raw_out:
                    ends = si + 1 == len(plan) or not _step_half_ok(plan[si + 1], cur.shape, weights)
                    if need_grad or ends:
                        leave_half()                        # the region ends here: gradients from outside meet on the fp32 side
                        found[name] = (cur, len(tape))
                    else:                                   # no gradient will come back: a copy for the caller, the chain goes on
                        found[name] = (ops.convert_storage(cur, torch.float32), len(tape))
                else:
                    found[name] = (cur, len(tape))
                hook = ops.ENDPOINT_HOOKS.get(name) if ops.ENDPOINT_HOOKS else None
                if hook is not None:                        # a consumer that can start on this endpoint now (ops.early_lane)
                    hook(found[name][0])
        missing = [e for e in endpoints if e not in found]
        if missing:
            raise RuntimeError(f"unknown endpoints {missing}")
        ctx.tape = tape
        ctx.meta = (scale, offs, [found[e][1] for e in endpoints], len(weights), x.requires_grad)
        ctx.weights = weights
        ctx.outs = [found[e][0] for e in endpoints]
        return tuple(ctx.outs)

    @staticmethod
    def backward(ctx, *douts):
        tape, weights = ctx.tape, ctx.weights
        scale, offs, out_pos, nw, need_dx = ctx.meta
        sc = lambda i: scale[offs[i]:offs[i + 1]]
        dws = [None] * nw
        # external gradients arrive raw at the output of tape step (pos-1); that output's ReLU mask and BN scale are
        # applied when the walk below reaches the step, in the same launch that moves the gradient into its buffer
        # (ops.masked_scale_copy: the gradient may be a permuted view, see ops.conv_dgrad_collapse)
        pending = {}
        for pos, g, z in zip(out_pos, douts, ctx.outs):
            if g is not None:
                pending.setdefault(pos, []).append((g, z, tape[pos - 1][-1]))
        zg = {}                     # gradient buffers [dh1 | dh2 | dY] of the mixed steps, keyed by tape position

        def out_grad_buffer(p, shape, like):
            """Where the gradient w.r.t. the OUTPUT of tape step p (1-based; 0 = the network input) is written.  For a
            mixed step it is the tail of a larger buffer, so that the fused 1x1 backward reads one channel range.
            `like`: a tensor of the dtype the gradient is stored in (= the dtype of that output)."""
            if p >= 1 and tape[p - 1][0] == "mixed":
                o13 = tape[p - 1][10][2]
                B, ct, T, H, W = shape
                buf = torch.empty((B, o13 + ct, T, H, W), dtype=like.dtype, device=like.device)
                zg[p] = buf
                return buf[:, o13:]
            return torch.empty(tuple(shape), dtype=like.dtype, device=like.device)

        # weight gradients run on a second stream beside the data-gradient chain (ops.SideWgrads); the main stream joins
        # them where a data-parallel trainer hands a module's gradients to RCCL (grads_ready -> bucket flush), at the end
        # of the node, or -- a single-GPU trainer writing into its arena -- not before the optimizer step
        side = ops.side_wgrads(weights[0].device)
        in_slots = True
        dcur = None
        cloned = True               # False while dcur still aliases an incoming gradient tensor
        # where only the stem's tail is left (no Inception module below this position): the trainer may start the optimizer
        # step for every other parameter beside these last weight gradients (ops.late_mark)
        mixed_pos = [p for p in range(1, len(tape) + 1) if tape[p - 1][0] == "mixed"]
        tail_below = min(mixed_pos) if mixed_pos and not need_dx else 0
        for pos in range(len(tape), 0, -1):
            if pos == tail_below - 1 and not any(pending.get(q) for q in range(1, pos + 1)):
                ops.late_mark([weights[st[1]] for st in tape[:pos] if st[0] == "conv"])
            for g, z, zs in pending.pop(pos, ()):
                ops.join_pending(g)                         # (a gradient its producer still writes on another lane)
                dense = g.dim() == 5 and (g.shape[4] == 1 or g.stride(4) == 1) and (g.shape[3] == 1 or g.stride(3) == g.shape[4])
                if zs is not None and not dense:
                    g = g.contiguous()
                if dcur is None:
                    if zs is not None:
                        dcur = out_grad_buffer(pos, g.shape, g)
                        ops.masked_scale_copy(g, z, zs, dcur)
                    elif tape[pos - 1][0] == "mixed":
                        dcur = out_grad_buffer(pos, g.shape, g)
                        dcur.copy_(g)
                    else:
                        dcur = g.contiguous()       # only read from here on unless another gradient is added below
                        cloned = False
                elif zs is not None:
                    ops.masked_scale_copy(g, z, zs, dcur, accumulate=True)
                else:
                    if not cloned:                  # never modify a caller's tensor in place
                        dcur, cloned = dcur.clone(), True
                    dcur.add_(g)
            if dcur is None:
                continue
            step = tape[pos - 1]
            first = pos == 1
            if step[0] == "cvt":
                # the fp32 gradient of the converted copy -> the bf16 gradient of the stored tensor (already masked / scaled
                # by its producers when that tensor is a conv / mixed output: the convention does not change at the seam)
                _, xh, _, _ = step
                dcur = ops.convert_storage(dcur.contiguous(), xh.dtype, out=out_grad_buffer(pos - 1, xh.shape, xh))
            elif step[0] == "conv":
                _, wi, k, s, xin, y, in_scale, _ = step
                slot = ops.grad_slot(weights[wi])
                in_slots = in_slots and slot is not None
                if first and not need_dx and ops.LAST_WGRAD_MAIN:
                    # the walk's LAST weight gradient (Conv3d_1a: its input needs no gradient, nothing follows on this lane) runs
                    # HERE, beside what the weight-gradient lane still holds (the Mixed_3 backlog, Conv3d_2c, 2b) -- ops.LAST_WGRAD_MAIN
                    side.issue()
                    dws[wi] = ops.conv_wgrad(xin, dcur, weights[wi].shape, k, s, out=slot)
                else:
                    dws[wi] = side.wgrad(xin, dcur, weights[wi].shape, k, s, out=slot)
                if first and not need_dx:
                    dcur = None
                else:
                    dcur = ops.conv_dgrad(dcur, weights[wi], xin.shape, k, s, out=out_grad_buffer(pos - 1, xin.shape, xin),
                                          out_mask=xin if in_scale is not None else None, out_scale=in_scale)
                side.flush()
                ops.grads_ready([(weights[wi], dws[wi])])
            elif step[0] == "pool":
                _, k, s, xin, (arg, bits, grad_half), in_scale, _ = step
                if dcur.dtype == torch.bfloat16:            # bf16 chain: dy, dx and the (sign-bit) mask alike
                    masked = in_scale is not None and bits is not None
                    dcur = ops.maxpool3d_backward(dcur, arg, xin.shape, k, s, out=out_grad_buffer(pos - 1, xin.shape, xin),
                                                  out_scale=in_scale if masked else None, out_signbits=bits if masked else None)
                elif grad_half:                             # Conv3d_1a's bf16-stored output: its gradient is stored the same way
                    dcur = ops.maxpool3d_backward(dcur, arg, xin.shape, k, s, out_scale=in_scale, out_signbits=bits, half_out=True)
                else:
                    # (a bf16-stored xin behind an fp32 gradient -- ops.HALF_ACT_DIRECT without the chain -- keeps an fp32 dx)
                    dcur = ops.maxpool3d_backward(dcur, arg, xin.shape, k, s, out=out_grad_buffer(pos - 1, xin.shape, dcur),
                                                  out_mask=xin if (in_scale is not None and bits is None) else None,
                                                  out_scale=in_scale, out_signbits=bits)
            else:
                _, w0, (c1, c2, c3), xin, h1, h2, pm, argm, Y, in_scale, (wf, o1, o13), _ = step
                Zg = zg.pop(pos)            # [dh1 | dh2 | dY]; dcur is its tail
                dY = Zg[:, o13:]
                dX = out_grad_buffer(pos - 1, xin.shape, xin)
                xm = xin if in_scale is not None else None
                sl = (slice(0, c1), slice(c1, c2), slice(c2, c3), slice(c3, Y.shape[1]))
                for a, b, hid, sli, dst in ((w0 + 1, w0 + 2, h1, sl[1], Zg[:, :o1]), (w0 + 3, w0 + 4, h2, sl[2], Zg[:, o1:o13])):
                    g = dY[:, sli]
                    slot = ops.grad_slot(weights[b])
                    in_slots = in_slots and slot is not None
                    dws[b] = side.wgrad(hid, g, weights[b].shape, THREE, ONE, out=slot)
                    ops.conv_dgrad(g, weights[b], hid.shape, THREE, ONE, out=dst, out_mask=hid, out_scale=sc(a))
                gf = Zg[:, :o13 + c1]       # gradients of the fused 1x1 outputs: dh1, dh2, dY[:, :c1]
                slot = ops.grad_slot(wf)
                in_slots = in_slots and slot is not None
                dwf = side.wgrad(xin, gf, wf.shape, ONE, ONE, out=slot)
                dws[w0 + 1], dws[w0 + 3], dws[w0] = dwf[:o1], dwf[o1:o13], dwf[o13:]
                ops.conv_dgrad(gf, wf, xin.shape, ONE, ONE, out=dX, out_mask=xm, out_scale=in_scale)
                g3 = dY[:, sl[3]]
                slot = ops.grad_slot(weights[w0 + 5])
                in_slots = in_slots and slot is not None
                dws[w0 + 5] = side.wgrad(pm, g3, weights[w0 + 5].shape, ONE, ONE, out=slot)
                dpm = ops.conv_dgrad(g3, weights[w0 + 5], pm.shape, ONE, ONE)
                ops.maxpool3d_backward(dpm, argm, xin.shape, THREE, ONE, out=dX, accumulate=True,
                                       out_mask=xm, out_scale=in_scale)
                dcur = dX
                side.flush()
                ops.grads_ready([(weights[w0 + q], dws[w0 + q]) for q in range(6)])      # this module's six weights are final
        side.node_end(in_slots)
        ctx.tape = None
        ctx.outs = None
        dx = dcur if need_dx else None
        return (dx, None, None, None, None, None) + tuple(dws)
