"""Layer primitives of the detection path on MI355X, with the reference's constructor signatures
and state-dict key names (AFSD/common/layers.py): Unit1D (:178-214), Unit3D (:106-175),
MaxPool3dSamePadding (:9-35).  The arithmetic runs in libopental_hip.so (implicit-GEMM
convolution on MFMA, fused GroupNorm+ReLU, virtual SAME padding); nn.Conv1d / nn.Conv3d /
nn.GroupNorm objects are kept only as parameter containers so checkpoints stay interchangeable.

Extra, not in the reference: `levels=` (a level table) runs a stride-1 Unit1D over a packed
(B,C,sum t_l) pyramid buffer as six independent SAME-padded convolutions in one launch, and
`ConvGNReLU` fuses the reference's nn.Sequential(Unit, GroupNorm(32,C), ReLU) blocks.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops


def _tuple3(k):
    return (k, 1, 1) if isinstance(k, int) else tuple(k)


class ConvSameFunction(Function):
    """y = conv_SAME(x, w) + b for (B,C,T) or (B,C,T,H,W); dx/dw through the MFMA GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b, k, s, spatial_valid, levels):
        ctx.cfg = (_tuple3(k), _tuple3(s), spatial_valid, levels)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return ops.conv_forward(x, w, ctx.cfg[0], ctx.cfg[1], shift=b, spatial_valid=spatial_valid, levels=levels)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, s, sv, lev = ctx.cfg
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (ops.conv_dgrad_collapse(dy, w, x.shape) if ops.is_full_collapse(x.shape, k, s, sv)
                  else ops.conv_dgrad(dy, w, x.shape, k, s, spatial_valid=sv, levels=lev))
        dw = None
        if ctx.needs_input_grad[1]:         # on the weight-gradient stream, beside the data gradient (ops.SideWgrads)
            slot, side = ops.grad_slot(w), ops.side_wgrads(x.device)
            dw = side.wgrad(x, dy, w.shape, k, s, spatial_valid=sv, levels=lev, out=slot)
            side.node_end(slot is not None)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=[0] + list(range(2, dy.dim())))
        return dx, dw, db, None, None, None, None


class ConvGNReLUFunction(Function):
    """relu(GroupNorm_32(conv_SAME(x, w) + b)); the conv output must be 1-D in space (H=W=1 after
    the convolution), which covers every Unit1D/Unit3D + GroupNorm + ReLU block of BDNet.py."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, k, s, spatial_valid, levels, groups, eps):
        k, s = _tuple3(k), _tuple3(s)
        c = ops.conv_forward(x, w, k, s, shift=b, spatial_valid=spatial_valid, levels=levels)
        c3 = c.view(c.shape[0], c.shape[1], c.shape[2]) if c.dim() == 5 else c
        y, stats = ops.gn_relu_forward(c3, gamma, beta, groups, eps, True, levels)
        ctx.cfg = (k, s, spatial_valid, levels, groups)
        ctx.save_for_backward(x, w, c3, gamma, beta, stats, *([b] if b is not None else []))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, c3, gamma, beta, stats = ctx.saved_tensors[:6]
        bias = ctx.saved_tensors[6] if len(ctx.saved_tensors) > 6 else None
        k, s, sv, lev, groups = ctx.cfg
        # dy is read in place when it is a channel slice of a wider map (the gradient of torch.cat); the three batch sums
        # are deferred to the trainer's next bucket flush when it is running (ops.gn_relu_backward)
        dc, dgamma, dbeta, dbias = ops.gn_relu_backward(dy, c3, gamma, beta, stats, groups, True, lev, bias=bias)
        dc5 = dc.view(dc.shape[0], dc.shape[1], dc.shape[2], 1, 1) if x.dim() == 5 else dc
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (ops.conv_dgrad_collapse(dc5, w, x.shape) if ops.is_full_collapse(x.shape, k, s, sv)
                  else ops.conv_dgrad(dc5, w, x.shape, k, s, spatial_valid=sv, levels=lev))
        slot, side = ops.grad_slot(w), ops.side_wgrads(x.device)
        dw = side.wgrad(x, dc5, w.shape, k, s, spatial_valid=sv, levels=lev, out=slot)
        side.node_end(slot is not None)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None, None


class ConvGNReLUPairFunction(Function):
    """Two ConvGNReLU blocks of ONE geometry -- sibling layers the model applies side by side: the loc / conf towers, the
    two ProposalBranches -- as one autograd node whose five kernels each carry both problems (pair launches,
    include/opental_hip.h): (y0, y1) = (block0(x0), block1(x1)).  Every value is what ConvGNReLUFunction gives for the block
    on its own (a workgroup never sees the other problem); where the library has no pair kernel for the geometry the two
    problems are launched one after the other."""

    @staticmethod
    def forward(ctx, x0, x1, w0, w1, b0, b1, g0, g1, be0, be1, k, s, levels, groups, eps):
        k, s = _tuple3(k), _tuple3(s)
        cs = ops.conv_forward_pair((x0, x1), (w0, w1), k, s, (b0, b1), levels)
        if cs is None:
            cs = [ops.conv_forward(x, w, k, s, shift=b, levels=levels) for x, w, b in ((x0, w0, b0), (x1, w1, b1))]
        ys = ops.gn_relu_forward_pair(cs, (g0, g1), (be0, be1), groups, eps, True, levels)
        if ys is None:
            ys = [ops.gn_relu_forward(c, g, be, groups, eps, True, levels) for c, g, be in ((cs[0], g0, be0), (cs[1], g1, be1))]
        ctx.cfg = (k, s, levels, groups)
        ctx.save_for_backward(x0, x1, w0, w1, cs[0], cs[1], g0, g1, be0, be1, ys[0][1], ys[1][1], b0, b1)
        return ys[0][0], ys[1][0]

    @staticmethod
    def backward(ctx, dy0, dy1):
        x0, x1, w0, w1, c0, c1, g0, g1, be0, be1, st0, st1, b0, b1 = ctx.saved_tensors
        k, s, lev, groups = ctx.cfg
        r = ops.gn_relu_backward_pair((dy0, dy1), (c0, c1), (g0, g1), (be0, be1), (st0, st1), groups, True, lev, (b0, b1))
        if r is None:
            r = [ops.gn_relu_backward(dy, c, g, be, st, groups, True, lev, bias=b)
                 for dy, c, g, be, st, b in ((dy0, c0, g0, be0, st0, b0), (dy1, c1, g1, be1, st1, b1))]
        (dc0, dg0, dbe0, db0), (dc1, dg1, dbe1, db1) = r
        dxs = (None, None)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dxs = ops.conv_dgrad_pair((dc0, dc1), (w0, w1), x0.shape, k, s, lev)
            if dxs is None:
                dxs = [ops.conv_dgrad(dc, w, x0.shape, k, s, levels=lev) for dc, w in ((dc0, w0), (dc1, w1))]
        slots = (ops.grad_slot(w0), ops.grad_slot(w1))
        side = ops.side_wgrads(x0.device)
        dws = side.wgrad_pair((x0, x1), (dc0, dc1), w0.shape, k, s, lev, slots)
        side.node_end(slots[0] is not None and slots[1] is not None)
        return (dxs[0], dxs[1], dws[0], dws[1], db0, db1, dg0, dg1, dbe0, dbe1, None, None, None, None, None)


def conv_gn_relu_pair(block0, block1, x0, x1, levels=None):
    """(block0(x0, levels), block1(x1, levels)) for two ConvGNReLU blocks around Unit1D layers of the same shape."""
    u0, n0, u1, n1 = block0[0], block0[1], block1[0], block1[1]
    same = (isinstance(u0, Unit1D) and isinstance(u1, Unit1D) and u0.conv1d.weight.shape == u1.conv1d.weight.shape
            and u0._kernel_shape == u1._kernel_shape and u0._stride == u1._stride == 1 and n0.num_groups == n1.num_groups
            and n0.eps == n1.eps and x0.shape == x1.shape and (u0.conv1d.bias is None) == (u1.conv1d.bias is None))
    if not same or not ops.PAIR_LAUNCHES or not x0.is_cuda:
        return block0(x0, levels), block1(x1, levels)
    return ConvGNReLUPairFunction.apply(x0, x1, u0.conv1d.weight, u1.conv1d.weight, u0.conv1d.bias, u1.conv1d.bias,
                                        n0.weight, n1.weight, n0.bias, n1.bias, u0._kernel_shape, u0._stride, levels,
                                        n0.num_groups, n0.eps)


class Unit1D(nn.Module):
    def __init__(self, in_channels, output_channels, kernel_shape=1, stride=1, padding='same',
                 activation_fn=F.relu, use_bias=True):
        super(Unit1D, self).__init__()
        if padding != 'same':
            raise NotImplementedError("only padding='same' is used on the hot path")
        self.conv1d = nn.Conv1d(in_channels, output_channels, kernel_shape, stride, padding=0, bias=use_bias)
        self._activation_fn = activation_fn
        self._padding = padding
        self._stride = stride
        self._kernel_shape = kernel_shape

    def forward(self, x, levels=None):
        x = ConvSameFunction.apply(x, self.conv1d.weight, self.conv1d.bias, self._kernel_shape, self._stride,
                                   False, levels)
        if self._activation_fn is not None:
            x = self._activation_fn(x)
        return x


class Unit3D(nn.Module):
    """Pyramid-projection Unit3D (layers.py:106-175): temporal SAME pad, spatially 'valid'."""

    def __init__(self, in_channels, output_channels, kernel_shape=(1, 1, 1), stride=(1, 1, 1),
                 padding='spatial_valid', activation_fn=F.relu, use_batch_norm=False, use_bias=False):
        super(Unit3D, self).__init__()
        if use_batch_norm:
            raise NotImplementedError("the pyramid Unit3D never carries BatchNorm (BDNet.py:129-155)")
        if padding not in ('spatial_valid', 'same'):
            raise NotImplementedError(padding)
        self._kernel_shape = tuple(kernel_shape)
        self._stride = tuple(stride)
        self._activation_fn = activation_fn
        self.padding = padding
        self.conv3d = nn.Conv3d(in_channels, output_channels, self._kernel_shape, self._stride, padding=0,
                                bias=use_bias)

    def forward(self, x):
        x = ConvSameFunction.apply(x, self.conv3d.weight, self.conv3d.bias, self._kernel_shape, self._stride,
                                   self.padding == 'spatial_valid', None)
        if self._activation_fn is not None:
            x = self._activation_fn(x)
        return x


class ConvGNReLU(nn.Sequential):
    """nn.Sequential(Unit1D | Unit3D, nn.GroupNorm(32, C), nn.ReLU) with the reference's child
    indices (so keys read `<name>.0.conv1d.weight`, `<name>.1.weight`), executed as one fused
    autograd node: conv (+bias) -> GroupNorm statistics per level -> ReLU."""

    def __init__(self, unit, channels, groups=32):
        super(ConvGNReLU, self).__init__(unit, nn.GroupNorm(groups, channels), nn.ReLU(inplace=True))

    def forward(self, x, levels=None):
        unit, gn = self[0], self[1]
        if isinstance(unit, Unit1D):
            conv, k, s, sv = unit.conv1d, unit._kernel_shape, unit._stride, False
        else:
            conv, k, s, sv = unit.conv3d, unit._kernel_shape, unit._stride, unit.padding == 'spatial_valid'
        return ConvGNReLUFunction.apply(x, conv.weight, conv.bias, gn.weight, gn.bias, k, s, sv, levels,
                                        gn.num_groups, gn.eps)


class MaxPool3dFunction(Function):
    @staticmethod
    def forward(ctx, x, k, s):
        y, arg = ops.maxpool3d_forward(x, k, s)
        ctx.cfg = (tuple(k), tuple(s), x.shape)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        k, s, shape = ctx.cfg
        return ops.maxpool3d_backward(dy.contiguous(), arg, shape, k, s), None, None


class MaxPool3dSamePadding(nn.MaxPool3d):
    """Zero-padded SAME max-pool (layers.py:9-35) without materialising the padded tensor."""

    def forward(self, x):
        return MaxPool3dFunction.apply(x, tuple(self.kernel_size), tuple(self.stride))
