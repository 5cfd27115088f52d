"""BoundaryMaxPooling on MI355X -- same Python surface as the reference's
AFSD/prop_pooling/boundary_pooling_op.py:7-32 (BoundaryMaxPoolingFunction / BoundaryMaxPooling),
backed by the C ABI otal_bmp_* (include/opental_hip.h) instead of the pybind CUDA extension.

Differences, all deliberate:
  * backward is a deterministic gather (no atomics) and, by default, mathematically correct.
    ``compat_reference_bwd=True`` reproduces the reference launcher's tscale=N addressing
    (boundary_max_pooling_kernel.cu:121) bit for bit, for gradient-parity checks.
  * segments must have the same batch as the features (the reference reads out of bounds
    otherwise); a RuntimeError is raised instead.
  * ``*_levels`` pools all pyramid levels in one launch.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib as L

# module-level switch used by the model builder for gradient-parity runs
COMPAT_REFERENCE_BWD = False


def bmp_forward(input, segments):
    L.require_device(input, segments)
    if segments.dtype != torch.float32:
        raise RuntimeError("segments must be float32")
    B, C, T = input.shape
    N = segments.shape[1]
    out = torch.empty((B, C, N), dtype=input.dtype, device=input.device)
    L.check(L.lib().otal_bmp_fwd(L.ptr(input), L.ptr(segments), L.ptr(out), B, C, T, N,
                                 segments.shape[0], L.dtype_code(input), L.stream()), "otal_bmp_fwd")
    return out


def bmp_backward(grad_output, input, segments, compat_reference_bwd=False):
    L.require_device(grad_output, input, segments)
    B, C, T = input.shape
    N = segments.shape[1]
    grad_input = torch.empty_like(input)
    L.check(L.lib().otal_bmp_bwd(L.ptr(grad_output), L.ptr(input), L.ptr(segments), L.ptr(grad_input),
                                 B, C, T, N, segments.shape[0], int(bool(compat_reference_bwd)),
                                 L.dtype_code(input), L.stream()), "otal_bmp_bwd")
    return grad_input


class BoundaryMaxPoolingFunction(Function):
    @staticmethod
    def forward(ctx, input, segments, compat_reference_bwd=None):
        ctx.compat = COMPAT_REFERENCE_BWD if compat_reference_bwd is None else compat_reference_bwd
        ctx.save_for_backward(input, segments)
        return bmp_forward(input, segments)

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_contiguous():
            grad_output = grad_output.contiguous()
        input, segments = ctx.saved_tensors
        return bmp_backward(grad_output, input, segments, ctx.compat), None, None


class BoundaryMaxPooling(nn.Module):
    def __init__(self):
        super(BoundaryMaxPooling, self).__init__()

    def forward(self, input, segments):
        return BoundaryMaxPoolingFunction.apply(input, segments)


# ----------------------------------------------------------------------------- level-batched
def bmp_forward_levels(input, segments, t_start, n_start):
    L.require_device(input, segments)
    B, C, Tt = input.shape
    assert Tt == t_start[-1] and segments.shape[1] == n_start[-1] and segments.shape[0] == B
    out = torch.empty((B, C, n_start[-1]), dtype=input.dtype, device=input.device)
    L.check(L.lib().otal_bmp_fwd_levels(L.ptr(input), L.ptr(segments), L.ptr(out), B, C, len(t_start) - 1,
                                        L.int_array(t_start), L.int_array(n_start),
                                        L.dtype_code(input), L.stream()), "otal_bmp_fwd_levels")
    return out


def bmp_backward_levels(grad_output, input, segments, t_start, n_start):
    L.require_device(grad_output, input, segments)
    B, C, _ = input.shape
    grad_input = torch.empty_like(input)
    L.check(L.lib().otal_bmp_bwd_levels(L.ptr(grad_output), L.ptr(input), L.ptr(segments), L.ptr(grad_input),
                                        B, C, len(t_start) - 1, L.int_array(t_start), L.int_array(n_start),
                                        L.dtype_code(input), L.stream()), "otal_bmp_bwd_levels")
    return grad_input


def _slice_stride(t, B, C, N):
    """Batch stride of a (B,C,N) fp32 tensor that may be a channel slice of a wider buffer (rows dense)."""
    if tuple(t.shape) != (B, C, N) or t.dtype != torch.float32 or not t.is_cuda or t.stride(2) != 1 or t.stride(1) != N:
        raise RuntimeError("BoundaryMaxPooling: the pooled tensor must be fp32 (B,C,N) with dense rows")
    return t.stride(0) if B > 1 else C * N


def bmp_forward_levels_to(input, segments, t_start, n_start, out):
    """bmp_forward_levels writing into `out`, a channel slice of a concatenation buffer."""
    import ctypes
    L.require_device(input, segments)
    B, C, Tt = input.shape
    N = n_start[-1]
    L.check(L.lib().otal_bmp_fwd_levels_to(L.ptr(input), L.ptr(segments), L.ptr(out), ctypes.c_int64(_slice_stride(out, B, C, N)),
                                           B, C, len(t_start) - 1, L.int_array(t_start), L.int_array(n_start), L.stream()),
            "otal_bmp_fwd_levels_to")
    return out


def bmp_backward_levels_from(grad_output, input, segments, t_start, n_start):
    """bmp_backward_levels reading `grad_output` in place where it is a channel slice of a wider gradient buffer."""
    import ctypes
    L.require_device(input, segments)
    B, C, _ = input.shape
    N = n_start[-1]
    grad_input = torch.empty_like(input)
    L.check(L.lib().otal_bmp_bwd_levels_from(L.ptr(grad_output), ctypes.c_int64(_slice_stride(grad_output, B, C, N)), L.ptr(input),
                                             L.ptr(segments), L.ptr(grad_input), B, C, len(t_start) - 1, L.int_array(t_start),
                                             L.int_array(n_start), L.stream()), "otal_bmp_bwd_levels_from")
    return grad_input


class BoundaryMaxPoolingLevelsFunction(Function):
    """All pyramid levels in one launch: input (B,C,sum t_l), segments (B,sum n_l,4) in
    level-local coordinates.  Equivalent to the per-level calls of BDNet.py:386-389."""

    @staticmethod
    def forward(ctx, input, segments, t_start, n_start):
        ctx.tabs = (tuple(t_start), tuple(n_start))
        ctx.save_for_backward(input, segments)
        return bmp_forward_levels(input, segments, ctx.tabs[0], ctx.tabs[1])

    @staticmethod
    def backward(ctx, grad_output):
        input, segments = ctx.saved_tensors
        return bmp_backward_levels(grad_output.contiguous(), input, segments, *ctx.tabs), None, None, None
