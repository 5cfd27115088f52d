"""INTEGRATION.md section 1 as a tested file: integration/boundary_max_pooling_cuda.py is loaded under the NAME the
reference imports (`import boundary_max_pooling_cuda`, AFSD/prop_pooling/boundary_pooling_op.py:4) and driven through the
reference's own wrapper pattern (autograd Function: save_for_backward(input, segments), backward -> (grad, None),
boundary_pooling_op.py:7-24) against the CPU oracle -- float, double and half as the reference dispatches them."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(REPO, "integration", "boundary_max_pooling_cuda.py")


def _load():
    spec = importlib.util.spec_from_file_location("boundary_max_pooling_cuda", STUB)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["boundary_max_pooling_cuda"] = mod
    spec.loader.exec_module(mod)
    return mod


def _ref64(x, seg, gy):
    """boundary_max_pooling_kernel.cu:17-82 restated in float64 loops (tiny case): trunc -> clamp -> first maximum by strict >,
    l > r -> in[l]; backward adds grad_out into grad_in[arg-max] (T-strided rows: the correct addressing)."""
    x, seg, gy = x.double().numpy(), seg.numpy(), gy.double().numpy()
    B, C, T = x.shape
    N = seg.shape[1]
    out, gin = np.zeros((B, C, N)), np.zeros((B, C, T))
    for n in range(B):
        for c in range(C):
            w = 0 if c < C // 2 else 2
            for k in range(N):
                l = min(max(0, int(seg[n, k, w])), T - 1)
                r = min(max(0, int(seg[n, k, w + 1])), T - 1)
                a, best = l, x[n, c, l]
                for i in range(l + 1, r + 1):
                    if x[n, c, i] > best:
                        a, best = i, x[n, c, i]
                out[n, c, k] = best
                gin[n, c, a] += gy[n, c, k]
    return torch.from_numpy(out), torch.from_numpy(gin)


def test_stub_loads_and_exposes_the_pybind_surface():
    try:
        mod = _load()
        assert callable(mod.forward) and callable(mod.backward)
        with pytest.raises(RuntimeError):
            mod.forward(torch.zeros(1, 2, 4), torch.zeros(1, 3, 4))         # CHECK_CUDA
    finally:
        sys.modules.pop("boundary_max_pooling_cuda", None)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16])
def test_reference_wrapper_over_the_stub_matches_the_oracle(dtype):
    from oracle import afsd_oracle as O
    try:
        _load()
        import boundary_max_pooling_cuda                                    # what the reference's module does

        class BoundaryMaxPoolingFunction(torch.autograd.Function):          # the reference's wrapper, restated
            @staticmethod
            def forward(ctx, input, segments):
                output = boundary_max_pooling_cuda.forward(input, segments)
                ctx.save_for_backward(input, segments)
                return output

            @staticmethod
            def backward(ctx, grad_output):
                if not grad_output.is_contiguous():
                    grad_output = grad_output.contiguous()
                input, segments = ctx.saved_tensors
                return boundary_max_pooling_cuda.backward(grad_output, input, segments), None

        g = torch.Generator().manual_seed(3)
        B, C, T, N = 2, 16, 37, 11
        x = torch.randn(B, C, T, generator=g).to(dtype)
        x[0, 3, 5:9] = x[0, 3, 5]                                           # a tie: the first maximum wins
        seg = torch.rand(B, N, 4, generator=g) * (T + 8) - 4                # windows reaching outside the map
        seg[0, 0] = torch.tensor([9.0, 3.0, 20.0, 12.0])                    # l > r
        gy = torch.randn(B, C, N, generator=g).to(dtype)
        want, want_g = _ref64(x, seg, gy)
        if dtype == torch.float32:      # ... which the C oracle (pinned by tests/test_oracle_bmp.py) agrees with
            assert torch.equal(O.bmp_forward(x, seg).double(), want)
            assert float((O.bmp_backward(gy, x, seg).double() - want_g).abs().max()) <= 1e-6 * float(want_g.abs().max())
        xd = x.cuda().requires_grad_(True)
        y = BoundaryMaxPoolingFunction.apply(xd, seg.cuda())
        assert y.dtype == dtype and torch.equal(y.cpu().double(), want)     # an output IS an input value: exact in any dtype
        y.backward(gy.cuda())
        got = xd.grad.cpu().double()
        if dtype == torch.float16:      # sums of half gradients: accumulated in fp32, rounded once
            assert float((got - want_g).abs().max()) <= 2e-3 * float(want_g.abs().max())
        else:
            tol = 1e-6 if dtype == torch.float32 else 1e-14
            assert float((got - want_g).abs().max()) <= tol * float(want_g.abs().max())
        assert torch.equal(got != 0, want_g != 0)
    finally:
        sys.modules.pop("boundary_max_pooling_cuda", None)
