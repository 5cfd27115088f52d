"""Pins oracle/bmp_ref.c (the restatement of boundary_max_pooling_kernel.cu:17-82) with
hand-computed known answers and a brute-force Python loop.  CPU only."""
import numpy as np
import torch

from oracle import afsd_oracle as O


def test_known_answers_forward():
    # B=1, C=2 (channel 0 = start window, channel 1 = end window), T=6
    x = torch.tensor([[[1., 5., 2., 5., 0., -1.],
                       [3., -2., 7., 7., 9., 4.]]])
    seg = torch.tensor([[[0., 2., 3., 5.],      # plain windows
                         [4., 1., 2., 2.],      # l > r on the start side -> in[l]; single element on the end side
                         [-3., 1.9, 4.2, 99.],  # clamp low, truncation 1.9 -> 1, clamp high
                         [1., 3., 2., 3.]]])    # tie: max value 5 twice / 7 twice
    out = O.bmp_forward(x, seg)
    expect = torch.tensor([[[5., 0., 5., 5.],
                            [9., 7., 9., 7.]]])
    assert torch.equal(out, expect)


def test_known_answers_backward_ties_and_clamp():
    x = torch.tensor([[[1., 5., 2., 5., 0., -1.],
                       [3., -2., 7., 7., 9., 4.]]])
    seg = torch.tensor([[[0., 2., 3., 5.], [4., 1., 2., 2.], [-3., 1.9, 4.2, 99.], [1., 3., 2., 3.]]])
    g = torch.tensor([[[1., 10., 100., 1000.], [2., 20., 200., 2000.]]])
    gin = O.bmp_backward(g, x, seg)
    # channel 0 argmax: k0 -> 1, k1 -> 4 (l>r keeps l), k2 -> 1, k3 -> 1 (tie keeps the lowest index)
    # channel 1 argmax: k0 -> 4, k1 -> 2, k2 -> 4, k3 -> 2
    expect = torch.tensor([[[0., 1101., 0., 0., 10., 0.],
                            [0., 0., 2020., 0., 202., 0.]]])
    assert torch.equal(gin, expect)


def test_nan_never_replaces_and_negative_values():
    x = torch.tensor([[[-5., float("nan"), -7., -1.], [-1., -2., -3., -4.]]])
    seg = torch.tensor([[[0., 3., 0., 3.]]])
    out = O.bmp_forward(x, seg)
    assert out[0, 0, 0] == -1.0 and out[0, 1, 0] == -1.0
    x2 = torch.tensor([[[float("nan"), 3., 4., 1.], [0., 0., 0., 0.]]])
    assert torch.isnan(O.bmp_forward(x2, seg)[0, 0, 0])  # a NaN at l sticks (val > NaN is false)


def test_c_matches_python_bruteforce():
    rs = np.random.RandomState(0)
    for (B, C, T, N) in ((2, 8, 16, 5), (1, 4, 7, 9), (3, 6, 33, 12)):
        x = torch.from_numpy(rs.randn(B, C, T).astype(np.float32))
        seg = torch.from_numpy(rs.uniform(-4, T + 4, size=(B, N, 4)).astype(np.float32))
        assert torch.equal(O.bmp_forward(x, seg), O.bmp_forward_py(x, seg))


def test_backward_is_adjoint_of_forward_selection():
    rs = np.random.RandomState(1)
    B, C, T, N = 2, 6, 20, 7
    x = torch.from_numpy(rs.randn(B, C, T).astype(np.float32))
    seg = torch.from_numpy(np.sort(rs.uniform(0, T, size=(B, N, 2, 2)), -1).reshape(B, N, 4).astype(np.float32))
    g = torch.from_numpy(rs.randn(B, C, N).astype(np.float32))
    gin = O.bmp_backward(g, x, seg)
    # distinct random values -> max is differentiable: compare with autograd over an explicit gather
    xr = x.clone().requires_grad_(True)
    outs = []
    for n in range(B):
        for c in range(C):
            w = 0 if c < C // 2 else 2
            for k in range(N):
                l, r = int(seg[n, k, w]), int(seg[n, k, w + 1])
                outs.append(xr[n, c, l:r + 1].max() * g[n, c, k])
    torch.stack(outs).sum().backward()
    assert torch.allclose(gin, xr.grad, atol=1e-6)


def test_compat_backward_reproduces_reference_stride():
    """boundary_max_pooling_kernel.cu:121 passes tscale = N: rows are addressed with stride N."""
    rs = np.random.RandomState(2)
    B, C, T, N = 1, 4, 16, 4
    x = torch.from_numpy(rs.randn(B, C, T).astype(np.float32))
    seg = torch.tensor([[[0., 3., 2., 9.], [1., 2., 3., 3.], [0., 0., 1., 2.], [2., 15., 0., 1.]]])
    g = torch.ones(B, C, N)
    gin = O.bmp_backward(g, x, seg, compat_reference_bwd=True).reshape(-1)
    xf = x.reshape(-1)
    expect = torch.zeros(B * C * T)
    for c in range(C):
        w = 0 if c < C // 2 else 2
        for k in range(N):
            l = min(max(int(seg[0, k, w]), 0), N - 1)
            r = min(max(int(seg[0, k, w + 1]), 0), N - 1)
            row = xf[c * N:(c + 1) * N]
            arg = l + int(torch.argmax(row[l:r + 1])) if r >= l else l
            expect[c * N + arg] += 1
    assert torch.equal(gin, expect)
    # and equals the correct backward when N == T
    seg2 = torch.from_numpy(rs.uniform(0, T, size=(1, T, 4)).astype(np.float32))
    g2 = torch.from_numpy(rs.randn(1, C, T).astype(np.float32))
    assert torch.equal(O.bmp_backward(g2, x, seg2, True), O.bmp_backward(g2, x, seg2, False))
