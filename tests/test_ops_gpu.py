"""GPU parity of the dense ops (through the C ABI) against the CPU oracle / torch-CPU fp32:
implicit-GEMM conv (fwd, dgrad, wgrad; SAME pads, strides, split-K, channel slices, level packing,
fused epilogues), GroupNorm+ReLU, MaxPool3dSamePadding, proposal windows (bit-exact), Adam."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F


_SWITCHES = ("OTAL_POOL_NO133", "OTAL_CONV_NO1A", "OTAL_CONV_NOW1D", "OTAL_CONV_1A_NOTILE", "OTAL_CONV_1A_WGS")


def _switch(monkeypatch, name, value):
    """Flip a kernel-selection switch of the library (otal_set_option; reset by the fixture below)."""
    from opental_amd import _lib as L
    assert name in _SWITCHES
    L.set_option(name, value)


@pytest.fixture(autouse=True)
def _switches_off_after_each_test():
    yield
    from opental_amd import _lib as L
    if L._lib is not None:
        for name in _SWITCHES:
            L.set_option(name, 0)


from oracle import afsd_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)   # north_star: 1e-4 fp32


def close(a, b, scale=None, tol=1e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    s = float(b.abs().max()) if scale is None else scale
    err = float((a - b).abs().max())
    assert err <= tol * max(s, 1e-6), f"max err {err:.3e} vs scale {s:.3e}"


def ref_conv(x, w, b, k, s, spatial_valid=False):
    if x.dim() == 3:
        return O.unit1d(x, w, b, s if isinstance(s, int) else s[0])
    pads = []
    for d in (2, 1, 0):
        f, bk = (0, 0) if (spatial_valid and d > 0) else O.same_pad(x.shape[2 + d], k[d], s[d])
        pads += [f, bk]
    return F.conv3d(F.pad(x, pads), w, b, stride=s)


CONV_CASES = [
    # (x shape, Cout, k, s, spatial_valid)
    ((1, 3, 16, 24, 24), 64, (7, 7, 7), (2, 2, 2), False),
    ((1, 3, 9, 13, 13), 64, (7, 7, 7), (2, 2, 2), False),
    ((1, 64, 6, 12, 12), 192, (3, 3, 3), (1, 1, 1), False),
    ((2, 192, 4, 6, 6), 16, (1, 1, 1), (1, 1, 1), False),
    ((1, 96, 4, 6, 6), 208, (3, 3, 3), (1, 1, 1), False),
    ((2, 480, 4, 3, 3), 112, (1, 1, 1), (1, 1, 1), False),
    ((2, 832, 8, 6, 6), 512, (1, 6, 6), (1, 1, 1), True),
    ((2, 1024, 4, 3, 3), 512, (1, 3, 3), (1, 1, 1), True),
    ((2, 512, 64), 512, 3, 1, False),
    ((2, 512, 32), 512, 3, 2, False),
    ((1, 512, 2), 512, 3, 2, False),
    ((2, 512, 64), 1024, 1, 1, False),
    ((1, 2048, 64), 512, 1, 1, False),
    ((2, 512, 64), 2, 3, 1, False),
    ((2, 512, 16), 15, 3, 1, False),
    ((8, 512, 256), 1, 3, 1, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_backward(case):
    from opental_amd.common.layers import ConvSameFunction
    shape, cout, k, s, sv = case
    rs = np.random.RandomState(abs(hash(str(case))) % 10000)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32))
    kk = (k,) if isinstance(k, int) else k
    w = torch.from_numpy((rs.randn(cout, shape[1], *kk) / np.sqrt(shape[1] * np.prod(kk))).astype(np.float32))
    b = torch.from_numpy(rs.randn(cout).astype(np.float32))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, br, k, s, sv)
    dy = torch.from_numpy(rs.randn(*yr.shape).astype(np.float32))
    yr.backward(dy)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ConvSameFunction.apply(xd, wd, bd, k, s, sv, None)
    assert y.shape == yr.shape
    y.backward(dy.cuda())
    close(y, yr)
    close(xd.grad, xr.grad)
    close(wd.grad, wr.grad)
    close(bd.grad, br.grad)


def test_conv_epilogue_mask_slices_accumulate():
    from opental_amd.common import ops
    rs = np.random.RandomState(4)
    B, Cin, Cout, T, H, W = 2, 24, 40, 4, 6, 6
    x = torch.from_numpy(rs.randn(B, Cin, T, H, W).astype(np.float32))
    w = torch.from_numpy((rs.randn(Cout, Cin, 3, 3, 3) / 25).astype(np.float32))
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
    sh = torch.from_numpy(rs.uniform(-0.3, 0.3, Cout).astype(np.float32))
    ref = F.relu(F.conv3d(F.pad(x, [1] * 6), w) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
    # input and output are channel slices of wider buffers
    xb = torch.zeros(B, Cin + 5, T, H, W, device="cuda")
    xb[:, 3:3 + Cin] = x.cuda()
    yb = torch.full((B, Cout + 7, T, H, W), 9.0, device="cuda")
    ops.conv_forward(xb[:, 3:3 + Cin], w.cuda(), (3, 3, 3), (1, 1, 1), scale=sc.cuda(), shift=sh.cuda(), relu=True,
                     out=yb[:, 2:2 + Cout])
    close(yb[:, 2:2 + Cout], ref)
    assert bool((yb[:, :2] == 9).all()) and bool((yb[:, 2 + Cout:] == 9).all())
    # backward on a pre-masked gradient (dz = dy * (y > 0) * scale), accumulating into an existing dx slice
    dy = torch.from_numpy(rs.randn(*ref.shape).astype(np.float32))
    dz = dy * (ref > 0) * sc.view(1, -1, 1, 1, 1)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv3d(F.pad(xr, [1] * 6), wr).backward(dz)
    dzb = torch.zeros_like(yb)
    dzb[:, 2:2 + Cout] = dz.cuda()
    dxb = torch.full((B, Cin + 5, T, H, W), 1.0, device="cuda")
    ops.conv_dgrad(dzb[:, 2:2 + Cout], w.cuda(), (B, Cin, T, H, W), (3, 3, 3), (1, 1, 1), out=dxb[:, 3:3 + Cin],
                   accumulate=True)
    close(dxb[:, 3:3 + Cin] - 1.0, xr.grad)
    assert bool((dxb[:, :3] == 1).all())
    dw = ops.conv_wgrad(xb[:, 3:3 + Cin], dzb[:, 2:2 + Cout], w.shape, (3, 3, 3), (1, 1, 1))
    close(dw, wr.grad)
    dw2 = ops.conv_wgrad(xb[:, 3:3 + Cin], dzb[:, 2:2 + Cout], w.shape, (3, 3, 3), (1, 1, 1), out=dw.clone(),
                         accumulate=True)
    close(dw2, 2 * wr.grad)
    # producer-side masking in the store epilogue: dx * (x > 0) * in_scale[ci], accumulated
    isc = torch.from_numpy(rs.uniform(0.5, 1.5, Cin).astype(np.float32))
    dxb2 = torch.full((B, Cin + 5, T, H, W), 1.0, device="cuda")
    ops.conv_dgrad(dzb[:, 2:2 + Cout], w.cuda(), (B, Cin, T, H, W), (3, 3, 3), (1, 1, 1), out=dxb2[:, 3:3 + Cin],
                   accumulate=True, out_mask=xb[:, 3:3 + Cin], out_scale=isc.cuda())
    close(dxb2[:, 3:3 + Cin] - 1.0, xr.grad * (x > 0) * isc.view(1, -1, 1, 1, 1))
    from oracle import afsd_oracle as O
    xp = x.clone().requires_grad_(True)
    yp = O.maxpool3d_same(xp, (3, 3, 3), (1, 1, 1))
    gp = torch.from_numpy(rs.randn(*yp.shape).astype(np.float32))
    yp.backward(gp)
    yq, arg = ops.maxpool3d_forward(x.cuda(), (3, 3, 3), (1, 1, 1))
    dq = ops.maxpool3d_backward(gp.cuda(), arg, x.shape, (3, 3, 3), (1, 1, 1), out_mask=x.cuda(), out_scale=isc.cuda())
    close(dq, xp.grad * (x > 0) * isc.view(1, -1, 1, 1, 1))


def test_wgrad_long_k_split():
    """weight gradient with a long reduction (split-K slabs) -- Conv3d_2c-like."""
    from opental_amd.common import ops
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.randn(1, 16, 16, 24, 24).astype(np.float32))
    w = torch.zeros(24, 16, 3, 3, 3)
    dy = torch.from_numpy(rs.randn(1, 24, 16, 24, 24).astype(np.float32))
    wr = w.clone().requires_grad_(True)
    F.conv3d(F.pad(x, [1] * 6), wr).backward(dy)
    dw = ops.conv_wgrad(x.cuda(), dy.cuda(), w.shape, (3, 3, 3), (1, 1, 1))
    close(dw, wr.grad)


LEV = [0, 64, 96, 112, 120, 124, 126]


def test_level_packed_conv_gn_block():
    from opental_amd.common.layers import ConvGNReLUFunction
    rs = np.random.RandomState(6)
    B, C = 2, 512
    x = torch.from_numpy(rs.randn(B, C, 126).astype(np.float32))
    w = torch.from_numpy((rs.randn(C, C, 3) / 40).astype(np.float32))
    b = torch.from_numpy(rs.randn(C).astype(np.float32) * 0.1)
    ga = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32))
    be = torch.from_numpy(rs.uniform(-0.2, 0.2, C).astype(np.float32))
    leaves = [t.clone().requires_grad_(True) for t in (x, w, b, ga, be)]
    outs = []
    for i in range(6):
        seg = leaves[0][:, :, LEV[i]:LEV[i + 1]]
        outs.append(O.gn_relu(O.unit1d(seg, leaves[1], leaves[2]), leaves[3], leaves[4]))
    yr = torch.cat(outs, 2)
    dy = torch.from_numpy(rs.randn(*yr.shape).astype(np.float32))
    yr.backward(dy)
    dl = [t.cuda().requires_grad_(True) for t in (x, w, b, ga, be)]
    y = ConvGNReLUFunction.apply(dl[0], dl[1], dl[2], dl[3], dl[4], 3, 1, False, tuple(LEV), 32, 1e-5)
    y.backward(dy.cuda())
    close(y, yr)
    for got, ref in zip(dl, leaves):
        close(got.grad, ref.grad)


@pytest.mark.parametrize("shape", [(2, 512, 64), (1, 1024, 32), (2, 512, 256), (3, 64, 2)])
def test_groupnorm_relu(shape):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape))
    x = torch.from_numpy((rs.randn(*shape) * 2 + 0.5).astype(np.float32))
    ga = torch.from_numpy(rs.uniform(0.5, 1.5, shape[1]).astype(np.float32))
    be = torch.from_numpy(rs.uniform(-0.5, 0.5, shape[1]).astype(np.float32))
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, ga, be))
    yr = O.gn_relu(xr, gr, br)
    dy = torch.from_numpy(rs.randn(*shape).astype(np.float32))
    yr.backward(dy)
    y, stats = ops.gn_relu_forward(x.cuda(), ga.cuda(), be.cuda())
    close(y, yr)
    dx, dg, db, dbias = ops.gn_relu_backward(dy.cuda(), x.cuda(), ga.cuda(), be.cuda(), stats)
    close(dx, xr.grad)
    close(dg, gr.grad)
    close(db, br.grad)
    close(dbias, xr.grad.sum((0, 2)), scale=float(xr.grad.abs().sum((0, 2)).max()))


POOLS = [((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (1, 1, 1)), ((3, 3, 3), (2, 2, 2)), ((2, 2, 2), (2, 2, 2))]


@pytest.mark.parametrize("ks", POOLS)
@pytest.mark.parametrize("relu_input", [False, True])
def test_maxpool3d_same(ks, relu_input):
    from opental_amd.common.layers import MaxPool3dFunction
    k, s = ks
    rs = np.random.RandomState(11)
    for shape in ((2, 5, 8, 12, 12), (1, 3, 7, 9, 11), (1, 4, 4, 6, 6)):
        x = torch.from_numpy(rs.randn(*shape).astype(np.float32))
        if relu_input:
            x = x.clamp(min=0) - 0.0   # exact zeros tie with the zero padding and with each other
        xr = x.clone().requires_grad_(True)
        yr = O.maxpool3d_same(xr, k, s)
        dy = torch.from_numpy(rs.randn(*yr.shape).astype(np.float32))
        yr.backward(dy)
        xd = x.cuda().requires_grad_(True)
        y = MaxPool3dFunction.apply(xd, k, s)
        y.backward(dy.cuda())
        assert torch.equal(y.cpu(), yr.detach())
        if relu_input:   # gradient routed to a zero input is killed by the preceding ReLU anyway
            m = (x > 0)
            assert torch.allclose(xd.grad.cpu()[m], xr.grad[m], rtol=1e-6, atol=1e-6)   # sums differ by fp32 add order only
        else:
            assert torch.allclose(xd.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-6)


def test_proposal_windows_bit_exact():
    from opental_amd.common import ops
    rs = np.random.RandomState(3)
    for lens, frames in (([64, 32, 16, 8, 4, 2], 256.0), ([96, 48, 24, 12, 6, 3], 768.0)):
        lev = [0]
        for t in lens:
            lev.append(lev[-1] + t)
        for B in (1, 3):
            loc = torch.from_numpy(np.exp(rs.uniform(-1, 5.5, size=(B, lev[-1], 2))).astype(np.float32))
            seg, fseg = ops.proposal_windows(loc.cuda(), lev, frames)
            for i, t in enumerate(lens):
                s_ref, f_ref = O.proposal_windows(loc[:, lev[i]:lev[i + 1]], t, frames)
                assert torch.equal(seg[:, lev[i]:lev[i + 1]].cpu(), s_ref), (lens, i)
                assert torch.equal(fseg[:, lev[i]:lev[i + 1]].cpu(), f_ref), (lens, i)


def test_adam_flat_matches_torch():
    from opental_amd.common import ops
    rs = np.random.RandomState(9)
    n = 100003
    p0 = torch.from_numpy(rs.randn(n).astype(np.float32))
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=1e-3)
    p = p0.cuda(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.from_numpy(rs.randn(n).astype(np.float32))
        pr.grad = g.clone()
        opt.step()
        ops.adam_flat(p, g.cuda(), m, v, step, 1e-3, weight_decay=1e-3)
    close(p, pr, tol=1e-6)
    mo, vo = torch.zeros(n), torch.zeros(n)
    pp = p0.clone()
    O.adam_step(pp, g, mo, vo, 1, 1e-3, 1e-3)   # the oracle's own Adam agrees with torch for one step


def _bf16_round(t):
    return t.bfloat16().float()


BF16_CASES = [((1, 3, 16, 24, 24), 64, (7, 7, 7), (2, 2, 2), False), ((1, 64, 6, 12, 12), 192, (3, 3, 3), (1, 1, 1), False),
              ((2, 192, 4, 6, 6), 16, (1, 1, 1), (1, 1, 1), False), ((1, 96, 4, 6, 6), 208, (3, 3, 3), (1, 1, 1), False),
              ((2, 832, 8, 6, 6), 512, (1, 6, 6), (1, 1, 1), True), ((2, 512, 64), 512, 3, 1, False),
              ((2, 512, 32), 512, 3, 2, False), ((2, 512, 64), 15, 3, 1, False), ((1, 2048, 64), 512, 1, 1, False),
              # vector-gather paths: 4 positions per load (W % 4 == 0) and 2 (W % 2 == 0), batch > 1, 1x1x1 and 3x1x1 taps
              ((2, 16, 3, 24, 24), 40, (3, 3, 3), (1, 1, 1), False), ((2, 24, 5, 6, 6), 64, (3, 3, 3), (1, 1, 1), False),
              ((2, 64, 4, 12, 12), 96, (1, 1, 1), (1, 1, 1), False), ((1, 32, 6, 8, 8), 48, (3, 1, 1), (1, 1, 1), False),
              # 1x1x1 weight gradients on 8-position vectors that cross row ends (W = 6 and W = 3 planes)
              ((2, 64, 8, 6, 6), 80, (1, 1, 1), (1, 1, 1), False), ((2, 96, 32, 3, 3), 64, (1, 1, 1), (1, 1, 1), False),
              # direct 3x3x3 kernel (P % 256 == 0, channels % 16 == 0): W = 8 / 12 / 24 / 6, BM 96 / 64 / padded rows, batch 2
              ((2, 16, 4, 8, 8), 96, (3, 3, 3), (1, 1, 1), False), ((1, 32, 16, 12, 12), 64, (3, 3, 3), (1, 1, 1), False),
              ((1, 64, 4, 24, 24), 192, (3, 3, 3), (1, 1, 1), False), ((2, 48, 64, 6, 6), 112, (3, 3, 3), (1, 1, 1), False)]


@pytest.mark.parametrize("case", BF16_CASES)
def test_conv_bf16_operands_fp32_accumulate(case):
    """precision=1: operands are rounded to bf16 (RNE) when staged, products are exact in fp32 and
    accumulated in fp32 -- so the result must equal an fp32 convolution of the bf16-ROUNDED operands
    up to summation order (1e-4 of scale; stated tolerance of the bf16 path)."""
    from opental_amd.common import ops
    shape, cout, k, s, sv = case
    rs = np.random.RandomState(abs(hash(str(case))) % 10000)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32))
    kk = (k,) if isinstance(k, int) else k
    w = torch.from_numpy((rs.randn(cout, shape[1], *kk) / np.sqrt(shape[1] * np.prod(kk))).astype(np.float32))
    xq, wq = _bf16_round(x), _bf16_round(w)
    xr, wr = xq.clone().requires_grad_(True), wq.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, None, k, s, sv)
    dy = torch.from_numpy(rs.randn(*yr.shape).astype(np.float32))
    dyq = _bf16_round(dy)
    yr.backward(dyq)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        y = ops.conv_forward(x.cuda(), w.cuda(), k, s, spatial_valid=sv)
        dx = ops.conv_dgrad(dy.cuda(), w.cuda(), x.shape, k, s, spatial_valid=sv)
        dw = ops.conv_wgrad(x.cuda(), dy.cuda(), w.shape, k, s, spatial_valid=sv)
    finally:
        ops.CONV_PRECISION = old
    close(y, yr)
    close(dx, xr.grad)
    close(dw, wr.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,k,s", [((1, 8, 2, 8, 8), 16, (3, 3, 3), (1, 1, 1)), ((1, 8, 2, 4, 4), 16, (3, 3, 3), (1, 1, 1)),
                                            ((1, 8, 2, 6, 6), 16, (3, 3, 3), (1, 1, 1)), ((1, 3, 4, 16, 16), 8, (7, 7, 7), (2, 2, 2)),
                                            ((1, 16, 4, 8, 8), 64, (3, 3, 3), (1, 1, 1)),
                                            ((2, 3, 8, 32, 32), 64, (7, 7, 7), (2, 2, 2))])
def test_conv_bf16_vector_starting_before_tensor(shape, cout, k, s):
    """Vector gathers whose first element lies in FRONT of the tensor (channel 0, first row, tap shifted left) are
    rejected as a whole by the buffer bounds check; the kernels re-fetch the remaining elements.  Large values in
    the first row make a dropped element flagrant (fwd, dgrad through the same gather on dy, wgrad)."""
    from opental_amd.common import ops
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32))
    x[0, 0, 0, 0, :] = 64.0
    w = _bf16_round(torch.from_numpy((rs.randn(cout, shape[1], *k) * 0.25).astype(np.float32)))
    xq = _bf16_round(x)
    xr, wr = xq.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, None, k, s, False)
    dy = _bf16_round(torch.from_numpy(rs.randn(*yr.shape).astype(np.float32)))
    dy[0, 0, 0, 0, :] = 64.0
    yr.backward(dy)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        y = ops.conv_forward(xq.cuda(), w.cuda(), k, s)
        dx = ops.conv_dgrad(dy.cuda(), w.cuda(), x.shape, k, s)
        dw = ops.conv_wgrad(xq.cuda(), dy.cuda(), w.shape, k, s)
    finally:
        ops.CONV_PRECISION = old
    close(y, yr)
    close(dx, xr.grad)
    close(dw, wr.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
def test_fused_1x1_launches_match_separate_ones(prec):
    """The backbone runs the three 1x1 convolutions that read a module's input (b1a, b2a, b0) as ONE launch over a shared
    buffer [h1 | h2 | Y] (common/i3d_backbone.py).  Weight gradient and data gradient of the fused launch against
    separate launches and against fp64: exact in fp32 (1e-6), bit-identical in bf16 mode."""
    from opental_amd.common import ops
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = prec
    try:
        torch.manual_seed(3)
        B, Cin, T, H, W = 1, 48, 16, 12, 12
        o1, o3, c0, ctot = 32, 16, 24, 96
        x = torch.randn(B, Cin, T, H, W, device="cuda")
        Zg = torch.randn(B, o1 + o3 + ctot, T, H, W, device="cuda") * 0.1
        gf = Zg[:, :o1 + o3 + c0]
        wshape = (o1 + o3 + c0, Cin, 1, 1, 1)
        dwf = ops.conv_wgrad(x, gf, wshape, (1, 1, 1), (1, 1, 1))
        sep = torch.cat([ops.conv_wgrad(x, Zg[:, lo:hi].contiguous(), (hi - lo, Cin, 1, 1, 1), (1, 1, 1), (1, 1, 1))
                         for lo, hi in ((0, o1), (o1, o1 + o3), (o1 + o3, o1 + o3 + c0))], 0)
        ref = torch.einsum("bcthw,bkthw->kc", x.double(), gf.double()).view(wshape)
        tol = 2e-6 if prec == 0 else 1e-2
        assert float((dwf.double() - ref).abs().max()) <= tol * float(ref.abs().max())
        assert float((dwf - sep).abs().max()) <= 2e-6 * float(ref.abs().max())
        wf = torch.randn(*wshape, device="cuda") * 0.05
        dx = ops.conv_dgrad(gf, wf, x.shape, (1, 1, 1), (1, 1, 1))
        dref = torch.einsum("bkthw,kc->bcthw", gf.double(), wf.double().view(wshape[0], Cin))
        assert float((dx.double() - dref).abs().max()) <= tol * float(dref.abs().max())
        # forward into the shared buffer: the fused output range ends inside Y
        Z = torch.zeros(B, o1 + o3 + ctot, T, H, W, device="cuda")
        ops.conv_forward(x, wf, (1, 1, 1), (1, 1, 1), out=Z[:, :o1 + o3 + c0])
        yref = torch.einsum("bcthw,kc->bkthw", x.double(), wf.double().view(wshape[0], Cin))
        assert float((Z[:, :o1 + o3 + c0].double() - yref).abs().max()) <= tol * float(yref.abs().max())
        assert float(Z[:, o1 + o3 + c0:].abs().max()) == 0.0
    finally:
        ops.CONV_PRECISION = old


@pytest.mark.parametrize("k,s", [((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (2, 2, 2))])
@pytest.mark.parametrize("shape", [(2, 3, 6, 48, 48), (1, 2, 4, 24, 24), (1, 5, 8, 12, 12), (1, 1, 2, 2, 4)])
def test_strided_pool_fast_path_equals_generic_kernels(shape, k, s, monkeypatch):
    """MaxPool3d_2a / 3a (kernel (1,3,3), stride (1,2,2)) and 4a ((3,3,3) / (2,2,2)): the float4 kernels against the
    generic ones -- outputs and winner taps bit for bit, the fused mask / scale / accumulate backward within fp32 add
    order; dx may be a channel slice of a larger buffer."""
    from opental_amd.common import ops
    rs = np.random.RandomState(21)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).clamp(min=0).cuda()        # zeros tie with the padding
    x[0, 0, 0] = -1.0                                                                   # a plane where the zero padding wins
    y1, a1 = ops.maxpool3d_forward(x, k, s)
    _switch(monkeypatch, "OTAL_POOL_NO133", 1)
    y0, a0 = ops.maxpool3d_forward(x, k, s)
    _switch(monkeypatch, "OTAL_POOL_NO133", 0)
    assert torch.equal(y0, y1) and torch.equal(a0, a1)
    dy = torch.from_numpy(rs.randn(*y0.shape).astype(np.float32)).cuda()
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, shape[1]).astype(np.float32)).cuda()
    big = torch.from_numpy(rs.randn(shape[0], shape[1] + 4, *shape[2:]).astype(np.float32)).cuda()
    outs = []
    for generic in (True, False):
        if generic:
            _switch(monkeypatch, "OTAL_POOL_NO133", 1)
        else:
            _switch(monkeypatch, "OTAL_POOL_NO133", 0)
        buf = big.clone()
        xm = torch.zeros_like(big)
        xm[:, 4:] = x                                   # the mask source shares dx's (sliced) layout, as in the backbone
        ops.maxpool3d_backward(dy, a0, x.shape, k, s, out=buf[:, 4:], accumulate=True, out_mask=xm[:, 4:], out_scale=sc)
        plain = ops.maxpool3d_backward(dy, a0, x.shape, k, s)
        outs.append((buf, plain))
    assert torch.equal(outs[0][0][:, :4], outs[1][0][:, :4])
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-6)
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-6, atol=1e-6)
    # the ReLU mask carried as sign bits by the forward kernel instead of re-read from the activations: identical dx
    y2, a2, bits = ops.maxpool3d_forward(x, k, s, signbits=True)
    assert torch.equal(y2, y0) and torch.equal(a2, a0)
    if shape[-1] % 4 == 0:
        assert bits is not None and bits.numel() == x.numel() // 8
        d_bits = ops.maxpool3d_backward(dy, a0, x.shape, k, s, out_scale=sc, out_signbits=bits)
        d_mask = ops.maxpool3d_backward(dy, a0, x.shape, k, s, out_mask=x, out_scale=sc)
        assert torch.equal(d_bits, d_mask)


@pytest.mark.parametrize("shape,cout", [((2, 3, 8, 96, 96), 64), ((1, 3, 12, 20, 96), 64), ((1, 3, 4, 8, 96), 96),
                                        ((1, 3, 16, 24, 96), 64), ((2, 3, 8, 8, 96), 96)])
def test_conv1a_direct_kernel_matches_the_gather_kernel(shape, cout, monkeypatch):
    """Conv3d_1a_7x7 (7x7x7, stride 2, 3 channels, 96-wide planes): the LDS-patch kernel against the kw-vector gather
    kernel (same bf16-rounded operands, fp32 accumulation in a different order) and against torch on the rounded
    operands; scale / shift / ReLU epilogue included."""
    from opental_amd.common import ops
    rs = np.random.RandomState(41)
    k, s = (7, 7, 7), (2, 2, 2)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(cout, 3, 7, 7, 7) / 30).astype(np.float32)).cuda()
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32)).cuda()
    sh = torch.from_numpy(rs.uniform(-0.3, 0.3, cout).astype(np.float32)).cuda()
    monkeypatch.setattr(ops, "CONV_PRECISION", 1)
    y1 = ops.conv_forward(x, w, k, s, scale=sc, shift=sh, relu=True)
    _switch(monkeypatch, "OTAL_CONV_NO1A", 1)
    y0 = ops.conv_forward(x, w, k, s, scale=sc, shift=sh, relu=True)
    _switch(monkeypatch, "OTAL_CONV_NO1A", 0)
    assert float((y1 - y0).abs().max()) <= 2e-5 * float(y0.abs().max())
    xr, wr = x.to(torch.bfloat16).float().cpu(), w.to(torch.bfloat16).float().cpu()
    ref = F.conv3d(F.pad(xr, [2, 3, 2, 3, 2, 3]), wr, stride=2) * sc.cpu().view(1, -1, 1, 1, 1) + sh.cpu().view(1, -1, 1, 1, 1)
    close(y1, ref.clamp(min=0))


@pytest.mark.parametrize("shape,cout", [((2, 3, 8, 96, 96), 64), ((1, 3, 16, 24, 96), 64), ((2, 3, 8, 8, 96), 96), ((1, 3, 24, 16, 96), 40)])
def test_conv1a_tile_kernel_matches_the_2x2_tile_kernel_bit_for_bit(shape, cout, monkeypatch):
    """conv1a_tile_fwd_kernel (csrc/conv1a_tile.hip: 4 x 4 x 48 output tiles, planes staged while the K loop runs) against
    conv1a_direct_fwd_kernel (2 x 2 x 48 tiles): same operand roundings and the same accumulation order per output, so
    fp32 and bf16-stored outputs are equal bit for bit; the bf16 output is the rounding of the fp32 one."""
    from opental_amd.common import ops
    rs = np.random.RandomState(43)
    k, s = (7, 7, 7), (2, 2, 2)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(cout, 3, 7, 7, 7) / 30).astype(np.float32)).cuda()
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32)).cuda()
    sh = torch.from_numpy(rs.uniform(-0.3, 0.3, cout).astype(np.float32)).cuda()
    monkeypatch.setattr(ops, "CONV_PRECISION", 1)
    got = [ops.conv_forward(x, w, k, s, scale=sc, shift=sh, relu=relu, half_out=h) for relu in (True, False) for h in (False, True)]
    _switch(monkeypatch, "OTAL_CONV_1A_NOTILE", 1)
    want = [ops.conv_forward(x, w, k, s, scale=sc, shift=sh, relu=relu, half_out=h) for relu in (True, False) for h in (False, True)]
    _switch(monkeypatch, "OTAL_CONV_1A_NOTILE", 0)
    for g, t in zip(got, want):
        assert g.dtype == t.dtype and torch.equal(g, t)
    # few persistent workgroups: every workgroup walks SEVERAL tiles (the next tile's planes prefetched under the epilogue,
    # the patch re-zeroed and re-filled, the last range shorter than the others)
    for wgs in (1, 4):
        _switch(monkeypatch, "OTAL_CONV_1A_WGS", wgs)
        again = [ops.conv_forward(x, w, k, s, scale=sc, shift=sh, relu=relu, half_out=h) for relu in (True, False) for h in (False, True)]
        for g, t in zip(again, want):
            assert torch.equal(g, t)
    _switch(monkeypatch, "OTAL_CONV_1A_WGS", 0)
    assert torch.equal(got[1], got[0].to(torch.bfloat16)) and torch.equal(got[3], got[2].to(torch.bfloat16))
    assert float(got[2].abs().max()) > 0.1 and bool((got[2] < 0).any())


@pytest.mark.parametrize("lev,strides,K", [((0, 64, 96, 112, 120, 124, 126), None, 15), ((0, 96, 144, 168, 180, 186, 189), (4, 8, 16, 32, 64, 128), 150)])
def test_head_output_tails_match_the_torch_formulation(lev, strides, K):
    """ScaleExp (per level, x fpn stride), permute(0,2,1).contiguous() and Dirichlet uncertainty of the head maps in one
    launch against the reference's op sequence (BDNet.py:337-353,:538-556), forward and backward incl. d scale."""
    from opental_amd.common import ops
    rs = np.random.RandomState(51)
    B, N = 3, lev[-1]
    mk = lambda c, sd: torch.from_numpy((rs.randn(B, c, N) * sd).astype(np.float32))
    raws = [mk(2, 1.0), mk(K, 4.0), mk(1, 1.0)]                      # logits beyond +-10 exercise the clamp
    scales = torch.from_numpy(rs.uniform(0.8, 1.2, len(lev) - 1).astype(np.float32))
    # reference formulation
    rl = [r.clone().requires_grad_(True) for r in raws]
    sl = scales.clone().requires_grad_(True)
    cols = torch.cat([sl[i].expand(lev[i + 1] - lev[i]) for i in range(len(lev) - 1)])
    loc = torch.exp(rl[0] * cols).permute(0, 2, 1).contiguous()
    if strides is not None:
        loc = loc * torch.cat([torch.full((lev[i + 1] - lev[i],), float(s)) for i, s in enumerate(strides)]).view(1, -1, 1)
    conf = rl[1].permute(0, 2, 1).contiguous()
    act = rl[2].permute(0, 2, 1).contiguous()
    unct = K / (torch.exp(torch.clamp(conf, -10, 10)) + 1).sum(-1)
    gs = [torch.from_numpy(rs.randn(*t.shape).astype(np.float32)) for t in (loc, conf, act, unct)]
    (loc * gs[0]).sum().add((conf * gs[1]).sum()).add((act * gs[2]).sum()).add((unct * gs[3]).sum()).backward()
    # HIP
    rd = [r.clone().cuda().requires_grad_(True) for r in raws]
    sds = [scales[i:i + 1].clone().cuda().requires_grad_(True) for i in range(len(lev) - 1)]
    o = ops.HeadOutputsFunction.apply(tuple(lev), strides, (1, 2, 0), *sds, *rd)
    for got, want in zip(o, (loc, conf, act, unct)):
        close(got, want.detach(), tol=2e-6)
    sum((a * g.cuda()).sum() for a, g in zip(o, gs)).backward()
    for a, b_ in zip(rd, rl):
        close(a.grad, b_.grad, tol=2e-5)
    close(torch.cat([t.grad for t in sds]), sl.grad, tol=2e-5)
    assert all(t.grad.data_ptr() % 16 == 0 for t in sds)          # aligned: keeps multi-tensor gradient copies vectorised
    # gradients that never arrive (uncertainty is not part of the training loss) are zeros, not garbage
    rd2 = [r.clone().cuda().requires_grad_(True) for r in raws]
    o2 = ops.HeadOutputsFunction.apply(tuple(lev), strides, (1, 2, 0), *[t.detach() for t in sds], *rd2)
    o2[1].sum().backward()
    assert float(rd2[0].grad.abs().max()) == 0.0 and float(rd2[2].grad.abs().max()) == 0.0
    assert torch.equal(rd2[1].grad, torch.ones_like(rd2[1].grad))


@pytest.mark.parametrize("B,C,T,Tm,row0,step", [(2, 512, 256, 256, 0, 1), (3, 1024, 64, 256, 0, 4), (2, 1024, 96, 768, 1, 8)])
def test_boundary_bce_matches_calc_bce_loss(B, C, T, Tm, row0, step):
    """tanh + channel mean + BCE of both halves of a boundary map in one launch, read in place from a T-slice of a wider
    map, against calc_bce_loss (train.py:152-161) on the permuted copies; gradients included."""
    from opental_amd.common import ops
    rs = np.random.RandomState(61)
    wide = torch.from_numpy(np.abs(rs.randn(B, C, T + 30)).astype(np.float32) * 0.6)      # post-ReLU features
    mask = torch.from_numpy((rs.rand(B, row0 + 2, Tm) < 0.2).astype(np.float32))
    xr = wide.clone().requires_grad_(True)
    xs = xr[:, :, :T]
    half = C // 2
    st = torch.tanh(xs[:, :half].permute(0, 2, 1).contiguous()).mean(-1)
    en = torch.tanh(xs[:, half:].permute(0, 2, 1).contiguous()).mean(-1)
    ms = mask[:, :, ::step][:, :, :T]
    ls = F.binary_cross_entropy(st.reshape(-1), ms[:, row0].contiguous().view(-1))
    le = F.binary_cross_entropy(en.reshape(-1), ms[:, row0 + 1].contiguous().view(-1))
    (ls * 0.7 + le * 1.3).backward()
    xd = wide.clone().cuda().requires_grad_(True)
    gs, ge = ops.BoundaryBCEFunction.apply(xd[:, :, :T], mask.cuda(), row0, step)
    close(gs, ls.detach(), tol=1e-5)
    close(ge, le.detach(), tol=1e-5)
    (gs * 0.7 + ge * 1.3).backward()
    close(xd.grad, xr.grad, tol=2e-5)
    assert float(xd.grad[:, :, T:].abs().max()) == 0.0


@pytest.mark.parametrize("B,Cin,Cout,T,k,lev", [(2, 64, 64, 126, 3, LEV), (3, 128, 15, 126, 3, LEV), (2, 192, 96, 126, 1, None),
                                                  (2, 64, 128, 256, 3, None), (1, 64, 2, 189, 3, [0, 96, 144, 168, 180, 186, 189]),
                                                  (2, 64, 64, 130, 3, None)])
def test_wgrad_1d_kernel_matches_torch(B, Cin, Cout, T, k, lev, monkeypatch):
    """Weight gradient of the 1-D pyramid / head layers (H = W = 1): the K-contiguous LDS kernel against torch on the
    bf16-rounded operands (per level when level-packed) and against the generic tap-table kernel."""
    from opental_amd.common import ops
    rs = np.random.RandomState(71)
    x = torch.from_numpy(rs.randn(B, Cin, T).astype(np.float32))
    dy = torch.from_numpy(rs.randn(B, Cout, T).astype(np.float32))
    monkeypatch.setattr(ops, "CONV_PRECISION", 1)
    got = ops.conv_wgrad(x.cuda(), dy.cuda(), (Cout, Cin, k), k, 1, levels=lev)
    _switch(monkeypatch, "OTAL_CONV_NOW1D", 1)
    old = ops.conv_wgrad(x.cuda(), dy.cuda(), (Cout, Cin, k), k, 1, levels=lev)
    _switch(monkeypatch, "OTAL_CONV_NOW1D", 0)
    xr, dr = x.to(torch.bfloat16).float(), dy.to(torch.bfloat16).float()
    w = torch.zeros(Cout, Cin, k, requires_grad=True)
    bounds = lev if lev is not None else [0, T]
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        F.conv1d(xr[:, :, lo:hi], w, padding=k // 2).backward(dr[:, :, lo:hi])
    close(got, w.grad, tol=2e-5)
    close(got, old, tol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,slices", [((1, 64, 4, 24, 24), 192, False), ((2, 96, 3, 12, 12), 128, False),
                                               ((1, 64, 2, 24, 24), 208, True),      # Cout not a multiple of the 64-row tile
                                               ((2, 80, 5, 12, 12), 64, True),       # Cin not a multiple of the 32-channel block
                                               ((3, 128, 8, 12, 12), 192, False),
                                               ((2, 96, 16, 6, 6), 208, False),      # 6x6 planes: four planes per K step, ring of 16 slabs
                                               ((3, 144, 8, 6, 6), 96, True), ((1, 64, 4, 6, 6), 64, False), ((2, 70, 40, 6, 6), 130, True),
                                               ((2, 48, 32, 3, 3), 128, False),      # 3x3 planes: sixteen planes per K step, two slabs
                                               ((3, 160, 16, 3, 3), 96, True), ((2, 64, 32, 3, 3), 130, True),
                                               ((1, 34, 48, 3, 3), 70, False)])      # Cin % 4 != 0: the epilogue's 4-byte row stores
def test_direct_wgrad_3x3x3_matches_reference_and_vector_kernel(shape, cout, slices):
    """conv3_wgrad_direct_kernel (LDS-staged receptive field, transposed LDS reads; csrc/conv_wgrad_direct.inc) on the
    backbone's 24x24 / 12x12 / 6x6 / 3x3 layer shapes: equal to an fp32 convolution's weight gradient on the bf16-ROUNDED operands
    (1e-4 of scale), equal to the vector kernel it replaces (same products, other summation order), with x / dy taken as
    channel slices of larger buffers (Inception concat layout), several samples / planes (zero borders in t and h), and
    split-K over the positions."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(abs(hash(str((shape, cout)))) % 10000)
    B, Cin, T, H, W = shape
    xb = torch.from_numpy(rs.randn(B, Cin + (8 if slices else 0), T, H, W).astype(np.float32)).cuda()
    dyb = torch.from_numpy(rs.randn(B, cout + (16 if slices else 0), T, H, W).astype(np.float32)).cuda()
    x = xb[:, 4:4 + Cin] if slices else xb
    dy = dyb[:, 8:8 + cout] if slices else dyb
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        L.set_option("OTAL_CONV_NOWDIRECT", 0)
        dw = ops.conv_wgrad(x, dy, (cout, Cin, 3, 3, 3), (3, 3, 3), (1, 1, 1))
        L.set_option("OTAL_WDIRECT_BLOCKS", 3)                 # few workgroups: long split-K ranges crossing samples
        dw_few = ops.conv_wgrad(x, dy, (cout, Cin, 3, 3, 3), (3, 3, 3), (1, 1, 1))
        L.set_option("OTAL_WDIRECT_BLOCKS", 0)
        L.set_option("OTAL_CONV_NOWDIRECT", 1)
        dw_vec = ops.conv_wgrad(x, dy, (cout, Cin, 3, 3, 3), (3, 3, 3), (1, 1, 1))
    finally:
        L.set_option("OTAL_CONV_NOWDIRECT", 0)
        L.set_option("OTAL_WDIRECT_BLOCKS", 0)
        ops.CONV_PRECISION = old
    xr = _bf16_round(x.cpu()).contiguous()
    dr = _bf16_round(dy.cpu()).contiguous()
    w = torch.zeros(cout, Cin, 3, 3, 3, requires_grad=True)
    F.conv3d(xr, w, padding=1).backward(dr)
    close(dw, w.grad)
    close(dw_few, w.grad)
    close(dw_vec, w.grad)
    scale = float(w.grad.abs().max())
    assert float((dw - dw_vec).abs().max()) <= 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout", [((2, 64, 2, 24, 24), 96), ((1, 96, 4, 12, 12), 64), ((2, 64, 8, 6, 6), 80), ((2, 40, 16, 3, 3), 64)])
def test_direct_wgrad_accumulates_into_dw_without_a_split(shape, cout):
    """One workgroup per tile (OTAL_WDIRECT_BLOCKS=1: no split-K): the direct kernels' row epilogue adds to the dW it finds
    (accumulate=True) -- equal to dW0 + the weight gradient, as the split-K reduction does it."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(7)
    B, Cin, T, H, W = shape
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    dy = torch.from_numpy(rs.randn(B, cout, T, H, W).astype(np.float32)).cuda()
    dw0 = torch.from_numpy(rs.randn(cout, Cin, 3, 3, 3).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        plain = ops.conv_wgrad(x, dy, tuple(dw0.shape), (3, 3, 3), (1, 1, 1))
        L.set_option("OTAL_WDIRECT_BLOCKS", 1)
        one = ops.conv_wgrad(x, dy, tuple(dw0.shape), (3, 3, 3), (1, 1, 1))
        acc = ops.conv_wgrad(x, dy, tuple(dw0.shape), (3, 3, 3), (1, 1, 1), out=dw0.clone(), accumulate=True)
    finally:
        L.set_option("OTAL_WDIRECT_BLOCKS", 0)
        ops.CONV_PRECISION = old
    scale = float(plain.abs().max())
    assert float((one - plain).abs().max()) <= 2e-5 * scale
    assert torch.equal(acc, dw0 + one)


LEV126 = (0, 64, 96, 112, 120, 124, 126)


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,T,k,lev", [(8, 512, 512, 126, 3, LEV126), (3, 512, 1024, 126, 1, None), (2, 2048, 512, 126, 1, None),
                                                (2, 512, 512, 256, 3, None), (2, 512, 512, 256, 1, None), (4, 512, 15, 126, 3, LEV126),
                                                (2, 128, 40, 64, 3, (0, 32, 48, 64)), (1, 256, 512, 131, 3, None)])
def test_conv1d_tile_kernel_matches_reference_and_tiled_kernels(B, cin, cout, T, k, lev):
    """conv1d_tile_kernel (csrc/conv1d_tile.inc: one launch, whole K streamed through LDS, no split-K) -- forward with
    bias + ReLU epilogue and data gradient -- on the temporal-pyramid shapes: equal to fp32 conv1d on the bf16-ROUNDED
    operands per level (1e-4 of scale) and to the tiled kernel + split-K reduce it replaces; x and dy may be channel
    slices of larger buffers; an odd row length exercises the 4-byte aligned vector loads."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(cin + cout + T + k)
    xb = torch.from_numpy(rs.randn(B, cin + 6, T).astype(np.float32)).cuda()
    x = xb[:, 2:2 + cin]
    w = torch.from_numpy((rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    dyb = torch.from_numpy(rs.randn(B, cout + 4, T).astype(np.float32)).cuda()
    dy = dyb[:, 1:1 + cout]
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        res = []
        for off in (0, 1):
            L.set_option("OTAL_CONV_NO1DTILE", off)
            y = ops.conv_forward(x, w.view(cout, cin, k, 1, 1), (k, 1, 1), (1, 1, 1), shift=bias, relu=True, levels=lev)
            dx = ops.conv_dgrad(dy, w.view(cout, cin, k, 1, 1), x.shape, (k, 1, 1), (1, 1, 1), levels=lev)
            res.append((y, dx))
    finally:
        L.set_option("OTAL_CONV_NO1DTILE", 0)
        ops.CONV_PRECISION = old
    xr = _bf16_round(x.cpu()).contiguous().requires_grad_(True)
    wr = _bf16_round(w.cpu())
    bounds = lev if lev is not None else (0, T)
    ys = [F.conv1d(xr[:, :, bounds[i]:bounds[i + 1]], wr, bias.cpu(), padding=k // 2) for i in range(len(bounds) - 1)]
    yref = torch.cat(ys, 2)
    yref.backward(_bf16_round(dy.cpu()))                    # d/dx of the convolution itself (the ReLU is applied below)
    close(res[0][0], yref.detach().clamp(min=0))
    close(res[0][1], xr.grad)
    close(res[1][0], yref.detach().clamp(min=0))
    close(res[1][1], xr.grad)
    for a, b_ in zip(res[0], res[1]):
        assert float((a - b_).abs().max()) <= 2e-5 * float(b_.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,splits", [((1, 3, 8, 96, 96), 64, 0), ((2, 3, 4, 96, 96), 48, 5), ((1, 3, 12, 96, 96), 64, 1)])
def test_conv1a_direct_wgrad_matches_reference_and_pair_kernel(shape, cout, splits):
    """conv1a_wgrad_direct_kernel (csrc/conv1a_wgrad.inc) -- the 7x7x7 stride-2 weight gradient from an LDS-staged
    channel-last patch with transposed LDS reads -- equals the fp32 weight gradient on bf16-rounded operands (1e-4 of
    scale) and the pair-mode vector kernel it replaces; zero padding in t / h / w, Cout < 64, few / one split."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    B, _, T, H, W = shape
    dy = torch.from_numpy(rs.randn(B, cout, T // 2, H // 2, W // 2).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        L.set_option("OTAL_W1A_SPLITS", splits)
        dw = ops.conv_wgrad(x, dy, (cout, 3, 7, 7, 7), (7, 7, 7), (2, 2, 2))
        L.set_option("OTAL_CONV_NO1AW", 1)
        dw_pair = ops.conv_wgrad(x, dy, (cout, 3, 7, 7, 7), (7, 7, 7), (2, 2, 2))
    finally:
        L.set_option("OTAL_CONV_NO1AW", 0)
        L.set_option("OTAL_W1A_SPLITS", 0)
        ops.CONV_PRECISION = old
    w = torch.zeros(cout, 3, 7, 7, 7, requires_grad=True)
    F.conv3d(F.pad(_bf16_round(x.cpu()), [2, 3, 2, 3, 2, 3]), w, stride=2).backward(_bf16_round(dy.cpu()))
    close(dw, w.grad)
    close(dw_pair, w.grad)
    assert float((dw - dw_pair).abs().max()) <= 2e-5 * float(w.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout", [((2, 832, 64, 6, 6), 512), ((1, 64, 8, 6, 6), 40), ((3, 96, 70, 6, 6), 64)])
def test_projection_gemm_forward(shape, cout):
    """proj_fwd_kernel (csrc/proj_gemm.inc): the [1,6,6] spatial_valid projection as a K-contiguous GEMM on the fp32 weights
    in place -- equal to conv3d on bf16-rounded operands (1e-4 of scale) and to the tiled gather kernel it replaces; bias
    + ReLU epilogue through the split-K reduce; M < 64, T not a multiple of 64, channel-sliced x."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, cin, T, H, W = shape
    xb = torch.from_numpy(rs.randn(B, cin + 8, T, H, W).astype(np.float32)).cuda()
    x = xb[:, 4:4 + cin]
    w = torch.from_numpy((rs.randn(cout, cin, 1, 6, 6) / np.sqrt(cin * 36)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        y = ops.conv_forward(x, w, (1, 6, 6), (1, 1, 1), shift=bias, relu=True, spatial_valid=True)
        L.set_option("OTAL_CONV_NOPROJ", 1)
        y_old = ops.conv_forward(x, w, (1, 6, 6), (1, 1, 1), shift=bias, relu=True, spatial_valid=True)
    finally:
        L.set_option("OTAL_CONV_NOPROJ", 0)
        ops.CONV_PRECISION = old
    ref = F.conv3d(_bf16_round(x.cpu()), _bf16_round(w.cpu()), bias.cpu()).clamp(min=0)
    assert tuple(y.shape) == tuple(ref.shape) == (B, cout, T, 1, 1)
    close(y, ref)
    close(y_old, ref)
    assert float((y - y_old).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k,cout,accumulate", [((2, 832, 64, 6, 6), (1, 6, 6), 512, False), ((2, 1024, 32, 3, 3), (1, 3, 3), 512, False),
                                                     ((1, 40, 32, 6, 6), (1, 6, 6), 72, True), ((3, 20, 96, 3, 3), (1, 3, 3), 200, True),
                                                     ((2, 16, 64, 2, 2), (1, 2, 2), 130, False)])
def test_projection_weight_gradient(shape, k, cout, accumulate):
    """proj_wgrad_kernel (csrc/proj_gemm.inc): the weight gradient of the spatial_valid pyramid projections
    (AFSD/thumos14/BDNet.py:129-155) as one 128 x 128-tile GEMM over the whole K = B * T, no split-K, straight into the
    gradient -- equal to the fp32 conv3d weight gradient on bf16-rounded operands (1e-4 of scale) and to the tiled gather
    kernel it replaces; both model shapes (6 x 6: 16-byte column quads; 3 x 3: per-element loads), ragged M / N tiles,
    channel-sliced x, accumulate."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, cin, T, H, W = shape
    xb = torch.from_numpy(rs.randn(B, cin + 8, T, H, W).astype(np.float32)).cuda()
    x = xb[:, 4:4 + cin]
    dc = torch.from_numpy(rs.randn(B, cout, T, 1, 1).astype(np.float32)).cuda()
    base = torch.from_numpy(rs.randn(cout, cin, *k).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        dw = ops.conv_wgrad(x, dc, (cout, cin) + k, k, (1, 1, 1), spatial_valid=True, out=base.clone() if accumulate else None,
                            accumulate=accumulate)
        L.set_option("OTAL_CONV_NOPROJW", 1)
        dw_old = ops.conv_wgrad(x, dc, (cout, cin) + k, k, (1, 1, 1), spatial_valid=True, out=base.clone() if accumulate else None,
                                accumulate=accumulate)
    finally:
        L.set_option("OTAL_CONV_NOPROJW", 0)
        ops.CONV_PRECISION = old
    xr, dr = _bf16_round(x.cpu()).double(), _bf16_round(dc.cpu()).double()
    ref = torch.einsum("bot,bcthw->ochw", dr.view(B, cout, T), xr).view(cout, cin, *k)
    if accumulate:
        ref = ref + base.cpu().double()
    close(dw, ref.float())
    close(dw_old, ref.float())
    assert float((dw - dw_old).abs().max()) <= 3e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,accumulate", [((2, 256, 4, 12, 12), 288, False), ((1, 192, 2, 24, 24), 176, True), ((3, 832, 8, 6, 6), 624, False),
                                                   ((2, 40, 32, 3, 3), 24, False), ((1, 480, 8, 6, 6), 304, False), ((1, 64, 1, 8, 4), 448, False)])
def test_wide_1x1_wgrad_matches_reference_and_vector_kernel(shape, cout, accumulate):
    """wgrad1x1_wide_kernel (csrc/wgrad1x1.inc): one workgroup holds up to 288 x 256 of dW and streams positions -- equal to
    the fp32 weight gradient on bf16-rounded operands (1e-4 of scale) and to the tiled vector kernel it replaces; channel
    slices of larger buffers, Cout / Cin that are not multiples of 32, several row / column blocks, one split, accumulate."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, cin, T, H, W = shape
    xb = torch.from_numpy(rs.randn(B, cin + 8, T, H, W).astype(np.float32)).cuda()
    x = xb[:, 4:4 + cin]
    dyb = torch.from_numpy(rs.randn(B, cout + 4, T, H, W).astype(np.float32)).cuda()
    dy = dyb[:, 1:1 + cout]
    base = torch.from_numpy(rs.randn(cout, cin, 1, 1, 1).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        res = []
        for off in (0, 1):
            L.set_option("OTAL_CONV_NOW1X1", off)
            out = base.clone() if accumulate else None
            res.append(ops.conv_wgrad(x, dy, (cout, cin, 1, 1, 1), (1, 1, 1), (1, 1, 1), out=out, accumulate=accumulate))
    finally:
        L.set_option("OTAL_CONV_NOW1X1", 0)
        ops.CONV_PRECISION = old
    ref = torch.einsum("bmthw,bnthw->mn", _bf16_round(dy.cpu()).double(), _bf16_round(x.cpu()).double()).float().view(cout, cin, 1, 1, 1)
    if accumulate:
        ref = ref + base.cpu()
    close(res[0], ref)
    close(res[1], ref)
    assert float((res[0] - res[1]).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout", [((2, 32, 8, 8, 8), 96), ((1, 96, 8, 24, 24), 192), ((2, 96, 32, 12, 12), 96), ((3, 192, 128, 2, 2), 96)])
def test_direct_conv_two_position_tiles_per_wave(shape, cout):
    """conv3_direct_kernel<BM, MODE, 512, 2> (512 positions per workgroup, each weight fragment shared by two position tiles
    of a wave): forward with scale / shift / ReLU and the masked data gradient equal the fp32 convolution of the bf16-rounded
    operands (1e-4 of scale) and the 256-position variant bit for bit (same products, same summation order per output);
    96-row tiles (the variant's domain) in forward (Cout) and data gradient (Cin), planes of 8 / 24 / 12 / 2, tiles crossing
    t planes and rows."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, cin, T, H, W = shape
    assert (T * H * W) % 512 == 0
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float32)).cuda()
    sc = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)).cuda()
    sh = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    sci = torch.from_numpy((rs.rand(cin) + 0.5).astype(np.float32)).cuda()
    dy = torch.from_numpy(rs.randn(B, cout, T, H, W).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        res = []
        for min512 in (1, 1 << 30):
            L.set_option("OTAL_CONV_DIRECT_MINTILES512", min512)
            y = ops.conv_forward(x, w, (3, 3, 3), (1, 1, 1), scale=sc, shift=sh, relu=True)
            dx = ops.conv_dgrad(dy, w, x.shape, (3, 3, 3), (1, 1, 1), out_mask=x, out_scale=sci)
            res.append((y, dx))
    finally:
        L.set_option("OTAL_CONV_DIRECT_MINTILES512", 512)
        ops.CONV_PRECISION = old
    xr = _bf16_round(x.cpu()).requires_grad_(True)
    yr = F.conv3d(xr, _bf16_round(w.cpu()), padding=1)
    yr.backward(_bf16_round(dy.cpu()))
    yref = (yr.detach() * sc.cpu().view(1, -1, 1, 1, 1) + sh.cpu().view(1, -1, 1, 1, 1)).clamp(min=0)
    dxref = xr.grad * (x.cpu() > 0) * sci.cpu().view(1, -1, 1, 1, 1)
    close(res[0][0], yref)
    close(res[0][1], dxref)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout", [((2, 16, 8, 8, 8), 96),       # 3 K steps (odd: the unrolled loop's tail), 96 x 256 tiles
                                        ((1, 48, 4, 24, 24), 64),      # 9 K steps forward; data gradient: 12 steps, M = 48
                                        ((1, 96, 8, 24, 24), 192),     # 96 x 512 forward tiles (two position tiles per wave)
                                        ((2, 64, 64, 6, 6), 128),      # 6x6 planes, 64-row tiles, tiles crossing t planes
                                        ((1, 16, 4, 8, 8), 64)])       # ONE channel block: 3 steps, the second load re-reads step 2
def test_direct_conv_positions_two_steps_ahead_equal_one_step(shape, cout):
    """conv3_direct_kernel<..., XPF2 = true> (gathered positions loaded TWO K steps ahead through a second register set, K loop
    unrolled by two) gives bit for bit what the one-step-ahead kernel gives -- same products, same order -- for odd and even
    step counts, every tile shape that has the variant (OTAL_CONV_DIRECT_XPF2 = 7 forces all of them), forward with scale /
    shift / ReLU and masked data gradient; and both equal the fp32 convolution of the bf16-rounded operands."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, cin, T, H, W = shape
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float32)).cuda()
    sc = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)).cuda()
    sh = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    sci = torch.from_numpy((rs.rand(cin) + 0.5).astype(np.float32)).cuda()
    dy = torch.from_numpy(rs.randn(B, cout, T, H, W).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        res = []
        L.set_option("OTAL_CONV_DIRECT_MINTILES512", 1)
        for xpf in (0, 7):
            L.set_option("OTAL_CONV_DIRECT_XPF2", xpf)
            y = ops.conv_forward(x, w, (3, 3, 3), (1, 1, 1), scale=sc, shift=sh, relu=True)
            dx = ops.conv_dgrad(dy, w, x.shape, (3, 3, 3), (1, 1, 1), out_mask=x, out_scale=sci)
            res.append((y, dx))
    finally:
        L.set_option("OTAL_CONV_DIRECT_XPF2", 3)
        L.set_option("OTAL_CONV_DIRECT_MINTILES512", 512)
        ops.CONV_PRECISION = old
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    xr = _bf16_round(x.cpu()).requires_grad_(True)
    yr = F.conv3d(xr, _bf16_round(w.cpu()), padding=1)
    yr.backward(_bf16_round(dy.cpu()))
    close(res[1][0], (yr.detach() * sc.cpu().view(1, -1, 1, 1, 1) + sh.cpu().view(1, -1, 1, 1, 1)).clamp(min=0))
    close(res[1][1], xr.grad * (x.cpu() > 0) * sci.cpu().view(1, -1, 1, 1, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,min_tiles", [
    ((2, 32, 64, 6, 6), 128, 30),      # Mixed_4f.b2b: dgrad M = 32 on 128-position tiles (18 tiles of 256 < 30 <= 36 of 128), fwd M = 128
    ((2, 24, 64, 6, 6), 64, 30),       # Mixed_4c.b2b: dgrad M = 24 (rows 24 .. 31 are padding); fwd: gather kernel (Cin % 16)
    ((1, 16, 32, 12, 12), 32, 1),      # Mixed_3b.b2b: dgrad M = 16 AND fwd M = 32, 256-position tiles
    ((2, 48, 8, 8, 8), 16, 1)])        # fwd M = 16, dgrad M = 48 (64-row tile, padded)
def test_direct_conv_32_row_tiles(shape, cout, min_tiles):
    """conv3_direct_kernel<32, MODE, 128 / 256, 1> (one 32-row MFMA tile per wave: the data gradient of the Inception b2b
    layers, M = Cin = 16 / 24 / 32, and forward layers with <= 32 output channels): forward with scale / shift / ReLU and the
    masked data gradient equal the fp32 convolution of the bf16-rounded operands (1e-4 of scale), and the gather kernel
    (OTAL_CONV_NODIRECT) to summation-order accuracy."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, cin, T, H, W = shape
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float32)).cuda()
    sc = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)).cuda()
    sh = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    sci = torch.from_numpy((rs.rand(cin) + 0.5).astype(np.float32)).cuda()
    dy = torch.from_numpy(rs.randn(B, cout, T, H, W).astype(np.float32)).cuda()
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        res = []
        L.set_option("OTAL_CONV_DIRECT_MINTILES", min_tiles)
        for nodirect in (0, 1):
            L.set_option("OTAL_CONV_NODIRECT", nodirect)
            y = ops.conv_forward(x, w, (3, 3, 3), (1, 1, 1), scale=sc, shift=sh, relu=True)
            dx = ops.conv_dgrad(dy, w, x.shape, (3, 3, 3), (1, 1, 1), out_mask=x, out_scale=sci)
            res.append((y, dx))
    finally:
        L.set_option("OTAL_CONV_NODIRECT", 0)
        L.set_option("OTAL_CONV_DIRECT_MINTILES", int(os.environ.get("OTAL_CONV_DIRECT_MINTILES", "140")))     # conftest: 1
        ops.CONV_PRECISION = old
    xr = _bf16_round(x.cpu()).requires_grad_(True)
    yr = F.conv3d(xr, _bf16_round(w.cpu()), padding=1)
    yr.backward(_bf16_round(dy.cpu()))
    yref = (yr.detach() * sc.cpu().view(1, -1, 1, 1, 1) + sh.cpu().view(1, -1, 1, 1, 1)).clamp(min=0)
    dxref = xr.grad * (x.cpu() > 0) * sci.cpu().view(1, -1, 1, 1, 1)
    for y, dx in res:
        close(y, yref)
        close(dx, dxref)
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * float(dxref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,T,H,W", [(2, 48, 8, 6, 6), (3, 40, 5, 3, 3), (1, 7, 3, 1, 1)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_masked_scale_copy_matches_aten(B, C, T, H, W, accumulate):
    """otal_masked_scale_copy = aten threshold_backward * scale (+ add) on a PERMUTED source view ([(b,t)][c][hw], what
    ops.conv_dgrad_collapse returns) with channel-sliced mask and destination: bit-identical (one multiply, one add)."""
    from opental_amd.common import ops
    rs = np.random.RandomState(B * 100 + C + T)
    src = torch.from_numpy(rs.randn(B, T, C, H, W).astype(np.float32)).cuda().permute(0, 2, 1, 3, 4)
    zb = torch.from_numpy(rs.randn(B, C + 12, T, H, W).astype(np.float32)).cuda().clamp(min=0)
    z = zb[:, 8:8 + C]
    scale = torch.from_numpy(rs.uniform(0.5, 2.0, C).astype(np.float32)).cuda()
    db = torch.from_numpy(rs.randn(B, C + 4, T, H, W).astype(np.float32)).cuda()
    dst = db[:, 4:]
    before = db.clone()
    want = torch.ops.aten.threshold_backward(src.contiguous(), z.contiguous(), 0.0) * scale.view(1, -1, 1, 1, 1)
    if accumulate:
        want = before[:, 4:] + want
    ops.masked_scale_copy(src, z, scale, dst, accumulate=accumulate)
    assert torch.equal(dst, want)
    assert torch.equal(db[:, :4], before[:, :4])


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("shape,cout,k", [((2, 832, 64, 6, 6), 512, (1, 6, 6)), ((2, 1024, 32, 3, 3), 512, (1, 3, 3)), ((3, 40, 7, 6, 6), 24, (1, 6, 6))])
def test_projection_data_gradient_swapped_roles(shape, cout, k, prec):
    """ops.conv_dgrad_collapse (the weight as the GEMM's activation map, dy as its weights, result returned as a permuted
    view) against conv3d's input gradient -- 1e-4 on the fp32 path, and on bf16-rounded operands on the bf16 path."""
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, cin, T, H, W = shape
    w = torch.from_numpy((rs.randn(cout, cin, *k) / np.sqrt(cin * H * W)).astype(np.float32))
    dy = torch.from_numpy(rs.randn(B, cout, T, 1, 1).astype(np.float32))
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = prec
    try:
        dx = ops.conv_dgrad_collapse(dy.cuda(), w.cuda(), shape)
    finally:
        ops.CONV_PRECISION = old
    assert tuple(dx.shape) == tuple(shape)
    wr, dyr = (_bf16_round(w), _bf16_round(dy)) if prec else (w, dy)
    x = torch.zeros(shape, requires_grad=True)
    F.conv3d(x, wr).backward(dyr)
    close(dx, x.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,N,lev", [(8, 512, 126, (0, 64, 96, 112, 120, 124, 126)), (2, 64, 37, None), (3, 48, 16, (0, 8, 12, 14, 15, 16))])
@pytest.mark.parametrize("stage", ["coarse", "refined"])
def test_fused_head_convs_match_conv1d(B, C, N, lev, stage):
    """csrc/headconv.hip: the heads of a CoarsePyramid stage in one launch forward / two backward against F.conv1d with
    SAME zero padding per pyramid level (fp32, 1e-5 of scale): outputs, the summed input gradients, weight and bias
    gradients; one head without an incoming gradient (dy = None -> zeros)."""
    from opental_amd.common import ops
    rs = np.random.RandomState(B * 1000 + C + N)
    heads = [(0, 2, 3), (1, 15, 3), (1, 1, 3)] if stage == "coarse" else [(0, 2, 1), (1, 15, 1), (0, 1, 3), (1, 1, 1)]
    xs = [torch.from_numpy(rs.randn(B, C, N).astype(np.float32)) for _ in range(2)]
    ws = [torch.from_numpy((rs.randn(co, C, k) / np.sqrt(C * k)).astype(np.float32)) for _, co, k in heads]
    bs = [torch.from_numpy(rs.randn(co).astype(np.float32)) for _, co, _ in heads]
    dys = [torch.from_numpy(rs.randn(B, co, N).astype(np.float32)) for _, co, _ in heads]
    skip = len(heads) - 1                       # the last head receives no gradient
    bounds = list(lev) if lev is not None else [0, N]

    def reference():
        xr = [x.clone().requires_grad_(True) for x in xs]
        wr = [w.clone().requires_grad_(True) for w in ws]
        br = [b.clone().requires_grad_(True) for b in bs]
        ys = []
        for (j, co, k), w, b in zip(heads, wr, br):
            ys.append(torch.cat([F.conv1d(F.pad(xr[j][:, :, lo:hi], (k // 2, k // 2)), w, b) for lo, hi in zip(bounds[:-1], bounds[1:])], 2))
        torch.autograd.backward([y for i, y in enumerate(ys) if i != skip], [g for i, g in enumerate(dys) if i != skip])
        return ys, xr, wr, br

    ys_r, xr, wr, br = reference()
    assert ops.head_convs_supported(heads, 2, B, C, N)
    xd = [x.cuda().requires_grad_(True) for x in xs]
    wd = [w.cuda().requires_grad_(True) for w in ws]
    bd = [b.cuda().requires_grad_(True) for b in bs]
    ys = ops.HeadConvsFunction.apply(lev, tuple((j, k) for j, _, k in heads), 2, *xd, *wd, *bd)
    torch.autograd.backward([y for i, y in enumerate(ys) if i != skip], [g.cuda() for i, g in enumerate(dys) if i != skip])
    for y, yr in zip(ys, ys_r):
        close(y, yr, tol=1e-5)
    for a, b_ in zip(xd, xr):
        close(a.grad, b_.grad, tol=1e-5)
    for i, (a, b_) in enumerate(zip(wd, wr)):
        if i == skip:
            assert float(a.grad.abs().max()) == 0.0
        else:
            close(a.grad, b_.grad, tol=2e-5)
    for i, (a, b_) in enumerate(zip(bd, br)):
        if i != skip:
            close(a.grad, b_.grad, tol=2e-5)


@pytest.mark.gpu
def test_fused_head_convs_decline_wide_heads():
    """150-class ActivityNet heads do not fit (more than 21 rows on one input): the caller runs them one by one."""
    from opental_amd.common import ops
    assert not ops.head_convs_supported([(0, 2, 3), (1, 150, 3), (1, 1, 3)], 2, 2, 512, 189)


@pytest.mark.gpu
def test_groupnorm_backward_reads_sliced_gradients_and_defers_batch_sums():
    """otal_gn_relu_bwd with dy as a channel slice of a wider map (the gradient of torch.cat) is bit-identical to the
    contiguous copy; otal_sum_partials (many layers per launch, ascending b) equals the per-layer batch sums."""
    import ctypes
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(7)
    B, C, T, lev = 8, 512, 126, (0, 64, 96, 112, 120, 124, 126)
    x = torch.from_numpy(rs.randn(B, C, T).astype(np.float32)).cuda()
    ga = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)).cuda()
    be = torch.from_numpy(rs.randn(C).astype(np.float32)).cuda()
    wide = torch.from_numpy(rs.randn(B, C + 1024 + 512, T).astype(np.float32)).cuda()
    dy = wide[:, 1024:1024 + C]
    assert not dy.is_contiguous()
    _, stats = ops.gn_relu_forward(x, ga, be, levels=lev)
    a = ops.gn_relu_backward(dy, x, ga, be, stats, levels=lev)
    b = ops.gn_relu_backward(dy.contiguous(), x, ga, be, stats, levels=lev)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    parts = [torch.from_numpy(rs.randn(bb, 3, cc).astype(np.float32)).cuda() for bb, cc in ((8, 512), (8, 1024), (3, 40), (1, 7))] * 10
    outs = [[torch.full((p.shape[2],), 7.0, device="cuda") for _ in range(3)] for p in parts]
    outs[2][1] = None                                   # a skipped row
    n = len(parts)
    VP = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    L.check(L.lib().otal_sum_partials(n, VP(parts), VP([o[0] for o in outs]), VP([o[1] for o in outs]), VP([o[2] for o in outs]),
                                      L.int_array([p.shape[2] for p in parts]), L.int_array([p.shape[0] for p in parts]), L.stream()),
            "otal_sum_partials")
    for p, o in zip(parts, outs):
        want = p.double().sum(0)
        for r in range(3):
            if o[r] is not None:
                assert float((o[r].double() - want[r]).abs().max()) < 1e-5


@pytest.mark.gpu
def test_boundary_losses_fused_tails_match_calc_bce_loss():
    """ops.BoundaryLossesFunction (three otal_boundary_bce launches + one launch for the means / weighted sums + one for the
    backward of all maps) against the reference formulation of train.py:193-201 built from calc_bce_loss."""
    from opental_amd.common import ops
    from opental_amd.thumos14.train import calc_bce_loss
    rs = np.random.RandomState(3)
    B = 4
    post_relu = lambda *shape: torch.from_numpy(np.abs(rs.randn(*shape)).astype(np.float32) * 0.6)      # post-ReLU features
    maps = [post_relu(B, 512, 256), post_relu(B, 1024, 64), post_relu(B, 1024, 64)]
    mask = torch.from_numpy((rs.rand(B, 2, 256) < 0.2).astype(np.float32))
    xr = [m.clone().requires_grad_(True) for m in maps]
    half = lambda x: (x[:, :x.shape[1] // 2].permute(0, 2, 1), x[:, x.shape[1] // 2:].permute(0, 2, 1))
    s0, e0 = calc_bce_loss(*half(xr[0]), mask)
    a, b_ = calc_bce_loss(*half(xr[1]), mask[:, :, ::4])
    c, d = calc_bce_loss(*half(xr[2]), mask[:, :, ::4])
    ls, le = s0 + 0.1 * (a + c), e0 + 0.1 * (b_ + d)
    (2.0 * ls + 3.0 * le).backward()
    xd = [m.cuda().requires_grad_(True) for m in maps]
    gs, ge = ops.BoundaryLossesFunction.apply(mask.cuda(), (1, 4, 4), (1.0, 0.1, 0.1), *xd)
    (2.0 * gs + 3.0 * ge).backward()
    assert abs(float(gs) - float(ls)) < 1e-5 * abs(float(ls)) and abs(float(ge) - float(le)) < 1e-5 * abs(float(le))
    for u, v in zip(xd, xr):
        close(u.grad, v.grad, tol=1e-4)


@pytest.mark.parametrize("k,cin,cout,T,levels", [(3, 512, 512, 126, (0, 64, 96, 112, 120, 124, 126)), (1, 512, 1024, 126, None),
                                                 (1, 2048, 512, 126, None), (3, 512, 512, 256, None)])
def test_pair_launches_equal_the_single_launches_bit_for_bit(k, cin, cout, T, levels):
    """ABI 20: two sibling Conv + GroupNorm + ReLU blocks in one set of launches (ConvGNReLUPairFunction: conv fwd, GN fwd,
    GN bwd, dgrad, wgrad each carry both problems) give exactly the bits of the two blocks on their own -- outputs,
    input gradients and every parameter gradient."""
    from opental_amd.common import ops
    from opental_amd.common.layers import ConvGNReLU, Unit1D, conv_gn_relu_pair
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        torch.manual_seed(3)
        blocks = [ConvGNReLU(Unit1D(cin, cout, k, use_bias=True, activation_fn=None), cout).cuda() for _ in range(2)]
        for b in blocks:
            with torch.no_grad():
                b[1].weight.uniform_(0.5, 1.5); b[1].bias.uniform_(-0.3, 0.3)
        xs = [torch.randn(4, cin, T, device="cuda", requires_grad=True) for _ in range(2)]
        gys = [torch.randn(4, cout, T, device="cuda") for _ in range(2)]

        def run(pair):
            for b in blocks:
                b.zero_grad(set_to_none=True)
            for x in xs:
                x.grad = None
            ops.PAIR_LAUNCHES = pair
            ys = conv_gn_relu_pair(blocks[0], blocks[1], xs[0], xs[1], levels)
            torch.autograd.backward(ys, gys)
            out = [y.detach().clone() for y in ys] + [x.grad.clone() for x in xs]
            for b in blocks:
                out += [p.grad.clone() for p in b.parameters()]
            return out
        single, paired = run(False), run(True)
        assert len(single) == len(paired) == 12
        for a, b in zip(single, paired):
            assert torch.equal(a, b)
        # one tensor feeding both blocks (the towers' first stage): autograd adds the two input gradients
        x = torch.randn(4, cin, T, device="cuda", requires_grad=True)
        ops.PAIR_LAUNCHES = True
        y0, y1 = conv_gn_relu_pair(blocks[0], blocks[1], x, x, levels)
        (y0.sum() + 2 * y1.sum()).backward()
        g_pair = x.grad.clone()
        x.grad = None
        ops.PAIR_LAUNCHES = False
        y0, y1 = conv_gn_relu_pair(blocks[0], blocks[1], x, x, levels)
        (y0.sum() + 2 * y1.sum()).backward()
        assert torch.equal(g_pair, x.grad)
    finally:
        ops.CONV_PRECISION = old
        ops.PAIR_LAUNCHES = True


@pytest.mark.parametrize("shape", [(2, 8, 45, 12, 12), (1, 4, 128, 12, 12), (2, 8, 83, 6, 6), (1, 16, 64, 6, 6), (1, 4, 3, 12, 12)])
def test_row_per_thread_333_pool_equals_the_cell_kernel(shape):
    """maxpool333_rows_fwd_kernel (one row per thread, vector LDS exchange) gives the bits of maxpool333_sep_fwd_kernel --
    values AND tap bytes -- on planes of 12 x 12 and 6 x 6, tile boundaries and short clips included; ties with the zero
    padding (ReLU outputs) and NaN included."""
    from opental_amd import _lib as L
    from opental_amd.common import ops
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(*shape, generator=g).clamp(min=0)          # exact zeros tie with the padding and with each other
    x[0, 0, shape[2] // 2, 3, 2] = float("nan")
    xd = x.cuda()
    try:
        L.set_option("OTAL_POOL_NOROWS", 1)
        y0, a0 = ops.maxpool3d_forward(xd, (3, 3, 3), (1, 1, 1))
        L.set_option("OTAL_POOL_NOROWS", 0)
        y1, a1 = ops.maxpool3d_forward(xd, (3, 3, 3), (1, 1, 1))
    finally:
        L.set_option("OTAL_POOL_NOROWS", 0)
    assert torch.equal(a0, a1)
    assert torch.equal(y0.view(torch.int32), y1.view(torch.int32))
    # backward: the row kernel adds the same terms in the same order (plain store, fused ReLU / BN mask, accumulate)
    dy = torch.randn(y0.shape, generator=g).cuda()
    sc = (torch.rand(shape[1], generator=g) + 0.5).cuda()
    base = torch.randn(*shape, generator=g).cuda()
    outs = []
    for rows_off in (1, 0):
        L.set_option("OTAL_POOL_NOROWS", rows_off)
        try:
            plain = ops.maxpool3d_backward(dy, a0, xd.shape, (3, 3, 3), (1, 1, 1))
            masked = ops.maxpool3d_backward(dy, a0, xd.shape, (3, 3, 3), (1, 1, 1), out_mask=xd, out_scale=sc)
            acc = ops.maxpool3d_backward(dy, a0, xd.shape, (3, 3, 3), (1, 1, 1), out=base.clone(), accumulate=True, out_mask=xd, out_scale=sc)
        finally:
            L.set_option("OTAL_POOL_NOROWS", 0)
        outs.append((plain, masked, acc))
    for u, v in zip(*outs):
        assert torch.equal(u.view(torch.int32), v.view(torch.int32))


def test_deferred_reduction_record_survives_two_issuing_threads():
    """The C ABI's deferred split-K record is process-wide (include/opental_hip.h, preamble): two host threads that issue
    weight gradients at the same time (ctypes releases the GIL during the calls) must not corrupt it -- every recorded
    reduction runs exactly once at the flush and every result equals the immediate-reduce result.  VERDICT r3 weak #10."""
    import ctypes
    import threading
    from opental_amd import _lib as L
    from opental_amd.common import ops
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        lib = L.lib()
        torch.manual_seed(3)
        shape, cout = (2, 64, 8, 6, 6), 48
        xs = [torch.randn(*shape, device="cuda") for _ in range(2)]
        dys = [torch.randn(2, cout, 8, 6, 6, device="cuda") for _ in range(2)]
        want = [ops.conv_wgrad(x, dy, (cout, 64, 1, 1, 1), (1, 1, 1), (1, 1, 1)) for x, dy in zip(xs, dys)]
        torch.cuda.synchronize()
        g, ga, sa, _, _ = ops._plan(2, xs[0], dys[0], cout, (1, 1, 1), (1, 1, 1), None, False, "x", "dy")
        rounds = 10                 # 20 records in all: below the 24 at which a launch flushes by itself (on ITS stream)
        outs = [[torch.zeros(cout, 64, 1, 1, 1, device="cuda") for _ in range(rounds)] for _ in range(2)]
        wss = [torch.empty(rounds * (8 << 20), dtype=torch.uint8, device="cuda") for _ in range(2)]
        streams = [torch.cuda.Stream() for _ in range(2)]
        torch.cuda.synchronize()
        L.check(lib.otal_conv_defer_reduces(1), "defer")
        errs = []

        def issue(t):
            try:
                for r in range(rounds):        # every launch gets its own slab range: the record only keeps pointers
                    rc = lib.otal_conv_wgrad(ga, sa, L.ptr(xs[t]), L.ptr(dys[t]), L.ptr(outs[t][r]), 0, 1, None,
                                             ctypes.c_void_p(wss[t].data_ptr() + r * (8 << 20)), ctypes.c_size_t(8 << 20),
                                             ctypes.c_void_p(streams[t].cuda_stream))
                    if rc:
                        errs.append(rc)
            except Exception as e:              # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=issue, args=(t,)) for t in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        torch.cuda.synchronize()                # the GEMMs of both streams are done: their slabs are complete
        L.check(lib.otal_conv_flush_reduces(ctypes.c_void_p(streams[0].cuda_stream)), "flush")
        assert lib.otal_conv_deferred_count() == 0
        L.check(lib.otal_conv_defer_reduces(0), "defer off")
        torch.cuda.synchronize()
        for t in range(2):
            for r in range(rounds):
                assert torch.equal(outs[t][r], want[t]), (t, r)
    finally:
        ops.CONV_PRECISION = old
        L.lib().otal_conv_flush_reduces(None)
        L.lib().otal_conv_defer_reduces(0)
