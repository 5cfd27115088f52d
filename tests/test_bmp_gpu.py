"""GPU parity: HIP BoundaryMaxPooling (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import afsd_oracle as O

pytestmark = pytest.mark.gpu


def _mk(rs, B, C, T, N, wide=False):
    x = torch.from_numpy(rs.randn(B, C, T).astype(np.float32))
    if wide:   # out-of-range and inverted windows
        seg = rs.uniform(-0.3 * T - 3, 1.3 * T + 3, size=(B, N, 4))
    else:      # l <= r windows like the model produces (rounded floats)
        a = np.sort(rs.uniform(-2, T + 2, size=(B, N, 2, 2)), -1).reshape(B, N, 4)
        seg = np.round(a)
    g = torch.from_numpy(rs.randn(B, C, N).astype(np.float32))
    return x, torch.from_numpy(seg.astype(np.float32)), g


SHAPES = [(1, 1024, 64, 64), (2, 1024, 2, 2), (1, 512, 256, 64), (2, 512, 256, 4), (8, 1024, 32, 32),
          (1, 512, 768, 96), (3, 6, 33, 7), (1, 2, 5, 300), (2, 20, 1000, 3), (1, 1024, 256, 3)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("wide", [False, True])
def test_forward_backward_bit_exact(shape, wide):
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    rs = np.random.RandomState(hash(shape) % 1000 + int(wide))
    x, seg, g = _mk(rs, *shape, wide=wide)
    xd, sd, gd = x.cuda(), seg.cuda(), g.cuda()
    out = bp.bmp_forward(xd, sd)
    assert torch.equal(out.cpu(), O.bmp_forward(x, seg))
    gin = bp.bmp_backward(gd, xd, sd)
    assert torch.equal(gin.cpu(), O.bmp_backward(g, x, seg))
    if shape[3] <= shape[2]:
        ginc = bp.bmp_backward(gd, xd, sd, compat_reference_bwd=True)
        assert torch.equal(ginc.cpu(), O.bmp_backward(g, x, seg, compat_reference_bwd=True))


def test_ties_keep_lowest_index_and_relu_zeros():
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    rs = np.random.RandomState(3)
    x = torch.from_numpy(np.maximum(rs.randn(2, 64, 128), 0).astype(np.float32))  # many exact zeros / ties
    x[:, :, 10:20] = 1.5
    seg = torch.from_numpy(np.sort(rs.randint(0, 128, size=(2, 40, 2, 2)), -1).reshape(2, 40, 4).astype(np.float32))
    g = torch.from_numpy(rs.randn(2, 64, 40).astype(np.float32))
    assert torch.equal(bp.bmp_forward(x.cuda(), seg.cuda()).cpu(), O.bmp_forward(x, seg))
    assert torch.equal(bp.bmp_backward(g.cuda(), x.cuda(), seg.cuda()).cpu(), O.bmp_backward(g, x, seg))


def test_backward_denormal_gradients_are_not_flushed():
    """Subnormal gradients, and sums that land in the subnormal range, must come out as in the serial fp32 loop of the
    oracle (which does not flush them): guards the LDS accumulation of the backward against flush-to-zero variants
    (an LDS atomic-add version was tried; it passed this test too but was slower)."""
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    rs = np.random.RandomState(11)
    x, seg, g = _mk(rs, 2, 64, 48, 40)
    g = (g * 1e-41).float()                     # subnormal inputs, subnormal sums
    g[:, ::2] = torch.from_numpy((rs.randn(2, 32, 40) * 3e-38).astype(np.float32))      # near the normal / subnormal boundary: sums cancel into it
    assert float(g.abs().min()) < 1e-38 and bool((g != 0).all())
    want = O.bmp_backward(g, x, seg)
    assert bool(((want != 0) & (want.abs() < 1.1754944e-38)).any())
    got = bp.bmp_backward(g.cuda(), x.cuda(), seg.cuda()).cpu()
    assert torch.equal(got, want)


def test_autograd_function_and_module():
    from opental_amd.prop_pooling.boundary_pooling_op import BoundaryMaxPooling
    rs = np.random.RandomState(5)
    x, seg, g = _mk(rs, 2, 512, 256, 64)
    xd = x.cuda().requires_grad_(True)
    y = BoundaryMaxPooling()(xd, seg.cuda())
    y.backward(g.cuda())
    xo = x.clone().requires_grad_(True)
    yo = O.boundary_max_pool(xo, seg)
    yo.backward(g)
    assert torch.equal(y.detach().cpu(), yo.detach()) and torch.equal(xd.grad.cpu(), xo.grad)


def test_batch_mismatch_raises():
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    x = torch.zeros(2, 4, 8, device="cuda")
    seg = torch.zeros(1, 3, 4, device="cuda")
    with pytest.raises(RuntimeError, match="batch"):
        bp.bmp_forward(x, seg)
    with pytest.raises(RuntimeError):
        bp.bmp_forward(torch.zeros(2, 4, 8), torch.zeros(2, 3, 4))  # CPU tensors: no fallback


def test_levels_batched_equals_per_level():
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    rs = np.random.RandomState(9)
    lens = [64, 32, 16, 8, 4, 2]
    starts = [0]
    for t in lens:
        starts.append(starts[-1] + t)
    B, C = 2, 1024
    xs, segs, gs = [], [], []
    for t in lens:
        x, s, g = _mk(rs, B, C, t, t, wide=True)
        xs.append(x); segs.append(s); gs.append(g)
    X, S, G = torch.cat(xs, 2).contiguous(), torch.cat(segs, 1).contiguous(), torch.cat(gs, 2).contiguous()
    out = bp.bmp_forward_levels(X.cuda(), S.cuda(), starts, starts).cpu()
    gin = bp.bmp_backward_levels(G.cuda(), X.cuda(), S.cuda(), starts, starts).cpu()
    for i, t in enumerate(lens):
        sl = slice(starts[i], starts[i + 1])
        assert torch.equal(out[:, :, sl], O.bmp_forward(xs[i], segs[i]))
        assert torch.equal(gin[:, :, sl], O.bmp_backward(gs[i], xs[i], segs[i]))
    # frame-level pooling of every level's proposals in one plain call (N = 126, T = 256)
    x, s, g = _mk(rs, B, 512, 256, 126)
    assert torch.equal(bp.bmp_forward(x.cuda(), s.cuda()).cpu(), O.bmp_forward(x, s))
    assert torch.equal(bp.bmp_backward(g.cuda(), x.cuda(), s.cuda()).cpu(), O.bmp_backward(g, x, s))


def test_bf16_forward_exact_backward_close():
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    rs = np.random.RandomState(11)
    x, seg, g = _mk(rs, 2, 512, 256, 64)
    xb, gb = x.bfloat16(), g.bfloat16()
    out = bp.bmp_forward(xb.cuda(), seg.cuda())
    assert out.dtype == torch.bfloat16
    assert torch.equal(out.float().cpu(), O.bmp_forward(xb.float(), seg))   # max of bf16 values is exact
    gin = bp.bmp_backward(gb.cuda(), xb.cuda(), seg.cuda()).float().cpu()
    ref = O.bmp_backward(gb.float(), xb.float(), seg)
    assert torch.allclose(gin, ref, rtol=1e-2, atol=1e-2)   # sums rounded once to bf16


def test_full_size_property_max_bounds():
    """BASELINE-size property check (b=8, all 24 calls' shapes): out <= row max, out >= in[l]."""
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    g = torch.Generator(device="cuda").manual_seed(1)
    for (C, T, N) in ((1024, 64, 64), (512, 256, 126)):
        x = torch.randn(8, C, T, device="cuda", generator=g)
        a = torch.sort(torch.rand(8, N, 2, 2, device="cuda", generator=g) * T, -1)[0].reshape(8, N, 4).floor()
        out = bp.bmp_forward(x, a)
        assert bool((out <= x.max(2, keepdim=True)[0]).all())
        l0 = a[:, :, 0].long().clamp(0, T - 1)
        first = torch.gather(x[:, : C // 2], 2, l0[:, None, :].expand(8, C // 2, N))
        assert bool((out[:, : C // 2] >= first).all())
        gin = bp.bmp_backward(torch.ones_like(out), x, a)
        assert abs(float(gin.sum()) - out.numel()) < 1e-3 * out.numel()   # every grad lands exactly once
