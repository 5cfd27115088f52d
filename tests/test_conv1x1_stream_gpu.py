"""GPU: the streaming 1x1x1 kernel for bf16-stored tensors (csrc/conv1x1_stream.inc; VERDICT r4 next #1b) -- forward and
data gradient of the reference's Unit3D layers with kernel_shape [1,1,1] (AFSD/common/i3d_backbone.py:33-43, the fused
b1a | b2a | b0 branches and b3b of an InceptionModule, :90-121) at the shapes the backbone runs them on.

Checked against (a) torch's own convolution of the same bf16-valued operands with fp32 accumulation (F.conv3d), i.e. NOT
against a kernel of this library: every output within one bf16 rounding step of the reference; (b) the chunked kernel it
replaces (library option OTAL_CONV_NO1X1STREAM): identical bf16 values except where the two fp32 sums -- whole K in one
workgroup here, split-K slabs there -- land on different sides of a rounding boundary (a handful of elements, one step apart);
(c) channel-sliced views on both sides, the ReLU mask of the data gradient, neighbours of the output slice untouched."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
ONE = (1, 1, 1)

# (x shape, Cout): Conv3d_2b (two position tiles per wave where the grid is large), Mixed_3b / 3c fused and b3b, Mixed_4b / 4c /
# 4f fused (K tails of 16 and 8 channels in the data gradient, 2 and 3 row groups), a b3b with a single half-empty row tile
CASES = [((2, 64, 16, 24, 24), 64), ((4, 64, 128, 24, 24), 64), ((2, 192, 32, 12, 12), 176), ((1, 256, 32, 12, 12), 288),
         ((2, 480, 64, 6, 6), 304), ((1, 512, 64, 6, 6), 296), ((1, 528, 64, 6, 6), 448), ((1, 192, 32, 12, 12), 32),
         ((1, 528, 64, 6, 6), 128), ((2, 512, 64, 6, 6), 280)]


@pytest.fixture(autouse=True)
def _bf16_mode():
    from opental_amd import _lib as L
    from opental_amd.common import ops
    old = (ops.CONV_PRECISION, ops.HALF_STORAGE)
    ops.CONV_PRECISION, ops.HALF_STORAGE = 1, True
    yield
    ops.CONV_PRECISION, ops.HALF_STORAGE = old
    L.set_option("OTAL_CONV_NO1X1STREAM", 0)


def _t(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.randn(*shape) * scale).astype(np.float32)).cuda()


def _sliced(t, lo, hi):
    B, C = t.shape[:2]
    big = torch.full((B, lo + C + hi) + tuple(t.shape[2:]), 7.0, dtype=t.dtype, device=t.device)
    big[:, lo:lo + C] = t
    return big[:, lo:lo + C]


def _within_one_bf16_step(got, ref):
    """|got - ref| <= one bf16 spacing at |ref| (2^-7 relative: a rounding step, in case the fp32 sums straddle a boundary)."""
    got, ref = got.float(), ref.float()
    tol = ref.abs() * 2.0 ** -7 + 1e-6 * float(ref.abs().max())
    bad = ((got - ref).abs() > tol) | torch.isnan(got)          # (a NaN compares False: count it explicitly)
    return int(bad.sum()), float(((got - ref).abs() / tol).max())


def _close_to_the_old_kernel(new, old):
    """bf16 tensors: equal except for rare differences of one rounding step of the value BEFORE the shift was added (an
    output close to zero is the difference of two larger numbers: its own spacing is not the yardstick)."""
    diff = new != old
    n = int(diff.sum())
    assert n <= max(4, new.numel() // 500), (n, new.numel())
    if n:
        a, b = new[diff].float(), old[diff].float()
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(old.float().abs().max())


@pytest.mark.parametrize("shape,cout", CASES)
@pytest.mark.parametrize("sliced", [False, True])
def test_forward(shape, cout, sliced):
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    xh = _t(rs, *shape).to(BF)
    w = _t(rs, cout, shape[1], 1, 1, 1, scale=0.05)
    sc, sh = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)).cuda(), _t(rs, cout, scale=0.3)
    assert ops.half_storage_ok(0, shape, cout, ONE, ONE, both=True)

    def run():
        if sliced:
            big = torch.zeros((shape[0], cout + 24) + tuple(shape[2:]), dtype=BF, device="cuda")
            y = ops.conv_forward(_sliced(xh, 8, 16), w, ONE, ONE, scale=sc, shift=sh, relu=True, out=big[:, 16:16 + cout])
            assert float(big[:, :16].abs().max()) == 0 and float(big[:, 16 + cout:].abs().max()) == 0
            return y
        return ops.conv_forward(xh, w, ONE, ONE, scale=sc, shift=sh, relu=True)
    y = run()
    assert y.dtype == BF
    ref = F.conv3d(xh.float(), w.to(BF).float())                # torch's convolution of the same bf16-valued operands, fp32
    ref = torch.relu(ref * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
    nbad, worst = _within_one_bf16_step(y, ref)
    assert nbad == 0, (nbad, worst)
    L.set_option("OTAL_CONV_NO1X1STREAM", 1)
    y_old = run()
    L.set_option("OTAL_CONV_NO1X1STREAM", 0)
    _close_to_the_old_kernel(y, y_old)


@pytest.mark.parametrize("shape,cout", CASES)
@pytest.mark.parametrize("masked", [False, True])
def test_data_gradient(shape, cout, masked):
    from opental_amd import _lib as L
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout + 1)
    B, cin = shape[:2]
    dyh = _t(rs, B, cout, *shape[2:]).to(BF)
    w = _t(rs, cout, cin, 1, 1, 1, scale=0.05)
    xh = torch.relu(_t(rs, *shape)).to(BF)
    esc = torch.from_numpy((rs.rand(cin) + 0.5).astype(np.float32)).cuda()
    assert ops.half_storage_ok(1, shape, cout, ONE, ONE, both=True)

    def run():
        big = torch.zeros((B, cin + 16) + tuple(shape[2:]), dtype=BF, device="cuda")
        kw = dict(out_mask=_sliced(xh, 8, 8), out_scale=esc) if masked else {}
        dx = ops.conv_dgrad(_sliced(dyh, 16, 8), w, shape, ONE, ONE, out=big[:, 8:8 + cin], **kw)
        assert float(big[:, :8].abs().max()) == 0 and float(big[:, 8 + cin:].abs().max()) == 0
        return dx
    dx = run()
    assert dx.dtype == BF
    ref = F.conv_transpose3d(dyh.float(), w.to(BF).float())
    if masked:
        ref = ref * (xh > 0) * esc.view(1, -1, 1, 1, 1)
    nbad, worst = _within_one_bf16_step(dx, ref)
    assert nbad == 0, (nbad, worst)
    L.set_option("OTAL_CONV_NO1X1STREAM", 1)
    dx_old = run()
    L.set_option("OTAL_CONV_NO1X1STREAM", 0)
    _close_to_the_old_kernel(dx, dx_old)
