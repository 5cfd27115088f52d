"""bf16 STORAGE of the backbone's activations and data gradients (precision bits 2 + 3 of the C ABI, otal_maxpool3d_*_io,
otal_convert_storage; ops.HALF_STORAGE): every kernel of the chain against the fp32-tensor kernel it was derived from.

The H kernels change loaders and epilogues only: the MFMA operands are the bf16 values themselves where the fp32-tensor
kernels round fp32 values to bf16, the accumulation order is untouched, and the output is rounded to nearest even once.
So for inputs that ARE bf16 values the contract is exact: H(x) == rne(F(float(x))) bit for bit -- checked here on the
layer shapes of the model (scaled down in T and batch), on channel-sliced views (Inception concat buffers) and with the
ReLU mask read from the bf16 activation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(autouse=True)
def _bf16_mode():
    from opental_amd.common import ops
    old = (ops.CONV_PRECISION, ops.HALF_STORAGE)
    ops.CONV_PRECISION, ops.HALF_STORAGE = 1, True
    yield
    ops.CONV_PRECISION, ops.HALF_STORAGE = old


def _t(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.randn(*shape) * scale).astype(np.float32)).cuda()


def _sliced(t, pad_lo, pad_hi):
    """t copied into channels [pad_lo, pad_lo + C) of a wider buffer: a channel-sliced VIEW with t's values."""
    B, C = t.shape[:2]
    big = torch.full((B, pad_lo + C + pad_hi) + tuple(t.shape[2:]), 7.0, dtype=t.dtype, device=t.device)
    big[:, pad_lo:pad_lo + C] = t
    return big[:, pad_lo:pad_lo + C]


# (x shape, Cout, kernel): 1x1x1 layers on 24^2 / 12^2 / 6^2 planes (chunked kernel, 4-position loads), the direct 3x3x3 kernel
# with 96 / 64 / 32-row tiles and 512 / 256 / 128-position tiles, and the chunked kernel's 3x3x3 form (Cin = 24: 2-position loads)
FWD_CASES = [((2, 64, 8, 24, 24), 64, 1), ((1, 192, 4, 12, 12), 176, 1), ((1, 480, 8, 6, 6), 304, 1), ((1, 256, 4, 12, 12), 64, 1),
             ((2, 64, 8, 24, 24), 192, 3), ((1, 96, 16, 12, 12), 128, 3), ((1, 16, 16, 12, 12), 32, 3), ((1, 96, 64, 6, 6), 208, 3),
             ((1, 32, 64, 6, 6), 64, 3), ((1, 24, 64, 6, 6), 64, 3), ((1, 128, 8, 12, 12), 192, 3)]


@pytest.mark.parametrize("shape,cout,k", FWD_CASES)
@pytest.mark.parametrize("sliced", [False, True])
def test_forward_on_bf16_tensors_equals_the_rounded_fp32_tensor_kernel(shape, cout, k, sliced):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout + k)
    kk = (k, k, k)
    xh = _t(rs, *shape).to(BF)
    w = _t(rs, cout, shape[1], k, k, k, scale=0.05)
    sc, sh = (torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)).cuda(), _t(rs, cout))
    if not ops.half_storage_ok(0, shape, cout, kk, (1, 1, 1), both=True):
        pytest.skip("no bf16-tensor kernel for this geometry")
    y_ref = ops.conv_forward(xh.float(), w, kk, (1, 1, 1), scale=sc, shift=sh, relu=True)
    if sliced:
        xin = _sliced(xh, 8, 16)
        big = torch.zeros((shape[0], cout + 24) + tuple(y_ref.shape[2:]), dtype=BF, device="cuda")
        y = ops.conv_forward(xin, w, kk, (1, 1, 1), scale=sc, shift=sh, relu=True, out=big[:, 16:16 + cout])
        assert float(big[:, :16].abs().max()) == 0 and float(big[:, 16 + cout:].abs().max()) == 0      # neighbours untouched
    else:
        y = ops.conv_forward(xh, w, kk, (1, 1, 1), scale=sc, shift=sh, relu=True)
    assert y.dtype == BF and torch.equal(y, y_ref.to(BF))


@pytest.mark.parametrize("shape,cout,k", FWD_CASES)
@pytest.mark.parametrize("masked", [False, True])
def test_data_gradient_on_bf16_tensors_equals_the_rounded_fp32_tensor_kernel(shape, cout, k, masked):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout + k + 1)
    kk = (k, k, k)
    B, cin = shape[:2]
    dyh = _t(rs, B, cout, *shape[2:]).to(BF)
    w = _t(rs, cout, cin, k, k, k, scale=0.05)
    if not ops.half_storage_ok(1, shape, cout, kk, (1, 1, 1), both=True):
        pytest.skip("no bf16-tensor kernel for this geometry")
    xh = torch.relu(_t(rs, *shape)).to(BF)                     # the activation whose ReLU mask the epilogue applies
    esc = torch.from_numpy((rs.rand(cin) + 0.5).astype(np.float32)).cuda()
    kw = dict(out_mask=xh.float(), out_scale=esc) if masked else {}
    dx_ref = ops.conv_dgrad(dyh.float(), w, shape, kk, (1, 1, 1), **kw)
    # channel-sliced on both sides, as in the Inception backward: dy a slice of the module's gradient buffer, dx a slice of dh
    dyv = _sliced(dyh, 16, 8)
    big = torch.zeros((B, cin + 16) + tuple(shape[2:]), dtype=BF, device="cuda")
    mk = _sliced(xh, 8, 8) if masked else None
    kw = dict(out_mask=mk, out_scale=esc) if masked else {}
    dx = ops.conv_dgrad(dyv, w, shape, kk, (1, 1, 1), out=big[:, 8:8 + cin], **kw)
    assert dx.dtype == BF and torch.equal(dx, dx_ref.to(BF))
    assert float(big[:, :8].abs().max()) == 0 and float(big[:, 8 + cin:].abs().max()) == 0


# direct 3x3x3 weight gradient (24^2 rows of 4 / 8, whole 12^2 planes, four 6^2 planes per step), the wide 1x1 kernel, the vector
# kernel with 8 / 4 / 2-position groups (small channel counts)
WGRAD_CASES = [((2, 64, 8, 24, 24), 192, 3), ((1, 64, 4, 12, 24), 64, 3), ((1, 96, 16, 12, 12), 128, 3), ((1, 96, 64, 6, 6), 208, 3),
               ((2, 64, 8, 24, 24), 64, 1), ((1, 192, 4, 12, 12), 176, 1), ((1, 480, 8, 6, 6), 304, 1),
               ((1, 16, 16, 12, 12), 32, 3), ((1, 24, 64, 6, 6), 64, 3), ((1, 32, 4, 24, 24), 48, 3)]


@pytest.mark.parametrize("shape,cout,k", WGRAD_CASES)
def test_weight_gradient_from_bf16_tensors_equals_the_fp32_tensor_kernel(shape, cout, k):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout + k + 2)
    kk = (k, k, k)
    B, cin = shape[:2]
    xh = _t(rs, *shape).to(BF)
    dyh = _t(rs, B, cout, *shape[2:]).to(BF)
    if not ops.half_storage_ok(2, shape, cout, kk, (1, 1, 1), both=True):
        pytest.skip("no bf16-tensor kernel for this geometry")
    dw_ref = ops.conv_wgrad(xh.float(), dyh.float(), (cout, cin, k, k, k), kk, (1, 1, 1))
    dw = ops.conv_wgrad(_sliced(xh, 8, 8), _sliced(dyh, 24, 8), (cout, cin, k, k, k), kk, (1, 1, 1))
    assert dw.dtype == torch.float32 and torch.equal(dw, dw_ref)      # same bf16 operands, same summation order


@pytest.mark.parametrize("shape,k,s", [((2, 5, 6, 48, 48), (1, 3, 3), (1, 2, 2)), ((1, 8, 4, 24, 24), (1, 3, 3), (1, 2, 2)),
                                       ((2, 6, 8, 12, 12), (3, 3, 3), (2, 2, 2)), ((1, 3, 6, 12, 12), (3, 3, 3), (2, 2, 2))])
def test_strided_pools_on_bf16_tensors(shape, k, s):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape))
    x32 = _t(rs, *shape)
    xh = x32.to(BF)
    y_ref, arg_ref, bits_ref = ops.maxpool3d_forward(xh.float(), k, s, signbits=True)
    y, arg, bits = ops.maxpool3d_forward(xh, k, s, signbits=True, half_out=True)
    assert y.dtype == BF and torch.equal(y.float(), y_ref) and torch.equal(arg, arg_ref) and torch.equal(bits, bits_ref)
    y32, _, _ = ops.maxpool3d_forward(x32, k, s, signbits=True)
    assert torch.equal(y, y32.to(BF))                          # pooling commutes with the rounding
    dyh = _t(rs, *y.shape).to(BF)
    scale = torch.from_numpy((rs.rand(shape[1]) + 0.5).astype(np.float32)).cuda()
    dx_ref = ops.maxpool3d_backward(dyh.float(), arg, shape, k, s, out_scale=scale, out_signbits=bits)
    dx = ops.maxpool3d_backward(dyh, arg, shape, k, s, out_scale=scale, out_signbits=bits)
    assert dx.dtype == BF and torch.equal(dx, dx_ref.to(BF))


@pytest.mark.parametrize("shape", [(2, 16, 8, 12, 12), (1, 8, 20, 12, 12), (2, 24, 16, 6, 6), (1, 8, 44, 6, 6)])
def test_branch_pools_on_bf16_tensors_and_the_accumulating_backward(shape):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape))
    k, s = (3, 3, 3), (1, 1, 1)
    xh = torch.relu(_t(rs, *shape)).to(BF)
    y_ref, arg_ref = ops.maxpool3d_forward(xh.float(), k, s)
    xv = _sliced(xh, 8, 8)
    y, arg = ops.maxpool3d_forward(xv, k, s, half_out=True)
    assert y.dtype == BF and torch.equal(y.float(), y_ref) and torch.equal(arg, arg_ref)
    dyh = _t(rs, *shape).to(BF)
    scale = torch.from_numpy((rs.rand(shape[1]) + 0.5).astype(np.float32)).cuda()
    first = _t(rs, *shape).to(BF)                              # what the fused 1x1 data gradient stored before
    ref = first.float().clone()
    ops.maxpool3d_backward(dyh.float(), arg, shape, k, s, out=ref, accumulate=True, out_mask=xh.float(), out_scale=scale)
    big = torch.zeros((shape[0], shape[1] + 16) + tuple(shape[2:]), dtype=BF, device="cuda")
    dx = big[:, 8:8 + shape[1]]
    dx.copy_(first)
    ops.maxpool3d_backward(dyh, arg, shape, k, s, out=dx, accumulate=True, out_mask=xv, out_scale=scale)
    assert torch.equal(dx, ref.to(BF))                         # read bf16, add in fp32, round once
    plain_ref = ops.maxpool3d_backward(dyh.float(), arg, shape, k, s, out_mask=xh.float(), out_scale=scale)
    plain = ops.maxpool3d_backward(dyh, arg, shape, k, s, out_mask=xh, out_scale=scale)
    assert torch.equal(plain, plain_ref.to(BF))


def test_storage_conversion_of_channel_slices():
    from opental_amd.common import ops
    rs = np.random.RandomState(5)
    x = _t(rs, 2, 40, 4, 6, 6)
    big = torch.zeros((2, 64, 4, 6, 6), dtype=BF, device="cuda")
    ops.convert_storage(x, BF, out=big[:, 8:48])
    assert torch.equal(big[:, 8:48], x.to(BF)) and float(big[:, :8].abs().max()) == 0 and float(big[:, 48:].abs().max()) == 0
    back = ops.convert_storage(big[:, 8:48], torch.float32)
    assert back.dtype == torch.float32 and torch.equal(back, x.to(BF).float())


@pytest.mark.parametrize("shape", [(2, 5, 6, 48, 48), (1, 8, 4, 24, 24), (1, 3, 2, 16, 8), (2, 4, 3, 48, 96)])
def test_ordered_key_pool_equals_the_scanning_kernel_on_relu_outputs(shape):
    """Round 5: for the output of a conv + ReLU -- what MaxPool3d_2a / 3a read -- the (1,3,3)/(1,2,2) pool
    runs on ordered integer keys (io bit 2).  Values, winner bytes and sign bits must be those of the scanning kernel, ties
    included (half of the inputs are exact zeros, the rest bf16 values with many equal neighbours), at plane borders (SAME
    padding = a zero row / column that may tie with real zeros) and on channel slices."""
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape))
    x = torch.relu(_t(rs, *shape)).to(BF)
    x = (x * 4).round() / 4                                     # few distinct values: ties inside most windows
    x = x.to(BF)
    k, s = (1, 3, 3), (1, 2, 2)
    y0, a0, b0 = ops.maxpool3d_forward(x, k, s, signbits=True, half_out=True)
    xv = _sliced(x, 8, 8)
    y1, a1, b1 = ops.maxpool3d_forward(xv, k, s, signbits=True, half_out=True, nonneg=True)
    assert y1.dtype == BF and torch.equal(y1, y0) and torch.equal(a1, a0) and torch.equal(b1, b0)
    assert int((a1 == 255).sum()) == 0                          # the padding never wins over a non-negative tap 0
    ref = torch.nn.functional.max_pool3d(torch.nn.functional.pad(x.float(), (0, 1, 0, 1, 0, 0)), k, s)
    assert torch.equal(y1.float(), ref)
