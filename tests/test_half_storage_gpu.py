"""bf16 STORAGE of Conv3d_1a's output and of its gradient (precision bit 2 of the C ABI; ops.HALF_STORAGE):
the tensors are consumed through bf16 roundings only, so every kernel of the chain is checked bit for bit against the fp32
tensors rounded to nearest even, and the model's forward values do not move."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16_mode():
    from opental_amd.common import ops
    old = (ops.CONV_PRECISION, ops.HALF_STORAGE)
    ops.CONV_PRECISION, ops.HALF_STORAGE = 1, True
    yield
    ops.CONV_PRECISION, ops.HALF_STORAGE = old


def _rne(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("shape,cout", [((2, 3, 8, 96, 96), 64), ((1, 3, 4, 8, 96), 40)])
def test_conv1a_bf16_output_is_the_rounded_fp32_output(shape, cout):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(cout, 3, 7, 7, 7) * 0.05).astype(np.float32)).cuda()
    sc = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)).cuda()
    sh = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    assert ops.half_storage_ok(0, shape, cout, (7, 7, 7), (2, 2, 2))
    y32 = ops.conv_forward(x, w, (7, 7, 7), (2, 2, 2), scale=sc, shift=sh, relu=True)
    y16 = ops.conv_forward(x, w, (7, 7, 7), (2, 2, 2), scale=sc, shift=sh, relu=True, half_out=True)
    assert y16.dtype == torch.bfloat16 and y16.shape == y32.shape
    assert torch.equal(y16, _rne(y32))
    assert not ops.half_storage_ok(0, (1, 64, 8, 24, 24), 64, (1, 1, 1), (1, 1, 1))       # other kernels: fp32 tensors
    with pytest.raises(RuntimeError):
        ops.conv_forward(torch.randn(1, 64, 4, 24, 24, device="cuda"), torch.randn(64, 64, 1, 1, 1, device="cuda"), (1, 1, 1), (1, 1, 1),
                         half_out=True)


@pytest.mark.parametrize("shape,cout", [((2, 64, 8, 24, 24), 192), ((1, 64, 4, 8, 16), 192), ((1, 32, 2, 16, 8), 96),
                                        ((1, 96, 16, 12, 12), 208)])
def test_direct_3x3x3_bf16_output_is_the_rounded_fp32_output(shape, cout):
    """Conv3d_2c -> MaxPool3d_3a: the LDS-direct 3x3x3 kernel's bf16 epilogue (affine + ReLU, rounded to nearest even)."""
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(cout, shape[1], 3, 3, 3) * 0.05).astype(np.float32)).cuda()
    sc = torch.from_numpy((rs.rand(cout) + 0.5).astype(np.float32)).cuda()
    sh = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    if not ops.half_storage_ok(0, shape, cout, (3, 3, 3), (1, 1, 1)):
        pytest.skip("geometry not served by the direct kernel")
    y16 = ops.conv_forward(x, w, (3, 3, 3), (1, 1, 1), scale=sc, shift=sh, relu=True, half_out=True)
    y32 = ops.conv_forward(x, w, (3, 3, 3), (1, 1, 1), scale=sc, shift=sh, relu=True)      # same kernel, fp32 epilogue
    assert y16.dtype == torch.bfloat16 and y16.shape == y32.shape
    assert torch.equal(y16, _rne(y32))


@pytest.mark.parametrize("shape", [(2, 5, 6, 48, 48), (1, 3, 2, 6, 8)])
def test_strided_pool_on_bf16_input_and_bf16_gradient(shape):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape))
    x32 = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    xh = _rne(x32)
    k, s = (1, 3, 3), (1, 2, 2)
    y_ref, arg_ref, bits_ref = ops.maxpool3d_forward(xh.float(), k, s, signbits=True)
    y, arg, bits = ops.maxpool3d_forward(xh, k, s, signbits=True)
    assert y.dtype == torch.float32 and torch.equal(y, y_ref) and torch.equal(arg, arg_ref) and torch.equal(bits, bits_ref)
    # max-pool commutes with the (monotonic) rounding: pooling the rounded tensor = rounding the pooled fp32 tensor
    y32, _, _ = ops.maxpool3d_forward(x32, k, s, signbits=True)
    assert torch.equal(y, _rne(y32).float())
    dy = torch.from_numpy(rs.randn(*y.shape).astype(np.float32)).cuda()
    scale = torch.from_numpy((rs.rand(shape[1]) + 0.5).astype(np.float32)).cuda()
    dx_ref = ops.maxpool3d_backward(dy, arg, shape, k, s, out_scale=scale, out_signbits=bits)
    dx = ops.maxpool3d_backward(dy, arg, shape, k, s, out_scale=scale, out_signbits=bits, half_out=True)
    assert dx.dtype == torch.bfloat16 and torch.equal(dx, _rne(dx_ref))


@pytest.mark.parametrize("shape,cout", [((2, 3, 8, 96, 96), 64), ((1, 3, 4, 96, 96), 48)])
def test_conv1a_weight_gradient_from_a_bf16_gradient(shape, cout):
    from opental_amd.common import ops
    rs = np.random.RandomState(sum(shape) + cout)
    B, _, T, H, W = shape
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    dyh = _rne(torch.from_numpy(rs.randn(B, cout, T // 2, H // 2, W // 2).astype(np.float32)).cuda())
    assert ops.half_storage_ok(2, shape, cout, (7, 7, 7), (2, 2, 2))
    dw_ref = ops.conv_wgrad(x, dyh.float(), (cout, 3, 7, 7, 7), (7, 7, 7), (2, 2, 2))
    dw = ops.conv_wgrad(x, dyh, (cout, 3, 7, 7, 7), (7, 7, 7), (2, 2, 2))
    assert torch.equal(dw, dw_ref)          # same bf16 operands, same summation order


OUTSIDE_REL_BOUND, OUTSIDE_REL_MEDIAN = 0.4, 0.2


def test_model_forward_is_unchanged_and_only_backbone_gradients_move(golden_dir):
    """HALF_STORAGE on / off (round 4: with ops.HALF_CHAIN every activation and data gradient between Conv3d_1a and Mixed_4f is
    stored as bf16): identical features and losses, bit for bit -- every consumer rounds its operand to bf16 anyway, max-pools
    commute with the rounding.  Gradients: everything outside the backbone is identical (computed before the backbone's
    backward); inside it a bf16-stored gradient is what its consumer's operand loader would have rounded the fp32 tensor to,
    EXCEPT where a tensor has two producers -- a module's input gradient is stored by the fused 1x1 data gradient and added to
    by the branch pool's backward (read bf16, add in fp32, round once: one more rounding than the fp32-stored run, a 1e-3
    effect) -- and where two elements of a pool window became EQUAL after rounding: the first-maximum rule then routes the
    window's gradient to the other element.  That is the visible effect: ~1-2 % of the windows of each of the twelve pools
    between Mixed_4f and the clip, each a full-magnitude re-routing, both choices valid subgradients of the (unchanged)
    forward function.  It accumulates on the way down -- measured cosines with the fp32-stored run at b = 1: Mixed_4f 0.9996
    .. 0.99996, Mixed_4b 0.986 .. 0.998, Mixed_3b / 3c 0.971 .. 0.997, Conv3d_2c 0.958, 2b 0.911, 1a 0.878 -- against 0.77
    (Conv3d_1a) between the bf16-operand and the fp32 modes themselves (test_bf16_compute_mode_stays_close_to_fp32): inside
    what the bf16 mode already does to these ill-conditioned first-layer gradients.  tests/test_bf16_parity_gpu.py pins the
    bf16 backward against the operand-rounding oracle."""
    from oracle import arch
    from test_model_gpu import build, _criterion, W
    from opental_amd.common import ops
    from opental_amd.thumos14.train import forward_one_epoch, total_cost
    fx = np.load(os.path.join(golden_dir, "thumos_b1.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), 1)).cuda()
    targets = [torch.from_numpy(fx["target_0"]).cuda()]
    scores = torch.from_numpy(fx["scores"]).cuda()

    def run(half):
        ops.HALF_STORAGE = half
        net.zero_grad(set_to_none=True)
        crit = _criterion("edl", 0)
        out = net(x)
        losses = forward_one_epoch(net, crit, x, targets, scores, training=True, ssl=False)
        cost = total_cost(losses, W)
        cost.backward()
        return ({k: v.detach().clone() for k, v in out.items() if v is not None}, float(cost.detach()),
                {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    o0, c0, g0 = run(False)
    o1, c1, g1 = run(True)
    # Round 5: the 1x1x1 layers of the bf16-stored region run on the streaming kernel (csrc/conv1x1_stream.inc), which sums
    # the whole K in one workgroup where the fp32-tensor kernel splits K into slabs at this batch: same products, another
    # fp32 association, so a few activations per layer land one bf16 step apart; ~50 layers later the heads see it at the
    # 1e-3 level of their scale (measured: loc -- an exponential -- 7e-3 of its largest value), well inside what the bf16
    # mode does anyway (test_bf16_parity_gpu.py: 2e-2 against the operand-rounding oracle).
    # (Up to round 4 the two modes were bit-identical; with OTAL_CONV_NO1X1STREAM=1 they still are: checked below.)
    for k in o0:
        a, b = o1[k].double(), o0[k].double()
        if k.startswith("prop_") or k in ("center", "start_loc_prop", "end_loc_prop", "start_conf_prop", "end_conf_prop"):
            # behind BoundaryMaxPooling: an anchor whose proposal window rounds to the neighbouring frame pools other frames (a
            # discrete change, as in test_bf16_parity_gpu.py): bounded at the 95th percentile
            d = ((a - b).abs() / float(b.abs().max())).flatten()
            assert float(torch.quantile(d[:4_000_000].float(), 0.95)) <= 3.5e-2, k
            continue
        assert float((a - b).abs().max()) <= 3e-2 * float(b.abs().max()) + 1e-6, (k, float((a - b).abs().max()), float(b.abs().max()))
        rms = float(((a - b) ** 2).mean().sqrt()) / (float((b ** 2).mean().sqrt()) + 1e-12)
        assert rms <= 2e-2, (k, rms)
    assert abs(c0 - c1) <= 1e-2 * abs(c0), (c0, c1)
    from opental_amd import _lib as L
    L.set_option("OTAL_CONV_NO1X1STREAM", 1)
    try:
        o2, c2, g2 = run(True)
    finally:
        L.set_option("OTAL_CONV_NO1X1STREAM", 0)
    cosine = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
    outside = lambda k: not k.startswith("backbone.") or "Mixed_5" in k      # pyramid, heads, Mixed_5b / 5c: fp32 tensors
    # ---- leg 1, the chunked kernels on bf16-stored tensors (same fp32 association as on fp32 tensors): the storage format is
    # invisible in the forward pass AND in every gradient computed before the backbone's bf16 region -- bit for bit; inside
    # the region only the two documented effects act (one more rounding at two-producer tensors, pool-tie re-routing), with
    # the round-4 bounds (these are not widened)
    for k in o0:
        assert torch.equal(o0[k], o2[k]), k
    assert c0 == c2
    moved2 = [k for k in g0 if not torch.equal(g0[k], g2[k])]
    assert moved2 and not [k for k in moved2 if outside(k)], [k for k in moved2 if outside(k)]
    worst2 = {k: cosine(g0[k], g2[k]) for k in moved2}
    for k, cos in worst2.items():
        assert cos > (0.8 if "Conv3d_" in k else (0.94 if "Mixed_3" in k else 0.97)), (k, cos)
    # ---- leg 2, the default kernels (streaming 1x1x1: the whole K in one workgroup, another fp32 association -- pinned on
    # its own against F.conv3d in tests/test_conv1x1_stream_gpu.py).  A few activations per layer land one bf16 step apart, so
    # gradients OUTSIDE the region now move too: by the forward differences bounded above, i.e. small RELATIVE errors -- a
    # routing or epilogue bug in the new kernel would show as an O(1) error of a whole tensor, not as this
    moved = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert moved
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    out_rel = {k: rel(g1[k], g0[k]) for k in moved if outside(k)}
    worst = {k: cosine(g0[k], g1[k]) for k in moved if not outside(k)}
    print("outside the region, largest relative gradient differences:", sorted(out_rel.items(), key=lambda kv: -kv[1])[:4])
    print("bf16-stored vs fp32-stored gradients, lowest cosines: chunked", sorted(worst2.items(), key=lambda kv: kv[1])[:3],
          "default", sorted(worst.items(), key=lambda kv: kv[1])[:3])
    med = float(np.median(list(out_rel.values())))
    print("outside the region: median relative gradient difference", med, "of", len(out_rel), "tensors")
    # measured (b = 1): median 0.09 (= cosine 0.996), largest 0.19 (the t = 2 pyramid level and Mixed_5c's narrow branch: few anchors, one
    # proposal window that rounds to the neighbouring frame moves them) -- bounds at about twice that
    assert med < OUTSIDE_REL_MEDIAN, med
    for k, r in out_rel.items():
        assert r < OUTSIDE_REL_BOUND, (k, r)
    for k, cos in worst.items():
        # the same two effects + the other association; the kernels themselves are pinned layer by layer at cosine >= 0.99998
        # (tests/test_bf16_layer_pin_gpu.py), these end-to-end figures bound the mode's conditioning
        assert cos > (0.7 if "Conv3d_" in k else (0.9 if "Mixed_3" in k else 0.95)), (k, cos)


@pytest.mark.parametrize("act_direct", [False, True])
def test_half_storage_without_the_chain_runs_forward_and_backward(golden_dir, act_direct):
    """ops.HALF_CHAIN = False is the documented off-switch of the bf16 chain.  Conv3d_1a then still stores
    its output as bf16 for MaxPool3d_2a (round 2's HALF_STORAGE; with ops.HALF_ACT_DIRECT also Conv3d_2c for MaxPool3d_3a): the
    pool's bf16-in / fp32-out kernel must consume it WITHOUT a conversion seam, and the backward pass must come out with the
    dtypes every consumer expects.  Forward values equal the fp32-stored run bit for bit; only Conv3d_1a's (and, with
    HALF_ACT_DIRECT, nothing else's: fp32 gradients there) weight gradient moves, by the pool's tie re-routing (cosine 0.997 at
    round 2; bound 0.99)."""
    from oracle import arch
    from test_model_gpu import build, _criterion, W
    from opental_amd.common import ops
    from opental_amd.thumos14.train import forward_one_epoch, total_cost
    fx = np.load(os.path.join(golden_dir, "thumos_b1.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), 1)).cuda()
    targets = [torch.from_numpy(fx["target_0"]).cuda()]
    scores = torch.from_numpy(fx["scores"]).cuda()
    saved = (ops.HALF_CHAIN, ops.HALF_ACT_DIRECT)

    def run(half, chain, direct):
        ops.HALF_STORAGE, ops.HALF_CHAIN, ops.HALF_ACT_DIRECT = half, chain, direct
        net.zero_grad(set_to_none=True)
        losses = forward_one_epoch(net, _criterion("edl", 0), x, targets, scores, training=True, ssl=False)
        cost = total_cost(losses, W)
        cost.backward()
        return float(cost.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    try:
        c0, g0 = run(False, False, False)
        c1, g1 = run(True, False, act_direct)
    finally:
        ops.HALF_CHAIN, ops.HALF_ACT_DIRECT = saved
    assert c0 == c1
    moved = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    # MaxPool3d_2a routes a tied window differently on rounded inputs: Conv3d_1a's gradient moves; with HALF_ACT_DIRECT
    # MaxPool3d_3a does the same to everything in front of it (Conv3d_2c, 2b, 1a)
    stem = ["backbone._model.Conv3d_1a_7x7.conv3d.weight"] + (
        ["backbone._model.Conv3d_2b_1x1.conv3d.weight", "backbone._model.Conv3d_2c_3x3.conv3d.weight"] if act_direct else [])
    assert moved and set(moved) <= set(stem), moved
    for k in moved:
        cos = float(torch.nn.functional.cosine_similarity(g0[k].flatten().double(), g1[k].flatten().double(), dim=0))
        assert cos > (0.95 if act_direct else 0.99), (k, cos)
