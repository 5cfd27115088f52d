"""The pyramid as two hand-scheduled autograd nodes (opental_amd/thumos14/pyramid_fused.py) against the module-by-module
composition of AFSD/thumos14/BDNet.py:295-432 it replaces -- same kernels, so forward values are bit-identical and
gradients agree to fp32 re-association (a gradient that autograd used to add with separate kernels is now added inside
the consuming launch): cost, every output of the dict, every parameter gradient and the clip gradient, in the fp32 parity
mode and in the bf16-operand mode, at b = 2.  (The module-by-module path itself is pinned to the reference's golden
vectors in tests/test_model_gpu.py.)"""
import os

import numpy as np
import pytest
import torch

from oracle import arch

pytestmark = pytest.mark.gpu

EDL = dict(evidence='exp', loss_type='log', iou_aware=True, with_focal=False, alpha=0.25, gamma=2, with_ibm=True,
           ibm_start=10, momentum=0.99, num_bins=50)
W = dict(lw=1.0, cw=10.0, ctw=1.0, actw=1.0, ssl=0.001)


def _run(fx, fused, prec):
    from opental_amd.common import ops
    from opental_amd.thumos14.BDNet import BDNet
    from opental_amd.thumos14.multisegment_loss import MultiSegmentLoss
    from opental_amd.thumos14.train import forward_one_epoch, total_cost
    b = 2
    from opental_amd.thumos14 import pyramid_fused as PF
    old = (ops.FUSED_PYRAMID, ops.CONV_PRECISION)
    ops.FUSED_PYRAMID, ops.CONV_PRECISION = fused, prec
    calls = []
    real_trunk, real_branches = PF.trunk, PF.branches
    PF.trunk = lambda *a: (calls.append("trunk"), real_trunk(*a))[1]
    PF.branches = lambda *a: (calls.append("branches"), real_branches(*a))[1]
    try:
        net = BDNet(training=False, use_edl=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in arch.make_params(int(fx["param_seed"])).items()})
        net = net.cuda().train()
        x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), b)).cuda()
        targets = [torch.from_numpy(fx[f"target_{i}"]).cuda() for i in range(b)]
        scores = torch.from_numpy(fx["scores"]).cuda()
        crit = MultiSegmentLoss(15, 0.5, 1.0, cls_loss_type='edl', edl_config=EDL, os_head=True,
                                act_config=dict(margin=1.0, weight=0)).cuda()
        with torch.no_grad():
            out = {k: v.detach().clone() for k, v in net(x).items() if torch.is_tensor(v)}
        feats = net.backbone(x)
        leaf = {k: v.detach().requires_grad_(True) for k, v in feats.items()}     # the pyramid alone: gradients w.r.t. its inputs
        net.backbone.forward = lambda _x: leaf
        losses = forward_one_epoch(net, crit, x, targets, scores, training=True, ssl=False)
        cost = total_cost(losses, W)
        cost.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        for k, v in leaf.items():
            grads["input." + k] = v.grad.detach().clone()
        assert calls == (["trunk", "branches"] * 2 if fused else []), calls     # (the no-grad forward + the training forward)
        return float(cost.detach()), out, grads
    finally:
        ops.FUSED_PYRAMID, ops.CONV_PRECISION = old
        PF.trunk, PF.branches = real_trunk, real_branches


@pytest.mark.parametrize("prec", [0, 1])
def test_fused_pyramid_nodes_equal_the_module_by_module_path(golden_dir, prec):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    fx = np.load(os.path.join(golden_dir, "thumos_b2.npz"))
    cost_a, out_a, g_a = _run(fx, False, prec)
    cost_b, out_b, g_b = _run(fx, True, prec)
    assert abs(cost_a - cost_b) <= 1e-6 * abs(cost_a), (cost_a, cost_b)
    for k in out_a:
        assert torch.equal(out_a[k], out_b[k]), k          # same kernels, same operands: forward values do not move
    assert set(g_a) == set(g_b)
    worst = ("", 0.0)
    for n in g_a:
        a, b = g_a[n].double(), g_b[n].double()
        scale = float(a.abs().max())
        err = float((a - b).abs().max())
        if scale > 0 and err / scale > worst[1]:
            worst = (n, err / scale)
        assert err <= 2e-5 * max(scale, 1e-12), (n, err, scale)
    print("worst gradient difference:", worst)


@pytest.mark.parametrize("chunk", [3, 5, 6])
def test_lane_graph_capture_survives_any_weight_gradient_chunk_size(chunk):
    """A lane-graph capture is CUT wherever the weight-gradient lane takes a chunk (ops.SideWgrads.flush), and a capture
    cannot end while a forked stream has not rejoined it: the fused nodes must not flush between a branch-lane fork and its
    join.  With the default chunk (4) the counts happened to work out; 5 and 6 crashed hipStreamEndCapture.  Every chunk size
    must capture, and the replayed steps must leave the parameters of eager steps."""
    import bench
    from opental_amd.common import ops
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda", 0)
    old = (ops.CONV_PRECISION, ops.SideWgrads.CHUNK)
    ops.CONV_PRECISION = 1
    try:
        batch = bench.synth_batch(2, 1000, dev)
        ops.SideWgrads.CHUNK = 4
        ref = bench.build_trainer(dev, seed=21)
        for _ in range(3):
            ref.step(*batch)
        ops.SideWgrads.CHUNK = chunk
        tr = bench.build_trainer(dev, seed=21)
        tr.step(*batch)
        tr.capture_step(*batch, warmup=0, lanes=True)
        for _ in range(2):
            tr.step(*batch)
        torch.cuda.synchronize()
        assert tr.replayed_steps == 2
        assert torch.equal(tr.arena.flat, ref.arena.flat) and torch.equal(tr.arena.m, ref.arena.m)
    finally:
        ops.CONV_PRECISION, ops.SideWgrads.CHUNK = old
