"""GPU: the bf16 BACKWARD pinned LAYER BY LAYER IN ISOLATION (reference layers
AFSD/common/i3d_backbone.py:7-87 Unit3D, :90-121 InceptionModule, AFSD/common/layers.py:9-35 MaxPool3dSamePadding).

The end-to-end pin (test_bf16_parity_gpu.py) compares gradients after ~60 layers of compounding: a different fp32
summation order moves a few activations across bf16 rounding boundaries, pool ties move with them, and by Conv3d_1a the
cosine against the oracle is 0.80 -- a bound that cannot see a 10 % error in ONE early kernel.  Here every backbone block
runs ALONE, at the benchmark's b = 8 shapes, on the oracle's own data:

  * input  = the operand-rounding oracle's activation in front of the block (its b = 8 forward pass), as the benchmarked
    path stores it (bf16 between Conv3d_1a and Mixed_4f, fp32 elsewhere);
  * upstream gradient = a seeded normal tensor, bf16-representable, w.r.t. the block's output;
  * yardstick = `O.conv3d_bn_relu` / `O.mixed` / `O.maxpool3d_same` of that block under
    `operand_rounding("bf16", grads=True)` with the stored-pool semantics, differentiated by autograd on the CPU;
  * the HIP block = the SAME launches the training step issues for it (I3DFeaturesFunction on the block's slice of the
    plan: bf16-tensor kernels, fused 1x1 launch, direct 3x3x3 / planes6 / 1a kernels, pool kernels with sign bits).

With identical operands the two differ by fp32 summation order only (plus one bf16 rounding of a stored dx, 2^-9 relative
per element, and the rare intermediate that rounds the other way inside a module).  Measured: every weight gradient and data
gradient of every block at cosine >= 0.99998 (the end-to-end pin: 0.80 at Conv3d_1a), relative norm error <= 1.4e-3, and
weight gradients that are bit-equal sums wherever no intermediate is involved -- figures and bounds next to BOUNDS below.
tools/break_dgrad_tap.sh runs this file against a library with a mis-routed data-gradient tap: it fails on the first
3x3x3 layer."""
import os

import numpy as np
import pytest
import torch

from oracle import afsd_oracle as O
from oracle import arch

pytestmark = pytest.mark.gpu
BATCH = 8
# (name, plan steps by endpoint name): a conv / module alone; a strided pool rides with the layer in front of it (it applies
# that layer's ReLU mask from sign bits), MaxPool3d_5a with the module behind it
BLOCKS = [
    ("Conv3d_1a", ["Conv3d_1a_7x7", "MaxPool3d_2a_3x3"]),
    ("Conv3d_2b", ["Conv3d_2b_1x1"]),
    ("Conv3d_2c", ["Conv3d_2c_3x3", "MaxPool3d_3a_3x3"]),
    ("Mixed_3b", ["Mixed_3b"]),
    ("Mixed_3c", ["Mixed_3c", "MaxPool3d_4a_3x3"]),
    ("Mixed_4b", ["Mixed_4b"]),
    ("Mixed_4c", ["Mixed_4c"]),
    ("Mixed_4d", ["Mixed_4d"]),
    ("Mixed_4e", ["Mixed_4e"]),
    ("Mixed_4f", ["Mixed_4f"]),
    ("Mixed_5b", ["MaxPool3d_5a_2x2", "Mixed_5b"]),
    ("Mixed_5c", ["Mixed_5c"]),
    # two layers in a row: the second one's data gradient carries the first one's ReLU mask / BN scale in its epilogue
    ("Conv3d_2b+2c", ["Conv3d_2b_1x1", "Conv3d_2c_3x3", "MaxPool3d_3a_3x3"]),
    ("Mixed_4e+4f", ["Mixed_4e", "Mixed_4f"]),
]
STORED_LAST = "Mixed_4f"        # the bf16-stored region of the benchmarked path: Conv3d_1a's output .. Mixed_4f's output


@pytest.fixture(scope="module")
def world():
    """Parameters, the oracle's b = 8 activations at every endpoint (bf16-operand forward), and the HIP model."""
    from opental_amd.thumos14.BDNet import BDNet
    torch.set_num_threads(min(32, os.cpu_count()))
    params = arch.make_params(2020)
    P = O.to_torch(params)
    x = torch.from_numpy(arch.make_clip(31, BATCH))
    with torch.no_grad(), O.operand_rounding("bf16", grads=True, stored_until=STORED_LAST):
        acts = O.i3d_features(P, x, endpoints=None)
    acts = dict(acts)
    acts["__clip__"] = x
    net = BDNet(training=False, use_edl=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.cuda().train()
    model = net.backbone._model
    plan, units = model._make_plan()
    folded = [u.folded_bn() for u in units]
    offs = [0]
    for sc, _ in folded:
        offs.append(offs[-1] + sc.numel())
    fold = (torch.cat([sc for sc, _ in folded]), torch.cat([sh for _, sh in folded]), offs)
    return dict(params=params, acts=acts, net=net, plan=plan, units=units, fold=fold)


def _oracle_block(P, names, x):
    kinds = {n: (k, a) for n, k, a in arch.I3D_ENDPOINTS}
    for n in names:
        kind, args = kinds[n]
        if kind == "conv":
            x = O.conv3d_bn_relu(P, f"backbone._model.{n}", x, args[2], args[3])
        elif kind == "pool":
            x = O.maxpool3d_same(x, *args)
        else:
            x = O.mixed(P, f"backbone._model.{n}", x)
    return x


def _cmp(got, want):
    a, b = got.detach().double().cpu().flatten(), want.detach().double().cpu().flatten()
    return (abs(float(a.norm() / b.norm().clamp(min=1e-30)) - 1.0), float(torch.nn.functional.cosine_similarity(a, b, dim=0)),
            float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt().clamp(min=1e-30)))


@pytest.mark.parametrize("block", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_block_backward_alone_matches_the_operand_rounding_oracle(world, block):
    from opental_amd.common import ops
    from opental_amd.common.i3d_backbone import I3DFeaturesFunction
    label, names = block
    order = [n for n, _, _ in arch.I3D_ENDPOINTS]
    i0, i1 = order.index(names[0]), order.index(names[-1])
    assert [n for n in order[i0:i1 + 1]] == names
    stored = lambda i: 0 <= i <= order.index(STORED_LAST)       # is endpoint i's output a bf16-stored tensor?
    src = world["acts"]["__clip__"] if i0 == 0 else world["acts"][order[i0 - 1]]
    x_half = i0 > 0 and stored(i0 - 1)
    x_in = src.to(torch.bfloat16).float() if x_half else src.clone()
    # ---- oracle: the block alone, differentiated on the CPU
    keys = [k for k in world["params"] if any(k.startswith(f"backbone._model.{n}.") for n in names)]
    P = O.to_torch({k: world["params"][k] for k in keys}, requires_grad=True)
    xo = x_in.clone().requires_grad_(i0 > 0)
    saved = O._STORED_NOW
    with O.operand_rounding("bf16", grads=True, stored_until=STORED_LAST):
        O._STORED_NOW = stored(i0)                              # pools of the stored region pick winners among rounded values
        try:
            y = _oracle_block(P, names, xo)
            dy = torch.from_numpy(np.random.RandomState(1000 + i0).randn(*y.shape).astype(np.float32)).to(torch.bfloat16).float()
            y.backward(dy)
        finally:
            O._STORED_NOW = saved
    want = {k: v.grad for k, v in P.items() if v.grad is not None}
    # ---- HIP: the block's slice of the training step's plan
    plan, units, (scale, shift, offs) = world["plan"], world["units"], world["fold"]
    weights = [u.conv3d.weight for u in units]
    for w in weights:
        w.grad = None
    old = (ops.CONV_PRECISION, ops.HALF_STORAGE, ops.HALF_CHAIN)
    ops.CONV_PRECISION, ops.HALF_STORAGE, ops.HALF_CHAIN = 1, True, True
    try:
        xg = x_in.cuda().to(torch.bfloat16) if x_half else x_in.cuda()
        xg.requires_grad_(i0 > 0)
        (yg,) = I3DFeaturesFunction.apply(xg, plan[i0:i1 + 1], (names[-1],), scale, shift, offs, *weights)
        assert yg.dtype == torch.float32
        yg.backward(dy.cuda())
        torch.cuda.synchronize()
    finally:
        ops.CONV_PRECISION, ops.HALF_STORAGE, ops.HALF_CHAIN = old
    # forward first: same operands, fp32 accumulation -> a few bf16 ulps of the scale at most
    fwd = _cmp(yg, y)
    assert fwd[1] > 0.99999 and fwd[2] < 2e-3, (label, "forward", fwd)
    report = {}
    named = dict(world["net"].named_parameters())
    for k, g_want in want.items():
        g = named[k].grad
        assert g is not None, k
        report[k.replace("backbone._model.", "")] = _cmp(g, g_want)
    if i0 > 0:
        assert xg.grad is not None and xg.grad.dtype == xg.dtype
        report["dx"] = _cmp(xg.grad.float(), xo.grad)
    print(f"{label}: forward {tuple(round(v, 6) for v in fwd)}; (norm error, cosine, rms error) "
          + str({k: (round(v[0], 6), round(v[1], 7), round(v[2], 6)) for k, v in report.items()}))
    pair = "+" in label
    for k, (ne, cs, rms) in report.items():
        bn, bc, br = BOUNDS[("pair" if pair else "alone", "dx" if k == "dx" else "dw")]
        assert ne < bn and cs > bc and rms < br, (label, k, (ne, cs, rms), (bn, bc, br))


# (norm error <, cosine >, rms error <).  Measured on MI355X at b = 8 (round 5), worst case over the twelve single blocks:
# weight gradients -- 0 / 1.0 / 0 wherever the layer's gradient operand is the upstream gradient itself (Conv3d_2b, every b0 and
# b3b: same bf16 operands, and an fp32 sum that comes out bit-equal), otherwise norm 1.4e-3 (Mixed_4f.b2a), cosine 0.9999967,
# rms 2.6e-3 (Mixed_5c.b1a: 2304 positions, a handful of intermediate activations that round or clip the other way under a
# different summation order -- each a full-magnitude change of one gradient element); data gradients -- norm 8e-6, cosine
# 0.999984 (Mixed_3c + MaxPool3d_4a: pool winners that move with such an element), rms 5.6e-3, of which 1.7e-3 is the single
# bf16 rounding of the stored dx.  Two blocks in a row: norm 1.2e-3, cosine 0.99997, rms 7.3e-3.  Bounds = 2 x those.
# A mis-routed tap (tools/break_dgrad_tap.sh) sends the 3x3x3 layers' dx cosine below 0.95.
BOUNDS = {("alone", "dw"): (3e-3, 0.99999, 5e-3), ("alone", "dx"): (1e-3, 0.99996, 1.1e-2),
          ("pair", "dw"): (3e-3, 0.99994, 1.5e-2), ("pair", "dx"): (1e-3, 0.99994, 1.5e-2)}
