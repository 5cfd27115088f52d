"""The training step is meant to be bit-reproducible (fixed-order split-K / bucket reductions, gather backward passes, no
atomics).  A data race shows up as a run that differs: repeat forward + loss + backward from identical weights on the same
clip and require identical costs and an identical gradient arena every time (this is how the missing barrier in
conv1a_direct_fwd_kernel's weight-slice ring was found: 2-10 % of the runs had one wave reading a slice too late)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch,iters,prec", [(1, 40, 1), (2, 16, 1), (1, 8, 0)])
def test_repeated_forward_backward_is_bit_identical(batch, iters, prec):
    import bench
    from opental_amd.common import ops
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = prec
    try:
        dev = torch.device("cuda", 0)
        tr = bench.build_trainer(dev)
        clips, targets, scores = bench.synth_batch(batch, 1000, dev)
        ibm = tr._ibm_state()
        ibm0 = None if ibm is None else ibm.detach().clone()
        first = None
        for it in range(iters):
            if ibm0 is not None:
                tr._ibm_state().copy_(ibm0)
            tr.arena.grad.zero_()
            ops.activate_prologues(tr._prologues)
            try:
                cost, _ = tr.compute_cost(clips, targets, scores)
                tr.begin_backward(early=True)
                cost.backward()
                tr.end_backward()
            finally:
                ops.deactivate_prologues(); ops.GRAD_SLOTS = None; ops.GRAD_READY = None
            torch.cuda.synchronize()
            cur = (float(cost.detach()), tr.arena.grad.detach().clone())
            if first is None:
                first = cur
                continue
            assert cur[0] == first[0], f"run {it}: cost {cur[0]} vs {first[0]}"
            assert torch.equal(cur[1], first[1]), f"run {it}: {int((cur[1] != first[1]).sum())} gradient elements differ"
    finally:
        ops.CONV_PRECISION = old


@pytest.mark.parametrize("batch,prec", [(2, 1), (1, 0)])
def test_weight_gradients_on_the_side_stream_change_nothing(batch, prec):
    """ops.SideWgrads: the backbone's weight gradients run on a second stream beside the data-gradient chain.  Same
    kernels in the same order within each family: cost and gradient arena equal the one-stream run's, bit for bit, on
    every repetition (a missing wait between the two streams would show up as a differing run)."""
    import bench
    from opental_amd.common import ops
    old = (ops.CONV_PRECISION, ops.WGRAD_STREAM)
    ops.CONV_PRECISION = prec
    try:
        dev = torch.device("cuda", 0)
        tr = bench.build_trainer(dev)
        clips, targets, scores = bench.synth_batch(batch, 1000, dev)
        ibm = tr._ibm_state()
        ibm0 = None if ibm is None else ibm.detach().clone()
        runs = []
        for side in (False, True, True, False, True, True, True, True):
            ops.WGRAD_STREAM = side
            if ibm0 is not None:
                tr._ibm_state().copy_(ibm0)
            tr.arena.grad.zero_()
            ops.activate_prologues(tr._prologues)
            try:
                cost, _ = tr.compute_cost(clips, targets, scores)
                tr.begin_backward(early=True)
                cost.backward()
                tr.end_backward()
            finally:
                ops.deactivate_prologues(); ops.GRAD_SLOTS = None; ops.GRAD_READY = None
            torch.cuda.synchronize()
            runs.append((side, float(cost.detach()), tr.arena.grad.detach().clone()))
        for side, c, g in runs[1:]:
            assert c == runs[0][1]
            assert torch.equal(g, runs[0][2]), f"side={side}: {int((g != runs[0][2]).sum())} gradient elements differ"
    finally:
        ops.CONV_PRECISION, ops.WGRAD_STREAM = old
