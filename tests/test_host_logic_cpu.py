"""CPU: host-side logic added in round 2's second session -- gradient-slot hand-out / release, the backbone's stem / trunk
cut, and the arena buckets of the two-phase data-parallel step (no kernels involved)."""
import torch

from opental_amd.common import ops
from opental_amd.common.i3d_backbone import InceptionI3d


def test_grad_slots_take_release_and_spanning_views():
    flat = torch.zeros(100)
    grad = torch.arange(100, dtype=torch.float32)
    offsets, numels = [0, 10, 30, 60], [10, 20, 30, 40]
    slots = ops.GradSlots(flat, grad, offsets, numels)
    p1 = flat[10:30].view(4, 5)
    v = slots.take(p1)
    assert v is not None and v.shape == (4, 5) and v.data_ptr() == grad[10:30].data_ptr()
    assert slots.take(p1) is None                       # handed out once per step
    slots.release(p1)
    assert slots.take(p1) is not None                   # ... unless given back unwritten
    span = flat[30:100].view(70)                        # two adjacent parameters read as one fused weight
    w = slots.take(span)
    assert w is not None and w.numel() == 70
    assert slots.take(flat[60:100]) is None             # part of the span: taken
    assert slots.take(flat[5:15]) is None               # not a parameter boundary
    slots.reset()
    assert slots.take(flat[60:100]) is not None


def test_backbone_stem_cut_and_stem_parameters():
    m = InceptionI3d(final_endpoint='Mixed_5c')
    m.build()
    plan, units = m._make_plan()
    cut = m._stem_cut(plan, ('Mixed_4f', 'Mixed_5c'))
    names = [st[-1] for st in plan]
    assert names[cut - 1] == 'MaxPool3d_4a_3x3' and names[cut] == 'Mixed_4b'
    assert m._stem_cut(plan, ('Mixed_3c', 'Mixed_5c')) == 0        # an endpoint inside the stem: no cut
    stem = m.stem_parameters()
    # Conv3d_1a, 2b, 2c + 6 convolutions in each of Mixed_3b / 3c
    assert len(stem) == 3 + 12
    stem_ids = {id(p) for p in stem}
    by_name = dict(m.named_parameters())
    assert id(by_name['Conv3d_1a_7x7.conv3d.weight']) in stem_ids and id(by_name['Mixed_3c.b3b.conv3d.weight']) in stem_ids
    assert id(by_name['Mixed_4b.b0.conv3d.weight']) not in stem_ids
    n_stem = sum(p.numel() for p in stem)
    n_all = sum(p.numel() for n, p in by_name.items() if n.endswith('conv3d.weight'))
    assert 0.08 < n_stem / n_all < 0.15                 # a small share of the parameters, a third of the backward's time


def test_plan_cache_distinguishes_layouts():
    # the plan key holds shapes AND strides: a channel slice of a wider buffer is a different launch than a dense tensor
    x = torch.zeros(2, 8, 16)
    wide = torch.zeros(2, 12, 16)
    k1 = (0, x.shape, x.stride(), x.shape, x.stride(), 8, (1, 1, 1), (1, 1, 1), None, False, x.dtype, x.dtype)
    xs = wide[:, 2:10]
    k2 = (0, xs.shape, xs.stride(), x.shape, x.stride(), 8, (1, 1, 1), (1, 1, 1), None, False, xs.dtype, x.dtype)
    assert k1 != k2 and hash(k1) != hash(k2)


def test_evidence_loss_kinds_match_the_reference_formulas():
    """EvidenceLoss beyond the final recipe (VERDICT r2 missing #6): evidence relu / softplus / exp and loss types log /
    digamma / mse, against the formulas of AFSD/thumos14/cls_loss.py:186-285 restated with gathers over the positives
    (what the reference computes on its boolean-masked rows)."""
    import torch.nn.functional as F
    from opental_amd.thumos14.cls_loss import EvidenceLoss
    g = torch.Generator().manual_seed(4)
    K, N = 15, 64
    logit = torch.randn(N, K, generator=g) * 2
    target = torch.randint(0, K, (N,), generator=g)
    mask = torch.rand(N, generator=g) > 0.4
    for evidence in ('exp', 'relu', 'softplus'):
        ev = {'exp': lambda z: torch.exp(z.clamp(-10, 10)), 'relu': F.relu, 'softplus': F.softplus}[evidence]
        for loss_type in ('log', 'digamma', 'mse'):
            crit = EvidenceLoss(K, dict(loss_type=loss_type, evidence=evidence))
            got = crit(logit, target, mask)
            z, t = logit[mask], target[mask]
            alpha = ev(z) + 1
            S = alpha.sum(1, keepdim=True)
            y = torch.eye(K)[t]
            if loss_type == 'mse':
                want = ((y - alpha / S) ** 2).sum() + (alpha * (S - alpha) / (S * S * (S + 1))).sum()
            else:
                f = torch.log if loss_type == 'log' else torch.digamma
                want = (y * (f(S) - f(alpha))).sum()
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), (evidence, loss_type, float(got), float(want))


def test_step_log_jsonl_sink_writes_the_reference_scalar_tags(tmp_path):
    """StepLog -> JsonlSink: one JSON line per logged step with the scalar tags the reference gives tensorboardX
    (AFSD/thumos14/train.py:249-268); VERDICT r3 missing #4."""
    import json
    import torch
    from opental_amd.thumos14.train import StepLog, JsonlSink
    seen = []
    sink = JsonlSink(str(tmp_path / "training"), echo=lambda s, v: seen.append(s))
    log = StepLog(every=2, sink=sink)
    for step in range(1, 7):
        log.push(step, torch.arange(8, dtype=torch.float32) + step)
    log.poll(wait=True)
    sink.close()
    rows = [json.loads(l) for l in open(tmp_path / "training" / "scalars.jsonl")]
    assert [r['step'] for r in rows] == [2, 4, 6] == seen
    assert set(rows[0]) == {'step'} | {'Train/' + t for t in ('Total', 'loc', 'conf', 'prop_loc', 'prop_conf', 'IoU', 'start', 'end')}
    assert rows[1]['Train/Total'] == 4.0 and rows[1]['Train/end'] == 11.0


def test_one_element_parameters_as_one_view_and_their_slot_index():
    """Round 5: the six ScaleExp scales are read where they lie when they are adjacent IN ORDER in one storage (a packed copy
    otherwise), and their gradients are scattered into the arena slots with one index_copy_ whatever the slots' order -- the
    trainer's arena holds them in DESCENDING order."""
    arena = torch.arange(32, dtype=torch.float32)
    up = [arena[4 + i:5 + i] for i in range(6)]
    v = ops._adjacent_view(up)
    assert v is not None and v.shape == (6,) and v.data_ptr() == up[0].data_ptr() and torch.equal(v, arena[4:10])
    down = [arena[20 - i:21 - i] for i in range(6)]
    assert ops._adjacent_view(down) is None                       # adjacent, wrong direction
    assert ops._adjacent_view([arena[0:1], arena[2:3]]) is None   # a gap
    assert ops._adjacent_view([arena[0:1], torch.zeros(1)]) is None
    assert ops._adjacent_view([arena[0:2]]) is None               # not one element
    # the scatter HeadOutputsFunction.backward uses for slots in any order
    grad = torch.zeros(32)
    slots = [grad[20 - i:21 - i] for i in range(6)]
    idx = torch.tensor([(s.data_ptr() - grad.data_ptr()) // 4 for s in slots])
    grad.index_copy_(0, idx, torch.arange(1., 7.))
    assert [float(s) for s in slots] == [1., 2., 3., 4., 5., 6.]
