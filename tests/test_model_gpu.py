"""GPU parity of the full detection path against the golden vectors written by the imported
reference (tests/golden/thumos_b*.npz, oracle/pin_against_reference.py): forward outputs within
1e-4 (fp32), proposal windows bit-exact, losses, parameter gradients (correct backward vs the
oracle, reference-addressing backward vs the reference itself)."""
import os

import numpy as np
import pytest
import torch

from oracle import arch

pytestmark = pytest.mark.gpu

EDL = dict(evidence='exp', loss_type='log', iou_aware=True, with_focal=False, alpha=0.25, gamma=2, with_ibm=True,
           ibm_start=10, momentum=0.99, num_bins=50)
ACT = dict(margin=1.0, weight=0)
W = dict(lw=1.0, cw=10.0, ctw=1.0, actw=1.0, ssl=0.001)
SMALL = ('loc', 'conf', 'prop_loc', 'prop_conf', 'center', 'act', 'prop_act', 'unct', 'prop_unct')
BIG = ('start', 'end', 'start_loc_prop', 'end_loc_prop', 'start_conf_prop', 'end_conf_prop')


def strided(t, n=4096):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)].cpu().numpy()


def build(fx):
    from opental_amd.thumos14.BDNet import BDNet
    net = BDNet(training=False, use_edl=True)
    params = arch.make_params(int(fx["param_seed"]))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return net.cuda().train()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))


@pytest.fixture(scope="module", params=[1, 2, 4])
def run(request, golden_dir):
    b = request.param
    fx = np.load(os.path.join(golden_dir, f"thumos_b{b}.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), b)).cuda()
    return b, fx, net, x


def test_forward_outputs_and_windows(run):
    b, fx, net, x = run
    with torch.no_grad():
        out = net(x)
    seg, fseg = net.coarse_pyramid_detection._last_windows
    lev = net.coarse_pyramid_detection.levels
    for i in range(6):
        assert np.array_equal(seg[:, lev[i]:lev[i + 1]].cpu().numpy(), fx[f"segments_{i}"]), f"level windows {i}"
        assert np.array_equal(fseg[:, lev[i]:lev[i + 1]].cpu().numpy(), fx[f"frame_segments_{i}"]), f"frame windows {i}"
    for k in SMALL:
        assert rel_err(out[k].cpu().numpy(), fx["out_" + k]) < 1e-4, k
    for k in BIG:
        assert rel_err(strided(out[k]), fx["probe_" + k]) < 1e-4, k
        assert abs(float(out[k].double().sum()) - float(fx["sum_" + k])) < 1e-4 * abs(float(fx["sum_" + k]))
    assert tuple(out['priors'].shape) == (126, 1)


def test_backbone_endpoints(run):
    b, fx, net, x = run
    names = ("Conv3d_1a_7x7", "Conv3d_2c_3x3", "Mixed_3c", "Mixed_4f", "Mixed_5c")
    with torch.no_grad():
        feats = net.backbone._model.extract_features(x, endpoints=names)
    for n in names:
        assert rel_err(strided(feats[n]), fx["probe_" + n]) < 1e-4, n
        assert abs(float(feats[n].double().abs().mean()) - float(fx["absmean_" + n])) < 1e-5


def _criterion(mode, epoch=0):
    from opental_amd.thumos14.multisegment_loss import MultiSegmentLoss
    crit = MultiSegmentLoss(15, 0.5, 1.0, cls_loss_type=mode, edl_config=EDL, os_head=True, act_config=ACT).cuda()
    if mode == 'edl':
        crit.cls_loss.epoch = epoch
    return crit


def test_losses(run):
    b, fx, net, x = run
    targets = [torch.from_numpy(fx[f"target_{i}"]).cuda() for i in range(b)]
    with torch.no_grad():
        out = net(x)
    for mode, ep in (("edl", 0), ("edl", 12), ("focal", 0)):
        crit = _criterion(mode, ep)
        got = np.array([float(v) for v in crit(out, targets)])
        assert np.abs(got - fx[f"loss_{mode}{ep}"]).max() < 2e-4, (mode, ep, got, fx[f"loss_{mode}{ep}"])
        if ep >= 10:
            assert np.abs(crit.cls_loss.weight_accum.cpu().numpy() - fx["loss_edl12_weight_accum"]).max() < 1e-5


@pytest.mark.parametrize("compat", [False, True])
def test_training_cost_and_gradients(run, compat):
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    from opental_amd.thumos14.train import forward_one_epoch, total_cost
    b, fx, net, x = run
    targets = [torch.from_numpy(fx[f"target_{i}"]).cuda() for i in range(b)]
    scores = torch.from_numpy(fx["scores"]).cuda()
    crit = _criterion("edl", 0)
    net.zero_grad(set_to_none=True)
    bp.COMPAT_REFERENCE_BWD = compat
    try:
        losses = forward_one_epoch(net, crit, x, targets, scores, training=True, ssl=False)
        cost = total_cost(losses, W)
        cost.backward()
    finally:
        bp.COMPAT_REFERENCE_BWD = False
    assert abs(float(cost.detach()) - float(fx["cost_edl0"])) < 1e-4 * abs(float(fx["cost_edl0"]))
    grads = dict((k, p.grad) for k, p in net.named_parameters() if p.grad is not None)
    names = [str(n) for n in fx["grad_names"]]
    assert sorted(grads) == names
    # Yardstick: the fp64 run of the restatement.  fp32 gradients of this network are conditioned to
    # ~3e-5 (median) .. 5e-3 (first conv) -- that is how far the reference's own CPU fp32 gradients
    # sit from fp64 (grad32dist_*, measured when the fixture was written).  The HIP path must be
    # as close to fp64 as that, within a factor 10: its rounding errors are an independent draw of the
    # same size (MFMA k-ordered chains vs blocked CPU sums), and the worst of 161 tensors is tested.
    # (Measured worst case 8.4x, on the 16-channel bottleneck Mixed_3b.b2a, after the three 1x1 launches of a module were
    # fused: the fused launches themselves are exact -- tests/test_ops_gpu.py::test_fused_1x1_launches_match_separate_ones --
    # but every change of summation order anywhere downstream is a new draw for the layers upstream of it.)
    mode = "compat" if compat else "correct"
    d32 = fx[f"grad32dist_{mode}"]
    n64 = fx[f"grad64norm_{mode}"]
    got = np.array([float(grads[n].double().norm()) for n in names])
    # floor: a near-tie arg-max in BoundaryMaxPooling can flip with the rounding of the forward pass and
    # re-route ONE contribution; in reference-addressing mode the recomputed arg-max runs over
    # re-strided (semantically scrambled) rows, where near-ties are more frequent
    floor = 1e-3 if compat else 3e-4
    allowed = 10.0 * d32 + floor
    worst = np.abs(got - n64) / (n64 + 1e-30) / allowed
    iw = int(worst.argmax())
    assert worst.max() < 1.0, (names[iw], float(worst.max()), "rel |norm - fp64 norm|", float(abs(got[iw] - n64[iw]) / n64[iw]),
                               "reference fp32 distance", float(d32[iw]), "allowed", float(allowed[iw]))
    key = f"grad64probe_{mode}/"
    for k in fx.files:
        if k.startswith(key):
            name = k[len(key):]
            i = names.index(name)
            probe = strided(grads[name], 512).astype(np.float64)
            # probe error measured against the TENSOR's scale (a probe may sample a small-magnitude
            # slice, e.g. one weight column): expected probe norm = ||g64|| * sqrt(len(probe) / numel)
            scale = float(n64[i]) * np.sqrt(probe.size / grads[name].numel())
            dist = float(np.linalg.norm(probe - fx[k]) / scale)
            tol = 5.0 * float(d32[i]) + floor
            assert dist < tol, (name, dist, tol)
    if compat:   # and against the reference's own fp32 gradients (its launcher addresses rows with stride N)
        ref = fx["gradnorm_reference"]
        assert (np.abs(got - ref) / (ref + 1e-30) / (2 * allowed)).max() < 1.0


def test_eval_mode_inference_matches_train_mode_forward(run):
    """BN is frozen and dropout is 0, so eval() must give the same outputs."""
    b, fx, net, x = run
    net.eval()
    with torch.no_grad():
        out = net(x)
    net.train()
    assert rel_err(out['loc'].cpu().numpy(), fx["out_loc"]) < 1e-4


@pytest.mark.parametrize("si", [0, 1])
def test_ssl_triplet_branch(golden_dir, si):
    """a14 against the REFERENCE (fixture entries ssl{si}_* written by oracle/pin_against_reference.py::pin_ssl from the
    reference's BDNet.forward(ssl=True), BDNet.py:482-503, and the three TripletMarginLoss terms of train.py:177-184):
    anchor / positive / negative features, the weighted terms, and the gradients of the triplet cost through the second
    backbone pass -- reference addressing vs the reference's own fp32 gradient norms, correct backward vs fp64."""
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    from opental_amd.thumos14.train import forward_one_epoch
    import torch.nn as nn
    fx = np.load(os.path.join(golden_dir, "thumos_b1.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["ssl_clip_seed"]), 1)).cuda()
    props = [torch.from_numpy(fx["ssl_proposals"][si]).cuda()]
    t = f"ssl{si}_"
    with torch.no_grad():
        a, p, n = net(x, proposals=props, ssl=True)
    for nm, got in (("anchor", a), ("positive", p), ("negative", n)):
        for i in range(3):
            assert rel_err(got[i].cpu().numpy(), fx[f"{t}{nm}_{i}"]) < 1e-4, (nm, i)
    terms = [float(nn.TripletMarginLoss()(a[i], p[i], n[i]) * w) for i, w in enumerate((1, 0.1, 0.1))]
    assert np.abs(np.array(terms) - fx[t + "terms"]).max() < 1e-4
    names = [str(v) for v in fx[t + "grad_names"]]
    for compat in (True, False):
        net.zero_grad(set_to_none=True)
        bp.COMPAT_REFERENCE_BWD = compat
        try:
            cost = forward_one_epoch(net, None, x, props, training=True, ssl=True)
            cost.backward()
        finally:
            bp.COMPAT_REFERENCE_BWD = False
        assert abs(float(cost.detach()) - float(fx[t + "cost"])) < 1e-4 * max(1.0, abs(float(fx[t + "cost"])))
        grads = dict((k, q.grad) for k, q in net.named_parameters() if q.grad is not None)
        assert sorted(grads) == names
        mode = "compat" if compat else "correct"
        d32, n64 = fx[f"{t}grad32dist_{mode}"], fx[f"{t}grad64norm_{mode}"]
        got = np.array([float(grads[k].double().norm()) for k in names])
        live = n64 > 1e-7 * n64.max()                       # structurally zero gradients carry no relative error
        # the criterion of test_training_cost_and_gradients with the floor at 1e-3 in both modes: the whole cost hangs on
        # NINE pooled columns (3 proposals x 3 maps), so one near-tie arg-max that the forward pass's rounding flips moves a
        # tower gradient by a visible share (measured worst: loc_tower.1.1.bias 6.2e-4, where the CPU fp32 run happened to
        # sit 5e-6 from fp64)
        allowed = 10.0 * d32 + 1e-3
        worst = (np.abs(got - n64) / (n64 + 1e-30) / allowed)[live]
        assert worst.max() < 1.0, (mode, names[int(np.flatnonzero(live)[worst.argmax()])], float(worst.max()))
        if compat:      # ... and against the reference's own fp32 gradients
            ref = fx[t + "gradnorm_reference"]
            assert (np.abs(got - ref) / (ref + 1e-30) / (2 * allowed))[live].max() < 1.0


def test_bf16_compute_mode_stays_close_to_fp32(golden_dir):
    """bf16-operand GEMMs (fp32 accumulate, fp32 tensors): the stated tolerance of the throughput
    path.  Operand rounding is 2^-9 relative per element; through ~60 layers the outputs
    stay within 6e-2 of the fp32 golden vectors (relative to each tensor's scale), the training cost
    within 2 %, and parameter gradients keep a high cosine with the fp32 gradients (see below)."""
    from opental_amd.common import ops
    from opental_amd.thumos14.train import forward_one_epoch, total_cost
    fx = np.load(os.path.join(golden_dir, "thumos_b1.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), 1)).cuda()
    targets = [torch.from_numpy(fx["target_0"]).cuda()]
    scores = torch.from_numpy(fx["scores"]).cuda()

    def run(prec):
        ops.CONV_PRECISION = prec
        try:
            net.zero_grad(set_to_none=True)
            crit = _criterion("edl", 0)
            out = net(x)
            losses = forward_one_epoch(net, crit, x, targets, scores, training=True, ssl=False)
            cost = total_cost(losses, W)
            cost.backward()
            return ({k: v.detach().clone() for k, v in out.items() if v is not None}, float(cost.detach()),
                    {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        finally:
            ops.CONV_PRECISION = 0
    o32, c32, g32 = run(0)
    o16, c16, g16 = run(1)
    for k in ("loc", "conf", "act", "unct"):        # computed before any pooling: smooth in the rounding noise
        assert rel_err(o16[k].cpu().numpy(), fx["out_" + k]) < 6e-2, k      # measured 1e-2 .. 3e-2
    for k in ("prop_loc", "prop_conf", "prop_act", "center", "prop_unct"):
        # downstream of BoundaryMaxPooling: a proposal window whose rounding flips under the bf16 noise on
        # `loc` changes that anchor's pooled features discretely -> robust statistic instead of the max
        d = np.abs(o16[k].cpu().numpy().astype(np.float64) - fx["out_" + k]) / max(np.abs(fx["out_" + k]).max(), 1e-6)
        assert np.percentile(d, 90) < 6e-2, (k, float(np.percentile(d, 90)))
    assert abs(c16 - c32) < 2e-2 * abs(c32)
    gmax = max(float(g.norm()) for g in g32.values())
    cos = {k: float(torch.nn.functional.cosine_similarity(g16[k].flatten(), g32[k].flatten(), dim=0))
           for k in g32 if float(g32[k].norm()) > 1e-7 * gmax}          # skip structurally zero gradients
    assert len(cos) > 150
    # heads / pyramid: short backward paths -> tight; backbone: rounding noise is amplified towards the
    # input exactly as the fp32-vs-fp64 distance is (5e-3 at Conv3d_1a in fp32), so the first convs are
    # the noisiest (measured: Conv3d_1a 0.77) while the median stays > 0.99
    head = {k: v for k, v in cos.items() if k.startswith("coarse_pyramid_detection")}
    back = sorted(v for k, v in cos.items() if k.startswith("backbone"))
    assert min(head.values()) > 0.96, min(head, key=head.get)          # measured minimum 0.978 (a GroupNorm gamma)
    assert back[len(back) // 2] > 0.97 and back[0] > 0.6, (back[0], back[len(back) // 2])


@pytest.mark.gpu
def test_capture_after_eager_steps():
    """A trainer that already ran eager steps can still be captured (bench.py --graph auto does exactly that when the host
    turns out to be the bottleneck).  Regression: a graph-attached tensor parked on a module kept the previous step's
    AccumulateGrad nodes -- bound to the default stream -- alive, and hipStreamEndCapture crashed."""
    import bench
    from opental_amd.common import ops
    dev = torch.device("cuda", 0)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        clips, targets, scores = bench.synth_batch(1, 78, dev)
        tr = bench.build_trainer(dev, seed=12)
        eager = [float(tr.step(clips, targets, scores)[0]) for _ in range(2)]
        tr.capture_step(clips, targets, scores, warmup=1)
        replayed = [float(tr.step(clips, targets, scores)[0].clone()) for _ in range(2)]
        torch.cuda.synchronize()
        assert tr.step_count == 5 and all(np.isfinite(eager + replayed))
        assert replayed[-1] != eager[0]                       # the replays are real optimisation steps
    finally:
        ops.CONV_PRECISION = old


def test_graph_replayed_step_matches_eager_steps():
    """DetectorTrainer.capture_step: forward + losses + backward + Adam replayed from one HIP graph must walk the
    same parameter trajectory as eager launches (same kernels in the same order; only Adam's bias correction is
    read from device memory).  Float atomics inside torch ops (index_add_, gather backward) make even two EAGER
    runs differ in the last bits, so the graph run is held to a small multiple of that run-to-run spread."""
    import bench
    from opental_amd.common import ops
    dev = torch.device("cuda", 0)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        clips, targets, scores = bench.synth_batch(1, 77, dev)
        start = bench.build_trainer(dev, seed=11).arena.flat.detach().clone()

        def run(graphed):
            tr = bench.build_trainer(dev, seed=11)
            tr.lr = 1e-4
            if graphed:
                tr.capture_step(clips, targets, scores, warmup=1)
                c = [tr.step(clips, targets, scores)[0].clone() for _ in range(2)]
            else:
                c = [tr.step(clips, targets, scores)[0].clone() for _ in range(3)][1:]
            torch.cuda.synchronize()
            assert tr.step_count == 3
            return tr.arena.flat.detach().clone(), [float(v) for v in c]

        (fa, ca), (fb, cb), (fg, cg) = run(False), run(False), run(True)
        moved = float((fa - start).abs().max())
        assert moved > 1e-5
        spread = float((fa - fb).abs().max())
        d = float((fg - fa).abs().max())
        # measured: eager is run-to-run bit-stable here and the replayed trajectory is bit-identical to it; the bound
        # leaves room for the float atomics of torch ops on other inputs (bf16 operand rounding amplifies any last-bit
        # difference of a weight, so a loose bound would hide a real bug -- keep it at a small multiple of the spread)
        assert d <= 10.0 * spread + 1e-6 * moved, (d, spread, moved)
        cs = max(abs(x - y) for x, y in zip(ca, cb))
        assert max(abs(x - y) for x, y in zip(ca, cg)) <= 10.0 * cs + 1e-6 * abs(ca[0]), (ca, cb, cg)
    finally:
        ops.CONV_PRECISION = old


@pytest.mark.gpu
def test_persistent_prologues_do_not_change_the_trajectory():
    """ops.PrologueCache (tables + bf16-packed weights refreshed once per step, include/opental_hip.h "Persistent
    prologues") against launches that build their prologue themselves: same parameters after three steps, bit for bit,
    and the cache really is in use (regions registered, none for weights outside the arena)."""
    import bench
    from opental_amd.common import ops
    dev = torch.device("cuda", 0)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        clips, targets, scores = bench.synth_batch(1, 78, dev)

        def run(cached):
            tr = bench.build_trainer(dev, seed=12)
            tr.lr = 1e-4
            if not cached:
                tr._prologues = None            # activate_prologues(None): every launch packs for itself
            for _ in range(3):
                tr.step(clips, targets, scores)
            torch.cuda.synchronize()
            return tr

        a, b = run(False), run(True)
        assert torch.equal(a.arena.flat, b.arena.flat)
        regs = [v for v in b._prologues.entries.values() if v is not None]
        assert len(regs) > 100 and len(b._prologues.descs) > 80
        lo, hi = b._prologues.persistent_range
        assert hi - lo == 4 * b.arena.numel
    finally:
        ops.CONV_PRECISION = old


@pytest.mark.gpu
@pytest.mark.parametrize("ssl", [False, True])
def test_weight_gradients_written_into_the_arena_match_the_copied_ones(ssl, monkeypatch):
    """ops.GradSlots: during a trainer's backward the weight-gradient launches write into the gradient arena directly.
    The arena must hold bit-for-bit what the hand-over copy used to put there -- also when the self-supervised branch
    uses every backbone weight twice (slot handed out once, second gradient added by autograd)."""
    import bench
    from opental_amd.common import ops
    dev = torch.device("cuda", 0)
    clips, targets, scores = bench.synth_batch(1, 31, dev)
    extra = ()
    if ssl:
        ssl_clips, _, _ = bench.synth_batch(1, 32, dev)
        extra = (ssl_clips, [torch.tensor([[0.30, 0.55], [0.32, 0.52], [0.70, 0.90]], device=dev) * 256])
    grads, copied = [], []
    orig = torch._foreach_copy_
    for slots in (True, False):
        if slots:
            monkeypatch.delenv("OTAL_NO_GRAD_SLOTS", raising=False)
        else:
            monkeypatch.setenv("OTAL_NO_GRAD_SLOTS", "1")
        n = [0]

        def counting(dst, src, n=n):
            n[0] += sum(d.numel() for d in dst)
            return orig(dst, src)
        monkeypatch.setattr(torch, "_foreach_copy_", counting)
        tr = bench.build_trainer(dev, seed=12)
        tr.step(clips, targets, scores, *extra)
        grads.append(tr.arena.grad.clone())
        copied.append(n[0])
        assert ops.GRAD_SLOTS is None
    assert torch.equal(grads[0], grads[1])
    assert float(grads[0].abs().max()) > 0
    assert copied[1] > 0.9 * grads[0].numel()
    if not ssl:                 # (a weight used twice is summed by autograd into a new tensor, which is copied as before)
        assert copied[0] < 0.01 * grads[0].numel(), copied
