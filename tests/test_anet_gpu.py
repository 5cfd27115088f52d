"""GPU parity of the ActivityNet1.3 variant (BASELINE config 4: 768-frame clips, 150 classes, 189 anchors) against
tests/golden/anet_b*.npz, written from the reference's own AFSD/anet modules by oracle/pin_anet.py: forward outputs
within 1e-4 (fp32), proposal windows bit-exact, the loss 7-tuple at epoch 0 and past ibm_start, the training cost and
parameter gradients (fp64 yardstick, as in tests/test_model_gpu.py), and one optimisation step with the recipe's two
learning rates against the oracle's Adam."""
import os

import numpy as np
import pytest
import torch

from oracle import arch

pytestmark = pytest.mark.gpu

EDL = dict(evidence='exp', loss_type='log', iou_aware=True, with_ibm=True, ibm_start=10, momentum=0.99, num_bins=50)
W = dict(lw=1.0, cw=1.0, ctw=1.0, actw=1.0, ssl=0.1)
SMALL = ('loc', 'prop_loc', 'center', 'act', 'prop_act', 'unct', 'prop_unct')
BIG = ('start', 'end', 'start_loc_prop', 'end_loc_prop', 'start_conf_prop', 'end_conf_prop')


def strided(t, n=4096):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)].cpu().numpy()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))


def build(fx):
    from opental_amd.anet.BDNet import BDNet
    net = BDNet(training=False, use_edl=True)
    params = arch.make_params(int(fx["param_seed"]), arch.ANET)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return net.cuda().train()


@pytest.fixture(scope="module", params=[1, 2])
def run(request, golden_dir):
    b = request.param
    fx = np.load(os.path.join(golden_dir, f"anet_b{b}.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), b, frames=768)).cuda()
    return b, fx, net, x


def _criterion(fx, epoch=0):
    from opental_amd.anet.multisegment_loss import MultiSegmentLoss
    crit = MultiSegmentLoss(150, float(fx["piou"]), 1.0, cls_loss_type='edl', edl_config=EDL, os_head=True).cuda()
    crit.cls_loss.epoch = epoch
    return crit


def test_forward_outputs_and_windows(run):
    b, fx, net, x = run
    with torch.no_grad():
        out = net(x)
    seg, fseg = net.coarse_pyramid_detection._last_windows
    lev = net.coarse_pyramid_detection.levels
    assert lev[-1] == 189
    for i in range(6):
        assert np.array_equal(seg[:, lev[i]:lev[i + 1]].cpu().numpy(), fx[f"segments_{i}"]), f"level windows {i}"
        assert np.array_equal(fseg[:, lev[i]:lev[i + 1]].cpu().numpy(), fx[f"frame_segments_{i}"]), f"frame windows {i}"
    for k in SMALL:
        assert rel_err(out[k].cpu().numpy(), fx["out_" + k]) < 1e-4, k
    for k in ('conf', 'prop_conf'):
        assert rel_err(out[k][..., :16].cpu().numpy(), fx[f"out_{k}_first16"]) < 1e-4, k
        assert rel_err(out[k].double().sum(-1).cpu().numpy(), fx[f"out_{k}_rowsum"]) < 1e-4, k
    for k in BIG:
        assert rel_err(strided(out[k]), fx["probe_" + k]) < 1e-4, k
        assert abs(float(out[k].double().sum()) - float(fx["sum_" + k])) < 1e-4 * abs(float(fx["sum_" + k]))
    pri = out['priors'].cpu()
    assert tuple(pri.shape) == (189, 2) and pri[:, 1].tolist() == sum(([float(i)] * t for i, t in enumerate((96, 48, 24, 12, 6, 3))), [])


def test_losses(run):
    b, fx, net, x = run
    targets = [torch.from_numpy(fx[f"target_{i}"]).cuda() for i in range(b)]
    with torch.no_grad():
        out = net(x)
    pred = [out[k] for k in ("loc", "conf", "prop_loc", "prop_conf", "center", "priors", "act", "prop_act")]
    for ep in (0, 12):
        got = np.array([float(v) for v in _criterion(fx, ep)(pred, targets)])
        want = fx[f"loss_edl{ep}"]
        assert (np.abs(got - want) <= 2e-4 * np.maximum(1.0, np.abs(want))).all(), (ep, got, want)


def test_training_cost_and_gradients(golden_dir):
    from opental_amd.anet.train import forward_one_epoch, total_cost
    fx = np.load(os.path.join(golden_dir, "anet_b1.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), 1, frames=768)).cuda()
    targets = [torch.from_numpy(fx["target_0"]).cuda()]
    scores = torch.from_numpy(fx["scores"]).cuda()
    net.zero_grad(set_to_none=True)
    losses = forward_one_epoch(net, _criterion(fx, 0), x, targets, scores, training=True, ssl=False)
    cost = total_cost(losses, W)
    cost.backward()
    assert abs(float(cost.detach()) - float(fx["cost_edl0"])) < 1e-4 * abs(float(fx["cost_edl0"]))
    grads = dict((k, p.grad) for k, p in net.named_parameters() if p.grad is not None)
    names = [str(n) for n in fx["grad_names"]]
    assert sorted(grads) == names
    # the same criterion as tests/test_model_gpu.py: as close to the fp64 run as the reference's CPU fp32 gradients
    # are, within a factor 10 (independent rounding draws), plus the arg-max near-tie floor
    d32, n64 = fx["grad32dist_correct"], fx["grad64norm_correct"]
    got = np.array([float(grads[n].double().norm()) for n in names])
    allowed = 10.0 * d32 + 3e-4
    worst = np.abs(got - n64) / (n64 + 1e-30) / allowed
    iw = int(worst.argmax())
    assert worst.max() < 1.0, (names[iw], float(worst.max()), float(abs(got[iw] - n64[iw]) / n64[iw]), float(d32[iw]))
    for k in fx.files:
        if k.startswith("grad64probe_correct/"):
            name = k.split("/", 1)[1]
            want = fx[k]
            have = strided(grads[name], 512)
            tol = 10.0 * float(d32[names.index(name)]) + 3e-4
            assert np.linalg.norm(have - want) / (np.linalg.norm(want) + 1e-30) < tol, name


def test_one_step_with_the_two_learning_rates(golden_dir):
    """backbone at lr/10, pyramid at lr (anet/train.py:304-312), against the oracle's Adam on the same gradients."""
    from oracle import afsd_oracle as O
    from opental_amd.anet.train import make_trainer
    fx = np.load(os.path.join(golden_dir, "anet_b1.npz"))
    net = build(fx)
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), 1, frames=768)).cuda()
    targets = [torch.from_numpy(fx["target_0"]).cuda()]
    scores = torch.from_numpy(fx["scores"]).cuda()
    tr = make_trainer(net, _criterion(fx, 0), W, learning_rate=1e-4, weight_decay=1e-4, distributed=False)
    before = {k: p.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
    cost, _ = tr.step(x, targets, scores)
    assert abs(float(cost) - float(fx["cost_edl0"])) < 1e-4 * abs(float(fx["cost_edl0"]))
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        lr = 1e-5 if k.startswith("backbone.") else 1e-4
        want = before[k].cpu().clone()
        O.adam_step(want, p.grad.cpu().clone(), torch.zeros_like(want), torch.zeros_like(want), 1, lr, 1e-4)
        moved = float((want - before[k].cpu()).abs().max())
        assert float((p.detach().cpu() - want).abs().max()) <= 1e-3 * moved + 1e-9, k


def test_inference_postprocessing_equals_the_reference(golden_dir):
    """decode + thresholds + Soft-NMS + clipping of three one-clip videos against the proposal lists the reference's
    AFSD/anet/test.py produced for the same seeded head outputs (oracle/pin_anet_inference.py)."""
    from oracle.pin_anet_inference import VIDEOS, synthetic_heads
    from opental_amd.anet import test as T
    from opental_amd.thumos14.test import softnms_classes
    fx = np.load(os.path.join(golden_dir, "anet_inference.npz"))
    heads = [synthetic_heads(seed, cm) for _, seed, _, _, cm in VIDEOS]
    merged = {k: (torch.cat([h[k] for h in heads], 0).cuda() if k != 'priors' else heads[0][k].cuda()) for k in heads[0]}
    dec = T.decode_clips(merged, [v[2] for v in VIDEOS])
    rows, counts, _ = softnms_classes(dec, list(range(len(VIDEOS) + 1)), 5000, 0.85)
    for v, (name, _, _, duration, _) in enumerate(VIDEOS):
        got = T.get_video_prediction(rows[v], counts[v], duration)
        want_cls, want = fx[name + "_class"], fx[name + "_rows"]
        assert [p['label'] for p in got] == want_cls.tolist(), name           # same kept set, same order
        have = np.array([[p['segment'][0], p['segment'][1], p['score'], p['uncertainty'], p['actionness']] for p in got])
        assert np.abs(have - want).max() < 2e-4 * max(1.0, np.abs(want).max()), name
        assert all(0 <= p['segment'][0] < p['segment'][1] <= duration for p in got)


def test_short_video_is_padded_with_mid_grey():
    from opental_amd.anet.test import prepare_clip
    data = torch.randint(0, 256, (3, 100, 96, 96), dtype=torch.uint8, device="cuda")
    clip = prepare_clip(data, 0)
    assert tuple(clip.shape) == (1, 3, 768, 96, 96)
    assert torch.equal(clip[0, :, :100], (data.float() / 255.0) * 2.0 - 1.0)
    assert float(clip[0, :, 100:].abs().max()) == 0.0          # 127.5 / 255 * 2 - 1 == 0


def test_detect_batch_equals_the_per_video_order(golden_dir):
    """anet.test.detect_batch (windows of a pass prepared by one launch, all videos decoded and suppressed together)
    against the reference's order of operations: per video prepare_clip -> net -> decode -> Soft-NMS -> duration clip."""
    from opental_amd.anet import test as A
    from opental_amd.thumos14 import test as T
    fx = np.load(os.path.join(golden_dir, "anet_b1.npz"))
    net = build(fx).eval()
    rs = np.random.RandomState(5)
    videos = [torch.from_numpy(rs.randint(0, 256, size=(3, t, 96, 96)).astype(np.uint8)).cuda() for t in (768, 500, 90)]
    batch = T.prepare_windows(videos, [(0, 0), (1, 0), (2, 0)], 768)
    assert torch.equal(batch, torch.cat([A.prepare_clip(v, 0) for v in videos], 0))
    fps, durations = [4.0, 5.0, 6.0], [190.0, 99.0, 14.0]
    got = A.detect_batch(net, videos, fps, durations, batch_clips=1)
    total = 0
    for v, data in enumerate(videos):
        with torch.no_grad():
            out = net(A.prepare_clip(data, 0))
        dec = A.decode_clips(out, [fps[v]])
        rows, counts, _ = T.softnms_classes(dec, [0, 1], 5000, 0.85)
        ref = A.get_video_prediction(rows[0], counts[0], durations[v])
        assert got[v] == ref
        total += len(ref)
    assert total > 0


@pytest.mark.parametrize("case", ["mixed", "no_targets_in_one_sample", "single_positive_level", "ibm_off"])
def test_fused_anet_loss_equals_the_torch_formulation(case):
    """otal_detection_loss_anet (one workgroup per sample: matching with per-level bounds, EvidenceLoss with the closed-form
    influence-balanced weight, smooth-L1 refinement, quality BCE with the non-detached tIoU target, positive-unlabelled
    actionness BCE with its rank hinge, per-sample IoU calibration, per-sample normalisation) against this package's torch
    formulation of AFSD/anet/multisegment_loss.py -- which oracle/pin_anet.py pins to the reference with delta 0: the seven
    terms to 2e-5 and the gradient of a weighted sum w.r.t. every head output to 2e-5 of its scale.  Cases: three samples
    with 1-4 targets; a sample without any target (no positives: N = 1, thresholds fall back); targets that only one
    pyramid level accepts; the IBM weight off (epoch < ibm_start)."""
    from opental_amd.anet import multisegment_loss as M
    rs = np.random.RandomState({"mixed": 1, "no_targets_in_one_sample": 2, "single_positive_level": 3, "ibm_off": 4}[case])
    B, C = 3, 150
    lens = [96, 48, 24, 12, 6, 3]
    K = sum(lens)
    pri = np.concatenate([np.stack([(np.arange(t) + 0.5) / t, np.full(t, i)], 1) for i, t in enumerate(lens)]).astype(np.float32)
    priors = torch.from_numpy(pri).cuda()

    def seg(a, b, lab):
        return [a / 768.0, b / 768.0, float(lab)]
    if case == "single_positive_level":
        targets = [np.array([seg(100, 140, 5)], np.float32), np.array([seg(300, 330, 9), seg(500, 540, 2)], np.float32),
                   np.array([seg(20, 60, 150)], np.float32)]
    else:
        targets = [np.array([seg(50, 250, 3), seg(400, 430, 17), seg(600, 760, 150), seg(300, 320, 1)], np.float32),
                   np.array([seg(10, 700, 42)], np.float32),
                   np.array([seg(200, 260, 7), seg(220, 500, 8)], np.float32)]
    if case == "no_targets_in_one_sample":
        targets[1] = np.zeros((0, 3), np.float32)
    targets = [torch.from_numpy(t).cuda() for t in targets]
    mk = lambda *shape, scale=1.0: (torch.from_numpy((rs.randn(*shape) * scale).astype(np.float32)).cuda())
    loc0 = torch.from_numpy(np.abs(rs.randn(B, K, 2)).astype(np.float32) * 40 + 5).cuda()
    inputs = [loc0, mk(B, K, C, scale=2.0), mk(B, K, 2, scale=0.8), mk(B, K, C, scale=2.0), mk(B, K, 1), mk(B, K, 1), mk(B, K, 1)]
    wsum = torch.tensor([1.0, 0.7, 1.3, 0.9, 1.1, 0.6, 1.2], device="cuda")

    def run(fused):
        M.FUSED = fused
        crit = M.MultiSegmentLoss(C, 0.6, 1.0, cls_loss_type='edl', edl_config=EDL, os_head=True).cuda()
        crit.cls_loss.epoch = 0 if case == "ibm_off" else 12
        xs = [t.clone().requires_grad_(True) for t in inputs]
        loc, conf, pl, pc, cen, act, pact = xs
        terms = crit([loc, conf, pl, pc, cen, priors, act, pact], targets)
        (torch.stack([v.reshape(()) for v in terms]) * wsum).sum().backward()
        return np.array([float(v.detach()) for v in terms]), [x.grad.detach().cpu().numpy() for x in xs]
    try:
        t_ref, g_ref = run(False)
        t_hip, g_hip = run(True)
    finally:
        M.FUSED = True
    assert np.isfinite(t_hip).all()
    assert (np.abs(t_hip - t_ref) <= 2e-5 * np.maximum(1.0, np.abs(t_ref))).all(), (t_hip, t_ref)
    for name, a, b in zip(("loc", "conf", "prop_loc", "prop_conf", "center", "act", "prop_act"), g_hip, g_ref):
        scale = max(float(np.abs(b).max()), 1e-12)
        assert float(np.abs(a - b).max()) <= 2e-5 * scale + 1e-9, (name, float(np.abs(a - b).max()), scale)
