"""GPU: a MULTI-BATCH trajectory check of the benchmarked bf16 mode.  200 optimisation steps of the
reference's loop (AFSD/thumos14/train.py:204-303) over 16 different synthetic batches of 8 clips -- other clips, other
targets and other target counts every step, fed as fixed-shape label records in the epoch loop's launch mode
(`trainer.launch = 'lanes'`: eager once, captured, replayed) -- from identical weights, three times:

    fp32     exact-fp32 GEMMs, fp32 tensors              (the path the 1e-4 parity tests certify)
    bf16     bf16 MFMA operands + bf16-STORED backbone   (the benchmark configuration: ops.HALF_STORAGE + ops.HALF_CHAIN)
    nochain  bf16 MFMA operands, fp32-stored interior    (OTAL_HALF_CHAIN=0: what the stored-pool tie re-routing costs)

Compared: the cost curve smoothed over one pass of the 16 batches (12 windows of 16 steps + the tail), the parameter
displacement (direction and length), the influence-balanced loss's 50-bin state `weight_accum`.

What such a run can and cannot show.  From random initial weights at the recipe's learning rate this detector's first
few hundred steps are erratic in EVERY mode: with other seeds (OTAL_TRAJ_SEED=5, 33) the fp32 run itself does not improve
in 200 steps (window means 37.9 -> 38.0, 35.6 -> 62.5).  A free-running trajectory is therefore one sample of a chaotic
process and its bands are loose; the tight statement is the TEACHER-FORCED one below: at six points of the fp32 trajectory,
on the same weights, criterion state and batch, the gradient each mode computes is compared with the fp32 gradient --
that isolates what a mode does to one step from how the dynamics amplify it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import os
STEPS, NB, B = 200, 16, 8
SEED = int(os.environ.get('OTAL_TRAJ_SEED', '21'))
# measured on MI355X, seed 21 (the printed table), THREE trees that differ in the summation order of some weight
# gradients only: window means within 17-21 % (bf16) / 10-15 % (nochain) of fp32, tails 1.5-15.6 % / 0.7-10.8 %; displacement cosines
# 0.96 / 0.96, lengths 0.996-1.001 / 1.010-1.013, weight_accum 8.5-14 % / 6.1-8 %.  The tail moved from 8 % to 15.6 % when nothing but
# a split-K reduction's summation tree changed: a chaotic sample (see the docstring) -- bands at about twice the spread seen.
WINDOW_BAND, TAIL_BAND, COS_MIN, LEN_BAND, ACCUM_BAND = 0.40, 0.30, 0.90, 0.05, 0.25


def _run(mode, batches, ring):
    import bench
    from opental_amd.common import ops
    dev = torch.device("cuda", 0)
    saved = (ops.CONV_PRECISION, ops.HALF_CHAIN)
    ops.CONV_PRECISION = 0 if mode == "fp32" else 1
    ops.HALF_CHAIN = mode == "bf16"
    try:
        tr = bench.build_trainer(dev, seed=SEED)        # lr 1e-5, weight decay 1e-3, IBM active (epoch 12)
        tr.launch = 'lanes'
        start = tr.arena.flat.detach().clone()
        costs = torch.zeros(STEPS, device=dev)
        for i in range(STEPS):
            rec = ring[i % NB]
            cost, _ = tr.step(batches[i % NB], rec.targets, rec.scores)
            costs[i] = cost
        torch.cuda.synchronize()
        replayed = tr.replayed_steps
        out = (costs.cpu().numpy().astype(np.float64), (tr.arena.flat.detach() - start).double().cpu(),
               tr.criterion.cls_loss.weight_accum.detach().double().cpu(), replayed)
        del tr
        torch.cuda.empty_cache()
        return out
    finally:
        ops.CONV_PRECISION, ops.HALF_CHAIN = saved


def test_bf16_modes_follow_the_fp32_trajectory_over_sixteen_batches():
    import bench
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda", 0)
    batches = [bench.synth_batch(B, 3000 + 100 * (SEED - 21) + i, dev)[0] for i in range(NB)]
    ring = bench.synth_label_ring(B, 3000 + 100 * (SEED - 21), dev, n=NB)
    assert len({r.counts for r in ring}) >= 8                      # the batches differ in their target counts
    runs = {m: _run(m, batches, ring) for m in ("fp32", "bf16", "nochain")}
    for m, r in runs.items():
        assert np.all(np.isfinite(r[0])), m
        assert r[3] >= STEPS - 2, (m, r[3])                         # the epoch loop's launch mode: captured lane graphs replayed
    c32, d32, w32, _ = runs["fp32"]
    win = lambda c: c[: STEPS // NB * NB].reshape(-1, NB).mean(1)   # one pass over the 16 batches per window
    assert win(c32)[-1] < 0.6 * win(c32)[0]                         # the run optimises
    print("window means fp32   ", np.round(win(c32), 3).tolist())
    table = {}
    for m in ("bf16", "nochain"):
        c, d, w, _ = runs[m]
        rel = np.abs(win(c) - win(c32)) / win(c32)
        tail = abs(c[-32:].mean() - c32[-32:].mean()) / c32[-32:].mean()
        cos = float(torch.dot(d, d32) / (d.norm() * d32.norm()))
        ratio = float(d.norm() / d32.norm())
        acc = float((w - w32).norm() / w32.norm())
        table[m] = (float(rel.max()), float(tail), cos, ratio, acc)
        print(f"window means {m:8s}", np.round(win(c), 3).tolist())
        print(f"{m}: window max rel diff {rel.max():.4f}, tail rel diff {tail:.4f}, displacement cosine {cos:.4f}, "
              f"length ratio {ratio:.4f}, weight_accum rel diff {acc:.4f}")
    for m, (rmax, tail, cos, ratio, acc) in table.items():
        assert abs(runs[m][0][0] - c32[0]) < 2e-3 * c32[0], m       # step 1: identical weights
        assert rmax < WINDOW_BAND, (m, rmax)
        assert tail < TAIL_BAND, (m, tail)
        assert cos > COS_MIN, (m, cos)
        assert abs(ratio - 1.0) < LEN_BAND, (m, ratio)
        assert acc < ACCUM_BAND, (m, acc)


def _groups(tr):
    """Arena masks: stem (Conv3d_1a .. Mixed_3c), trunk (the rest of the backbone), pyramid + heads."""
    a = tr.arena
    names = {p.data_ptr(): n for n, p in tr.net.named_parameters()}
    stem_keys = ("Conv3d_1a", "Conv3d_2b", "Conv3d_2c", "Mixed_3b", "Mixed_3c")
    masks = {k: torch.zeros(a.numel, dtype=torch.bool, device=a.flat.device) for k in ("stem", "trunk", "pyramid+heads")}
    for p, off in zip(a.params, a.offsets):
        n = names[p.data_ptr()]
        g = "pyramid+heads" if not n.startswith("backbone.") else ("stem" if any(k in n for k in stem_keys) else "trunk")
        masks[g][off:off + p.numel()] = True
    return masks


def test_teacher_forced_gradients_along_the_fp32_trajectory():
    """At steps 0, 40, ..., 199 of the fp32 run: copy its weights and criterion state into a bf16 (chain) and a nochain
    trainer, run ONE step there with a zero learning rate on the same batch, compare the gradient arenas with the fp32
    run's own gradient of that step -- per parameter group, cosine and relative length."""
    import bench
    from opental_amd.common import ops
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda", 0)
    batches = [bench.synth_batch(B, 3000 + 100 * (SEED - 21) + i, dev)[0] for i in range(NB)]
    ring = bench.synth_label_ring(B, 3000 + 100 * (SEED - 21), dev, n=NB)
    saved = (ops.CONV_PRECISION, ops.HALF_CHAIN)
    modes = {"fp32": (0, True), "bf16": (1, True), "nochain": (1, False)}

    def use(m):
        ops.CONV_PRECISION, ops.HALF_CHAIN = modes[m]
    try:
        trs = {}
        for m in modes:
            use(m)
            trs[m] = bench.build_trainer(dev, seed=SEED)
        masks = _groups(trs["fp32"])
        lead = trs["fp32"]
        worst = {}
        for i in range(STEPS):
            rec = ring[i % NB]
            probe = i % 40 == 0 or i == STEPS - 1
            if probe:
                w0 = lead.arena.flat.detach().clone()
                acc0 = lead.criterion.cls_loss.weight_accum.detach().clone()
            use("fp32")
            lead.step(batches[i % NB], rec.targets, rec.scores)
            if not probe:
                continue
            g32 = lead.arena.grad.detach().double()
            for m in ("bf16", "nochain"):
                use(m)
                t = trs[m]
                t.arena.flat.copy_(w0)
                t.criterion.cls_loss.weight_accum.copy_(acc0)
                t.lr = 0.0
                t.step(batches[i % NB], rec.targets, rec.scores)
                g = t.arena.grad.detach().double()
                assert torch.equal(t.arena.flat, w0)                # a zero learning rate: the probe does not move the weights
                for gname, mk in masks.items():
                    a, b = g[mk], g32[mk]
                    cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
                    ratio = float(a.norm() / b.norm())
                    key = (m, gname)
                    worst[key] = (min(worst.get(key, (1.0, 0.0))[0], cos), max(worst.get(key, (1.0, 0.0))[1], abs(ratio - 1.0)))
                    print(f"step {i:3d} {m:8s} {gname:14s} cosine {cos:.5f} length ratio {ratio:.4f}")
    finally:
        ops.CONV_PRECISION, ops.HALF_CHAIN = saved
    print("lowest cosine / largest length error per (mode, group):", {k: (round(v[0], 4), round(v[1], 4)) for k, v in worst.items()})
    for (m, gname), (cos, dlen) in worst.items():
        assert cos > GRAD_COS[gname], (m, gname, cos)
        assert dlen < GRAD_LEN, (m, gname, dlen)
    for gname in masks:
        assert worst[("bf16", gname)][0] > worst[("nochain", gname)][0] - CHAIN_GAP, gname


# measured on MI355X (seed 21), lowest cosine / largest length error over the six probes (the steps late in the run are the
# ill-conditioned ones: at step 0 the stem's cosine is 0.975):
#   bf16 (chain)  stem 0.875 / 3.2 %   trunk 0.9637 / 1.3 %   pyramid + heads 0.9812 / 0.5 %
#   nochain       stem 0.882 / 4.0 %   trunk 0.9633 / 0.8 %   pyramid + heads 0.9822 / 0.7 %
# (on another 16-batch set: 0.921 / 0.988 / 0.9987 against 0.931 / 0.988 / 0.9987) -- i.e. bf16 STORAGE of the backbone interior
# (pool-tie re-routing, one more rounding at two-producer tensors) costs at most 0.01 of cosine in the stem on top of what
# bf16 OPERANDS do, and nothing measurable elsewhere.
GRAD_COS = {"stem": 0.82, "trunk": 0.94, "pyramid+heads": 0.97}
GRAD_LEN = 0.08
CHAIN_GAP = 0.03        # the chain's lowest cosine may sit this far below nochain's, per group
