"""oracle/pin_anet.py -- TEST INFRASTRUCTURE, runs in the BUILD container only (needs /root/reference).

Pins the ActivityNet1.3 branch of the restatement (cfg=arch.ANET: afsd_oracle.coarse_pyramid,
multisegment_loss_anet, train_cost_anet) against the reference's own AFSD/anet modules imported from
/root/reference (configs/anet_opental.yaml, flags of AFSD/anet/README.md:61 `--lw=1 --cw=1 --piou=0.6`), and writes
tests/golden/anet_b1.npz (forward + losses + gradients, one 768-frame clip) and tests/golden/anet_b2.npz
(forward + losses, the yaml's batch of 2: exercises the per-sample normalisation of the ANet loss).

    python -m oracle.pin_anet          # ~10 min on 8 cores, appends to tests/golden/PIN_REPORT.txt

The config is parsed when AFSD.common.config is imported, so this must be its own process (the THUMOS14 pin is
oracle/pin_against_reference.py).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"       # child processes too (joblib workers of the evaluation code)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn.functional as F

from oracle import arch
from oracle import afsd_oracle as O
from oracle.pin_against_reference import maxdiff, round_margin, strided

CFG = arch.ANET
PIOU, LW, CW = 0.6, 1.0, 1.0


def import_reference():
    sys.path.insert(0, REF)
    sys.argv = ["pin", os.path.join(REF, "configs/anet_opental.yaml"), "--open_set", "--split", "0",
                "--lw", str(LW), "--cw", str(CW), "--piou", str(PIOU)]
    fake = types.ModuleType("boundary_max_pooling_cuda")
    fake.forward = lambda inp, seg: O.bmp_forward(inp, seg)
    fake.backward = lambda g, inp, seg: O.bmp_backward(g, inp, seg, compat_reference_bwd=True)
    sys.modules["boundary_max_pooling_cuda"] = fake
    torch.Tensor.cuda = lambda self, *a, **k: self
    from AFSD.anet.BDNet import BDNet
    from AFSD.anet.multisegment_loss import MultiSegmentLoss
    import AFSD.anet.multisegment_loss as ref_loss_mod
    from AFSD.common.config import config
    return BDNet, MultiSegmentLoss, ref_loss_mod, config


def ref_bce(s, e, sc):     # restated from anet/train.py:136-144 (train.py itself needs tensorboardX)
    s = torch.tanh(s).mean(-1)
    e = torch.tanh(e).mean(-1)
    return (F.binary_cross_entropy(s.view(-1), sc[:, 1].contiguous().view(-1)),
            F.binary_cross_entropy(e.view(-1), sc[:, 2].contiguous().view(-1)))


def ref_total_cost(ref_out, ref7, scores):
    ls, le = ref_bce(ref_out["start"], ref_out["end"], scores)
    sc8 = F.interpolate(scores, scale_factor=1.0 / 8)          # torch 2.x branch of anet/train.py:174-178
    a, b_ = ref_bce(ref_out["start_loc_prop"], ref_out["end_loc_prop"], sc8)
    c, d_ = ref_bce(ref_out["start_conf_prop"], ref_out["end_conf_prop"], sc8)
    return ref7[0] * LW + ref7[1] * CW + ref7[2] * LW + ref7[3] * CW + ref7[4] * 1 + \
        (ls + 0.1 * (a + c)) + (le + 0.1 * (b_ + d_)) + ref7[5] * 1 + ref7[6] * 1


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    BDNet, MultiSegmentLoss, ref_loss_mod, config = import_reference()
    levels = arch.level_lengths(CFG)
    frame_num = CFG["frame_num"]
    report = []

    net = BDNet(training=False, use_edl=True)
    sd = net.state_dict()
    spec = arch.param_spec(CFG)
    assert [k for k, _ in spec] == list(sd.keys()), "state-dict key order differs"
    for k, shp in spec:
        assert tuple(sd[k].shape) == tuple(shp), (k, shp, tuple(sd[k].shape))
    report.append(f"anet: state_dict layout: {len(spec)} entries match")

    PARAM_SEED = 2021
    params_np = arch.make_params(PARAM_SEED, CFG)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params_np.items()})
    net.train()
    edl_cfg = config["training"]["edl_config"]

    def pick_clip_seed(batch, first, need):
        Pq = O.to_torch(params_np)
        for seed in range(first, first + 200):
            with torch.no_grad():
                o = O.bdnet_forward(Pq, torch.from_numpy(arch.make_clip(seed, batch, frames=frame_num)), cfg=CFG)
            m = round_margin(o["loc"], levels, frame_num)
            if m > need:
                return seed, m
        raise RuntimeError("no seed with a safe rounding margin")

    for batch, first, need, with_grads in ((1, 21, 5e-4, True), (2, 700, 3e-4, False)):
        tag = f"anet_b{batch}"
        clip_seed, m0 = pick_clip_seed(batch, first, need)
        report.append(f"{tag}: clip seed {clip_seed} (rounding margin {m0:.2e})")
        x = torch.from_numpy(arch.make_clip(clip_seed, batch, frames=frame_num))
        targets_np = arch.make_targets(clip_seed + 100, batch, num_classes=CFG["num_classes"], clip_length=frame_num)
        scores_np = arch.make_scores_anet(targets_np, frame_num)
        targets = [torch.from_numpy(t) for t in targets_np]
        scores = torch.from_numpy(scores_np)

        net.zero_grad()
        ref_out = net(x)
        P = O.to_torch(params_np, requires_grad=True)
        keep = {}
        out = O.bdnet_forward(P, x, cfg=CFG, compat_reference_bwd=True, keep=keep)
        worst = max(maxdiff(ref_out[k], out[k]) for k in ref_out if ref_out[k] is not None)
        report.append(f"{tag}: forward max|ref-oracle| over out_dict = {worst:.3e}")
        assert worst < 1e-5, worst
        margin = round_margin(out["loc"], levels, frame_num)
        assert margin > 0.9 * need

        losses = {}
        for epoch in (0, 12):
            ref_loss_mod.prior_lb = ref_loss_mod.prior_rb = None
            crit = MultiSegmentLoss(CFG["num_classes"], PIOU, 1.0, cls_loss_type="edl", edl_config=edl_cfg, os_head=True)
            crit.cls_loss.epoch = epoch
            pred = [ref_out[k] for k in ("loc", "conf", "prop_loc", "prop_conf", "center", "priors", "act", "prop_act")]
            ref7 = crit(pred, [t.clone() for t in targets])
            ora7 = O.multisegment_loss_anet(out, targets, CFG, PIOU, epoch, edl_cfg["ibm_start"])
            d = max(maxdiff(a, b) for a, b in zip(ref7, ora7))
            report.append(f"{tag}: loss 7-tuple edl@{epoch} max diff {d:.3e}  {[round(float(v.detach()), 5) for v in ora7]}")
            assert d < 1e-5, (epoch, d)
            losses[f"edl{epoch}"] = np.array([float(v) for v in ref7], np.float64)

        fx = {"param_seed": np.int64(PARAM_SEED), "clip_seed": np.int64(clip_seed), "batch": np.int64(batch),
              "round_margin": np.float64(margin), "piou": np.float64(PIOU), "scores": scores_np}
        for i, t in enumerate(targets_np):
            fx[f"target_{i}"] = t
        for k in ("loc", "prop_loc", "center", "act", "prop_act", "unct", "prop_unct"):
            fx["out_" + k] = ref_out[k].detach().numpy().copy()
        for k in ("conf", "prop_conf"):           # (b,189,150): keep every anchor of 16 classes + row sums
            fx["out_" + k + "_first16"] = ref_out[k][..., :16].detach().numpy().copy()
            fx["out_" + k + "_rowsum"] = ref_out[k].double().sum(-1).detach().numpy().copy()
        for k in ("start", "end", "start_loc_prop", "end_loc_prop", "start_conf_prop", "end_conf_prop"):
            fx["probe_" + k] = strided(ref_out[k])
            fx["sum_" + k] = np.float64(ref_out[k].double().sum())
        for i in range(len(levels)):
            fx[f"segments_{i}"] = keep["segments"][i].numpy().copy()
            fx[f"frame_segments_{i}"] = keep["frame_segments"][i].numpy().copy()
            fx[f"probe_pyramid_{i}"] = strided(keep["pyramid_feats"][i], 1024)
        fx["probe_frame_level_feat"] = strided(keep["frame_level_feat"])
        fx["probe_Mixed_5c"] = strided(keep["endpoints"]["Mixed_5c"])
        for k, v in losses.items():
            fx["loss_" + k] = v

        if with_grads:
            ref_loss_mod.prior_lb = ref_loss_mod.prior_rb = None
            crit = MultiSegmentLoss(CFG["num_classes"], PIOU, 1.0, cls_loss_type="edl", edl_config=edl_cfg, os_head=True)
            pred = [ref_out[k] for k in ("loc", "conf", "prop_loc", "prop_conf", "center", "priors", "act", "prop_act")]
            ref7 = crit(pred, [t.clone() for t in targets])
            ref_cost = ref_total_cost(ref_out, ref7, scores)
            ref_cost.backward()
            cost, parts = O.train_cost_anet(out, targets, scores, CFG, LW, CW, piou=PIOU, epoch=0)
            report.append(f"{tag}: total cost ref {float(ref_cost):.6f} oracle {float(cost):.6f}")
            assert abs(float(ref_cost) - float(cost)) < 1e-5
            cost.backward()
            ref_grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
            worst_rel = 0.0
            for k, g in ref_grads.items():
                og = P[k].grad
                assert og is not None, k
                worst_rel = max(worst_rel, maxdiff(g, og) / (float(g.abs().max()) + 1e-12))
            report.append(f"{tag}: backward (reference addressing) worst relative grad diff = {worst_rel:.3e} "
                          f"over {len(ref_grads)} tensors")
            assert worst_rel < 1e-3, worst_rel
            grads_compat = {k: P[k].grad.clone() for k in ref_grads}
            cost_val = float(ref_cost)
            del ref_out, out, P, keep, cost, ref_cost

            P2 = O.to_torch(params_np, requires_grad=True)
            out2 = O.bdnet_forward(P2, x, cfg=CFG, compat_reference_bwd=False)
            cost2, _ = O.train_cost_anet(out2, targets, scores, CFG, LW, CW, piou=PIOU, epoch=0)
            cost2.backward()
            grads_correct = {k: P2[k].grad.clone() for k in ref_grads}
            del P2, out2, cost2

            def grads64(compat):
                Pd = {k: (torch.from_numpy(v).double() if v.dtype == np.float32 else torch.from_numpy(v))
                      for k, v in params_np.items()}
                for k, v in Pd.items():
                    if v.is_floating_point() and ".bn." not in k:
                        v.requires_grad_(True)
                o = O.bdnet_forward(Pd, x.double(), cfg=CFG, compat_reference_bwd=compat)
                o["priors"] = o["priors"].double()
                c64, _ = O.train_cost_anet(o, [t.double() for t in targets], scores.double(), CFG, LW, CW,
                                           piou=PIOU, epoch=0)
                c64.backward()
                return {k: Pd[k].grad.clone() for k in ref_grads}, float(c64)
            g64_correct, c64 = grads64(False)
            g64_compat, _ = grads64(True)
            names = sorted(ref_grads)
            fx["cost_edl0"] = np.float64(cost_val)
            fx["cost_parts_edl0"] = np.array([float(parts[k]) for k in sorted(parts)], np.float64)
            fx["cost64_edl0"] = np.float64(c64)
            fx["grad_names"] = np.array(names)
            fx["gradnorm_compat"] = np.array([float(grads_compat[k].double().norm()) for k in names])
            fx["gradnorm_correct"] = np.array([float(grads_correct[k].double().norm()) for k in names])
            fx["grad64norm_compat"] = np.array([float(g64_compat[k].norm()) for k in names])
            fx["grad64norm_correct"] = np.array([float(g64_correct[k].norm()) for k in names])
            fx["grad32dist_compat"] = np.array([float((grads_compat[k].double() - g64_compat[k]).norm() /
                                                      (g64_compat[k].norm() + 1e-30)) for k in names])
            fx["grad32dist_correct"] = np.array([float((grads_correct[k].double() - g64_correct[k]).norm() /
                                                       (g64_correct[k].norm() + 1e-30)) for k in names])
            for k in ("coarse_pyramid_detection.loc_tower.0.0.conv1d.weight",
                      "coarse_pyramid_detection.pyramids.0.0.conv3d.weight",
                      "coarse_pyramid_detection.conf_head.conv1d.weight",
                      "backbone._model.Mixed_5b.b1b.conv3d.weight",
                      "backbone._model.Conv3d_1a_7x7.conv3d.weight"):
                fx["grad64probe_compat/" + k] = strided(g64_compat[k], 512)
                fx["grad64probe_correct/" + k] = strided(g64_correct[k], 512)
            report.append(f"{tag}: fp64 cost {c64:.8f}; CPU-fp32 vs fp64 gradient distance: median "
                          f"{np.median(fx['grad32dist_correct']):.2e}, max {fx['grad32dist_correct'].max():.2e} "
                          f"({names[int(fx['grad32dist_correct'].argmax())]})")
        np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **fx)
        report.append(f"{tag}: wrote tests/golden/{tag}.npz")

    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
