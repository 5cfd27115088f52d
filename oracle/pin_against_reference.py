"""oracle/pin_against_reference.py -- BUILD-CONTAINER ONLY (needs /root/reference).

Imports the reference's Python (never shipped, never copied), checks the functional
restatement in oracle/afsd_oracle.py against it on identical seeded inputs, and writes the
golden vectors the GPU-side parity tests use to tests/golden/*.npz.  Nothing in tests -m gpu,
smoke() or bench.py reads /root/reference; they read the fixtures written here.

How the reference is made importable on CPU:
  * sys.dont_write_bytecode, so nothing is written under /root/reference;
  * sys.argv carries the yaml + flags of experiments/opental/train_opental_final.sh because
    AFSD/common/config.py:101 parses them at import;
  * a stand-in module ``boundary_max_pooling_cuda`` backed by oracle/bmp_ref.c (the CUDA
    extension cannot be built: no nvcc, THC headers gone).  Its backward reproduces the
    launcher's tscale = N (boundary_max_pooling_kernel.cu:121);
  * torch.Tensor.cuda -> identity (EvidenceLoss.__init__ calls .cuda(), cls_loss.py:114).

Usage:  python -m oracle.pin_against_reference            (from the repo root)
        python -m oracle.pin_against_reference --batch=1  (one model fixture; b = 1 includes the ssl / triplet branch, a14)
        python -m oracle.pin_against_reference --ssl-only (a14 alone, merged into tests/golden/thumos_b1.npz)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"       # child processes too (joblib workers of the evaluation code)
REF = "/root/reference"
ARGV = list(sys.argv[1:])         # our own flags (sys.argv is replaced before the reference's config module is imported)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")

import numpy as np
import torch

from oracle import arch
from oracle import afsd_oracle as O


def import_reference():
    sys.path.insert(0, REF)
    sys.argv = ["pin", os.path.join(REF, "configs/thumos14_opental_final.yaml"), "--open_set",
                "--split", "0", "--lw", "1", "--cw", "10", "--ctw", "1", "--piou", "0.5",
                "--ssl", "0.001"]
    fake = types.ModuleType("boundary_max_pooling_cuda")
    fake.forward = lambda inp, seg: O.bmp_forward(inp, seg)
    fake.backward = lambda g, inp, seg: O.bmp_backward(g, inp, seg, compat_reference_bwd=True)
    sys.modules["boundary_max_pooling_cuda"] = fake
    torch.Tensor.cuda = lambda self, *a, **k: self
    from AFSD.thumos14.BDNet import BDNet
    from AFSD.thumos14.multisegment_loss import MultiSegmentLoss
    from AFSD.common.segment_utils import softnms_v2
    from AFSD.common.config import config
    import AFSD.thumos14.test as ref_test
    return BDNet, MultiSegmentLoss, softnms_v2, config, ref_test


def maxdiff(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


def round_margin(loc, levels, frame_num=256):
    """Smallest distance of any pre-rounding window value to a .5 tie (fixtures must not sit on one)."""
    worst, off = 1.0, 0
    for t in levels:
        l = loc[:, off:off + t].double()
        off += t
        pri = torch.tensor([(c + 0.5) / t for c in range(t)]).double().view(1, t, 1)
        seg = l / frame_num * t
        ctr = torch.round(pri * t - 0.5)
        plen = seg[:, :, :1] + seg[:, :, 1:]
        i_, o_ = plen.div(4).clamp(min=1), plen.div(10).clamp(min=1)
        lft, rgt = ctr - seg[:, :, :1], ctr + seg[:, :, 1:]
        d0, d1 = pri * frame_num - l[:, :, :1], pri * frame_num + l[:, :, 1:]
        pl2 = d1 - d0 + 1
        i2, o2 = pl2.div(4).clamp(min=1), pl2.div(10).clamp(min=1)
        vals = torch.cat([lft - o_, lft + i_, rgt - i_, rgt + o_, d0 - o2, d0 + i2, d1 - i2, d1 + o2], -1)
        frac = (vals - torch.floor(vals) - 0.5).abs()
        worst = min(worst, float(frac.detach().min()))
    return worst


def strided(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step].contiguous().numpy().copy()


SSL_CLIP_SEED = 77
# three (start, end) frame-space proposals per set, as thumos_dataset.py:110-141's splice writes them: the (moved)
# ground-truth piece, the piece it was cut from, and the seam; set 1 is the hand-picked triple of the first GPU test
SSL_PROPOSALS = ([[40., 90.], [100., 150.], [10., 30.]],
                 [[61., 118.], [142., 199.], [119., 141.]])


def pin_ssl(net, params_np, report):
    """a14: the reference's BDNet.forward(x, proposals, ssl=True) (BDNet.py:482-503) and the three TripletMarginLoss terms
    of forward_one_epoch(ssl=True) (train.py:177-184) at b = 1, next to O.ssl_triplets / O.triplet_cost; gradients of the
    triplet cost through the WHOLE network (second backbone pass), reference addressing (the extension's backward) and
    the correct one, with fp64 yardsticks.  -> fixture entries `ssl{set}_*`."""
    import torch.nn as nn
    fx = {"ssl_clip_seed": np.int64(SSL_CLIP_SEED), "ssl_proposals": np.array(SSL_PROPOSALS, np.float32)}
    x = torch.from_numpy(arch.make_clip(SSL_CLIP_SEED, 1))
    weights = [1, 0.1, 0.1]
    for si, props in enumerate(SSL_PROPOSALS):
        targets = [torch.tensor(props, dtype=torch.float32)]
        net.zero_grad()
        ra, rp, rn = net(x, proposals=targets, ssl=True)
        terms = [nn.TripletMarginLoss()(ra[i], rp[i], rn[i]) * weights[i] for i in range(3)]   # train.py:180-183
        trip = torch.stack(terms).sum(0)
        trip.backward()
        ref_grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        P = O.to_torch(params_np, requires_grad=True)
        oa, op, on = O.ssl_triplets(P, x, targets, compat_reference_bwd=True)
        worst = max(maxdiff(a, b) for ra_, oa_ in ((ra, oa), (rp, op), (rn, on)) for a, b in zip(ra_, oa_))
        cost = O.triplet_cost(oa, op, on, 1.0)
        report.append(f"ssl set {si}: anchor/positive/negative max|ref-oracle| = {worst:.3e}; triplet cost ref "
                      f"{float(trip):.6f} oracle {float(cost):.6f}; terms {[round(float(t), 6) for t in terms]}")
        assert worst < 1e-5 and abs(float(trip) - float(cost)) < 1e-5 * max(1.0, abs(float(trip)))
        cost.backward()
        rel = 0.0
        for k, g in ref_grads.items():
            assert P[k].grad is not None, k
            rel = max(rel, maxdiff(g, P[k].grad) / (float(g.abs().max()) + 1e-12))
        report.append(f"ssl set {si}: backward (reference addressing) worst relative grad diff = {rel:.3e} over "
                      f"{len(ref_grads)} tensors")
        assert rel < 1e-3, rel
        g_compat = {k: P[k].grad.clone() for k in ref_grads}
        P2 = O.to_torch(params_np, requires_grad=True)
        O.triplet_cost(*O.ssl_triplets(P2, x, targets, compat_reference_bwd=False), 1.0).backward()
        g_correct = {k: P2[k].grad.clone() for k in ref_grads}

        def grads64(compat):
            Pd = {k: (torch.from_numpy(v).double() if v.dtype == np.float32 else torch.from_numpy(v))
                  for k, v in params_np.items()}
            for k, v in Pd.items():
                if v.is_floating_point() and ".bn." not in k:
                    v.requires_grad_(True)
            c = O.triplet_cost(*O.ssl_triplets(Pd, x.double(), [t.double() for t in targets], compat_reference_bwd=compat), 1.0)
            c.backward()
            return {k: Pd[k].grad.clone() for k in ref_grads}, float(c)
        g64_correct, c64 = grads64(False)
        g64_compat, _ = grads64(True)
        names = sorted(ref_grads)
        t = f"ssl{si}_"
        for nm, tr3 in (("anchor", ra), ("positive", rp), ("negative", rn)):
            for i in range(3):
                fx[f"{t}{nm}_{i}"] = tr3[i].detach().numpy().copy()
        fx[t + "terms"] = np.array([float(v) for v in terms], np.float64)
        fx[t + "cost"] = np.float64(trip)
        fx[t + "cost64"] = np.float64(c64)
        fx[t + "grad_names"] = np.array(names)
        fx[t + "gradnorm_reference"] = np.array([float(ref_grads[k].double().norm()) for k in names])
        for mode, g32, g64 in (("compat", g_compat, g64_compat), ("correct", g_correct, g64_correct)):
            fx[f"{t}grad64norm_{mode}"] = np.array([float(g64[k].norm()) for k in names])
            fx[f"{t}grad32dist_{mode}"] = np.array([float((g32[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-30))
                                                    for k in names])
        report.append(f"ssl set {si}: fp64 cost {c64:.8f}; CPU-fp32 vs fp64 gradient distance median "
                      f"{np.median(fx[t + 'grad32dist_correct']):.2e}, max {fx[t + 'grad32dist_correct'].max():.2e}")
    net.zero_grad()
    return fx


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    BDNet, MultiSegmentLoss, ref_softnms, config, ref_test = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    levels = arch.level_lengths()
    report = []

    # ---- 1. state-dict layout
    net = BDNet(training=False, use_edl=True)
    sd = net.state_dict()
    spec = arch.param_spec()
    assert [k for k, _ in spec] == list(sd.keys()), "state-dict key order differs"
    for k, shp in spec:
        assert tuple(sd[k].shape) == tuple(shp), (k, shp, tuple(sd[k].shape))
    report.append(f"state_dict layout: {len(spec)} entries match")

    PARAM_SEED = 2020
    params_np = arch.make_params(PARAM_SEED)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params_np.items()})
    net.train()  # BN stays frozen (BDNet.py:39-49)

    edl_cfg = config["training"]["edl_config"]
    act_cfg = config["training"]["act_config"]

    def pick_clip_seed(batch, first, need):
        """A seeded clip whose window values all stay `need` away from a rounding tie, so that
        fp32 re-association noise (~1e-5 relative on loc) cannot flip a proposal index."""
        Pq = O.to_torch(params_np)
        for seed in range(first, first + 400):
            with torch.no_grad():
                o = O.bdnet_forward(Pq, torch.from_numpy(arch.make_clip(seed, batch)))
            m = round_margin(o["loc"], levels)
            if m > need:
                return seed, m
        raise RuntimeError("no seed with a safe rounding margin")

    only = [int(a.split("=")[1]) for a in ARGV if a.startswith("--batch=")]     # e.g. --batch=4: just that fixture
    if "--ssl-only" in ARGV:        # re-pin a14 alone: the other entries of thumos_b1.npz are kept as they are
        path = os.path.join(GOLD, "thumos_b1.npz")
        old = dict(np.load(path))
        old = {k: v for k, v in old.items() if not k.startswith("ssl")}
        old.update(pin_ssl(net, params_np, report))
        np.savez_compressed(path, **old)
        with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
            f.write("\n".join(report[1:]) + "\n")
        print("\n".join(report))
        leftovers = [os.path.join(d, n) for d, _, fs in os.walk(REF) for n in fs if n.endswith(".pyc")]
        assert not leftovers, leftovers
        return
    for batch, first, need in ((1, 11, 6e-4), (2, 500, 4e-4), (4, 900, 3e-4)):
        if only and batch not in only:
            continue
        tag = f"thumos_b{batch}"
        clip_seed, m0 = pick_clip_seed(batch, first, need)
        report.append(f"{tag}: clip seed {clip_seed} (rounding margin {m0:.2e})")
        x = torch.from_numpy(arch.make_clip(clip_seed, batch))
        targets_np = arch.make_targets(clip_seed + 100, batch)
        scores_np = arch.make_scores(targets_np)
        targets = [torch.from_numpy(t) for t in targets_np]
        scores = torch.from_numpy(scores_np)

        # ---- 2. forward parity (reference modules vs functional restatement)
        net.zero_grad()
        ref_out = net(x)
        P = O.to_torch(params_np, requires_grad=True)
        keep = {}
        out = O.bdnet_forward(P, x, compat_reference_bwd=True, keep=keep)
        worst = 0.0
        for k in ref_out:
            if ref_out[k] is None:
                continue
            worst = max(worst, maxdiff(ref_out[k], out[k]))
        report.append(f"{tag}: forward max|ref-oracle| over out_dict = {worst:.3e}")
        assert worst < 1e-5, worst
        margin = round_margin(out["loc"], levels)
        report.append(f"{tag}: min distance of window values to a rounding tie = {margin:.3e}")
        assert margin > 0.9 * need, "fixture sits on a rounding tie; pick another seed"

        # ---- 3. loss parity: edl epoch 0, edl epoch >= ibm_start, focal (as-shipped dispatch H2)
        losses = {}
        for mode, epoch in (("edl", 0), ("edl", 12), ("focal", 0)):
            crit = MultiSegmentLoss(15, 0.5, 1.0, cls_loss_type=mode, edl_config=edl_cfg,
                                    os_head=True, act_config=act_cfg)
            state = None
            if mode == "edl":
                crit.cls_loss.epoch = epoch
                state = O.EvidenceState(edl_cfg["num_bins"], edl_cfg["momentum"], edl_cfg["ibm_start"])
                state.epoch = epoch
            ref7 = crit({k: v for k, v in ref_out.items()}, targets)
            ora7 = O.multisegment_loss(out, targets, piou=0.5, cls_loss_type=mode, state=state,
                                       act_weight=act_cfg["weight"])
            d = max(maxdiff(a, b) for a, b in zip(ref7, ora7))
            report.append(f"{tag}: loss 7-tuple {mode}@{epoch} max diff {d:.3e}  "
                          f"{[round(float(v), 5) for v in ora7]}")
            assert d < 1e-5, (mode, epoch, d)
            name = f"{mode}{epoch}"
            losses[name] = np.array([float(v) for v in ref7], np.float64)
            if state is not None and epoch >= edl_cfg["ibm_start"]:
                wd = maxdiff(crit.cls_loss.weight_accum, state.weight_accum)
                assert wd < 1e-6, wd
                losses[name + "_weight_accum"] = state.weight_accum.numpy().copy()

        # ---- 4. backward parity through the full training cost (edl, epoch 0)
        crit = MultiSegmentLoss(15, 0.5, 1.0, cls_loss_type="edl", edl_config=edl_cfg,
                                os_head=True, act_config=act_cfg)
        ref7 = crit(ref_out, targets)
        import torch.nn.functional as F

        def ref_bce(s, e, sc):  # restated from train.py:152-161 (train.py itself needs tensorboardX)
            s = torch.tanh(s).mean(-1)
            e = torch.tanh(e).mean(-1)
            return (F.binary_cross_entropy(s.view(-1), sc[:, 0].contiguous().view(-1)),
                    F.binary_cross_entropy(e.view(-1), sc[:, 1].contiguous().view(-1)))
        ls, le = ref_bce(ref_out["start"], ref_out["end"], scores)
        sc4 = F.interpolate(scores, scale_factor=0.25)
        a, b_ = ref_bce(ref_out["start_loc_prop"], ref_out["end_loc_prop"], sc4)
        c, d_ = ref_bce(ref_out["start_conf_prop"], ref_out["end_conf_prop"], sc4)
        ref_cost = ref7[0] * 1 + ref7[1] * 10 + ref7[2] * 1 + ref7[3] * 10 + ref7[4] * 1 + \
            (ls + 0.1 * (a + c)) + (le + 0.1 * (b_ + d_)) + ref7[5] * 1 + ref7[6] * 1
        ref_cost.backward()
        cost, parts = O.train_cost(out, targets, scores, piou=0.5, cls_loss_type="edl",
                                   state=O.EvidenceState(), act_weight=act_cfg["weight"])
        report.append(f"{tag}: total cost ref {float(ref_cost):.6f} oracle {float(cost):.6f}")
        assert abs(float(ref_cost) - float(cost)) < 1e-5
        cost.backward()
        ref_grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
        worst_rel = 0.0
        for k, g in ref_grads.items():
            og = P[k].grad
            assert og is not None, k
            rel = maxdiff(g, og) / (float(g.abs().max()) + 1e-12)
            worst_rel = max(worst_rel, rel)
        report.append(f"{tag}: backward (reference addressing) worst relative grad diff = {worst_rel:.3e} "
                      f"over {len(ref_grads)} tensors")
        assert worst_rel < 1e-3, worst_rel
        grads_compat = {k: P[k].grad.clone() for k in ref_grads}

        # correct-backward variant (oracle only; the reference cannot produce it)
        P2 = O.to_torch(params_np, requires_grad=True)
        out2 = O.bdnet_forward(P2, x, compat_reference_bwd=False)
        cost2, _ = O.train_cost(out2, targets, scores, piou=0.5, cls_loss_type="edl",
                                state=O.EvidenceState(), act_weight=act_cfg["weight"])
        cost2.backward()
        grads_correct = {k: P2[k].grad.clone() for k in ref_grads}

        # ---- 4b. fp64 runs of the restatement: the yardstick for gradient parity.  fp32 gradients of
        # this 60-layer network are only conditioned to ~3e-3 at the first conv (ReLU masks and
        # summation order); the GPU path is required to be as close to fp64 as the CPU fp32 path is.
        def grads64(compat):
            Pd = {k: (torch.from_numpy(v).double() if v.dtype == np.float32 else torch.from_numpy(v))
                  for k, v in params_np.items()}
            for k, v in Pd.items():
                if v.is_floating_point() and ".bn." not in k:
                    v.requires_grad_(True)
            o = O.bdnet_forward(Pd, x.double(), compat_reference_bwd=compat)
            o["priors"] = o["priors"].double()
            c64, _ = O.train_cost(o, [t.double() for t in targets], scores.double(), piou=0.5,
                                  cls_loss_type="edl", state=O.EvidenceState(), act_weight=act_cfg["weight"])
            c64.backward()
            return {k: Pd[k].grad.clone() for k in ref_grads}, float(c64)
        g64_correct, c64 = grads64(False)
        g64_compat, _ = grads64(True)
        report.append(f"{tag}: fp64 cost {c64:.8f}")

        # ---- 5. fixture
        fx = {"param_seed": np.int64(PARAM_SEED), "clip_seed": np.int64(clip_seed),
              "batch": np.int64(batch), "round_margin": np.float64(margin)}
        for i, t in enumerate(targets_np):
            fx[f"target_{i}"] = t
        fx["scores"] = scores_np
        for k in ("loc", "conf", "prop_loc", "prop_conf", "center", "act", "prop_act", "unct", "prop_unct"):
            fx["out_" + k] = ref_out[k].detach().numpy().copy()
        for k in ("start", "end", "start_loc_prop", "end_loc_prop", "start_conf_prop", "end_conf_prop"):
            fx["probe_" + k] = strided(ref_out[k])
            fx["sum_" + k] = np.float64(ref_out[k].double().sum())
        for i in range(len(levels)):
            fx[f"segments_{i}"] = keep["segments"][i].numpy().copy()
            fx[f"frame_segments_{i}"] = keep["frame_segments"][i].numpy().copy()
            fx[f"probe_pyramid_{i}"] = strided(keep["pyramid_feats"][i], 1024)
            fx[f"probe_loc_feat_{i}"] = strided(keep["loc_feat"][i], 1024)
        fx["probe_frame_level_feat"] = strided(keep["frame_level_feat"])
        for name in ("Conv3d_1a_7x7", "Conv3d_2c_3x3", "Mixed_3c", "Mixed_4f", "Mixed_5c"):
            fx["probe_" + name] = strided(keep["endpoints"][name])
            fx["absmean_" + name] = np.float64(keep["endpoints"][name].double().abs().mean())
        for k, v in losses.items():
            fx["loss_" + k] = v
        fx["cost_edl0"] = np.float64(ref_cost)
        fx["cost_parts_edl0"] = np.array([float(parts[k]) for k in sorted(parts)], np.float64)
        names = sorted(ref_grads)
        fx["grad_names"] = np.array(names)
        fx["gradnorm_compat"] = np.array([float(grads_compat[k].double().norm()) for k in names])
        fx["gradnorm_correct"] = np.array([float(grads_correct[k].double().norm()) for k in names])
        fx["gradnorm_reference"] = np.array([float(ref_grads[k].double().norm()) for k in names])
        for k in ("coarse_pyramid_detection.loc_tower.0.0.conv1d.weight",
                  "coarse_pyramid_detection.deconv.0.conv1d.weight",
                  "coarse_pyramid_detection.conf_proposal_branch.roi_conv.0.conv1d.weight",
                  "coarse_pyramid_detection.pyramids.0.0.conv3d.weight",
                  "backbone._model.Mixed_4f.b1b.conv3d.weight",
                  "backbone._model.Conv3d_1a_7x7.conv3d.weight"):
            fx["gradprobe_compat/" + k] = strided(grads_compat[k], 512)
            fx["gradprobe_correct/" + k] = strided(grads_correct[k], 512)
            fx["grad64probe_compat/" + k] = strided(g64_compat[k], 512)
            fx["grad64probe_correct/" + k] = strided(g64_correct[k], 512)
        fx["grad64norm_compat"] = np.array([float(g64_compat[k].norm()) for k in names])
        fx["grad64norm_correct"] = np.array([float(g64_correct[k].norm()) for k in names])
        # per-tensor distance of the CPU fp32 gradient from the fp64 one: ||g32 - g64|| / ||g64||
        fx["grad32dist_compat"] = np.array([float((grads_compat[k].double() - g64_compat[k]).norm() /
                                                  (g64_compat[k].norm() + 1e-30)) for k in names])
        fx["grad32dist_correct"] = np.array([float((grads_correct[k].double() - g64_correct[k]).norm() /
                                                   (g64_correct[k].norm() + 1e-30)) for k in names])
        fx["cost64_edl0"] = np.float64(c64)
        report.append(f"{tag}: CPU-fp32 vs fp64 gradient distance: median "
                      f"{np.median(fx['grad32dist_correct']):.2e}, max {fx['grad32dist_correct'].max():.2e} "
                      f"({names[int(fx['grad32dist_correct'].argmax())]})")
        if batch == 1:
            fx.update(pin_ssl(net, params_np, report))       # a14: the self-supervised triplet branch
        np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **fx)
        report.append(f"{tag}: wrote tests/golden/{tag}.npz")

    if only:            # a single model fixture was requested: keep the other files, append to the report
        with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
            f.write("\n".join(report[1:]) + "\n")
        print("\n".join(report))
        leftovers = [os.path.join(d, n) for d, _, fs in os.walk(REF) for n in fs if n.endswith(".pyc")]
        assert not leftovers, leftovers
        return

    # ---- 6. Soft-NMS parity + fixtures
    rs = np.random.RandomState(5)
    nms_fx = {}
    for n in (0, 1, 2, 50, 400, 2000):
        centres = rs.uniform(5, 300, size=max(1, min(12, n)))
        c = centres[rs.randint(0, len(centres), size=n)] + rs.normal(0, 3.0, size=n)
        w = np.abs(rs.normal(8, 4, size=n)) + 0.5
        seg = np.stack([c - w / 2, c + w / 2, rs.beta(0.5, 2.0, size=n), rs.uniform(0, 1, n),
                        rs.uniform(0.5, 1, n)], -1).astype(np.float32).reshape(n, 5)
        t = torch.from_numpy(seg)
        ref_rows, ref_cnt, ref_mask = ref_softnms(t.clone(), sigma=0.5, top_k=5000,
                                                  score_threshold=0.001, use_edl=True, os_head=True,
                                                  get_mask=True)
        rows, cnt, mask = O.softnms_v2(t.clone())
        rows_c, cnt_c, mask_c = O.softnms_v2_c(t.clone())
        assert int(ref_cnt) == cnt == cnt_c, (n, int(ref_cnt), cnt, cnt_c)
        assert bool((ref_mask == mask).all()) and bool((ref_mask == mask_c).all())
        if cnt:
            assert maxdiff(ref_rows, rows) == 0.0
            assert maxdiff(ref_rows, rows_c) < 1e-6
        nms_fx[f"in_{n}"] = seg
        nms_fx[f"rows_{n}"] = ref_rows.numpy().reshape(-1, 5).copy()
        nms_fx[f"mask_{n}"] = ref_mask.numpy().copy()
        report.append(f"softnms n={n}: kept {int(ref_cnt)} (torch restatement exact, C restatement same set)")
    # top_k cut
    t = torch.from_numpy(nms_fx["in_400"])
    ref_rows, ref_cnt, ref_mask = ref_softnms(t.clone(), sigma=0.5, top_k=20, score_threshold=0.001,
                                              use_edl=True, os_head=True, get_mask=True)
    rows_c, cnt_c, mask_c = O.softnms_v2_c(t.clone(), top_k=20)
    assert int(ref_cnt) == cnt_c and bool((ref_mask == mask_c).all())
    nms_fx["rows_400_top20"] = ref_rows.numpy().copy()
    nms_fx["mask_400_top20"] = ref_mask.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "softnms.npz"), **nms_fx)

    # ---- 7. decode + filtering parity (test.py:79-162) on the b=2 outputs
    out_layer = ref_test.DirichletLayer(evidence="exp", dim=-1)
    dec_fx = {}
    with torch.no_grad():
        for idx, (offset, fps) in enumerate(((0, 10.0), (384, 10.0))):
            single = {k: (v[idx:idx + 1] if (v is not None and k != "priors") else v) for k, v in ref_out.items()}
            parsed = ref_test.parse_output(single, use_edl=True, os_head=True)
            loc, conf, ploc, pconf, center, priors, unct, punct, act, pact = parsed
            seg_r, score_r, unct_r, act_r = ref_test.decode_predictions(
                loc, ploc, priors, conf, pconf, unct, punct, act, pact, center, offset, fps, 256, 15,
                score_func=out_layer, use_edl=True, os_head=True)
            seg_o, score_o, unct_o, act_o = O.decode_predictions(out, idx, offset, fps)
            d = max(maxdiff(seg_r, seg_o), maxdiff(score_r, score_o), maxdiff(unct_r, unct_o), maxdiff(act_r, act_o))
            assert d < 1e-6, d
            dec_fx[f"seg_{idx}"] = seg_r.numpy().copy()
            dec_fx[f"score_{idx}"] = score_r.numpy().copy()
            dec_fx[f"unct_{idx}"] = unct_r.numpy().copy()
            dec_fx[f"act_{idx}"] = act_r.numpy().copy()
            for cl in (0, 7, 14):
                fr = ref_test.filtering(seg_r, score_r[cl], unct_r, act_r, 0.01, use_edl=True, os_head=True)
                fo = O.filtering(seg_o, score_o[cl], unct_o, act_o, 0.01)
                assert (fr is None) == (fo is None)
                if fr is not None:
                    assert fr.shape == fo.shape and maxdiff(fr, fo) < 1e-6
                    dec_fx[f"filtered_{idx}_{cl}"] = fr.numpy().copy()
            report.append(f"decode/filter sample {idx}: max diff {d:.2e}")
    np.savez_compressed(os.path.join(GOLD, "decode_b2.npz"), **dec_fx)

    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "w") as f:
        f.write("Produced by oracle/pin_against_reference.py in the build container "
                "(reference imported from /root/reference, torch %s).\n" % torch.__version__)
        f.write("\n".join(report) + "\n")
    print("\n".join(report))
    leftovers = [os.path.join(d, n) for d, _, fs in os.walk(REF) for n in fs if n.endswith(".pyc")]
    assert not leftovers, leftovers


if __name__ == "__main__":
    main()
