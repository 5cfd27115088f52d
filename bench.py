#!/usr/bin/env python
"""bench.py -- clips/sec of one OpenTAL THUMOS14 training step on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward (I3D + temporal pyramid + heads) + MultiSegmentLoss (EDL/IBM + actionness + tIoU
quality + boundary BCE) + backward + bucketed gradient all-reduce (RCCL, overlapped) + Adam, on a
batch of synthetic 256x3x96x96 clips already resident in HBM.  `value` = world_size * batch /
max-over-ranks step time.  Rank 0 prints ONE JSON line with `roofline` (the implicit-GEMM
convolution, the kernel that dominates the step) and `cpu_baseline` (the CPU oracle timed on the
host cores of this box on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

EDL = dict(evidence='exp', loss_type='log', iou_aware=True, with_focal=False, alpha=0.25, gamma=2, with_ibm=True,
           ibm_start=10, momentum=0.99, num_bins=50)
ACT = dict(margin=1.0, weight=0)
W = dict(lw=1.0, cw=10.0, ctw=1.0, actw=1.0, ssl=0.001)   # experiments/opental/train_opental_final.sh
PEAK_TFLOPS = {"f32": 157.3,     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak
               "bf16": 2500.0}   # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), not the 2:1-sparsity headline


def synth_batch(batch, seed, device, frames=256, classes=15, score_rows=2):
    """Synthetic clips / targets of the THUMOS14 shape (no dataset in the container); frames=768, classes=150,
    score_rows=3 ([action, start, end], AFSD/common/anet_dataset.py:250-255) for the ActivityNet recipe."""
    rs = np.random.RandomState(seed)
    g = torch.Generator(device=device).manual_seed(seed)
    clips = torch.randint(0, 256, (batch, 3, frames, 96, 96), device=device, generator=g, dtype=torch.uint8)
    clips = (clips.float() / 255.0) * 2.0 - 1.0
    targets, scores = [], np.zeros((batch, score_rows, frames), np.float32)
    for i in range(batch):
        rows = []
        for _ in range(rs.randint(1, 4)):
            length = rs.uniform(8.0 / frames, 0.6)
            start = rs.uniform(0.0, 1.0 - length)
            rows.append([start, start + length, float(rs.randint(1, classes + 1))])
            s_f, e_f = start * frames, (start + length) * frames
            d = max((e_f - s_f) / 10.0, 2.0)
            for ch, c in ((score_rows - 2, s_f), (score_rows - 1, e_f)):
                lo = int(np.clip(int(round(c - d / 2)), 0, frames - 1)); hi = int(np.clip(int(round(c + d / 2)), 0, frames - 1)) + 1
                scores[i, ch, lo:hi] = 1.0
            if score_rows == 3:
                scores[i, 0, int(s_f):int(np.ceil(e_f))] = 1.0
        targets.append(torch.tensor(rows, dtype=torch.float32, device=device))
    return clips, targets, torch.from_numpy(scores).to(device)


def synth_label_ring(batch, seed, device, n=8, frames=256, classes=15, score_rows=2, max_targets=4):
    """`n` different label batches (1-3 targets per clip, drawn like synth_batch's) as fixed-shape device records
    (opental_amd.common.input_pipeline.LabelRecord, padded to `max_targets` rows + validity): what run_one_epoch hands the
    step for real data, where every batch has other target counts (AFSD/thumos14/train.py:221-224).  Record 0 holds the
    labels of synth_batch(batch, seed)."""
    from opental_amd.common.input_pipeline import LabelRecord
    ring = []
    for i in range(n):
        rs = np.random.RandomState(seed + 7919 * i)
        host = LabelRecord(batch, max_targets, score_rows, frames)
        samples = []
        for _ in range(batch):
            rows, sc = [], np.zeros((score_rows, frames), np.float32)
            for _ in range(rs.randint(1, 4)):
                length = rs.uniform(8.0 / frames, 0.6)
                start = rs.uniform(0.0, 1.0 - length)
                rows.append([start, start + length, float(rs.randint(1, classes + 1))])
                s_f, e_f = start * frames, (start + length) * frames
                d = max((e_f - s_f) / 10.0, 2.0)
                for ch, c in ((score_rows - 2, s_f), (score_rows - 1, e_f)):
                    lo = int(np.clip(int(round(c - d / 2)), 0, frames - 1)); hi = int(np.clip(int(round(c + d / 2)), 0, frames - 1)) + 1
                    sc[ch, lo:hi] = 1.0
                if score_rows == 3:
                    sc[0, int(s_f):int(np.ceil(e_f))] = 1.0
            samples.append({'target': np.asarray(rows, np.float32), 'scores': sc})
        host.fill(samples)
        rec = LabelRecord(batch, max_targets, score_rows, frames, device=device)
        rec.flat.copy_(host.flat)
        rec.counts = tuple(len(s['target']) for s in samples)
        ring.append(rec)
    return ring


def build_anet_trainer(device, seed=2020, force_collectives=False):
    """BASELINE configs[3], read from configs/anet_opental.yaml (768-frame clips, 150 classes + background, batch 2, lr 1e-4 /
    backbone 1e-5, wd 1e-4) with the flags of AFSD/anet/README.md:61 (--lw=1 --cw=1 --piou=0.6), through the recipe's own
    builder (opental_amd.anet.train.build_training: the path `python -m opental_amd.anet.train <yaml>` takes)."""
    from opental_amd.anet.train import build_training
    from opental_amd.common import config as C
    cfg = C.get_config([os.path.join(REPO, "configs", "anet_opental.yaml"), "--open_set", "--split", "0",
                        "--lw", "1", "--cw", "1", "--piou", "0.6", "--ssl", "0.1"])
    torch.manual_seed(seed)
    net, crit, trainer = build_training(cfg, device, random_init=True, force_collectives=force_collectives)
    crit.cls_loss.epoch = 12            # past ibm_start: the IBM re-weighting runs inside the timed step
    trainer.yaml_batch = int(cfg['training']['batch_size'])
    return trainer


def build_trainer(device, seed=2020, force_collectives=False):
    from opental_amd.thumos14.BDNet import BDNet
    from opental_amd.thumos14.multisegment_loss import MultiSegmentLoss
    from opental_amd.thumos14.train import DetectorTrainer
    torch.manual_seed(seed)
    net = BDNet(in_channels=3, training=False, use_edl=True)      # no pretrained file in the container
    net.backbone._model.apply(BDNet.weight_init)                  # random-init weights of the architecture
    net = net.to(device).train()
    crit = MultiSegmentLoss(15, 0.5, 1.0, cls_loss_type='edl', edl_config=EDL, os_head=True, act_config=ACT).to(device)
    crit.cls_loss.epoch = 12            # past ibm_start: the IBM re-weighting runs inside the timed step
    return DetectorTrainer(net, crit, W, lr=1e-5, weight_decay=1e-3, force_collectives=force_collectives)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget=20.0, threads=32, batch=1):
    """The CPU oracle (oracle/afsd_oracle.py: the restatement pinned against the imported reference)
    doing the same training step -- forward, loss, backward, Adam -- at batch 1 on this box's host
    cores.  Bounded: the first step doubles as warm-up and is the sample if it alone exceeds the
    budget; otherwise further steps are timed until the budget is spent.  torch-CPU conv3d does not
    scale past a few dozen threads (256 threads measured 392 s/step on the EPYC 9575F box), so the
    thread count is capped and reported as `cores`."""
    from oracle import afsd_oracle as O, arch
    torch.set_num_threads(min(threads, os.cpu_count()))
    P = O.to_torch(arch.make_params(2020), requires_grad=True)
    train = [k for k, v in P.items() if v.requires_grad]
    m = {k: torch.zeros_like(P[k]) for k in train}
    v = {k: torch.zeros_like(P[k]) for k in train}
    x = torch.from_numpy(arch.make_clip(3, batch))
    tg = [torch.from_numpy(t) for t in arch.make_targets(5, batch)]
    sc = torch.from_numpy(arch.make_scores(arch.make_targets(5, batch)))
    st = O.EvidenceState()
    st.epoch = 12

    def one(step):
        t0 = time.time()
        for k in train:
            P[k].grad = None
        out = O.bdnet_forward(P, x)
        cost, _ = O.train_cost(out, tg, sc, state=st)
        cost.backward()
        with torch.no_grad():
            for k in train:
                O.adam_step(P[k], P[k].grad, m[k], v[k], step, 1e-5, 1e-3)
        return time.time() - t0
    first = one(1)
    times, spent = [], first
    while spent < seconds_budget and len(times) < 5:
        times.append(one(len(times) + 2))
        spent += times[-1]
    dt = float(np.median(times)) if times else first
    what = f"median of {len(times)} steps after 1 warm-up" if times else "1 cold step (it alone exceeded the budget)"
    return {"value": round(batch / dt, 5), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "batch": batch,
            "sample": f"full training step (fwd+loss+bwd+Adam) at batch {batch}, 256x3x96x96, fp32, {what}; "
                      f"CPU oracle = torch-CPU restatement pinned to the reference; {dt:.2f} s/step"}


def cpu_baseline_all_threads(timeout_s=60):
    """One training step of the CPU oracle at os.cpu_count() threads, in a child process that is killed after `timeout_s`:
    torch-CPU's conv3d gets SLOWER past a few dozen threads on the many-core hosts of the GPU boxes (256 threads: 392 s
    per step measured), so the all-threads figure is reported as measured-or-timed-out next to the 32-thread one."""
    import subprocess
    code = ("import sys, json, os; sys.path.insert(0, %r); import bench; "
            "r = bench.cpu_baseline(seconds_budget=0.0, threads=os.cpu_count(), batch=1); print('CPUALL ' + json.dumps(r))" % REPO)
    t0 = time.time()
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s).stdout
        for line in out.splitlines():
            if line.startswith("CPUALL "):
                r = json.loads(line[7:])
                return {"value": r["value"], "unit": "clips/s", "cores": r["cores"], "sample": r["sample"]}
        return {"value": None, "cores": os.cpu_count(), "note": "child process produced no result"}
    except subprocess.TimeoutExpired:
        return {"value": None, "cores": os.cpu_count(), "timeout_s": timeout_s,
                "note": f"one cold step at {os.cpu_count()} threads did not finish within {timeout_s} s (< {1.0 / timeout_s:.4f} clips/s)"}
    except Exception as e:                          # noqa: BLE001
        return {"value": None, "cores": os.cpu_count(), "note": f"{type(e).__name__}: {str(e)[:120]} after {time.time() - t0:.0f} s"}


def _timed_steps(trainer, batch, steps, warm, extra=()):
    for _ in range(warm):
        trainer.step(*batch, *extra)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step(*batch, *extra)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def extras(device, batch):
    """The other BASELINE.json configs, measured in the same run (short legs; N = 1 only): the exact-fp32 parity path,
    the step with the ssl branch on every iteration, the ActivityNet recipe (configs[3]) at the yaml's batch, the
    inference path over 213 synthetic videos (configs[4]) with Soft-NMS timed against the CPU C oracle, and the
    pinned / double-buffered input pipeline."""
    from opental_amd.common import ops
    out = {}
    saved = ops.CONV_PRECISION

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:                      # noqa: BLE001 -- an extra must never take the headline down
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def fp32():
        ops.CONV_PRECISION = 0
        tr = build_trainer(device)
        dt = _timed_steps(tr, synth_batch(batch, 1000, device), 5, 2)
        return {"clips_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 2), "batch": batch,
                "what": "the exact-fp32 path every 1e-4 parity test runs (v_mfma_f32_32x32x2_f32)"}

    def b1():
        ops.CONV_PRECISION = 1
        tr = build_trainer(device)
        b = synth_batch(1, 1000, device)
        eager = _timed_steps(tr, b, 20, 5)
        # one clip per step is launch-bound: the step replays from captured HIP graphs -- a sequence of them on two streams
        # (main lane | weight-gradient lane, DetectorTrainer.capture_step(lanes=True)); the one-graph step is timed beside it
        tr.capture_step(*b)
        one = _timed_steps(tr, b, 30, 10)
        tr.capture_step(*b, warmup=0, lanes=True)
        dt = _timed_steps(tr, b, 100, 20)           # 20 warm-up + 100 timed steps
        return {"clips_per_s": round(1 / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": 1, "steps": 100, "warmup": 20,
                "launch": "lane graphs: a sequence of captured HIP graphs per step on two streams", "eager_ms_per_step": round(eager * 1e3, 3),
                "one_graph_ms_per_step": round(one * 1e3, 3),
                "what": "BASELINE configs[1] at the yaml's own batch_size: 1 (configs/thumos14_opental_final.yaml), bf16 operands"}

    def ssl():
        ops.CONV_PRECISION = 1
        tr = build_trainer(device)
        b = synth_batch(batch, 1000, device)
        ssl_clips, _, _ = synth_batch(batch, 2000, device)
        tg = [torch.tensor([[0.30, 0.55], [0.32, 0.52], [0.70, 0.90]], device=device) * 256 for _ in range(batch)]
        dt = _timed_steps(tr, b, 8, 3, (ssl_clips, tg))
        return {"clips_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 2), "batch": batch,
                "what": "ssl / triplet branch on EVERY step (a second backbone pass; the reference runs it when flags[0])"}

    def anet():
        ops.CONV_PRECISION = 1
        tr = build_anet_trainer(device)
        nb = getattr(tr, "yaml_batch", 2)
        b = synth_batch(nb, 1000, device, frames=768, classes=150, score_rows=3)
        eager = _timed_steps(tr, b, 6, 3)
        tr.capture_step(*b, lanes=True)             # two clips per step: the host cannot issue ~400 launches in the step's GPU time
        dt = _timed_steps(tr, b, 10, 3)
        best = min(eager, dt)
        return {"clips_per_s": round(nb / best, 1), "ms_per_step": round(best * 1e3, 2), "batch": nb,
                "launch": "lane graphs (captured HIP graphs on two streams)" if dt <= eager else "eager launches",
                "eager_ms_per_step": round(eager * 1e3, 2), "graph_ms_per_step": round(dt * 1e3, 2),
                "what": "BASELINE configs[3]: configs/anet_opental.yaml (read through opental_amd.anet.train.build_training), 768-frame "
                        "clips, 150 classes, the yaml's per-GPU batch"}

    def inference():
        from opental_amd.thumos14 import test as T
        from opental_amd.thumos14.BDNet import BDNet
        ops.CONV_PRECISION = 1
        torch.manual_seed(0)
        net = BDNet(training=False, use_edl=True)
        net.backbone._model.apply(BDNet.weight_init)
        net = net.to(device).eval()
        rs = np.random.RandomState(0)
        nvid = 213                                  # THUMOS14 test videos with temporal annotations
        frames = rs.randint(600, 4001, size=nvid)
        g = torch.Generator(device=device).manual_seed(0)
        nclips = sum(len(T.get_offsets(int(f), 256, 128)) for f in frames)
        warm = [torch.randint(0, 256, (3, 700, 96, 96), device=device, generator=g, dtype=torch.uint8)]
        T.detect_batch(net, warm, 10.0, batch_clips=32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kept = 0
        for i in range(0, nvid, 16):                # 16 videos' frames resident at a time (uint8, <= 1.8 GB)
            vids = [torch.randint(0, 256, (3, int(f), 96, 96), device=device, generator=g, dtype=torch.uint8) for f in frames[i:i + 16]]
            rows, counts, _, _ = T.detect_batch(net, vids, 10.0, batch_clips=32)
            kept += int(counts.sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res = {"videos": nvid, "windows": nclips, "seconds": round(dt, 3), "proposals_per_s": round(126 * nclips / dt, 1),
               "what": "BASELINE configs[4]: forward (bf16 operands) + decode + filter + Soft-NMS over 213 synthetic videos of "
                       "600-4000 frames, 126 proposals per 256-frame window; includes generating the frames on the device"}
        # Soft-NMS alone on head outputs with realistic overlap, against the CPU C oracle on a bounded sample
        V, C, A, K = 213, 24, 126, 15
        n = V * C
        ctr = torch.rand(n, 12, device=device, generator=g) * 300
        pick = torch.randint(0, 12, (n, A), device=device, generator=g)
        c = torch.gather(ctr, 1, pick) + torch.randn(n, A, device=device, generator=g) * 3
        w = torch.randn(n, A, device=device, generator=g).abs() * 4 + 6
        sd = dict(seg=torch.stack([c - w / 2, c + w / 2], -1).contiguous(),
                  score=torch.distributions.Beta(0.5, 2.0).sample((n, K, A)).to(device).contiguous(),
                  unct=torch.rand(n, A, device=device, generator=g), actn=torch.rand(n, A, device=device, generator=g) * 0.6 + 0.4)
        sd["flag"] = ((sd["score"] > 0.01) & (sd["actn"][:, None, :] > 0.5)).to(torch.uint8)
        cs = list(range(0, n + 1, C))
        T.softnms_classes(sd, cs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); T.softnms_classes(sd, cs); e1.record(); torch.cuda.synchronize()
        cand = int(sd["flag"].sum())
        res.update({"softnms_candidates": cand, "softnms_ms": round(e0.elapsed_time(e1), 3),
                    "softnms_candidates_per_s": round(cand / e0.elapsed_time(e1) * 1e3, 1)})
        from oracle import afsd_oracle as O
        flag = sd["flag"][:8 * C].cpu().numpy().astype(bool); seg = sd["seg"][:8 * C].cpu().numpy(); sc = sd["score"][:8 * C].cpu().numpy()
        t0 = time.perf_counter(); ncpu = 0
        for v in range(8):
            for k in range(K):
                rows_ = [np.concatenate([seg[ci][flag[ci, k]], sc[ci, k][flag[ci, k], None]], -1) for ci in range(cs[v], cs[v + 1])]
                cnd = torch.from_numpy(np.concatenate(rows_, 0))
                ncpu += len(cnd); O.softnms_v2_c(cnd)
        res["cpu_softnms_candidates_per_s"] = round(ncpu / (time.perf_counter() - t0), 1)
        res["cpu_sample"] = f"{8 * K} (video, class) problems, C oracle (oracle/bmp_ref.c), 1 core"
        return res

    def input_pipeline():
        from opental_amd.common import thumos_dataset as D
        ops.CONV_PRECISION = 1
        rs = np.random.RandomState(1)
        vids = [torch.from_numpy(rs.randint(0, 256, (1200, 112, 112, 3)).astype(np.uint8)).pin_memory() for _ in range(4)]
        st = D.ClipStager(batch, 256, 112, 112, 96, device=device, max_targets=8, score_rows=2,
                          copy_stream=ops.side_wgrads(device).side)     # as the drivers build it (thumos14.train.main)

        def samples(k):
            r = np.random.RandomState(k)
            out = []
            for _ in range(batch):
                n = int(r.randint(1, 7))                    # 1-6 targets per clip: every batch has other counts
                a = np.sort(r.uniform(0.0, 1.0, (n, 2)), 1)
                a[:, 1] = np.maximum(a[:, 1], a[:, 0] + 8.0 / 256)
                tg = np.concatenate([a, r.randint(1, 16, (n, 1))], 1).astype(np.float32)
                sc = (r.uniform(size=(2, 256)) < 0.05).astype(np.float32)
                out.append({"video": vids[int(r.randint(4))], "offset": int(r.randint(0, 900)), "frame_map": None,
                            "crop": (int(r.randint(17)), int(r.randint(17)), bool(r.randint(2))), "target": tg, "scores": sc})
            return out
        for k in range(3):
            st.submit(samples(k)); st.collect()
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        st.submit(samples(100))
        for k in range(n):
            clips, _ = st.collect()
            st.submit(samples(101 + k))
        st.collect()
        torch.cuda.synchronize()
        alone = batch * (n + 1) / (time.perf_counter() - t0)
        # the epoch loop's own sequence (thumos14.train.run_one_epoch): labels travel as one fixed-shape pinned record next to
        # the frames, the plain step replays the captured lane graphs, the clip kernel writes into the captured clip buffer
        tr = build_trainer(device)
        tr.launch = 'lanes'
        pre = [samples(200 + k) for k in range(48)]         # the decisions of 48 batches (a dataset's `decide`, host-only)
        st.submit(pre[0])
        for k in range(12):                                 # eager step, capture, ten replays (a graph's first launches are slow)
            static = tr.static_inputs()
            clips, _ = st.collect(out=None if static is None else static[0])
            rec = st.labels()
            st.submit(pre[k + 1])
            tr.step(clips, rec.targets, rec.scores)
            st.release()
        torch.cuda.synchronize()
        replayed = tr.replayed_steps
        t0 = time.perf_counter()
        for k in range(30):
            clips, _ = st.collect(out=tr.static_inputs()[0])
            rec = st.labels()
            st.submit(pre[13 + k])
            tr.step(clips, rec.targets, rec.scores)         # the step consumes the staged batch while the next one crosses PCIe
            st.release()
        st.collect()
        torch.cuda.synchronize()
        fed = batch * 30 / (time.perf_counter() - t0)
        return {"prepared_clips_per_s_alone": round(alone, 1), "clips_per_s_training_fed_by_the_pipeline": round(fed, 1),
                "replayed_steps": tr.replayed_steps - replayed, "launch": "lane graphs, as opental_amd.thumos14.train.run_one_epoch issues them",
                "what": "uint8 256x112x112x3 clip slices from PINNED host videos + one fixed-shape label record (1-6 targets per clip, "
                        "other counts every batch) -> async H2D on a copy stream (double buffered) -> otal_prepare_clips_map (crop 96, "
                        "flip, normalise, THWC->CTHW) straight into the captured step's clip buffer; 9.6 MB of PCIe traffic per clip"}

    leg("b1", b1)
    leg("fp32_parity", fp32)
    leg("ssl_on", ssl)
    leg("anet", anet)
    leg("inference", inference)
    leg("input_pipeline", input_pipeline)
    ops.CONV_PRECISION = saved
    return out


def inference_sharded(device, rank, world, nvid=213):
    """BASELINE configs[4] at N ranks: the video list is sharded (every world-th video), no collective in the
    data path; per-rank result dicts are gathered on rank 0 (opental_amd.thumos14.test.gather_results); proposals/s =
    126 x all windows / slowest rank's time between two barriers."""
    from opental_amd.common import ops
    from opental_amd.thumos14 import test as T
    from opental_amd.thumos14.BDNet import BDNet
    saved = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        torch.manual_seed(0)
        net = BDNet(training=False, use_edl=True)
        net.backbone._model.apply(BDNet.weight_init)
        net = net.to(device).eval()
        frames = np.random.RandomState(0).randint(600, 4001, size=nvid)
        names = [f"video_{i:04d}" for i in range(nvid)]
        mine = list(range(nvid))[rank::world]
        g = torch.Generator(device=device).manual_seed(rank)
        T.detect_batch(net, [torch.randint(0, 256, (3, 700, 96, 96), device=device, generator=g, dtype=torch.uint8)], 10.0, batch_clips=32)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        results = {}
        for i in range(0, len(mine), 16):
            part = mine[i:i + 16]
            vids = [torch.randint(0, 256, (3, int(frames[v]), 96, 96), device=device, generator=g, dtype=torch.uint8) for v in part]
            rows, counts, _, _ = T.detect_batch(net, vids, 10.0, batch_clips=32)
            for k, v in enumerate(part):
                results[names[v]] = T.get_video_detections(rows[k], counts[k], None, 5000)
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        merged = T.gather_results(results, names, rank, world, device)
        nwin = sum(len(T.get_offsets(int(f), 256, 128)) for f in frames)
        if rank != 0:
            return None
        return {"videos": nvid, "videos_in_merged_result": len(merged), "windows": nwin, "ranks": world,
                "seconds": round(float(t[0]), 3), "proposals_per_s": round(126 * nwin / float(t[0]), 1),
                "detections": sum(len(v) for v in merged.values()),
                "what": "BASELINE configs[4] at N ranks: video list sharded, per-rank result dicts gathered on rank 0; includes "
                        "generating the frames on the device and the host-side result lists"}
    finally:
        ops.CONV_PRECISION = saved


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:               # noqa: BLE001
        pass
    sys.stdout.flush()


def _mute_stdout():
    """Nothing may follow the JSON line on stdout (library banners flushed at exit, teardown chatter)."""
    _flush_c_stdio()
    fd = os.open(os.devnull, os.O_WRONLY)
    os.dup2(fd, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default 8: configs[2] batch 8/GPU; anet: 2, the yaml's)")
    ap.add_argument("--recipe", choices=["thumos14", "anet"], default="thumos14",
                    help="thumos14 = the headline workload (BASELINE configs[1]/[2]); anet = configs[3], 768-frame clips")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16",
                    help="arithmetic type of the convolution GEMMs (BASELINE.json configs[1]: bf16); f32 = parity path")
    ap.add_argument("--graph", choices=["auto", "on", "off", "lanes"], default="auto",
                    help="replay the training step from one captured HIP graph (auto: fall back to eager launches if capture fails)")
    ap.add_argument("--ssl", action="store_true",
                    help="also run the self-supervised triplet branch every step (train.py:237-242 runs it when flags[0]): a second "
                         "backbone pass on the spliced clip + 3 BoundaryMaxPooling calls + triplet losses; eager launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short legs for the other BASELINE configs (fp32 parity path, ssl on, ActivityNet, inference, input pipeline)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-hbm-kernels", action="store_true",
                    help="skip the achieved-GB/s table of the HBM-bound kernel classes (tools/bench_hbm_kernels.py)")
    args = ap.parse_args()
    anet = args.recipe == "anet"
    if args.batch is None:
        args.batch = 2 if anet else 8

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher of N ranks (one process per GPU) and exit with their
        # status; the ranks re-enter this file with RANK / LOCAL_RANK / WORLD_SIZE set (as under torch.distributed.run).
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s)")
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not os.environ.get("OTAL_FORCE_DIST"):
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N does it by itself)")
    if os.environ.get("OTAL_ONE_GPU"):      # test hook: every rank on cuda:0 (with OTAL_DIST_BACKEND=gloo: the N > 1 control
        local = 0                           # flow of this file on a one-GPU box; RCCL refuses two ranks on one device)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dist = bool(os.environ.get("OTAL_FORCE_DIST"))      # exercise the RCCL path on a single rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("OTAL_DIST_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"RCCL reports {dist.get_world_size()} ranks, expected {world}")
        # RCCL prints a version / host banner through C stdio when its communicator comes up (at the first collective).
        # With stdout redirected to a file that text sits in libc's buffer until the process exits -- i.e. it would land
        # BEHIND the JSON line the driver parses.  Bring the communicator up now, flush libc, and silence the stdout of
        # every rank but 0 (rank 0 closes its own right after the JSON line).
        warm = torch.zeros(1, device=device)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        _flush_c_stdio()
        if rank != 0:
            _mute_stdout()
    from opental_amd.common import ops as _ops
    _ops.CONV_PRECISION = 1 if args.dtype == "bf16" else 0
    if anet:
        trainer = build_anet_trainer(device, force_collectives=force_dist)
        clips, targets, scores = synth_batch(args.batch, 1000 + rank, device, frames=768, classes=150, score_rows=3)
    else:
        trainer = build_trainer(device, force_collectives=force_dist)
        clips, targets, scores = synth_batch(args.batch, 1000 + rank, device)
    # EVERY step gets other labels with other target counts, as a real epoch does: a ring of fixed-shape label records
    # (record 0 = the labels drawn above); the clips stay the resident synthetic batch
    ring = synth_label_ring(args.batch, 1000 + rank, device, frames=768 if anet else 256, classes=150 if anet else 15,
                            score_rows=3 if anet else 2)
    fed = [0]

    def step():
        rec = ring[fed[0] % len(ring)]
        fed[0] += 1
        return trainer.step(clips, rec.targets, rec.scores, *ssl_args)

    ssl_args = ()
    if args.ssl:
        ssl_clips, _, _ = synth_batch(args.batch, 2000 + rank, device, frames=768 if anet else 256, classes=150 if anet else 15,
                                      score_rows=3 if anet else 2)
        frames = 768 if anet else 256
        # anchor / positive / negative segments in frames, as the datasets' augment() emits them (thumos_dataset.py:160-237)
        ssl_targets = [torch.tensor([[0.30, 0.55], [0.32, 0.52], [0.70, 0.90]], device=device) * frames for _ in range(args.batch)]
        ssl_args = (ssl_clips, ssl_targets)
        args.graph = "off"

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    graphed = False
    launch_probe = None
    multi = world > 1 or force_dist
    backend = dist.get_backend() if multi else None

    def replica_check(where):
        """Data-parallel self-check: every rank must hold bit-identical parameters and Adam moments (identical updates of
        identical all-reduced gradients).  MIN and MAX over ranks of three checksums; a difference aborts the run -- a
        throughput number of diverged replicas would be worthless."""
        if not multi:
            return None
        a = trainer.arena
        cs = torch.stack([a.flat.double().sum(), a.flat.double().abs().sum(), a.m.double().sum(), a.v.double().sum()])
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit(f"[bench] rank {rank}: replicas DIVERGED {where}: checksum min {lo.tolist()} max {hi.tolist()}")
        return True

    def exposed_of(events):
        if not events:
            return None
        v = sum(a.elapsed_time(b) for a, b in events) / len(events)
        if multi:
            t = torch.tensor([v], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            v = float(t[0])
        return round(v, 3)

    # Launch mode.  The step is GPU-bound since the loss / gradient hand-over stopped issuing ~800 tiny kernels: eager
    # launches and a replayed HIP graph give the same throughput when the host has slack.  `auto` checks for that slack on
    # every rank (the probe steps are ordinary data-parallel steps, all ranks take part) and reports it; both forms -- eager
    # launches and the lane graphs -- are then built and TIMED (MAX over ranks) and the faster one runs.  Decisions and the
    # outcome of the capture are agreed between the ranks (MAX / MIN all-reduce), so all ranks replay or none does.
    want_graph = args.graph in ("on", "lanes")
    lanes = args.graph in ("lanes", "auto")     # auto: the lane graphs where a capture is wanted at all
    lanes_note = "a sequence of captured HIP graphs per step on two streams (main lane / weight-gradient lane)"
    # A step that issues RCCL collectives is never captured as ONE graph: ProcessGroupNCCL's watchdog thread queries the
    # events of the all-reduces it was handed, and a query of an event recorded in a capturing stream is an error
    # (hipErrorCapturedEvent) that takes the process down (measured with one forced rank).  Data-parallel runs capture the
    # step as TWO graphs with the collectives issued between them (DetectorTrainer.capture_step(split=True)).
    if args.graph == "auto" and not args.ssl:
        for _ in range(3):
            step()
        barrier()
        trainer.measure_exposed, trainer.exposed_events = multi, []
        t = time.perf_counter()
        for _ in range(6):
            step()
        t_issue = time.perf_counter() - t
        torch.cuda.synchronize()
        t_total = time.perf_counter() - t
        trainer.measure_exposed = False
        per_rank_issue = None
        if multi:
            mine = torch.tensor([t_issue / 6 * 1e3], device=device, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(allr, mine)
            per_rank_issue = [round(float(v), 3) for v in allr]
            tt = torch.tensor([t_issue, t_total], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_issue, t_total = float(tt[0]), float(tt[1])
        launch_probe = {"eager_ms": round(t_total / 6 * 1e3, 3), "host_issue_ms": round(t_issue / 6 * 1e3, 3)}
        if multi:
            launch_probe["host_issue_ms_per_rank"] = per_rank_issue
            launch_probe["allreduce_exposed_ms_eager"] = exposed_of(trainer.exposed_events)
        # the lane graphs also overlap what eager launches cannot (the weight pack beside Conv3d_1a's forward, Adam beside the
        # stem's weight gradients): always built, kept only if measured faster than the eager steps (below)
        want_graph = True
    if want_graph:
        ok = 1
        try:
            trainer.capture_step(clips, ring[0].targets, ring[0].scores, split=multi and not lanes, lanes=lanes)
        except Exception as e:                      # noqa: BLE001 -- any capture failure means eager launches
            if args.graph in ("on", "lanes"):
                raise
            ok = 0
            torch.cuda.synchronize()
            print(f"[bench] rank {rank}: HIP graph capture unavailable ({type(e).__name__}: {str(e)[:200]}); eager launches", file=sys.stderr)
        if multi:
            okt = torch.tensor([ok], device=device, dtype=torch.int32)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = int(okt)
        graphed = bool(ok)
        if graphed and launch_probe is not None:
            # the captured step must actually beat the eager one it replaces (both timed here, MAX over ranks); otherwise
            # back to eager launches
            for _ in range(2):
                step()
            barrier()
            trainer.measure_exposed, trainer.exposed_events = multi, []
            t = time.perf_counter()
            for _ in range(6):
                step()
            torch.cuda.synchronize()
            tt = torch.tensor([(time.perf_counter() - t) / 6], device=device, dtype=torch.float64)
            trainer.measure_exposed = False
            if multi:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            launch_probe["lane_graphs_ms" if lanes else "two_graph_ms"] = round(float(tt[0]) * 1e3, 3)
            if multi:
                launch_probe["allreduce_exposed_ms_graphs"] = exposed_of(trainer.exposed_events)
            if float(tt[0]) * 1e3 > launch_probe["eager_ms"]:
                graphed = False
        if not graphed:
            trainer.drop_graph()
    inputs_note = ("clips resident in HBM; labels: a different fixed-shape record every step out of a ring of %d "
                   "(target counts per clip e.g. %s / %s)" % (len(ring), ring[0].counts, ring[1].counts))
    if graphed and trainer.static_inputs() is not None:
        # the synthetic batch is resident in HBM either way; a captured step replays from ITS buffers, and a producer that
        # fills those (the clip kernel writes wherever it is told to) saves the device-to-device copy of the batch
        clips = trainer.static_inputs()[0]
        inputs_note = ("clips resident in HBM in the captured step's clip buffer (where the clip kernel writes them in a real epoch); "
                       "labels: a DIFFERENT fixed-shape record every step out of a ring of %d (target counts per clip e.g. %s / %s), "
                       "one device-to-device copy of %d bytes per step" % (len(ring), ring[0].counts, ring[1].counts, ring[0].flat.numel()))
    for _ in range(args.warmup):
        step()
    replicas_ok = replica_check("after the warm-up steps")
    trainer.measure_exposed = multi      # HIP events around the wait for the gradient all-reduces (eager or between the two graphs)
    trainer.exposed_events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    trainer.measure_exposed = False
    exposed_ms = None
    if trainer.exposed_events:
        exposed_ms = sum(a.elapsed_time(b) for a, b in trainer.exposed_events) / len(trainer.exposed_events)
    per_rank_ms = None
    if multi:
        mine = torch.tensor([dt / args.steps * 1e3], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        per_rank_ms = [round(float(v), 3) for v in allr]        # every rank's own clock around the same K steps
        t = torch.tensor([dt, exposed_ms if exposed_ms is not None else -1.0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        exposed_ms = float(t[1]) if float(t[1]) >= 0 else None
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    # A data-parallel step whose gradient all-reduce is not hidden under the backward pass is a scaling bug, not a number to
    # report quietly: more than 10 % of the step spent waiting for the collectives is flagged in the line AND shouted on
    # stderr (OTAL_BENCH_STRICT=1: the run fails with exit code 4 after printing the line).
    allreduce_check = None
    if multi and exposed_ms is not None:
        frac = exposed_ms / ms_per_step
        allreduce_check = {"exposed_frac_of_step": round(frac, 4), "ok": bool(frac <= 0.10)}
        if frac > 0.10 and rank == 0:
            print("\n" + "!" * 100 + f"\n[bench] GRADIENT ALL-REDUCE EXPOSED: {exposed_ms:.3f} ms of a {ms_per_step:.3f} ms step "
                  f"({100 * frac:.1f} % > 10 %) -- the collectives are NOT hidden under the backward pass at N = {world}\n" + "!" * 100,
                  file=sys.stderr, flush=True)
    replicas_ok = replica_check("after the timed steps") and replicas_ok
    # one more step with the all-reduce trace on: per rank, where on the step's timeline every bucket's all-reduce is issued
    # (HIP event on the issuing lane, ms after the step's first launch) and where the step waits for them -- the first N > 1
    # hardware run is then diagnosable from this line alone (which bucket is late, which lane issued it, how long the wait is)
    allreduce_timeline = None
    if multi:
        try:
            trainer.bucket_trace = []
            step()
            torch.cuda.synchronize()
            rec, trainer.bucket_trace = trainer.bucket_trace, None
            t_host, ev0 = rec[0][3], rec[0][4]
            mine = [{"what": k, "bucket": b, "MB": None if nb is None else round(nb / 1e6, 2),
                     "gpu_ms": round(ev0.elapsed_time(ev), 3), "host_ms": round((th - t_host) * 1e3, 3)} for k, b, nb, th, ev in rec]
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            allreduce_timeline = {f"rank{r}": g for r, g in enumerate(gathered)}
        except Exception as e:                      # noqa: BLE001 -- a diagnostic must never take the headline down
            trainer.bucket_trace = None
            allreduce_timeline = {"error": f"{type(e).__name__}: {str(e)[:160]}"}

    nbuckets = len(trainer.arena.buckets)
    infer_n = None
    if world > 1 and not args.no_extras and not anet:
        try:
            infer_n = inference_sharded(device, rank, world)
        except Exception as e:                      # noqa: BLE001 -- an extra must never take the headline down
            infer_n = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
            torch.cuda.synchronize()
    roofline = None
    if rank == 0 and not args.no_roofline:
        from opental_amd.common import ops
        saved_graph, trainer._graph = trainer._graph, None      # per-launch HIP events need eager launches
        # rank 0 runs these two profiling steps ALONE: no gradient all-reduce in them (the other ranks are not there to join)
        saved_coll, trainer.collectives = trainer.collectives, False
        # The per-op HIP events only bracket GPU time while the GPU is the slower side: with the host lagging (it issues
        # two extra event records per op here) an idle GPU timestamps the start marker the moment it arrives and the op's
        # "duration" then contains the host's time between the marker and the launch (10-20 us per op, 2-3 ms per step on
        # a slow host).  A spin kernel keeps the stream busy while the host issues the whole instrumented step ahead.
        spin = None
        try:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
            per_cycle_ms = e0.elapsed_time(e1) / 2_000_000
            spin = int(min(30.0 / max(per_cycle_ms, 1e-9), 2e9))          # ~30 ms of GPU time in front of each step
        except Exception:                           # noqa: BLE001 -- no spin kernel: measure as before
            spin = None
        ops.CONV_PROFILE = []
        for _ in range(2):
            if spin:
                torch.cuda._sleep(spin)
            step()
            torch.cuda.synchronize()
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        trainer._graph, trainer.collectives = saved_graph, saved_coll
        by = {}
        for mode, flops, a, b in prof:
            e = by.setdefault(mode, [0.0, 0.0, 0])
            e[0] += flops; e[1] += a.elapsed_time(b) * 1e-3; e[2] += 1
        tot_f = sum(e[0] for e in by.values()); tot_t = sum(e[1] for e in by.values()); tot_n = sum(e[2] for e in by.values())
        ach = tot_f / tot_t / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        roofline = {"bound": "mfma", "kernel": ("implicit-GEMM convolution family: conv3_direct / conv_gemm_bf16c / conv1a_tile / conv1d_tile (fwd, dgrad), "
                                          "conv3_wgrad_direct / conv1a_wgrad_direct / conv_wgrad_bf16v / conv_wgrad1d (wgrad), conv_gemm_kernel (irregular geometries)"
                               if args.dtype == "bf16" else "conv_gemm_kernel") +
                              f": implicit-GEMM convolution, {args.dtype} MFMA operands, fp32 accumulate; per-op time incl. prologue and split-K reduce",
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None,      # PMC counters cannot be read from inside the run:
                    "launches_per_step": tot_n // 2, "avg_launch_us": round(tot_t / tot_n * 1e6, 1),
                    "conv_time_ms_per_step": round(tot_t / 2 * 1e3, 2),
                    "by_mode_TFLOPs": {k: round(e[0] / e[1] / 1e12, 2) for k, e in by.items()}}
        # ... the committed rocprofv3 --pmc passes over this very command (tools/pmc_step.sh) supply it for the default
        # workload -- as long as the kernels and the launch path are still the ones that were profiled (tools/source_stamp.py)
        pmcs = sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_pmc_step_traffic.json"))
        if pmcs and args.dtype == "bf16" and not anet and args.batch == 8 and not args.ssl:
            sys.path.insert(0, os.path.join(REPO, "tools"))
            from source_stamp import source_stamp
            with open(os.path.join(REPO, "profiles", pmcs[-1])) as f:
                t = json.load(f)
            now = source_stamp()
            if t.get("source_stamp") == now:
                roofline["traffic"] = int(t["conv_traffic_MB_per_launch"] * 1e6)
                roofline["traffic_note"] = ("bytes of HBM traffic per convolution launch, FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 "
                                            f"--pmc passes over this command (profiles/{pmcs[-1][:-5]}.txt, measured on source stamp {now}: "
                                            f"the tree running now); whole step {t['step_traffic_MB'] / 1e3:.1f} GB")
            else:
                roofline["traffic_note"] = (f"null: profiles/{pmcs[-1]} was measured on source stamp {t.get('source_stamp')}, this tree is "
                                            f"{now} (kernels or launch path changed since: re-run tools/pmc_step.sh)")
    hbm = None
    if rank == 0 and world == 1 and not args.no_hbm_kernels and not anet:
        # the bandwidth-bound kernel classes in isolation, at the shapes of this step: algorithmic bytes / launch time
        # (HIP events around a graph replay of back-to-back launches) against the 8 TB/s HBM peak
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from bench_hbm_kernels import measure as measure_hbm_kernels
        from opental_amd.common import ops as _o
        saved_prec = _o.CONV_PRECISION
        hbm = {k: {f: v[f] for f in ("us", "algorithmic_MB", "GB/s", "frac_of_8TBps", "resident_in_infinity_cache")}
               for k, v in measure_hbm_kernels(args.batch).items()}
        _o.CONV_PRECISION = saved_prec
        # rocprofv3-counter bytes per launch of the same kernels (tools/pmc_hbm_kernels.sh -> profiles/*_pmc_hbm_kernels.json),
        # reported only while the file's source stamp is this tree's -- like roofline.traffic
        hk = sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_pmc_hbm_kernels.json"))
        if hk:
            from source_stamp import source_stamp as _stamp
            with open(os.path.join(REPO, "profiles", hk[-1])) as f:
                t = json.load(f)
            fresh = t.get("source_stamp") == _stamp()
            for k, v in hbm.items():
                c = t.get("kernels", {}).get(k)
                v["counter_MB"] = c["counter_MB"] if (fresh and c) else None
                v["counter_over_algorithmic"] = c["ratio"] if (fresh and c) else None
            hbm["_counter_source"] = (f"profiles/{hk[-1]} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)" if fresh else
                                      f"null: profiles/{hk[-1]} was measured on source stamp {t.get('source_stamp')}, not this tree")
        torch.cuda.empty_cache()
    extra = None
    if rank == 0 and world == 1 and not args.no_extras and not anet and not args.ssl and args.dtype == "bf16":
        del trainer
        torch.cuda.empty_cache()
        extra = extras(device, args.batch)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not anet:
        cpu = cpu_baseline(seconds_budget=12.0)
        if not args.no_extras:
            cpu["batch8"] = cpu_baseline(seconds_budget=10.0, batch=8)
            cpu["all_threads"] = cpu_baseline_all_threads()
    if world > 1:
        dist.barrier()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps({
            "metric": "clips/sec training step, 768-frame ActivityNet1.3 clips" if anet else
                      "clips/sec training step, 256-frame THUMOS14 clips", "value": round(value, 3),
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "OpenTAL ActivityNet1.3 training step (configs/anet_opental.yaml, EDL+IBM loss, per-sample "
                                   "normalisation, ssl branch " + ("ON" if args.ssl else "off") + "), 768x3x96x96 clips, random-init weights" if anet else
                                   "OpenTAL THUMOS14 split_0 training step (configs/thumos14_opental_final.yaml, "
                                   "EDL+IBM loss, ssl branch " + ("ON" if args.ssl else "off") + "), 256x3x96x96 clips, random-init weights",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "backend": backend, "ranks": dist.get_world_size() if multi else 1,
                       "rccl_ranks": dist.get_world_size() if (multi and backend == "nccl") else (1 if not multi else 0),
                       "replicas_bit_identical": replicas_ok,
                       "arena_buckets": nbuckets,
                       "grad_allreduce": "RCCL sum over xGMI of the flat fp32 gradient arena in %d contiguous buckets, issued from inside "
                                         "the backward pass (the backbone hands its finished layers over while it runs); "
                                         "exposed = compute-stream wait for the collectives after backward" % nbuckets,
                       "allreduce_exposed_ms": None if exposed_ms is None else round(exposed_ms, 3),
                       "allreduce_timeline": allreduce_timeline,
                       "allreduce_check": allreduce_check,
                       "ms_per_step_per_rank": per_rank_ms,
                       "rank_spread_ms": None if not per_rank_ms else round(max(per_rank_ms) - min(per_rank_ms), 3),
                       "ssl_branch": bool(args.ssl), "inputs": inputs_note, "launch_probe": launch_probe,
                       "launch": (lanes_note if lanes else
                                  "two captured HIP graphs per step, the gradient all-reduces issued between them" if multi else
                                  "one captured HIP graph per step") if graphed else "eager launches"},
            "roofline": roofline, "hbm_kernels": hbm,
            "other_configs": extra if extra is not None else ({"inference": infer_n} if infer_n is not None else None),
            "cpu_baseline": cpu}), flush=True)
        _mute_stdout()
    if world > 1 or force_dist:
        dist.barrier()              # every rank leaves together (rank 0 was busy with the roofline steps)
        dist.destroy_process_group()
    if allreduce_check is not None and not allreduce_check["ok"] and os.environ.get("OTAL_BENCH_STRICT"):
        sys.exit(4)


if __name__ == "__main__":
    main()
