"""GPU: the data-parallel training step of the REAL detector on TWO ranks.

The build and test boxes have one GPU, and RCCL refuses two ranks on one device -- so the two rank processes share cuda:0
and the collectives run on gloo (device tensors).  Everything else is the product path: the flat arena and its buckets,
the fixed issue order, the backbone's early hand-over, the deferred reductions / sums, the IBM state average, Adam with
grad_scale 1 / world, and the two-graph step with the collectives issued BETWEEN the graphs (capture_step(split=True)).
Checked against a single-process restatement of what data parallelism must compute: per-rank gradients of the
per-rank losses, summed, scaled by 1 / world; the IBM EMA averaged over ranks."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

RANK_CODE = textwrap.dedent('''
    import os, sys, torch
    import torch.distributed as dist
    sys.path.insert(0, os.getcwd())
    import bench
    from opental_amd.common import ops
    ops.CONV_PRECISION = 1
    rank, world, mode, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1], sys.argv[2]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    anet = mode.startswith("anet")
    tr = bench.build_anet_trainer(dev, seed=5) if anet else bench.build_trainer(dev, seed=5)
    assert tr.collectives and tr.world == world
    tr.lr = 1e-4
    kw = dict(frames=768, classes=150, score_rows=3) if anet else {}
    clips, targets, scores = bench.synth_batch(2, 1000 + rank, dev, **kw)
    ring = bench.synth_label_ring(2, 1000 + rank, dev, n=3, **kw)      # mode "driver*": other labels (other counts) every step
    steps = 3
    ssl_args = ()
    if mode.startswith("mixed") and rank == 0:
        # rank 0's sampler flag is set (train.py:237 `if flags[0]`), rank 1's is not: rank 0 runs the triplet branch (a
        # second backbone pass, an eager step with early=False) while rank 1 runs the plain step -- or replays its graphs
        ssl_clips, _, _ = bench.synth_batch(2, 2000, dev)
        ssl_targets = [torch.tensor([[0.30, 0.55], [0.32, 0.52], [0.70, 0.90]], device=dev) * 256 for _ in range(2)]
        ssl_args = (ssl_clips, ssl_targets)
    if mode == "mixed_split":
        tr.capture_step(clips, targets, scores, warmup=1, split=True)       # both ranks: same collectives in the warm-up
        assert tr._graph[0] == "split"
        done = 1
    elif mode == "split":
        tr.capture_step(clips, targets, scores, warmup=1, split=True)       # one real (eager) step, then the capture
        assert tr._graph[0] == "split"
        done = 1
    elif mode in ("lanes", "anet_lanes"):
        tr.capture_step(clips, targets, scores, warmup=1, lanes=True)       # the lane graphs: collectives issued between them
        assert tr._graph[0] == "lanes"
        done = 1
    else:
        done = 0
    costs = []
    if mode == "driver":
        # what run_one_epoch does: fixed-shape label records, trainer.launch = 'lanes' -- step 1 eager, step 2 captured and
        # replayed, step 3 replayed, each with other labels
        tr.launch = "lanes"
        for k in range(steps):
            costs.append(float(tr.step(clips, ring[k].targets, ring[k].scores)[0]))
        assert tr._graph is not None and tr._graph[0] == "lanes" and tr.replayed_steps == steps - 1, tr.replayed_steps
    else:
        for _ in range(steps - done):
            costs.append(float(tr.step(clips, targets, scores, *ssl_args)[0]))
    torch.cuda.synchronize()
    assert tr.step_count == steps
    wa = getattr(tr.criterion.cls_loss, "weight_accum", None)
    torch.save({"flat": tr.arena.flat.cpu(), "m": tr.arena.m.cpu(), "ibm": wa.cpu() if wa is not None else torch.zeros(1),
                "costs": costs, "order": list(tr._flush_order), "buckets": list(tr.arena.buckets)}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()
''')


def _run(mode, tmp_path, port):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"res_{mode}")
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, "-c", RANK_CODE, mode, out], cwd=root, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    logs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError(f"{mode}: the ranks did not finish")
        logs.append((p.returncode, o[-1500:], e[-3000:]))
    assert all(rc == 0 for rc, _, _ in logs), logs
    import torch
    return [torch.load(out + f".{r}") for r in range(2)]


def _reference(steps=3, ssl_rank0_from=None, anet=False, label_ring=False):
    """What two data-parallel ranks must compute, in ONE process without collectives.  `ssl_rank0_from`: first step
    (0-based) from which rank 0's cost includes the triplet branch.  `label_ring`: step k uses record k of each rank's
    label ring (the drivers' mode) instead of the rank's fixed labels."""
    import torch
    import bench
    from opental_amd.common import ops
    dev = torch.device("cuda", 0)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 1
    try:
        tr = bench.build_anet_trainer(dev, seed=5) if anet else bench.build_trainer(dev, seed=5)
        tr.lr = 1e-4
        tr.world = 2                                    # Adam's grad_scale = 1 / world
        kw = dict(frames=768, classes=150, score_rows=3) if anet else {}
        batches = [bench.synth_batch(2, 1000 + r, dev, **kw) for r in range(2)]
        rings = [bench.synth_label_ring(2, 1000 + r, dev, n=3, **kw) for r in range(2)] if label_ring else None
        ssl_clips, _, _ = bench.synth_batch(2, 2000, dev)
        ssl_targets = [torch.tensor([[0.30, 0.55], [0.32, 0.52], [0.70, 0.90]], device=dev) * 256 for _ in range(2)]
        ibm = getattr(tr.criterion.cls_loss, "weight_accum", None)
        if ibm is None or tr._ibm_state() is None:      # (the ActivityNet recipe's closed-form IBM weight has no state)
            ibm = torch.zeros(1, device=dev)
        for k in range(steps):
            state = ibm.detach().clone()
            total, states = torch.zeros_like(tr.arena.grad), []
            for r, b in enumerate(batches):
                if rings is not None:
                    b = (b[0], rings[r][k].targets, rings[r][k].scores)
                ibm.copy_(state)
                ops.activate_prologues(tr._prologues)
                with_ssl = r == 0 and ssl_rank0_from is not None and k >= ssl_rank0_from
                try:
                    cost, _ = tr.compute_cost(*b, *((ssl_clips, ssl_targets) if with_ssl else ()))
                    tr.begin_backward(early=not with_ssl)
                    cost.backward()
                    tr.end_backward()
                finally:
                    ops.deactivate_prologues()
                total += tr.arena.grad
                states.append(ibm.detach().clone())
            tr.arena.grad.copy_(total)
            ibm.copy_((states[0] + states[1]) / 2)
            tr.step_count += 1
            tr.optimizer_update()
        torch.cuda.synchronize()
        return tr.arena.flat.cpu(), tr.arena.m.cpu(), ibm.cpu()
    finally:
        ops.CONV_PRECISION = old


@pytest.mark.parametrize("mode", ["eager", "split", "lanes", "driver", "anet", "anet_lanes"])
def test_two_ranks_on_one_gpu_match_the_single_process_restatement(mode, tmp_path):
    """eager / split: round 2.  lanes: the LANE-graph data-parallel step -- the form bench.py and the
    drivers run -- with the bucket all-reduces issued between the graphs; driver: trainer.launch = 'lanes' fed other
    fixed-shape label records every step (eager step, capture, replays) as run_one_epoch feeds it; anet / anet_lanes: the
    ActivityNet recipe's criterion (per-sample normalisation, no IBM state) and its two optimizer groups on real kernels."""
    import torch
    res = _run(mode, tmp_path, 29551 + ["eager", "split", "lanes", "driver", "anet", "anet_lanes"].index(mode) * 7)
    # both ranks hold the same parameters, moments and IBM state, and issued their collectives in the same order
    assert torch.equal(res[0]["flat"], res[1]["flat"]) and torch.equal(res[0]["m"], res[1]["m"])
    assert torch.equal(res[0]["ibm"], res[1]["ibm"]) and res[0]["order"] == res[1]["order"] and res[0]["buckets"] == res[1]["buckets"]
    flat, m, ibm = _reference(anet=mode.startswith("anet"), label_ring=mode == "driver")
    scale = float(flat.abs().max())
    # the sum over ranks is one fp32 add per element either way; what may differ is the order of the two addends (none:
    # a + b) and the reference's extra accumulate through a zero tensor (0 + a + b): exact
    assert float((res[0]["flat"] - flat).abs().max()) <= 1e-6 * scale, float((res[0]["flat"] - flat).abs().max())
    assert float((res[0]["m"] - m).abs().max()) <= 1e-6 * float(m.abs().max())
    assert float((res[0]["ibm"] - ibm).abs().max()) <= 1e-6 * float(ibm.abs().max() + 1e-12)


@pytest.mark.parametrize("mode", ["mixed", "mixed_split"])
def test_mixed_ssl_and_plain_ranks_stay_in_step(mode, tmp_path):
    """run_one_epoch decides `use_ssl` per rank, so one rank may run the triplet branch (eager step,
    gradients final only when autograd returns them) while its peer runs the plain step or replays the two-graph step.
    The fixed collective order (IBM state, buckets in _flush_order, used-parameter flags) must keep them aligned, and
    the result must equal the single-process restatement: rank 0's gradient of (cost + ssl * triplet) plus rank 1's."""
    import torch
    res = _run(mode, tmp_path, 29653 if mode == "mixed" else 29654)
    assert torch.equal(res[0]["flat"], res[1]["flat"]) and torch.equal(res[0]["m"], res[1]["m"])
    assert torch.equal(res[0]["ibm"], res[1]["ibm"])
    # mixed_split: the warm-up step of the capture is a plain step on both ranks, the ssl steps follow
    flat, m, ibm = _reference(ssl_rank0_from=1 if mode == "mixed_split" else 0)
    scale = float(flat.abs().max())
    assert float((res[0]["flat"] - flat).abs().max()) <= 2e-6 * scale, float((res[0]["flat"] - flat).abs().max())
    assert float((res[0]["m"] - m).abs().max()) <= 2e-6 * float(m.abs().max())
    assert float((res[0]["ibm"] - ibm).abs().max()) <= 1e-6 * float(ibm.abs().max() + 1e-12)
