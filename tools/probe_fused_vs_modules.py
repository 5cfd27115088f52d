"""Trainer steps with the pyramid as two hand-scheduled nodes vs module by module: parameters after k steps, bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opental_amd.common import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "eager"
for prec in (0, 1):
    ops.CONV_PRECISION = prec
    res = {}
    for fused in (False, True, True):
        ops.FUSED_PYRAMID = fused
        tr = bench.build_trainer(dev, seed=21)
        tr.lr = 1e-5
        clips, targets, scores = bench.synth_batch(B, 1000, dev)
        costs = []
        if mode == "lanes":
            costs.append(float(tr.step(clips, targets, scores)[0]))
            tr.capture_step(clips, targets, scores, warmup=0, lanes=True)
        for _ in range(steps):
            costs.append(float(tr.step(clips, targets, scores)[0]))
        torch.cuda.synchronize()
        key = (fused, len([k for k in res if k[0] == fused]))
        res[key] = (costs, tr.arena.flat.detach().clone(), tr.arena.grad.detach().clone())
        del tr
        torch.cuda.empty_cache()
    a, b, c = res[(False, 0)], res[(True, 0)], res[(True, 1)]
    for name, u, v in (("modules vs fused", a, b), ("fused vs fused again", b, c)):
        dp = (u[1] - v[1]).abs().max().item()
        dg = (u[2] - v[2]).abs().max().item()
        print(f"prec {prec} {mode} {name}: max|dparam| {dp:.3e} max|dgrad| {dg:.3e} last costs {u[0][-1]:.6f} {v[0][-1]:.6f}", flush=True)
