"""Entry kinds of the captured lane-graph step (ops.LanePlan) and whether the early optimizer launch is in it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opental_amd.common import ops
dev = torch.device("cuda", 0)
ops.CONV_PRECISION = 1
tr = bench.build_trainer(dev, seed=21)
clips, targets, scores = bench.synth_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1000, dev)
tr.step(clips, targets, scores)
tr.capture_step(clips, targets, scores, warmup=0, lanes=True)
plan = tr._graph[1]
print("entries:", " ".join(e[0] for e in plan.entries))
