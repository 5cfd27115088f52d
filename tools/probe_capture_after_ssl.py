import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from opental_amd.common import ops
dev = torch.device('cuda', 0)
ops.CONV_PRECISION = 1
mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tr = bench.build_trainer(dev)
clips, targets, scores = bench.synth_batch(B, 1000, dev)
ring = bench.synth_label_ring(B, 1000, dev, n=3)
ssl_clips, _, _ = bench.synth_batch(B, 2000, dev)
ssl_t = [torch.tensor([[0.30, 0.55], [0.32, 0.52], [0.70, 0.90]], device=dev) * 256 for _ in range(B)]
tr.launch = 'lanes'
print("plain eager", flush=True)
tr.step(clips, ring[0].targets, ring[0].scores)
torch.cuda.synchronize()
if 'ssl' in mode:
    print("ssl eager", flush=True)
    tr.step(clips, ring[1].targets, ring[1].scores, ssl_clips, ssl_t)
    torch.cuda.synchronize()
print("capture", flush=True)
tr.step(clips, ring[2].targets, ring[2].scores)
torch.cuda.synchronize()
print("replayed", tr.replayed_steps, flush=True)
for i in range(3):
    tr.step(clips, ring[i].targets, ring[i].scores)
    if 'ssl' in mode:
        tr.step(clips, ring[1].targets, ring[1].scores, ssl_clips, ssl_t)
torch.cuda.synchronize()
print("ok", mode, B, tr.replayed_steps, flush=True)
