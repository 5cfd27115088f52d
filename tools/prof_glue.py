"""Attribute the non-HIP-library kernels of one training step to ATen ops (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from opental_amd.common import ops
ops.CONV_PRECISION = 1
dev = torch.device("cuda", 0)
tr = bench.build_trainer(dev)
clips, targets, scores = bench.synth_batch(8, 1000, dev)
for _ in range(3):
    tr.step(clips, targets, scores)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(clips, targets, scores)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print("aten ops with device time: total %.2f ms" % (tot / 1e3))
for e in rows[:25]:
    print(f"{e.device_time_total/1e3:7.3f} ms  n={e.count:4d}  {e.key}")
