"""Shapes of the device copies (aten::copy_ / cat / contiguous / clone) of one training step, by count and time."""
import os, sys, collections
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(clips, targets, scores)
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::copy_", "aten::cat", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::fill_", "aten::zero_", "aten::_foreach_copy_") and e.device_time_total > 0:
        k = (e.name, str(e.input_shapes)[:110])
        acc[k][0] += 1; acc[k][1] += e.device_time_total
for (name, shp), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t/1e3:7.3f} ms n={n:3d} {name:12s} {shp}")
