"""How much slack the host has at a given place of the eager training step: a busy-wait of X us is inserted there and the
step time is measured.  Slope 0 = the GPU has queued work to chew on (host ahead); slope 1 = the GPU is waiting for the host.
usage: python tools/host_slack.py [where=loss|start]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from opental_amd.common import ops
ops.CONV_PRECISION = 1
dev = torch.device("cuda", 0)
where = sys.argv[1] if len(sys.argv) > 1 else "loss"
tr = bench.build_trainer(dev)
batch = bench.synth_batch(8, 1000, dev)
delay = [0.0]


def spin():
    t = time.perf_counter() + delay[0]
    while time.perf_counter() < t:
        pass


if where == "loss":
    inner = tr.criterion.forward
    tr.criterion.forward = lambda *a, **k: (spin(), inner(*a, **k))[1]
else:
    inner = tr.compute_cost
    tr.compute_cost = lambda *a, **k: (spin(), inner(*a, **k))[1]
for us in (0, 250, 500, 1000, 2000, 0):
    delay[0] = us * 1e-6
    for _ in range(3):
        tr.step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(*batch)
    torch.cuda.synchronize()
    print(f"{where}: +{us:5d} us on the host -> {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
