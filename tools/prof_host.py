"""Where the HOST spends a training step (cProfile over eager steps): the launch-issue cost that decides eager vs graph.
usage: python tools/prof_host.py [dist]   -- `dist`: one forced RCCL rank (the data-parallel code path: bucket all-reduces)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import torch.distributed as dist
from opental_amd.common import ops
ops.CONV_PRECISION = 1
dev = torch.device("cuda", 0)
force = len(sys.argv) > 1 and sys.argv[1] == "dist"
if force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend="nccl", device_id=dev)
tr = bench.build_trainer(dev, force_collectives=force)
clips, targets, scores = bench.synth_batch(8, 1000, dev)
for _ in range(3):
    tr.step(clips, targets, scores)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(clips, targets, scores)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host issue {t_issue / 5 * 1e3:.2f} ms/step, step {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms (no profiler)")
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):      # backward on THIS thread, so that the profiler sees it
    tr.step(clips, targets, scores)
    pr.enable()
    for _ in range(5):
        tr.step(clips, targets, scores)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(70)
