"""Where the HOST spends a training step (cProfile over eager steps): the launch-issue cost that decides eager vs graph."""
import cProfile, os, pstats, sys
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
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(clips, targets, scores)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
