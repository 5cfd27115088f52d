import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from opental_amd.common import ops
ops.CONV_PRECISION = 1
dev = torch.device("cuda", 0)
clips, targets, scores = bench.synth_batch(1, 77, dev)
def mk():
    tr = bench.build_trainer(dev, seed=11); tr.lr = 1e-4; return tr
e = mk(); snaps_e = []
for i in range(3):
    c, _ = e.step(clips, targets, scores); torch.cuda.synchronize(); snaps_e.append((e.arena.flat.clone(), float(c), e.arena.grad.clone()))
g = mk()
g.capture_step(clips, targets, scores, warmup=1); torch.cuda.synchronize()
print("after warmup+capture: dflat", float((g.arena.flat - snaps_e[0][0]).abs().max()), "dgrad", float((g.arena.grad - snaps_e[0][2]).abs().max()))
for i in (1, 2):
    c, _ = g.step(clips, targets, scores); torch.cuda.synchronize()
    print(f"after replay {i}: dflat", float((g.arena.flat - snaps_e[i][0]).abs().max()), "dgrad", float((g.arena.grad - snaps_e[i][2]).abs().max()),
          "gradmax", float(snaps_e[i][2].abs().max()), "cost", float(c), snaps_e[i][1])
    dg = (g.arena.grad - snaps_e[i][2]).abs()
    j = int(dg.argmax())
    # which parameter
    off = 0
    for name, p in g.net.named_parameters():
        n = p.numel()
        if off <= j < off + n:
            print("   worst grad diff in", name, "idx", j - off, "eager", float(snaps_e[i][2][j]), "graph", float(g.arena.grad[j])); break
        off += n
