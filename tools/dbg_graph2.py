import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from opental_amd.common import ops
ops.CONV_PRECISION = int(os.environ.get("PREC", "1"))
dev = torch.device("cuda", 0)
clips, targets, scores = bench.synth_batch(1, 77, dev)
ibm = os.environ.get("IBM", "1") == "1"
def mk():
    tr = bench.build_trainer(dev, seed=11); tr.lr = 0.0; tr.wd = 0.0
    tr.criterion.cls_loss.with_ibm = ibm
    return tr
e = mk(); ge = []
for i in range(3):
    c, _ = e.step(clips, targets, scores); torch.cuda.synchronize(); ge.append((e.arena.grad.clone(), float(c)))
print("eager costs", [x[1] for x in ge], "eager grad step1-vs-step2 diff", float((ge[0][0] - ge[1][0]).abs().max()))
g = mk()
g.capture_step(clips, targets, scores, warmup=1); torch.cuda.synchronize()
names = [(n, p.numel()) for n, p in g.net.named_parameters()]
for i in (1, 2):
    c, _ = g.step(clips, targets, scores); torch.cuda.synchronize()
    dg = (g.arena.grad - ge[i][0]).abs()
    print(f"replay {i}: cost", float(c), "eager", ge[i][1], "max dgrad", float(dg.max()), "gradmax", float(ge[i][0].abs().max()))
    off = 0; bad = []
    for n, k in names:
        d = float(dg[off:off + k].max()); s = float(ge[i][0][off:off + k].abs().max())
        if d > 1e-4 * max(s, 1e-6): bad.append((n, d, s))
        off += k
    print("   tensors differing:", len(bad), "of", len(names)); 
    for b in bad[:12]: print("     ", b)
