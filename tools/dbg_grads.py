import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import arch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_model_gpu as T
from opental_amd.thumos14.train import forward_one_epoch, total_cost
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fx = np.load(f"tests/golden/thumos_b{b}.npz")
net = T.build(fx)
x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), b)).cuda()
targets = [torch.from_numpy(fx[f"target_{i}"]).cuda() for i in range(b)]
scores = torch.from_numpy(fx["scores"]).cuda()
for rep in range(2):
    crit = T._criterion("edl", 0)
    net.zero_grad(set_to_none=True)
    cost = total_cost(forward_one_epoch(net, crit, x, targets, scores, training=True, ssl=False), T.W)
    cost.backward()
    grads = dict((k, p.grad) for k, p in net.named_parameters() if p.grad is not None)
    names = [str(n) for n in fx["grad_names"]]
    d32, n64 = fx["grad32dist_correct"], fx["grad64norm_correct"]
    got = np.array([float(grads[n].double().norm()) for n in names])
    rel = np.abs(got - n64) / (n64 + 1e-30)
    ratio = rel / (5 * d32 + 1e-4)
    order = np.argsort(-ratio)[:12]
    print("rep", rep, "cost", float(cost.detach()), "golden", float(fx["cost_edl0"]))
    for i in order:
        print(f"  {names[i]:70s} rel {rel[i]:.2e} d32 {d32[i]:.2e} ratio {ratio[i]:.2f}")
