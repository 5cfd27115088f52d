"""Run the same forward + backward N times from identical weights and report every parameter whose gradient is not
bit-identical to the first run (the step is meant to be deterministic: fixed-order reductions, no atomics).
usage: python tools/determinism_check.py [iters] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opental_amd.common import ops


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ops.CONV_PRECISION = int(os.environ.get("OTAL_PREC", "1"))
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev)
    clips, targets, scores = bench.synth_batch(batch, 1000, dev)
    names = {id(p): n for n, p in tr.net.named_parameters()}
    a = tr.arena
    ibm0 = tr._ibm_state().detach().clone() if tr._ibm_state() is not None else None
    first = None
    seen = {}
    for it in range(iters):
        if ibm0 is not None:
            tr._ibm_state().copy_(ibm0)
        a.grad.zero_()
        ops.activate_prologues(tr._prologues)
        try:
            cost, losses = tr.compute_cost(clips, targets, scores)
            tr.begin_backward(early=True)
            cost.backward()
            tr.end_backward()
        finally:
            ops.deactivate_prologues(); ops.GRAD_SLOTS = None; ops.GRAD_READY = None
        torch.cuda.synchronize()
        g = a.grad.detach().clone()
        c = float(cost.detach())
        if first is None:
            first = (g, c)
            continue
        if c != first[1]:
            seen.setdefault("<cost>", []).append(it)
        if not torch.equal(g, first[0]):
            for p, off in zip(a.params, a.offsets):
                k = p.numel()
                if not torch.equal(g[off:off + k], first[0][off:off + k]):
                    d = float((g[off:off + k] - first[0][off:off + k]).abs().max())
                    seen.setdefault(names.get(id(p), "?"), []).append((it, d))
    if not seen:
        print(f"{iters} runs: bit-identical cost and gradients")
        return
    order = [names.get(id(p), "?") for p in a.params]           # registration (= forward) order
    per_it = {}
    for k, v in seen.items():
        for e in v:
            it = e if isinstance(e, int) else e[0]
            per_it.setdefault(it, []).append(k)
    for it in sorted(per_it):
        ks = per_it[it]
        fw = [n for n in order if n in ks]
        print(f"iteration {it}: cost {'DIFFERS' if '<cost>' in ks else 'same'}, {len(fw)} of {len(order)} gradients differ;"
              f" first in forward order: {fw[:3]}; last: {fw[-3:]}")


if __name__ == "__main__":
    main()
