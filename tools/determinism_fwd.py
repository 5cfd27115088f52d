"""Find the first forward op whose output is not bit-identical between repeated runs (same weights, same clip).
usage: python tools/determinism_fwd.py [iters] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opental_amd.common import ops

LOG = []


def wrap(name):
    fn = getattr(ops, name)

    def inner(*a, **k):
        out = fn(*a, **k)
        o = out[0] if isinstance(out, tuple) else out
        shp = tuple(a[0].shape) if torch.is_tensor(a[0]) else None
        LOG.append((name, shp, tuple(o.shape), o.detach().clone()))
        return out
    setattr(ops, name, inner)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ops.CONV_PRECISION = int(os.environ.get("OTAL_PREC", "1"))
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev)
    clips, targets, scores = bench.synth_batch(batch, 1000, dev)
    for n in ("conv_forward", "maxpool3d_forward", "gn_relu_forward"):
        wrap(n)
    ibm0 = tr._ibm_state().detach().clone() if tr._ibm_state() is not None else None
    first = None
    for it in range(iters):
        LOG.clear()
        if ibm0 is not None:
            tr._ibm_state().copy_(ibm0)
        ops.activate_prologues(tr._prologues)
        try:
            with torch.no_grad():
                cost, losses = tr.compute_cost(clips, targets, scores)
        finally:
            ops.deactivate_prologues()
        torch.cuda.synchronize()
        cur = list(LOG)
        if first is None:
            first = cur
            continue
        for i, (a, b) in enumerate(zip(first, cur)):
            if not torch.equal(a[3], b[3]):
                d = (a[3].float() - b[3].float()).abs()
                print(f"iteration {it}: first differing op #{i} {a[0]} in {a[1]} -> out {a[2]}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}")
                nz = (d > 0).nonzero()
                print("   first / last differing index:", nz[0].tolist(), nz[-1].tolist())
                break
    print("done")


if __name__ == "__main__":
    main()
