"""Micro-benchmark of the level-packed 1-D tower layer (512 -> 512, k = 3 and k = 1, (8,512,126)) in the three modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops
from tools.micro_conv import timeit
LEVELS = (0, 64, 96, 112, 120, 124, 126)


def main():
    ops.CONV_PRECISION = 1
    B = 8
    for cin, cout, k in ((512, 512, 3), (512, 512, 1), (2048, 512, 1), (512, 1024, 1), (512, 15, 3)):
        x = torch.randn(B, cin, 126, device="cuda")
        w = torch.randn(cout, cin, k, 1, 1, device="cuda") * 0.02
        dy = torch.randn(B, cout, 126, device="cuda")
        cache = ops.PrologueCache((w.data_ptr(), w.data_ptr() + 4 * w.numel()))
        ops.activate_prologues(cache)
        f = lambda: ops.conv_forward(x, w, (k, 1, 1), (1, 1, 1), levels=LEVELS)
        d = lambda: ops.conv_dgrad(dy, w, x.shape, (k, 1, 1), (1, 1, 1), levels=LEVELS)
        g = lambda: ops.conv_wgrad(x, dy, w.shape, (k, 1, 1), (1, 1, 1), levels=LEVELS)
        f(); d(); g()
        ops.activate_prologues(cache)
        tf, td, tg = timeit(f, 50), timeit(d, 50), timeit(g, 50)
        ops.deactivate_prologues()
        fl = 2.0 * B * cout * cin * k * 126
        print(f"{cin}->{cout} k{k}: fwd {tf*1e6:6.1f} us ({fl/tf/1e12:5.1f} TF)  dgrad {td*1e6:6.1f} us  wgrad {tg*1e6:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
