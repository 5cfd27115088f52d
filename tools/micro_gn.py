"""GroupNorm + ReLU forward / backward on the model's level-packed map (8,512,126), six levels vs one level (the whole
workgroup on it), HIP-event timed: how much of the kernel is the per-level wave imbalance (level 0 = half of the elements on
ONE wave).
The printed figures are host-bound (a launch per Python call); run it under tools/kernel_time.sh for kernel durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops

def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

x = torch.randn(8, 512, 126, device="cuda"); g = torch.rand(512, device="cuda") + 0.5; b = torch.randn(512, device="cuda")
dy = torch.randn_like(x)
for name, lev in (("six levels", (0, 64, 96, 112, 120, 124, 126)), ("one level", None)):
    y, st = ops.gn_relu_forward(x, g, b, 32, 1e-5, True, lev)
    tf = timeit(lambda: ops.gn_relu_forward(x, g, b, 32, 1e-5, True, lev))
    tb = timeit(lambda: ops.gn_relu_backward(dy, x, g, b, st, 32, True, lev))
    print(f"{name:12s} fwd {tf:6.2f} us  bwd {tb:6.2f} us   (back-to-back launches incl. the Python call)")
