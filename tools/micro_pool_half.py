"""bf16-tensor strided pools (the benchmarked path), HIP-event timed: scanning kernel vs ordered-key kernel (nonneg)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops
ops.CONV_PRECISION, ops.HALF_STORAGE = 1, True
LAYERS = {"2a": ((8, 64, 128, 48, 48), (1, 3, 3), (1, 2, 2)), "3a": ((8, 192, 128, 24, 24), (1, 3, 3), (1, 2, 2)),
          "4a": ((8, 480, 128, 12, 12), (3, 3, 3), (2, 2, 2))}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, (shape, k, s) in LAYERS.items():
    x = torch.relu(torch.randn(*shape, device="cuda")).to(torch.bfloat16)
    y, arg, bits = ops.maxpool3d_forward(x, k, s, signbits=True, half_out=True)
    mb = (x.numel() * 2 + y.numel() * 3 + (bits.numel() if bits is not None else 0)) / 1e6
    t0 = timeit(lambda: ops.maxpool3d_forward(x, k, s, signbits=True, half_out=True))
    t1 = timeit(lambda: ops.maxpool3d_forward(x, k, s, signbits=True, half_out=True, nonneg=True))
    dy = torch.randn_like(y.float()).to(torch.bfloat16)
    sc = torch.rand(shape[1], device="cuda") + 0.5
    tb = timeit(lambda: ops.maxpool3d_backward(dy, arg, x.shape, k, s, out_scale=sc, out_signbits=bits))
    print(f"{name}  fwd scan {t0:7.1f} us  keys {t1:7.1f} us  ({mb:.0f} MB: {mb / t0 / 1e0:.2f} -> {mb / t1:.2f} TB/s x1e-6)  bwd {tb:7.1f} us")
