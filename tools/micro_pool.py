"""Single-layer micro-benchmark of the max-pool kernels (fwd / bwd with fused mask), HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops

LAYERS = {
    "2a": ((8, 64, 128, 48, 48), (1, 3, 3), (1, 2, 2)),
    "3a": ((8, 192, 128, 24, 24), (1, 3, 3), (1, 2, 2)),
    "4a": ((8, 480, 128, 12, 12), (3, 3, 3), (2, 2, 2)),
    "3b": ((8, 192, 128, 12, 12), (3, 3, 3), (1, 1, 1)),
    "3c": ((8, 256, 128, 12, 12), (3, 3, 3), (1, 1, 1)),
    "4b": ((8, 480, 64, 6, 6), (3, 3, 3), (1, 1, 1)),
}


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(LAYERS)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for name in names:
    shape, k, s = LAYERS[name]
    x = torch.randn(*shape, device="cuda")
    y, arg = ops.maxpool3d_forward(x, k, s)
    dy = torch.randn_like(y)
    sc = torch.rand(shape[1], device="cuda") + 0.5
    dx = torch.empty_like(x)
    tf = timeit(lambda: ops.maxpool3d_forward(x, k, s), iters)
    tb = timeit(lambda: ops.maxpool3d_backward(dy, arg, x.shape, k, s, out=dx, out_mask=x, out_scale=sc), iters)
    bf = (x.numel() * 4 + y.numel() * 5) / tf / 1e12
    bb = (y.numel() * 5 + x.numel() * 8) / tb / 1e12
    extra = ""
    y2, a2, bits = ops.maxpool3d_forward(x, k, s, signbits=True)
    if bits is not None:
        tfb = timeit(lambda: ops.maxpool3d_forward(x, k, s, signbits=True), iters)
        tbb = timeit(lambda: ops.maxpool3d_backward(dy, arg, x.shape, k, s, out=dx, out_scale=sc, out_signbits=bits), iters)
        extra = f" | with sign bits: fwd {tfb*1e6:7.1f} us, bwd {tbb*1e6:7.1f} us"
    print(f"{name:4s} fwd {tf*1e6:7.1f} us {bf:5.2f} TB/s | bwd(+mask) {tb*1e6:7.1f} us {bb:5.2f} TB/s{extra}", flush=True)
