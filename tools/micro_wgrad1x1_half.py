"""The 1x1x1 weight gradient on bf16-stored tensors (wgrad1x1_wide_kernel<MT, true>), model shapes at b = 8, graph-replayed.
usage: python tools/micro_wgrad1x1_half.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops

def gtime(fn, reps=10, iters=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * reps) * 1e3

ops.CONV_PRECISION = 1
for cin, cout, T, H in ((512, 296, 64, 6), (512, 64, 64, 6), (528, 448, 64, 6), (256, 288, 128, 12), (192, 176, 128, 12), (64, 64, 128, 24), (480, 304, 64, 6)):
    x = torch.randn(8, cin, T, H, H, device="cuda").relu().to(torch.bfloat16)
    dy = torch.randn(8, cout, T, H, H, device="cuda").to(torch.bfloat16)
    out = torch.empty(cout, cin, 1, 1, 1, device="cuda")
    f = lambda: ops.conv_wgrad(x, dy, (cout, cin, 1, 1, 1), (1, 1, 1), (1, 1, 1), out=out)
    t = gtime(f)
    fl = 2.0 * 8 * T * H * H * cin * cout
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    print(f"{cin:4d}->{cout:4d} {T}x{H}x{H}: {t:6.1f} us  {fl / t / 1e6:6.1f} TF/s  {mb / t * 1e3 / 1e3:5.2f} TB/s", flush=True)
