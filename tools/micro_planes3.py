"""The 3x3x3 weight gradient on 3x3 planes (Mixed_5b / 5c b1b, b2b at b = 8): conv3_wgrad_planes3_kernel against the
generic gather kernel (OTAL_CONV_NOWDIRECT3=1 in a second process) and an fp64 reference on the bf16-rounded operands.
usage: python tools/micro_planes3.py [half]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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


half = len(sys.argv) > 1 and sys.argv[1] == "half"
ops.CONV_PRECISION = 1
for B, cin, cout, T in ((8, 192, 384, 32), (8, 160, 320, 32), (8, 48, 128, 32), (8, 32, 128, 32), (2, 160, 320, 32), (1, 34, 70, 48)):
    torch.manual_seed(1)
    x = torch.randn(B, cin, T, 3, 3, device="cuda").relu()
    dy = torch.randn(B, cout, T, 3, 3, device="cuda")
    if half:
        x, dy = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    out = torch.empty(cout, cin, 3, 3, 3, device="cuda")
    f = lambda: ops.conv_wgrad(x, dy, (cout, cin, 3, 3, 3), (3, 3, 3), (1, 1, 1), out=out)
    f(); torch.cuda.synchronize()
    xr, dyr = x.to(torch.bfloat16).double(), dy.to(torch.bfloat16).double()
    ref = torch.nn.grad.conv3d_weight(xr, (cout, cin, 3, 3, 3), dyr, stride=1, padding=1)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    t = gtime(f)
    fl = 2.0 * B * T * 9 * cin * cout * 27
    print(f"B{B} {cin:4d}->{cout:4d} T{T}: {t:6.1f} us  {fl / t / 1e6:6.1f} TF/s  rel err {err:.2e}", flush=True)
