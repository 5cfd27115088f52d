"""Direct 3x3x3 kernel on the 6x6 / 12x12 planes (Mixed_3b .. 4f b1b / b2b): forward and data gradient of every layer
under the tile-selection knobs given in the environment (HIP events, graph-free).  usage: micro_planes6.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops

LAYERS = [  # name, (B, Cin, T, H, W), Cout
    ("3b_b1b", (8, 96, 128, 12, 12), 128), ("3b_b2b", (8, 16, 128, 12, 12), 32),
    ("3c_b1b", (8, 128, 128, 12, 12), 192), ("3c_b2b", (8, 32, 128, 12, 12), 96),
    ("4b_b1b", (8, 96, 64, 6, 6), 208), ("4b_b2b", (8, 16, 64, 6, 6), 48),
    ("4c_b1b", (8, 112, 64, 6, 6), 224), ("4c_b2b", (8, 24, 64, 6, 6), 64),
    ("4d_b1b", (8, 128, 64, 6, 6), 256),
    ("4e_b1b", (8, 144, 64, 6, 6), 288), ("4e_b2b", (8, 32, 64, 6, 6), 64),
    ("4f_b1b", (8, 160, 64, 6, 6), 320), ("4f_b2b", (8, 32, 64, 6, 6), 128),
    ("5b_b1b", (8, 160, 32, 3, 3), 320), ("5c_b1b", (8, 192, 32, 3, 3), 384), ("5c_b2b", (8, 48, 32, 3, 3), 128),
]


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(2_000_000)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    ops.CONV_PRECISION = 1
    k, s = (3, 3, 3), (1, 1, 1)
    tot = [0.0, 0.0, 0.0]
    for name, shape, cout in LAYERS:
        if only and name not in only:
            continue
        x = torch.randn(*shape, device="cuda")
        w = torch.randn(cout, shape[1], 3, 3, 3, device="cuda") * 0.05
        sc = torch.rand(cout, device="cuda") + 0.5
        sci = torch.rand(shape[1], device="cuda") + 0.5
        y = ops.conv_forward(x, w, k, s, scale=sc, shift=sc, relu=True)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        fl = 2.0 * y.numel() * shape[1] * 27
        tf = timeit(lambda: ops.conv_forward(x, w, k, s, scale=sc, shift=sc, relu=True, out=y), iters)
        td = timeit(lambda: ops.conv_dgrad(dy, w, x.shape, k, s, out=dx, out_mask=x, out_scale=sci), iters)
        tw = timeit(lambda: ops.conv_wgrad(x, dy, w.shape, k, s, out=dw), iters)
        tot[0] += tf; tot[1] += td; tot[2] += tw
        print(f"{name:7s} fwd {tf:6.1f} us {fl/tf/1e6:6.0f} TF | dgrad {td:6.1f} us {fl/td/1e6:6.0f} TF | wgrad {tw:6.1f} us {fl/tw/1e6:6.0f} TF", flush=True)
    print(f"total   fwd {tot[0]:6.1f} us | dgrad {tot[1]:6.1f} us | wgrad {tot[2]:6.1f} us")


if __name__ == "__main__":
    main()
