"""Time one weight-gradient shape in isolation (run under rocprofv3 --kernel-trace --stats for per-kernel durations).
usage: python tools/micro_wgrad.py B Cin Cout T H W k [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops


def main():
    B, Cin, Cout, T, H, W, k = [int(v) for v in sys.argv[1:8]]
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
    ops.CONV_PRECISION = 1
    x = torch.randn(B, Cin, T, H, W, device="cuda")
    dy = torch.randn(B, Cout, T, H, W, device="cuda")
    for _ in range(3):
        ops.conv_wgrad(x, dy, (Cout, Cin, k, k, k), (k, k, k), (1, 1, 1))
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        ops.conv_wgrad(x, dy, (Cout, Cin, k, k, k), (k, k, k), (1, 1, 1))
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    print(f"{ms:.4f} ms  {2.0 * B * T * H * W * Cin * Cout * k ** 3 / ms / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
