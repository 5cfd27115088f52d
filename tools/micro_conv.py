"""Single-layer micro-benchmark of the implicit-GEMM conv (fwd / dgrad / wgrad), HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops

LAYERS = {
    "2c": ((8, 64, 128, 24, 24), 192, (3, 3, 3), (1, 1, 1)),
    "1a": ((8, 3, 256, 96, 96), 64, (7, 7, 7), (2, 2, 2)),   # dgrad unused by the model (input needs no gradient)
    "3c_b1b": ((8, 128, 128, 12, 12), 192, (3, 3, 3), (1, 1, 1)),
    "4f_b1b": ((8, 160, 64, 6, 6), 320, (3, 3, 3), (1, 1, 1)),
    "4b_b1b": ((8, 96, 64, 6, 6), 208, (3, 3, 3), (1, 1, 1)),
    "4c_b1b": ((8, 112, 64, 6, 6), 224, (3, 3, 3), (1, 1, 1)),
    "4d_b1b": ((8, 128, 64, 6, 6), 256, (3, 3, 3), (1, 1, 1)),
    "4e_b1b": ((8, 144, 64, 6, 6), 288, (3, 3, 3), (1, 1, 1)),
    "3b_b0": ((8, 192, 128, 12, 12), 64, (1, 1, 1), (1, 1, 1)),
    "2b": ((8, 64, 128, 24, 24), 64, (1, 1, 1), (1, 1, 1)),
    "3c_1x1": ((8, 256, 128, 12, 12), 288, (1, 1, 1), (1, 1, 1)),     # Mixed_3c: the fused [b1a | b2a | b0] launch
    "4e_1x1": ((8, 512, 64, 6, 6), 288, (1, 1, 1), (1, 1, 1)),
    "tower": ((8, 512, 126), 512, (3, 1, 1), (1, 1, 1)),
    "3c_b2b": ((8, 32, 128, 12, 12), 96, (3, 3, 3), (1, 1, 1)),
    "3b_b2b": ((8, 16, 128, 12, 12), 32, (3, 3, 3), (1, 1, 1)),
    "4b_b2b": ((8, 16, 64, 6, 6), 48, (3, 3, 3), (1, 1, 1)),
    "4c_b2b": ((8, 24, 64, 6, 6), 64, (3, 3, 3), (1, 1, 1)),
    "4f_b2b": ((8, 32, 64, 6, 6), 128, (3, 3, 3), (1, 1, 1)),
    "5b_b1b": ((8, 160, 32, 3, 3), 320, (3, 3, 3), (1, 1, 1)),
    "5c_b1b": ((8, 192, 32, 3, 3), 384, (3, 3, 3), (1, 1, 1)),
    "5c_b2b": ((8, 48, 32, 3, 3), 128, (3, 3, 3), (1, 1, 1)),
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


def main():
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(LAYERS)
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fwd", "dgrad", "wgrad"]
    ops.CONV_PRECISION = int(os.environ.get("OTAL_PREC", "0"))
    for name in names:
        shape, cout, k, s = LAYERS[name]
        half = os.environ.get("OTAL_HALF", "0") != "0"          # bf16-STORED tensors on both sides of the layer (ops.HALF_CHAIN kernels)
        x = torch.randn(*shape, device="cuda")
        if half:
            x = torch.relu(x).to(torch.bfloat16)
        kk = k[:1] if len(shape) == 3 else k
        w = torch.randn(cout, shape[1], *kk, device="cuda") * 0.05
        sc = torch.rand(cout, device="cuda") + 0.5
        sci = torch.rand(shape[1], device="cuda") + 0.5
        yh = os.environ.get("OTAL_HALF_OUT", "0") != "0"        # fp32 x, bf16-stored y / dy (the model's Conv3d_1a)
        y = ops.conv_forward(x, w, k, s, scale=sc, shift=sc, relu=True, half_out=yh)
        dy = torch.randn_like(y)
        wt = None if half else ops.pack_wt(w)
        flops = 2.0 * y.numel() * shape[1] * k[0] * k[1] * k[2]
        res = []
        if "fwd" in modes:
            t = timeit(lambda: ops.conv_forward(x, w, k, s, scale=sc, shift=sc, relu=True, out=y, half_out=yh), iters)
            res.append(f"fwd {t*1e3:7.3f} ms {flops/t/1e12:6.1f} TF")
        if "dgrad" in modes:
            dx = torch.empty_like(x)
            t = timeit(lambda: ops.conv_dgrad(dy, w, x.shape, k, s, out=dx, wt=wt, out_mask=x, out_scale=sci), iters)
            res.append(f"dgrad(+epi mask) {t*1e3:7.3f} ms {flops/t/1e12:6.1f} TF")
        if "wgrad" in modes:
            dw = torch.empty_like(w)        # (fp32 whatever the activations' storage)
            t = timeit(lambda: ops.conv_wgrad(x, dy, w.shape, k, s, out=dw), iters)
            res.append(f"wgrad {t*1e3:7.3f} ms {flops/t/1e12:6.1f} TF")
        print(f"{name:8s} " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
