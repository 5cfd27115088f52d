"""Micro-benchmark of the BoundaryMaxPooling kernels (HIP events on the launch stream)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.prop_pooling import boundary_pooling_op as bp


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def realistic_windows(B, lens, frame_num, gen):
    """Windows as the model produces them: loc = predicted half-lengths (log-uniform 1..40 frames),
    turned into level / frame windows by the product's own index kernel (BDNet.py:355-384)."""
    from opental_amd.common import ops
    lev = [0]
    for t in lens:
        lev.append(lev[-1] + t)
    loc = torch.exp(torch.rand(B, lev[-1], 2, device="cuda", generator=gen) * 3.7)
    return ops.proposal_windows(loc, lev, float(frame_num))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    g = torch.Generator(device="cuda").manual_seed(0)
    res = []
    lens = [64, 32, 16, 8, 4, 2]
    st = [0]
    for t in lens:
        st.append(st[-1] + t)
    for name, C, T, N, tabs in (("level_packed", 1024, 126, 126, (st, st)), ("frame", 512, 256, 126, None),
                                ("level0", 1024, 64, 64, None), ("frame_l0", 512, 256, 64, None)):
        x = torch.randn(B, C, T, device="cuda", generator=g)
        if tabs:
            seg = torch.cat([torch.sort(torch.rand(B, t, 2, 2, device="cuda", generator=g) * t, -1)[0].reshape(B, t, 4).round() for t in lens], 1).contiguous()
            f = lambda: bp.bmp_forward_levels(x, seg, st, st)
            go = torch.randn(B, C, N, device="cuda", generator=g)
            b = lambda: bp.bmp_backward_levels(go, x, seg, st, st)
        else:
            seg = torch.sort(torch.rand(B, N, 2, 2, device="cuda", generator=g) * T, -1)[0].reshape(B, N, 4).round()
            f = lambda: bp.bmp_forward(x, seg)
            go = torch.randn(B, C, N, device="cuda", generator=g)
            b = lambda: bp.bmp_backward(go, x, seg)
        tf, tb = timeit(f), timeit(b)
        bytes_f = 4 * B * (C * T + 4 * N + C * N)
        bytes_b = 4 * B * (C * N + C * T + 4 * N + C * T)
        res.append(dict(case=name, B=B, C=C, T=T, N=N, fwd_us=tf * 1e6, bwd_us=tb * 1e6,
                        fwd_GBps=bytes_f / tf / 1e9, bwd_GBps=bytes_b / tb / 1e9))
    # the four launches of one training-step forward/backward with realistic windows
    seg, fseg = realistic_windows(B, lens, 256, g)
    x1 = torch.randn(B, 1024, 126, device="cuda", generator=g).relu_()
    x2 = torch.randn(B, 512, 256, device="cuda", generator=g).relu_()
    g1 = torch.randn(B, 1024, 126, device="cuda", generator=g)
    g2 = torch.randn(B, 512, 126, device="cuda", generator=g)
    for name, f, b, C, T, N in (
            ("model_level_packed", lambda: bp.bmp_forward_levels(x1, seg, st, st), lambda: bp.bmp_backward_levels(g1, x1, seg, st, st), 1024, 126, 126),
            ("model_frame", lambda: bp.bmp_forward(x2, fseg), lambda: bp.bmp_backward(g2, x2, fseg), 512, 256, 126)):
        tf, tb = timeit(f), timeit(b)
        bytes_f = 4 * B * (C * T + 4 * N + C * N)
        bytes_b = 4 * B * (C * N + C * T + 4 * N + C * T)
        res.append(dict(case=name, B=B, C=C, T=T, N=N, fwd_us=tf * 1e6, bwd_us=tb * 1e6,
                        fwd_GBps=bytes_f / tf / 1e9, bwd_GBps=bytes_b / tb / 1e9))
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
