"""Micro-benchmark of the fused 1-D block launch (csrc/block1d.hip) against the launches it replaces (conv + GroupNorm),
b = 8, bf16 operands, both replayed from a HIP graph of 20 back-to-back launches (kernel + boundary time).
usage: python tools/micro_block1d.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops, block1d as B1
LEV = (0, 64, 96, 112, 120, 124, 126)


def graph_time(fn, reps=20, iters=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * reps) * 1e3      # us per call


def main():
    ops.CONV_PRECISION = 1
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = "cuda"
    cases = [("towers k3 512->512 x2", 512, 512, 3, 126, LEV, 2, "s1"),
             ("lr k1 512->1024 x2", 512, 1024, 1, 126, LEV, 2, "s1"),
             ("proposal k1 2048->512 x2", 2048, 512, 1, 126, LEV, 2, "s1"),
             ("cur_point k1 512->512 x2", 512, 512, 1, 126, LEV, 2, "s1"),
             ("deconv k3 T=256 x1", 512, 512, 3, 256, (0, 256), 1, "s1"),
             ("pyramid s2 32->16 x1", 512, 512, 3, 16, (0, 16), 1, "s2")]
    for name, cin, cout, kt, T, lev, npr, kind in cases:
        Tin = T * 2 if kind == "s2" else T
        xs = [torch.randn(B, cin, Tin, device=dev) for _ in range(npr)]
        ws = [torch.randn(cout, cin, kt, device=dev) * 0.03 for _ in range(npr)]
        bias = [torch.randn(cout, device=dev) * 0.1 for _ in range(npr)]
        g, be = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        levels = lev if len(lev) > 2 else None
        # ---- the launches in use today
        w5 = [w.view(cout, cin, kt, 1, 1) for w in ws]
        cache = ops.PrologueCache((min(w.data_ptr() for w in ws), max(w.data_ptr() + 4 * w.numel() for w in ws)))

        def old():
            k, s = (kt, 1, 1), ((2 if kind == "s2" else 1), 1, 1)
            cs = None
            if npr == 2:
                cs = ops.conv_forward_pair(xs, w5, k, s, bias, levels)
            if cs is None:
                cs = [ops.conv_forward(x, w, k, s, shift=b_, levels=levels) for x, w, b_ in zip(xs, w5, bias)]
            ys = ops.gn_relu_forward_pair(cs, (g, g), (be, be), 32, 1e-5, True, levels) if npr == 2 else None
            if ys is None:
                ys = [ops.gn_relu_forward(c, g, be, 32, 1e-5, True, levels) for c in cs]
            return ys
        ops.activate_prologues(cache)
        old()
        ops.activate_prologues(cache)
        t_old = graph_time(old)
        ops.deactivate_prologues()
        # ---- fused
        packs = [B1.Pack(w, dgrad=False) for w in ws]
        B1.PackSet(packs).refresh()
        res = {}
        for kc in (64, 128):
            for rg in ((None,) if len(lev) == 2 else (None, (0, 1, 6))):
                outs = [(torch.empty(B, cout, T, device=dev), torch.empty(B, cout, T, device=dev),
                         torch.empty(B, 32, len(lev) - 1, 2, device=dev)) for _ in range(npr)]
                probs = []
                for i in range(npr):
                    if kind == "s2":
                        sg = B1.seg(xs[i], packs[i].fwd, cin, 3, mul=2, off=0, Tv=Tin)
                    else:
                        sg = B1.seg(xs[i], packs[i].fwd, cin, kt, off=-(kt // 2), use_levels=True)
                    probs.append(B1.problem(B1.FWD, B, cout, T, [sg], outs[i][1], c=outs[i][0], stats=outs[i][2], gamma=g, beta=be,
                                            bias=bias[i], levels=lev, ranges=rg, kc=kc))
                ok = B1.launch(probs)
                if not ok:
                    res[(kc, rg is not None)] = float("nan")
                    continue
                res[(kc, rg is not None)] = graph_time(lambda: B1.launch(probs))
        print(f"{name:28s} old {t_old:6.1f} us | fused " + "  ".join(f"kc{k}{'/split' if r else ''} {v:6.1f}" for (k, r), v in res.items()), flush=True)


if __name__ == "__main__":
    main()
