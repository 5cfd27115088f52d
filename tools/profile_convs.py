"""Per-layer table of the implicit-GEMM launches of one training step (HIP events per launch)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opental_amd.common import ops


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ops.CONV_PRECISION = int(os.environ.get("OTAL_PREC", "1"))
    dev = torch.device("cuda", 0)
    if os.environ.get("OTAL_RECIPE") == "anet":         # configs/anet_opental.yaml: 768-frame clips, 150 classes
        tr = bench.build_anet_trainer(dev)
        clips, targets, scores = bench.synth_batch(batch, 1000, dev, frames=768, classes=150, score_rows=3)
    else:
        tr = bench.build_trainer(dev)
        clips, targets, scores = bench.synth_batch(batch, 1000, dev)
    for _ in range(2):
        tr.step(clips, targets, scores)
    rows = []
    orig = ops._prof_end

    def tagged(ev, mode, g, problems=1):       # problems = 2: a pair launch (two sibling layers in one grid)
        if ev is None:
            return
        end = torch.cuda.Event(enable_timing=True); end.record()
        B, Cin, Cout = g[0], g[1], g[2]
        flops = 2.0 * B * Cout * g[6] * g[7] * g[8] * Cin * g[9] * g[10] * g[11] * problems
        rows.append((mode, tuple(g[:18]), flops, ev, end))
    ops._prof_end = tagged
    ops.CONV_PROFILE = []
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); tr.step(clips, targets, scores); t1.record()
    torch.cuda.synchronize()
    ops.CONV_PROFILE = None
    ops._prof_end = orig
    agg = {}
    for mode, g, fl, a, b in rows:
        e = agg.setdefault((mode, g), [0, 0.0, 0.0])
        e[0] += 1; e[1] += a.elapsed_time(b); e[2] += fl
    tot = sum(e[1] for e in agg.values())
    print(f"step {t0.elapsed_time(t1):.2f} ms, conv {tot:.2f} ms, {len(rows)} launches")
    print("mode   n   ms     TF/s   B Cin Cout  Ti Hi Wi -> To Ho Wo  k  s")
    for (mode, g), e in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('OTAL_TOP', '60'))]:
        print(f"{mode:6s}{e[0]:3d} {e[1]:7.3f} {e[2]/e[1]/1e9:7.1f}  {g[0]} {g[1]:4d} {g[2]:4d}  {g[3]:3d} {g[4]:2d} {g[5]:2d} -> {g[6]:3d} {g[7]:2d} {g[8]:2d}  {g[9]}{g[10]}{g[11]} {g[12]}{g[13]}{g[14]}")
    for name, sel in (("1-D pyramid / heads (H=W=1 outputs)", lambda g: g[7] == 1 and g[8] == 1), ("3-D backbone", lambda g: not (g[7] == 1 and g[8] == 1))):
        t = sum(e[1] for (m, g), e in agg.items() if sel(g)); f = sum(e[2] for (m, g), e in agg.items() if sel(g)); n = sum(e[0] for (m, g), e in agg.items() if sel(g))
        print(f"{name}: {n} launches, {t:.2f} ms, {f/t/1e9:.1f} TF/s")
    for mode in ("fwd", "dgrad", "wgrad"):
        t = sum(e[1] for (m, _), e in agg.items() if m == mode); f = sum(e[2] for (m, _), e in agg.items() if m == mode)
        print(mode, f"{t:.2f} ms {f/t/1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
