"""Summarise a rocprofv3 kernel_stats.csv: total per step and the top kernels.  usage: kstats.py file.csv nsteps [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6/n:.2f} ms/step over {n:.0f} steps")
groups = {"conv gemm": ("conv_gemm", "conv_wgrad", "conv3_direct", "conv1a_direct", "conv1a_tile", "conv3_wgrad_direct", "conv1d_tile", "conv1a_wgrad", "wgrad1x1_wide", "conv3_wgrad_planes6", "proj_fwd", "conv1x1_stream", "head_convs_"), "split-K reduce": ("splitk_reduce",), "conv prologue": ("prep_chunks", "pack_wt", "pack_direct", "pack_conv1a", "build_"),
          "max-pool": ("maxpool",), "groupnorm": ("gn_relu",), "bmp": ("bmp_",), "adam": ("adam_flat",), "detection loss": ("detection_loss",), "torch/other": ()}
acc = {k: 0.0 for k in groups}
for r in rows:
    for k, pats in groups.items():
        if any(p in r["Name"] for p in pats):
            acc[k] += float(r["TotalDurationNs"]); break
    else:
        acc["torch/other"] += float(r["TotalDurationNs"])
for k, v in acc.items():
    print(f"  {k:16s} {v/1e6/n:7.2f} ms/step")
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs'])/1e6/n:8.3f} ms/step  n/step {int(r['Calls'])/n:7.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:120]}")
