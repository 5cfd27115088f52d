#!/bin/bash
# HBM-side traffic of the WHOLE training step (rocprofv3 FETCH_SIZE / WRITE_SIZE in separate passes, as
# MI355X_MICROARCH.md prescribes), summed per kernel class and divided by the number of steps in the run.
# usage: tools/pmc_step.sh out_prefix      (writes out_prefix.txt and out_prefix.json)
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_step}
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_step_$c
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- \
      python bench.py --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off --steps 4 --warmup 2 > $d.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
GROUPS = [("conv gemm", ("conv_gemm", "conv_wgrad", "conv3_direct", "conv1a_direct", "conv1a_tile", "conv3_wgrad_direct", "conv1d_tile", "conv1a_wgrad", "wgrad1x1_wide", "conv3_wgrad_planes", "proj_fwd", "proj_wgrad", "conv1x1_stream")), ("split-K reduce", ("splitk_reduce",)),
          ("conv prologue", ("prep_chunks", "pack_wt", "pack_direct", "pack_conv1a", "build_")), ("max-pool", ("maxpool",)),
          ("groupnorm", ("gn_relu",)), ("adam", ("adam_flat",)), ("bmp", ("bmp_",)), ("other", ())]
tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.defaultdict(int)
steps = 0
for c in tot:
    f = glob.glob(f"gpurun_out/pmc_step_{c}/*/*counter_collection.csv")[0]
    nstep = 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        n = r["Kernel_Name"]
        if "detection_loss_kernel" in n:
            nstep += 1
        for g, pats in GROUPS:
            if not pats or any(p in n for p in pats):
                tot[c][g] += float(r["Counter_Value"]) * 1e3          # KB -> bytes
                if c == "FETCH_SIZE":
                    cnt[g] += 1
                break
    steps = nstep
res = {"steps_in_run": steps, "unit": "MB per step", "classes": {}}
lines = [f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --no-extras --graph off --steps 4 --warmup 2`: {steps} training steps, b=8 bf16.",
         "FETCH x2 = the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte requests at 64 B); WRITE_SIZE 1:1.", "",
         f"{'kernel class':16s} {'launches/step':>13s} {'FETCH x2 MB/step':>17s} {'WRITE MB/step':>14s} {'total MB/step':>14s}"]
gt = 0.0
for g, _ in GROUPS:
    fe, wr = 2 * tot["FETCH_SIZE"][g] / steps / 1e6, tot["WRITE_SIZE"][g] / steps / 1e6
    res["classes"][g] = {"launches_per_step": round(cnt[g] / steps, 1), "fetch_x2_MB": round(fe, 1), "write_MB": round(wr, 1)}
    lines.append(f"{g:16s} {cnt[g] / steps:13.1f} {fe:17.1f} {wr:14.1f} {fe + wr:14.1f}")
    gt += fe + wr
lines.append(f"{'whole step':16s} {sum(cnt.values()) / steps:13.1f} {'':17s} {'':14s} {gt:14.1f}")
conv = res["classes"]["conv gemm"]
res["conv_traffic_MB_per_launch"] = round((conv["fetch_x2_MB"] + conv["write_MB"]) / conv["launches_per_step"], 2)
res["step_traffic_MB"] = round(gt, 1)
sys.path.insert(0, "tools")
from source_stamp import source_stamp
res["source_stamp"] = source_stamp()      # bench.py quotes these figures only for the tree they were measured on
lines.append("")
lines.append(f"convolution GEMM kernels: {res['conv_traffic_MB_per_launch']} MB of HBM traffic per launch on average")
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
PY
rm -rf gpurun_out/pmc_step_FETCH_SIZE gpurun_out/pmc_step_WRITE_SIZE
