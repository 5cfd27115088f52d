#!/bin/bash
# MFMA utilisation of the convolution kernels over the whole training step: one rocprofv3 --pmc pass (SQ + GRBM counters)
# over bench.py.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the gfx94x formula
# MI355X_MICROARCH.md says the derived metrics fall back to, with GRBM_GUI_ACTIVE -- which rocprofv3 reports summed over
# the 8 XCDs -- brought back to one clock (calibration: Conv3d_1a forward, 0.77 ms by HIP events, reads 14.2 M = 8 x 1.78 M
# cycles = 2.3 GHz; SQ_VALU_MFMA_BUSY_CYCLES = 32 x SQ_INSTS_MFMA as the guide states).  usage: tools/pmc_mfma.sh out.txt
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_mfma.txt}
d=gpurun_out/pmc_mfma
rm -rf $d
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace --output-format csv -d $d -- \
    python bench.py --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off --steps 4 --warmup 2 > $d.log 2>&1
python - "$d" "$out" <<'PY'
import csv, glob, sys, collections
d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
steps = 0
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    per[n][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[n].add(r["Dispatch_Id"])
    if "detection_loss_kernel" in n and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        steps += 1
CONV = ("conv_gemm", "conv_wgrad", "conv3_direct", "conv1a_direct", "conv1a_tile", "conv3_wgrad", "conv1a_wgrad", "wgrad1x1_wide", "conv1d_tile", "proj_fwd")
rows = [(n, c) for n, c in per.items() if any(p in n for p in CONV)]
rows.sort(key=lambda nc: -nc[1]["GRBM_GUI_ACTIVE"])
lines = [f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA over `bench.py --graph off --steps 4 --warmup 2`: {steps} training steps, b=8, bf16 operands.",
         "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)  [gfx94x derived-metric formula; GUI_ACTIVE is reported summed over the XCDs:",
         " Conv3d_1a forward reads 14.2 M for 0.77 ms = 8 x 2.3 GHz].  It counts ISSUED MFMA cycles (32 per 32x32x16 bf16 instruction), i.e. it includes K / tile padding;",
         " share = the kernel's part of the conv kernels' GPU-active cycles.", "",
         f"{'kernel':58s} {'launches/step':>13s} {'share':>6s} {'MfmaUtil':>9s} {'M MFMA/launch':>13s}"]
tot_gui = sum(c["GRBM_GUI_ACTIVE"] for _, c in rows)
tot_mfma = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for _, c in rows)
for n, c in rows[:24]:
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    k = len(calls[n])
    lines.append(f"{n[:58]:58s} {len(calls[n]) / steps:13.1f} {c['GRBM_GUI_ACTIVE'] / tot_gui:6.1%} {util:9.1%} {c['SQ_INSTS_MFMA'] / k / 1e6:11.2f}")
lines.append("")
lines.append(f"all convolution kernels: MfmaUtil {tot_mfma / (tot_gui / 8 * 256 * 4):.1%} over {sum(len(calls[n]) for n, _ in rows) / steps:.0f} launches per step")
allk_gui = sum(c["GRBM_GUI_ACTIVE"] for c in per.values())
lines.append(f"whole step (every kernel): MfmaUtil {sum(c['SQ_VALU_MFMA_BUSY_CYCLES'] for c in per.values()) / (allk_gui / 8 * 256 * 4):.1%}; the conv kernels are {tot_gui / allk_gui:.0%} of the GPU-active cycles")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $d
