#!/bin/bash
# HBM-side traffic of the training step PER KERNEL NAME (same two rocprofv3 --pmc passes as tools/pmc_step.sh: FETCH_SIZE and
# WRITE_SIZE separately, FETCH doubled -- the gfx950 correction of MI355X_MICROARCH.md), with the launch durations of the
# kernel trace: where a kernel moves more bytes than its tensors hold, or moves them slowly.
# usage (GPU box): tools/pmc_step_kernels.sh gpurun_out/r06_pmc_step_kernels.txt [top=60]
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_step_kernels.txt}
top=${2:-60}
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_sk_$c
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- \
      python bench.py --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off --steps 4 --warmup 2 > $d.log 2>&1
done
python - "$out" "$top" <<'PY'
import csv, glob, re, sys, collections
out, top = sys.argv[1], int(sys.argv[2])
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:78]
byt = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.defaultdict(int)
dur = collections.defaultdict(float)
steps = 0
for c in byt:
    f = glob.glob(f"/tmp/pmc_sk_{c}/*/*counter_collection.csv")[0]
    n = 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        n += "detection_loss_kernel" in k
        byt[c][k] += float(r["Counter_Value"]) * 1e3
        if c == "FETCH_SIZE":
            cnt[k] += 1
            if "End_Timestamp" in r and "Start_Timestamp" in r:
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    steps = n
rows = []
for k in cnt:
    fe, wr = 2 * byt["FETCH_SIZE"][k] / steps / 1e6, byt["WRITE_SIZE"][k] / steps / 1e6
    rows.append((fe + wr, k, cnt[k] / steps, fe, wr, dur[k] / steps))
rows.sort(reverse=True)
lines = [f"per kernel name, per step ({steps} steps, one stream, eager launches): FETCH_SIZE x 2 and WRITE_SIZE in MB, launch time under the counter pass in us",
         f"{'MB/step':>9s} {'n/step':>7s} {'fetch':>9s} {'write':>9s} {'MB/launch':>10s} {'us/step':>9s} {'TB/s':>6s}  kernel"]
for t, k, n, fe, wr, us in rows[:top]:
    lines.append(f"{t:9.1f} {n:7.1f} {fe:9.1f} {wr:9.1f} {t / n:10.2f} {us:9.1f} {(t / us if us else 0):6.2f}  {k}")
lines.append(f"{sum(r[0] for r in rows):9.1f} whole step")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
