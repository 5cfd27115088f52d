#!/bin/bash
# rocprofv3-counter HBM traffic of the HBM-bound kernel classes the north star names (BoundaryMaxPooling fwd / bwd, the 1-D
# convolution, GroupNorm, the strided max-pool, the bf16-tensor 1x1x1 convolution, Adam), per launch, next to their algorithmic
# bytes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one TCC pass; MI355X_MICROARCH.md), FETCH doubled
# (the gfx950 correction of the same guide: 128-byte requests are tallied at 64 B), WRITE 1:1.  Each pass runs
# tools/bench_hbm_kernels.py with plain (eager) launches, EAGER per entry; rows are attributed to the entries by kernel name in
# launch order.   usage (GPU box): tools/pmc_hbm_kernels.sh gpurun_out/r04_pmc_hbm_kernels   -> <prefix>.txt / .json
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_hbm_kernels}
repo=$(pwd)
N=6
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_hk_$c
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_hk_$c -o p -- python $repo/tools/bench_hbm_kernels.py 8 --eager $N > /tmp/pmc_hk_$c.json 2> /tmp/pmc_hk_$c.err )
done
python - "$out" $N <<'PY'
import csv, glob, json, os, sys
out, N = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from source_stamp import source_stamp
entries = json.load(open("/tmp/pmc_hk_FETCH_SIZE.json"))
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_hk_{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    pos = 0
    for name, e in entries.items():         # dict order = launch order
        pats = e["kernel"].split("|")
        got, start = [], pos
        while pos < len(rows) and len(got) < N:
            if any(p in rows[pos]["Kernel_Name"] for p in pats):
                got.append(float(rows[pos]["Counter_Value"]))
            pos += 1
        if not got:
            pos = start                     # (a renamed kernel must not swallow the rows of the entries behind it)
        tail = got[len(got) // 2:] or [float("nan")]
        per.setdefault(name, {})[c] = sum(tail) / len(tail)        # KB per launch (rocprofv3 unit), warm launches only
res = {"source_stamp": source_stamp(), "unit": "MB per launch", "kernels": {}}
lines = ["rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) per launch of the HBM-bound kernel classes at the b = 8 step's shapes",
         "(tools/bench_hbm_kernels.py --eager: plain launches, second half of each entry's launches).  FETCH x2 = the gfx950 correction of",
         "MI355X_MICROARCH.md; WRITE 1:1.  Working sets <= 40 MB stay in the 256 MB Infinity Cache between launches: the memory-side",
         "counters then see fabric requests, hits included (the guide's note on FETCH_SIZE), so `counter` can exceed DRAM traffic.",
         "", f"{'kernel':26s} {'algorithmic MB':>15s} {'FETCH x2 MB':>12s} {'WRITE MB':>10s} {'counter MB':>11s} {'ratio':>6s}"]
for name, e in entries.items():
    f2, w = 2 * per[name]["FETCH_SIZE"] / 1e3, per[name]["WRITE_SIZE"] / 1e3
    tot, alg = f2 + w, e["algorithmic_MB"]
    res["kernels"][name] = {"algorithmic_MB": alg, "fetch_x2_MB": round(f2, 3), "write_MB": round(w, 3), "counter_MB": round(tot, 3),
                            "ratio": round(tot / alg, 3)}
    lines.append(f"{name:26s} {alg:15.3f} {f2:12.3f} {w:10.3f} {tot:11.3f} {tot / alg:6.2f}")
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
PY
