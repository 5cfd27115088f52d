#!/bin/bash
# HBM-side traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes: the two do not
# fit one TCC pass) of the dominant kernels, per launch.  usage: tools/pmc_hbm.sh  (writes to stdout)
export TMPDIR=/tmp
pass() {  # counter, command...
  local c=$1; shift
  local d=gpurun_out/pmc_hbm_$c
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- "$@" > /dev/null 2>&1
  python - "$d" "$c" <<'PY'
import csv, glob, sys, collections
d, c = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(p in n for p in ("conv", "maxpool", "bmp_", "adam", "gn_relu")) and r["Counter_Name"] == c:
        agg[n.replace("(anonymous namespace)::", "")[:78]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[len(v) // 2:]          # skip the warm-up launches
    print(f"  {c:11s} {sum(v)/len(v)/1e3:12.1f} MB/launch (n={len(v):2d})  {k}")
PY
  rm -rf $d
}
echo "== Conv3d_2c (64->192, 3x3x3, x = 8x64x128x24x24): OTAL_PREC=1 python tools/micro_conv.py 2c 3 fwd,dgrad,wgrad"
for c in FETCH_SIZE WRITE_SIZE; do OTAL_PREC=1 pass $c python tools/micro_conv.py 2c 3 fwd,dgrad,wgrad; done
echo "== max-pools 2a / 3b: python tools/micro_pool.py 2a,3b 3"
for c in FETCH_SIZE WRITE_SIZE; do pass $c python tools/micro_pool.py 2a,3b 3; done
