import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        agg.setdefault((r["Kernel_Name"][:70], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k[35:]:40s} {c:26s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
