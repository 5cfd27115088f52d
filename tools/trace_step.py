"""Kernel timeline of ONE training step from a rocprofv3 --kernel-trace CSV: launch order, duration and the idle gap in front
of every kernel.  usage: trace_step.py kernel_trace.csv [step_index_from_end=1]
A step is delimited by its last Adam launch (adam_flat*)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
# a recipe with several optimizer groups launches Adam once per group: a step ends with the LAST Adam launch of a cluster
# (the next one starts more than 3 ms later)
ends = [i for n, i in enumerate(adam)
        if n + 1 == len(adam) or int(rows[adam[n + 1]]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) > 3_000_000]
hi = ends[-back]
lo = ends[-back - 1] + 1
step = rows[lo:hi + 1]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy = gap = 0
qids = {}
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = s - prev_end
    busy += e - s
    gap += max(g, 0)
    q = qids.setdefault(r.get("Queue_Id", "?"), len(qids))        # lane = order of first appearance of the HSA queue
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {g / 1e3:6.1f}  q{q}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
print(f"step: {len(step)} kernels, wall {(prev_end - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {gap / 1e6:.3f} ms")
