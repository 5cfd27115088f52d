import csv, sys
path, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", round(tot / 1e6 / steps, 2))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-92s n/step %7.1f ms/step %8.3f avg_us %8.1f" % (r["Name"][:92], int(r["Calls"]) / steps,
          float(r["TotalDurationNs"]) / steps / 1e6, float(r["AverageNs"]) / 1e3))
