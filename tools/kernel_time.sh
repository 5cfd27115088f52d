#!/bin/bash
# Kernel-only durations (rocprofv3 --kernel-trace --stats) of one command:  tools/kernel_time.sh <top n> -- <command...>
top=${1:-8}; shift; [ "$1" = "--" ] && shift
repo=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
d=/tmp/ktime_$$
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- "$@" > /dev/null 2>&1 )
f=$(find $d -name '*kernel_stats.csv' | head -1)
python - "$f" "$top" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2])]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x {int(r['Calls']):4d}  {r['Name'][:110]}")
PY
rm -rf $d
