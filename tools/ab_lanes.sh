#!/bin/bash
# A/B of the lane-graph step on ONE box (boxes differ by a few per cent): each configuration = env assignments, "-" = none
# usage: tools/ab_lanes.sh "-" "OTAL_CONV_DIRECT_NO32=1" "OTAL_CONV_DIRECT_NO32=1 OTAL_CONV_DIRECT_MINTILES=192"
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  for rep in 1 2; do
    r=$(env $cfg python bench.py --graph lanes --no-extras --no-cpu-baseline --no-hbm-kernels --no-roofline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "[$cfg] $r"
  done
done
