#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -- the b = 8 bench line's clips/s and ms/step for each value of one environment switch
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-hbm-kernels --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], d['ms_per_step'])"
done
