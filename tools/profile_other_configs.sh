#!/bin/bash
# rocprofv3 kernel statistics (one stream, eager launches) of the other BASELINE training configurations: the yaml's own batch
# (b = 1) and the ActivityNet recipe (b = 2, 768 frames).  usage (through gpurun): tools/profile_other_configs.sh r03d
tag=${1:-rXX}
export TMPDIR=/tmp
repo=$(pwd)
for cfg in "b1:--batch 1" "anet:--recipe anet"; do
  name=${cfg%%:*}; args=${cfg#*:}
  ( cd /tmp && rm -rf /tmp/prof_${tag}_$name && OTAL_WGRAD_STREAM=0 OTAL_BRANCH_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$name -o b -- \
      python $repo/bench.py $args --steps 15 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off > /tmp/prof_${tag}_$name.log 2>&1 )
  python tools/kstats.py $(find /tmp/prof_${tag}_$name -name "*kernel_stats.csv" | head -1) 18 40 > gpurun_out/${tag}_${name}_kernel_summary.txt
  head -11 gpurun_out/${tag}_${name}_kernel_summary.txt
done
