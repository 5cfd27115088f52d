#!/bin/bash
# A/B of two library builds on one box: the 1x1 / small layers in isolation, then the whole step.
# usage: tools/ab_libs.sh libA.so libB.so
for lib in "$@"; do
 echo "== $lib"
 env OTAL_PREC=1 OTAL_LIB_PATH=$lib python tools/micro_conv.py 3c_1x1,2b,4e_1x1,3b_b0,2c 20 fwd,dgrad 2>&1 | grep -v amdgpu
done
bash tools/ab_lanes.sh "OTAL_LIB_PATH=$1" "OTAL_LIB_PATH=$2" "OTAL_LIB_PATH=$1" "OTAL_LIB_PATH=$2"
