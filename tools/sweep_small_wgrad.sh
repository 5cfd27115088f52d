#!/bin/bash
# small-Cin 3x3x3 weight gradients (the b2b branches): direct kernel (32-channel tiles) against the vector kernel
cd "$(dirname "$0")/.."
for minc in 64 32 16; do
 for blocks in 0 64; do
 echo "== OTAL_WDIRECT_MINC=$minc OTAL_WDIRECT6_MINC=$minc OTAL_WDIRECT_MINM=32 OTAL_WDIRECT_BLOCKS=$blocks"
 OTAL_WDIRECT_MINM=32 OTAL_WDIRECT_MINC=$minc OTAL_WDIRECT6_MINC=$minc OTAL_WDIRECT_BLOCKS=$blocks OTAL_HALF=1 OTAL_PREC=1 python tools/micro_conv.py 3c_b2b,3b_b2b,4b_b2b,4c_b2b,4f_b2b 20 wgrad 2>&1 | grep -v amdgpu
 done
done
