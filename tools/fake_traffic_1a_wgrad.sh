#!/bin/bash
# Is conv1a_wgrad_direct_kernel waiting for the memory system?  Builds three instrumented libraries (compile-time switches,
# no run-time checks in the loop) in which every block re-reads the FIRST block's dy (1), x (2) or both (3) -- the loads,
# LDS traffic and MFMAs stay, the L2 / HBM traffic goes -- and times the layer.  Results are WRONG: timing only.
#   tools/fake_traffic_1a_wgrad.sh build   (where hipcc is)      tools/fake_traffic_1a_wgrad.sh   (on the GPU box)
cd "$(dirname "$0")/.."
L=opental_amd/lib
if [ "$1" = "build" ]; then
  for v in 1 2 3; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Iopental_amd/csrc -DOTAL_W1A_FAKE=$v -c opental_amd/csrc/conv_gemm.hip -o $L/obj/conv_gemm_fake$v.o &&
      objs=$(ls $L/obj/*.o | grep -v "_fake\|_ablate\|/conv_gemm.o") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libopental_fake$v.so $objs $L/obj/conv_gemm_fake$v.o && echo $L/libopental_fake$v.so ) &
  done
  wait
  exit
fi
echo "== product"; OTAL_HALF_OUT=1 OTAL_PREC=1 python tools/micro_conv.py 1a 20 wgrad 2>&1 | grep -v amdgpu
for v in 1 2 3; do
  echo "== fake traffic $v (1: dy, 2: x, 3: both)"
  OTAL_HALF_OUT=1 OTAL_LIB_PATH=$(pwd)/$L/libopental_fake$v.so OTAL_PREC=1 python tools/micro_conv.py 1a 20 wgrad 2>&1 | grep -v amdgpu
done
