#!/bin/bash
# Ablation of conv3_direct_kernel's phases on Conv3d_2c / Mixed_3c.b1b.  Results are WRONG under these flags: timing only.
# Builds an instrumented library next to the product one (-DOTAL_DIRECT_ABLATE: the product kernel has no such branches) --
# run the build step where hipcc is (it cross-compiles without a GPU), the timing step on the GPU box:
#   tools/ablate_direct.sh build      -> opental_amd/lib/libopental_ablate.so
#   [LAYERS=3c_1x1,2b FLAGS="0 4 64 68"] tools/ablate_direct.sh   -> the table (the chunked kernel of the 1x1 layers has
#                                        the 4 / 128 / 64 hooks too) (4 no position loads in the K loop, 128 no weight loads, 8 no LDS stores,
#                                        16 no barrier, 64 no epilogue)
# Round 3 (b = 8, one box; Conv3d_2c forward / data gradient, us): all on 517 / 481; no position loads 458 / 394; no weight
# loads 484 / 442; neither 446 / 370; no epilogue 424 / 454; no LDS stores 503 / 477; loads, stores, barrier and epilogue all
# off 327 / 296 -- i.e. 63 % of the forward launch is the LDS-read + MFMA loop itself (1200 TFLOP/s in place; 1858 in the
# micro-benchmark without masks and address selects), the 453 MB epilogue 18 %, the position loads 11 % (18 % of the data
# gradient) and NOT through their latency (two steps of cover: -1 %), the weight loads 6 %.
cd "$(dirname "$0")/.."
L=opental_amd/lib
if [ "$1" = "build" ]; then
  objs=""
  for f in opental_amd/csrc/*.hip; do
    o=$L/obj/$(basename ${f%.hip}).o
    if [ "$(basename $f)" = "conv_gemm.hip" ] || [ "$(basename $f)" = "conv_gemm_half.hip" ] || [ "$(basename $f)" = "conv1a_tile.hip" ]; then
      o=$L/obj/$(basename ${f%.hip})_ablate.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Iopental_amd/csrc -DOTAL_DIRECT_ABLATE -c $f -o $o || exit 1
    fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libopental_ablate.so $objs && echo $L/libopental_ablate.so
  exit
fi
layers=${LAYERS:-2c,3c_b1b}
for dbg in ${FLAGS:-0 4 128 132 64 8 24 220}; do
  echo "== OTAL_CONV_DEBUG=$dbg"
  OTAL_LIB_PATH=$(pwd)/$L/libopental_ablate.so OTAL_PREC=1 OTAL_CONV_DEBUG=$dbg python tools/micro_conv.py $layers 20 fwd,dgrad 2>&1 | grep -v amdgpu
done
