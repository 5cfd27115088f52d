#!/bin/bash
# Ablation of the Conv3d_1a forward kernels' phases (b = 8, bf16-stored output).  Results are WRONG under these flags: timing only.
#   tools/ablate_direct.sh build      (where hipcc is)   then, on the GPU box:   [NOTILE=1] tools/ablate_1a.sh
# OTAL_CONV_DEBUG bits: 4 no staging loads, 8 no staging LDS stores (2x2 kernel only), 128 no weight loads, 16 no K-loop
# barrier, 64 no epilogue, 256 no operand LDS reads, 512 no MFMAs.
cd "$(dirname "$0")/.."
L=opental_amd/lib
for dbg in ${FLAGS:-0 468 212 340 84 20 4 128 256 64 16 512}; do
  echo "== OTAL_CONV_DEBUG=$dbg"
  OTAL_CONV_1A_NOTILE=${NOTILE:-0} OTAL_HALF_OUT=1 OTAL_LIB_PATH=$(pwd)/$L/libopental_ablate.so OTAL_PREC=1 OTAL_CONV_DEBUG=$dbg python tools/micro_conv.py 1a 20 fwd 2>&1 | grep -v amdgpu
done
