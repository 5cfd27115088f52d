#!/bin/bash
# Per-kernel register / LDS / occupancy figures of one csrc file as hipcc reports them:
#   tools/kernel_regs.sh conv_gemm.hip [name filter]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Iopental_amd/csrc --cuda-device-only -c \
  opental_amd/csrc/$1 -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | \
  awk -v f="$2" '/Function Name/ {name=$0; sub(/.*Function Name: /,"",name)} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF} /LDS Size/ {if (f=="" || index(name,f)) printf "%-100s vgpr %s agpr %s scratch %s occ %s lds %s\n", substr(name,1,100), v, a, sc, o, $NF}' | c++filt
