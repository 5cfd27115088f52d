#!/bin/bash
# SQ counter passes over one micro_conv invocation; usage: tools/pmc_sq.sh <layer> <modes> [kernel-substring]
export TMPDIR=/tmp
L=${1:-2c}; M=${2:-fwd,dgrad,wgrad}; PAT=${3:-conv_gemm}
run() {  # name, counters...
  local d=gpurun_out/pmc_$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python tools/micro_conv.py $L 2 $M > /dev/null 2>&1
  python tools/pmc_summary.py $(dirname $(ls $d/*/*counter_collection.csv | head -1)) $PAT
  rm -rf $d
}
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run B SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run C SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR
