#!/bin/bash
export TMPDIR=/tmp
L=${1:-3c}
run() { local d=gpurun_out/pmc_$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python tools/micro_pool.py $L 2 > /dev/null 2>&1
  python tools/pmc_summary.py $(dirname $(ls $d/*/*counter_collection.csv | head -1)) pool
  rm -rf $d; }
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU
run B SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
