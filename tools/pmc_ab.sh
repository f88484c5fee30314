#!/bin/bash
# usage: tools/pmc_ab.sh <batch> <grid> <variant> ...
set -e
batch=$1; grid=$2; shift 2
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
B="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
for v in "$@"; do
  if [ "$v" = default ]; then unset TFHIP_LIBRARY; else export TFHIP_LIBRARY=$(pwd)/build_abl/libtfhip_$v.so; fi
  tools/pmc_pass.sh ${v}_a 4k $batch $A
  tools/pmc_pass.sh ${v}_b 4k $batch $B
done
for v in "$@"; do for p in a b; do python3 tools/pmc_top.py gpurun_out/pmc_${v}_$p flow_iter $grid; done; done > gpurun_out/pmc_ab.txt
cat gpurun_out/pmc_ab.txt
