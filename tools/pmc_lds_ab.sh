#!/bin/bash
# usage (GPU box, repo root): tools/pmc_lds_ab.sh <batch> <variant> ...: LDS and instruction-cache counters of the
# largest dispatches of the one-kernel iteration ("default" = the in-tree library)
set -e
batch=$1; shift
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
D="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_BUSY_CU_CYCLES"
for v in "$@"; do
  if [ "$v" = default ]; then unset TFHIP_LIBRARY; else export TFHIP_LIBRARY=$(pwd)/build_abl/libtfhip_$v.so; fi
  tools/pmc_pass.sh ${v}_lds 4k $batch $C
  tools/pmc_pass.sh ${v}_ic 4k $batch $D
done
for v in "$@"; do for p in lds ic; do python3 tools/pmc_top.py gpurun_out/pmc_${v}_$p flow_iter; done; done > gpurun_out/pmc_lds_ab.txt
cat gpurun_out/pmc_lds_ab.txt
