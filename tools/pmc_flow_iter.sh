#!/bin/bash
# usage (GPU box, repo root): tools/pmc_flow_iter.sh [variant ...]   ("default" = the in-tree library; others =
# build_abl/libtfhip_<variant>.so).  Three rocprofv3 --pmc passes (kernel trace only) over tools/kprof.py 4k 16 per
# variant; counters of the largest (level-0) dispatches of the fused iteration -> gpurun_out/pmc_flow_iter.txt
set -e
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
B="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
variants=${@:-default}
for v in $variants; do
  if [ "$v" = default ]; then unset TFHIP_LIBRARY; else export TFHIP_LIBRARY=$(pwd)/build_abl/libtfhip_$v.so; fi
  tools/pmc_pass.sh ${v}_a 4k 16 $A
  tools/pmc_pass.sh ${v}_b 4k 16 $B
  tools/pmc_pass.sh ${v}_c 4k 16 $C
done
for v in $variants; do for p in a b c; do python3 tools/pmc_top.py gpurun_out/pmc_${v}_$p flow_iter; done; done > gpurun_out/pmc_flow_iter.txt
cat gpurun_out/pmc_flow_iter.txt
