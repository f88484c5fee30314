#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel_mem.sh <kernel-substring> <batch> [out-name]: the memory path under one
# kernel of tools/kprof.py 4k <batch> (its largest grid): waves and their cycles, L1 and L2 requests and hits, fabric
# reads and their stalls, fetched / written bytes.  Two or three counters of one block per pass, every pass under a time limit.
k=$1; batch=${2:-32}; out=${3:-pmc_kmem}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY"
P2="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
P4="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
P5="TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum"
P6="FETCH_SIZE"
P7="WRITE_SIZE"
P8="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P9="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY"
for p in P1 P2 P3 P4 P5 P6 P7 P8 P9; do
  echo "pass $p: ${!p}"
  timeout -k 10 150 tools/pmc_pass.sh ${out}_$p 4k $batch ${!p} || echo "pass $p failed or timed out"
done
for p in P1 P2 P3 P4 P5 P6 P7 P8 P9; do python3 tools/pmc_top.py gpurun_out/pmc_${out}_$p $k || true; done > gpurun_out/${out}.txt
cat gpurun_out/${out}.txt
