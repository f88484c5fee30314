#!/bin/bash
# usage (GPU box, repo root): tools/pmc_mem_path.sh <batch>: which unit of the memory path the one-kernel iteration keeps
# busy -- address unit (TA), data return (TD), L1 (TCP), L2 (TCC), the fabric's read credits -- per level-0 launch.
# Two or three counters of one block per pass (more "exceeds the capabilities of the hardware"); every pass under a time limit.
batch=${1:-32}
P1="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
P2="TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
P4="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
P5="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
P6="TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum"
P7="TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
for p in P1 P2 P3 P4 P5 P6 P7; do
  echo "pass $p: ${!p}"
  timeout -k 10 150 tools/pmc_pass.sh mem_$p 4k $batch ${!p} || echo "pass $p failed or timed out"
done
for p in P1 P2 P3 P4 P5 P6 P7; do python3 tools/pmc_top.py gpurun_out/pmc_mem_$p flow_iter || true; done > gpurun_out/pmc_mem_path.txt
cat gpurun_out/pmc_mem_path.txt
