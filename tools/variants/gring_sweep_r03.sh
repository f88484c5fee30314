#!/bin/bash
# (round 3: drives the TF_PC_GRING variant of k_flow_iter_pc_r03.hip.txt; kept with it for the record, not runnable against the current kernel)
# usage (GPU box): sweep workgroups per CU (via dynamic LDS padding) and the launcher's slot count for the global-ring build
run() { echo "--- $*"; env "$@" bash tools/variant_bench.sh 4k 32 gring | grep -E "ms/step in|flow_iter.k0|flow_iter.k1"; }
echo "=== base"; bash tools/variant_bench.sh 4k 32 base | grep -E "ms/step in|flow_iter.k0|flow_iter.k1"
# LDS per CU 163840: static 15360; padding P gives floor(163840/(15360+P)) workgroups per CU
run TF_PC_DYNLDS=39000 TF_PC_SLOTS=768      # 3 per CU
run TF_PC_DYNLDS=25000 TF_PC_SLOTS=1024     # 4 per CU
run TF_PC_DYNLDS=17000 TF_PC_SLOTS=1280     # 5 per CU
run TF_PC_DYNLDS=11000 TF_PC_SLOTS=1536     # 6 per CU
run TF_PC_DYNLDS=0 TF_PC_SLOTS=1792         # as many as fit (7 by registers)
run TF_PC_DYNLDS=25000 TF_PC_SLOTS=768      # 4 per CU, 2 segments as in the base
