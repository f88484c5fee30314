#!/bin/bash
# usage (GPU box, repo root): tools/r03_evidence.sh <part>  ->  gpurun_out/r03_*
#   part a: the bench line of the driver's command; rocprofv3 --kernel-trace --stats of the same command
#           (--no-extra: every launch of a kernel belongs to the one workload) and the line printed under it;
#           per-level kernel times (tools/kprof.py).
#   part b: FETCH_SIZE / WRITE_SIZE passes over tools/calibrate_fetch.py -> the per-pixel traffic table;
#           three SQ counter passes over tools/kprof.py 4k 32 -> counters of the frame-expansion kernels and of
#           the iteration kernel (largest dispatch of each).
#   part c: the random-configuration fuzz with the exact mode; the drop-in path through host arrays, BGR frames in.
# rocprofv3 runs the program itself after `--` (python3 <script>), counters in passes of their own.
set -e
part=${1:-a}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
if [ "$part" = a ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r03_bench_4k_default.json 2> $out/r03_bench_4k_default.err
  tail -c 300 $out/r03_bench_4k_default.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_stats -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $out/r03_bench_4k_under_rocprof.json 2> $out/r03_bench_under_rocprof.err)
  cp $(find $out/r03_stats -name "*kernel_stats.csv" | head -1) $out/r03_bench_4k_kernel_stats.csv
  python3 tools/kprof.py 4k 32 > $out/r03_kprof_4k_batch32.txt 2>&1
  head -8 $out/r03_bench_4k_kernel_stats.csv
fi
if [ "$part" = b ]; then
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/r03_fetch -- python3 $root/tools/calibrate_fetch.py > $out/r03_fetch.log 2>&1)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/r03_write -- python3 $root/tools/calibrate_fetch.py > $out/r03_write.log 2>&1)
  python3 tools/traffic_from_pmc.py gpurun_out/r03_fetch gpurun_out/r03_write gpurun_out/r03_traffic.json
  A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
  B="SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  C="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_IFETCH SQ_INSTS_VALU_MFMA_I8"
  tools/pmc_pass.sh r03_a 4k 32 $A
  tools/pmc_pass.sh r03_b 4k 32 $B
  tools/pmc_pass.sh r03_c 4k 32 $C
  for k in k_level0_polyexp_t k_level1_polyexp_t k_level_rowpass k_level_colpass k_polyexp_t k_flow_iter_pc k_remap_step; do
    for p in a b c; do python3 tools/pmc_top.py gpurun_out/pmc_r03_$p $k; done
  done > $out/r03_sq_counters.txt
  tail -40 $out/r03_sq_counters.txt
  # what the exact mode costs: the same step with the box window summed in OpenCV's own order
  python3 tools/kprof.py 4k 8 > $out/r03_kprof_4k_batch8_default.txt 2>&1
  python3 tools/kprof.py 4k 8 fb_exact_sums=1 > $out/r03_kprof_4k_batch8_exact.txt 2>&1
  head -4 $out/r03_kprof_4k_batch8_default.txt $out/r03_kprof_4k_batch8_exact.txt
fi
if [ "$part" = c ]; then
  python3 tools/fuzz_fused.py 400 7 > $out/r03_fuzz_exact.txt 2>&1 || true
  tail -3 $out/r03_fuzz_exact.txt
  (python3 tools/bench_host_path.py 1080p 24 bgr; python3 tools/bench_host_path.py 4k 12 bgr; python3 tools/bench_host_path.py 1080p 24 grey; python3 tools/bench_host_path.py 4k 12 grey) > $out/r03_host_path.txt 2>&1
  cat $out/r03_host_path.txt
fi
