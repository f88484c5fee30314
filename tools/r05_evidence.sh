#!/bin/bash
# usage (GPU box, repo root): tools/r05_evidence.sh <part>  ->  gpurun_out/r05_*
#   part a: the bench line of the driver's command (4K, 128 pairs per pass, one lane) and rocprofv3 --kernel-trace --stats
#           of the same command (--no-extra --no-alone: every launch of a kernel belongs to the one workload, and with one
#           lane a launch's duration is the kernel's own); the same pair at 32 pairs per pass -- what a rank runs at 8 GPUs
#           (two lanes); per-level kernel times (tools/kprof.py) at 128 and 32.
#   part b: FETCH_SIZE / WRITE_SIZE passes over tools/calibrate_fetch.py -> the per-pixel traffic table, and over the
#           batched step at 128 pairs; SQ counter passes over tools/kprof.py 4k 128.
#   part c: the random-configuration fuzz; the drop-in path (host arrays, prefetching, device flows, batched look-ahead).
# rocprofv3 runs the program itself after `--` (python3 <script>), counters in passes of their own.
set -e
part=${1:-a}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
if [ "$part" = a ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_4k_default.json 2> $out/r05_bench_4k_default.err
  tail -c 300 $out/r05_bench_4k_default.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05_stats -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-alone > $out/r05_bench_4k_one_lane_under_rocprof.json 2> $out/r05_bench_under_rocprof.err)
  cp $(find $out/r05_stats -name "*kernel_stats.csv" | head -1) $out/r05_bench_4k_one_lane_kernel_stats.csv
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --batch 32 --no-extra > $out/r05_bench_4k_batch32_two_lanes.json 2> $out/r05_bench_4k_batch32.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05_stats32 -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --batch 32 --lanes 1 --no-extra --no-alone > $out/r05_bench_4k_batch32_one_lane_under_rocprof.json 2> $out/r05_bench_b32_under_rocprof.err)
  cp $(find $out/r05_stats32 -name "*kernel_stats.csv" | head -1) $out/r05_bench_4k_batch32_one_lane_kernel_stats.csv
  python3 tools/kprof.py 4k 128 > $out/r05_kprof_4k_batch128.txt 2>&1
  python3 tools/kprof.py 4k 32 > $out/r05_kprof_4k_batch32.txt 2>&1
  head -6 $out/r05_bench_4k_one_lane_kernel_stats.csv $out/r05_bench_4k_batch32_one_lane_kernel_stats.csv
fi
if [ "$part" = b ]; then
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/r05_fetch -- python3 $root/tools/calibrate_fetch.py > $out/r05_fetch.log 2>&1)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/r05_write -- python3 $root/tools/calibrate_fetch.py > $out/r05_write.log 2>&1)
  python3 tools/traffic_from_pmc.py gpurun_out/r05_fetch gpurun_out/r05_write gpurun_out/r05_traffic.json
  tools/pmc_pass.sh r05_bf 4k 128 FETCH_SIZE
  tools/pmc_pass.sh r05_bw 4k 128 WRITE_SIZE
  python3 tools/traffic_batched.py gpurun_out/pmc_r05_bf gpurun_out/pmc_r05_bw gpurun_out/r05_traffic.json 128
  A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
  B="SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  tools/pmc_pass.sh r05_a 4k 128 $A
  tools/pmc_pass.sh r05_b 4k 128 $B
  for k in k_flow_iter_pc k_level0_polyexp_t k_level1_polyexp_t k_remap_step; do
    for p in a b; do python3 tools/pmc_top.py gpurun_out/pmc_r05_$p $k; done
  done > $out/r05_sq_counters.txt
  tail -30 $out/r05_sq_counters.txt
fi
if [ "$part" = c ]; then
  python3 tools/fuzz_fused.py ${FUZZ_CASES:-1000} 12 > $out/r05_fuzz.txt 2>&1 || true
  tail -10 $out/r05_fuzz.txt
  (for a in "1080p 96 bgr" "1080p 96 bgr prefetch" "1080p 96 bgr prefetch device" "1080p 96 bgr prefetch device batch=8" "1080p 96 bgr prefetch device batch=16" "4k 48 bgr" "4k 48 bgr prefetch" "4k 48 bgr prefetch device" "4k 48 bgr prefetch device batch=2" "4k 48 bgr prefetch device batch=4" "4k 48 bgr exact prefetch device batch=2"; do python3 tools/bench_host_path.py $a reps=6; done) > $out/r05_host_path.txt 2>&1
  cat $out/r05_host_path.txt
fi
