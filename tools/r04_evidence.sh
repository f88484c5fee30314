#!/bin/bash
# usage (GPU box, repo root): tools/r04_evidence.sh <part>  ->  gpurun_out/r04_*
#   part a: the bench line of the driver's command (two lanes) and rocprofv3 --kernel-trace --stats of the same command
#           (--no-extra: every launch of a kernel belongs to the one workload); the same pair with --lanes 1 (one batch
#           in flight: a launch's duration is the kernel's own); per-level kernel times (tools/kprof.py, one lane).
#   part b: FETCH_SIZE / WRITE_SIZE passes over tools/calibrate_fetch.py -> the per-pixel traffic table; SQ counter passes
#           over tools/kprof.py 4k 32 -> counters of the iteration kernel and the frame-expansion kernels.
#   part c: the random-configuration fuzz; the drop-in path through host arrays (plain and prefetching); one lane
#           against two; the checking mode's kernels at 32 pairs.
# rocprofv3 runs the program itself after `--` (python3 <script>), counters in passes of their own.
set -e
part=${1:-a}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
if [ "$part" = a ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04_bench_4k_default.json 2> $out/r04_bench_4k_default.err
  tail -c 300 $out/r04_bench_4k_default.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r04_stats -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-alone > $out/r04_bench_4k_under_rocprof.json 2> $out/r04_bench_under_rocprof.err)
  cp $(find $out/r04_stats -name "*kernel_stats.csv" | head -1) $out/r04_bench_4k_kernel_stats.csv
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --lanes 1 --no-extra > $out/r04_bench_4k_one_lane.json 2> $out/r04_bench_4k_one_lane.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r04_stats1 -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --lanes 1 --no-extra --no-alone > $out/r04_bench_4k_one_lane_under_rocprof.json 2> $out/r04_bench_one_lane_under_rocprof.err)
  cp $(find $out/r04_stats1 -name "*kernel_stats.csv" | head -1) $out/r04_bench_4k_one_lane_kernel_stats.csv
  python3 tools/kprof.py 4k 32 > $out/r04_kprof_4k_batch32.txt 2>&1
  head -6 $out/r04_bench_4k_kernel_stats.csv $out/r04_bench_4k_one_lane_kernel_stats.csv
fi
if [ "$part" = b ]; then
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/r04_fetch -- python3 $root/tools/calibrate_fetch.py > $out/r04_fetch.log 2>&1)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/r04_write -- python3 $root/tools/calibrate_fetch.py > $out/r04_write.log 2>&1)
  python3 tools/traffic_from_pmc.py gpurun_out/r04_fetch gpurun_out/r04_write gpurun_out/r04_traffic.json
  tools/pmc_pass.sh r04_bf 4k 32 FETCH_SIZE
  tools/pmc_pass.sh r04_bw 4k 32 WRITE_SIZE
  python3 tools/traffic_batched.py gpurun_out/pmc_r04_bf gpurun_out/pmc_r04_bw gpurun_out/r04_traffic.json
  A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
  B="SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  tools/pmc_pass.sh r04_a 4k 32 $A
  tools/pmc_pass.sh r04_b 4k 32 $B
  for k in k_flow_iter_pc k_flow_carry_pc k_level0_polyexp_t k_level1_polyexp_t k_remap_step; do
    for p in a b; do python3 tools/pmc_top.py gpurun_out/pmc_r04_$p $k; done
  done > $out/r04_sq_counters.txt
  tail -30 $out/r04_sq_counters.txt
fi
if [ "$part" = c ]; then
  python3 tools/fuzz_fused.py ${FUZZ_CASES:-600} 11 > $out/r04_fuzz.txt 2>&1 || true
  tail -10 $out/r04_fuzz.txt
  (for a in "1080p 96 bgr" "1080p 96 bgr prefetch" "4k 72 bgr" "4k 72 bgr prefetch" "4k 72 bgr exact" "4k 72 bgr exact prefetch" "1080p 96 bgr exact prefetch"; do python3 tools/bench_host_path.py $a; done) > $out/r04_host_path.txt 2>&1
  cat $out/r04_host_path.txt
  (python3 tools/lanes_bench.py 4k 32 12; python3 tools/lanes_bench.py 4k 8 24; python3 tools/lanes_bench.py 1080p 64 24) > $out/r04_lanes.txt 2>&1
  cat $out/r04_lanes.txt
  python3 tools/kprof.py 4k 32 fb_exact_sums=1 reps=2 > $out/r04_kprof_4k_batch32_exact.txt 2>&1
  head -5 $out/r04_kprof_4k_batch32_exact.txt
  python3 tools/kprof.py 4k 1 fb_exact_sums=1 > $out/r04_kprof_4k_one_pair_exact.txt 2>&1
  python3 tools/kprof.py 4k 1 > $out/r04_kprof_4k_one_pair.txt 2>&1
  head -3 $out/r04_kprof_4k_one_pair_exact.txt $out/r04_kprof_4k_one_pair.txt
fi
