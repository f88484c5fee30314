#!/bin/bash
# usage (GPU box, repo root): tools/r06_evidence.sh <part>  ->  gpurun_out/r06_*
#   part a: the bench line of the driver's command (4K, 128 pairs per pass, one lane) and rocprofv3 --kernel-trace --stats
#           of the same command (--no-extra --no-alone: every launch of a kernel belongs to the one workload, and with one
#           lane a launch's duration is the kernel's own); per-level kernel times (tools/kprof.py) at 128.
#   part b: FETCH_SIZE / WRITE_SIZE passes over the whole step at 128 pairs (tools/kprof.py 4k 128: 7 identical steps)
#           -> gpurun_out/r06_traffic_step.json (every kernel of the step), which bench.py reads from profiles/.
#   part c: the random-configuration fuzz (its exit code is recorded); the drop-in path.
# rocprofv3 runs the program itself after `--` (python3 <script>), counters in passes of their own (kernel trace only).
set -e
part=${1:-a}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
if [ "$part" = a ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r06_bench_4k_default.json 2> $out/r06_bench_4k_default.err
  tail -c 300 $out/r06_bench_4k_default.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r06_stats -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-alone > $out/r06_bench_4k_one_lane_under_rocprof.json 2> $out/r06_bench_under_rocprof.err)
  cp $(find $out/r06_stats -name "*kernel_stats.csv" | head -1) $out/r06_bench_4k_one_lane_kernel_stats.csv
  python3 tools/kprof.py 4k 128 > $out/r06_kprof_4k_batch128.txt 2>&1
  head -6 $out/r06_bench_4k_one_lane_kernel_stats.csv
fi
if [ "$part" = b ]; then
  tools/pmc_pass.sh r06_sf 4k 128 FETCH_SIZE
  tools/pmc_pass.sh r06_sw 4k 128 WRITE_SIZE
  python3 tools/traffic_step.py gpurun_out/pmc_r06_sf gpurun_out/pmc_r06_sw gpurun_out/r06_traffic_step.json 4k 128 7 | tee $out/r06_traffic_step.txt
fi
if [ "$part" = c ]; then
  rc=0
  python3 tools/fuzz_fused.py ${FUZZ_CASES:-1000} 12 > $out/r06_fuzz.txt 2>&1 || rc=$?
  echo "# tools/fuzz_fused.py exit code: $rc" >> $out/r06_fuzz.txt
  tail -10 $out/r06_fuzz.txt
  (for a in "1080p 96 bgr prefetch device batch=16" "4k 48 bgr" "4k 48 bgr prefetch" "4k 48 bgr prefetch device" "4k 48 bgr prefetch device batch=2" "4k 48 bgr prefetch device batch=2 frame" "4k 48 bgr prefetch device batch=4 frame"; do python3 tools/bench_host_path.py $a reps=6; done) > $out/r06_host_path.txt 2>&1
  cat $out/r06_host_path.txt
  exit $rc
fi
