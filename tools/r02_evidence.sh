#!/bin/bash
# usage (GPU box, repo root): tools/r02_evidence.sh  ->  gpurun_out/r02_*
# 1. rocprofv3 --kernel-trace --stats of the driver's bench command (with --no-extra: no 1080p side workloads, so every launch of a kernel belongs to the one workload); 2. the bench line printed under it;
# 3. FETCH_SIZE / WRITE_SIZE passes over tools/calibrate_fetch.py and the per-pixel traffic table.
set -e
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r02_stats -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $out/r02_bench_under_rocprof.json 2> $out/r02_bench_under_rocprof.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/r02_fetch -- python3 $root/tools/calibrate_fetch.py > $out/r02_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/r02_write -- python3 $root/tools/calibrate_fetch.py > $out/r02_write.log 2>&1
cd $root
python3 tools/traffic_from_pmc.py gpurun_out/r02_fetch gpurun_out/r02_write gpurun_out/r02_traffic.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_4k_default.json 2> gpurun_out/r02_bench_4k_default.err
tail -c 300 gpurun_out/r02_bench_4k_default.err
