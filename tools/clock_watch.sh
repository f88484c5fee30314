#!/bin/bash
# usage (GPU box, repo root): tools/clock_watch.sh  ->  gpurun_out/clocks.txt
# Shader clock and package power sampled with rocm-smi while tools/kprof.py runs the 4K step in a loop: is the
# iteration kernel running at the part's peak clock or at a power-limited one?
out=gpurun_out/clocks.txt
python3 tools/kprof.py 4k 32 reps=1500 > gpurun_out/clock_kprof.txt 2>&1 &
pid=$!
sleep 14
for i in $(seq 1 12); do
  rocm-smi -c -P -u 2>/dev/null | grep -E "sclk|mclk|Power|GPU use" | tr '\n' ' '
  echo
  sleep 0.5
done > $out
wait $pid
echo "--- idle"; rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $out
cat $out
