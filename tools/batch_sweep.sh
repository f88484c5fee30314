#!/bin/bash
# frames/s of the Farneback + remap step by batch size and lane count (GPU box): tools/batch_sweep.sh [workload] "<batches>"
wl=${1:-4k}
for b in ${2:-16 32 48 64 96}; do
    echo "== $wl batch $b"
    python3 tools/lanes_bench.py $wl $b $(( 384 / b + 2 )) 2>&1 | tail -2
done
