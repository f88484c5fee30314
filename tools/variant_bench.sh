#!/bin/bash
# usage (GPU box, repo root): tools/variant_bench.sh <workload> <batch> <lib> [<lib> ...]
# Per-kernel times (tools/kprof.py) of each build_abl/libtfhip_<lib>.so; "default" = the in-tree library.
wl=$1; batch=$2; shift 2
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = default ]; then
    python3 tools/kprof.py $wl $batch 2>&1 | head -14
  else
    TFHIP_LIBRARY=build_abl/libtfhip_$v.so python3 tools/kprof.py $wl $batch 2>&1 | head -14
  fi
done
