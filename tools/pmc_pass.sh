#!/bin/bash
# usage (GPU box, repo root): tools/pmc_pass.sh <tag> <workload> <batch> <counter> [<counter> ...]
# One rocprofv3 --pmc pass (kernel trace only) over tools/kprof.py; CSVs land in gpurun_out/pmc_<tag>/.
set -e
tag=$1; wl=$2; batch=$3; shift 3
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$root/gpurun_out/pmc_$tag" -- python3 "$root/tools/kprof.py" "$wl" "$batch" > "$root/gpurun_out/pmc_$tag.log" 2>&1
