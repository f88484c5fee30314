#!/bin/bash
# usage: tools/build_variant.sh <name> [-DMACRO ...]  ->  build_abl/libtfhip_<name>.so (load with TFHIP_LIBRARY=...)
set -e
name=$1; shift
cd "$(dirname "$0")/../transflow_amd/csrc"
out=../../build_abl
mkdir -p $out/$name
for f in runtime remap farneback flowops batch; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result -DTF_EXPERIMENT "$@" -c $f.hip -o $out/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libtfhip_$name.so $out/$name/*.o -ldl
echo built $out/libtfhip_$name.so
