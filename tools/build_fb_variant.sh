#!/bin/bash
# usage: tools/build_fb_variant.sh <name> [-DMACRO ...]: only farneback.hip is recompiled, the other objects
# come from build_abl/base (made on first use)  ->  build_abl/libtfhip_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../transflow_amd/csrc"
out=../../build_abl
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result -DTF_EXPERIMENT"
mkdir -p $out/_common $out/$name
for f in runtime remap flowops batch; do
  if [ ! -f $out/_common/$f.o ] || [ $f.hip -nt $out/_common/$f.o ] || [ common.h -nt $out/_common/$f.o ]; then $CC -c $f.hip -o $out/_common/$f.o; fi
done
$CC "$@" -c farneback.hip -o $out/$name/farneback.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libtfhip_$name.so $out/_common/runtime.o $out/_common/remap.o $out/_common/flowops.o $out/_common/batch.o $out/$name/farneback.o -ldl
echo built $out/libtfhip_$name.so
