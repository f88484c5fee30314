#!/bin/bash
# usage: tools/build_fb_variant.sh <name> [-DMACRO ...]: the units of the Farneback path (farneback.hip, fb_*.hip) are
# recompiled with the macros (in parallel), the other objects come from build_abl/_common (made on first use)
#   ->  build_abl/libtfhip_<name>.so   (TFHIP_LIBRARY=build_abl/libtfhip_<name>.so picks it: transflow_amd/_lib.py)
set -e
name=$1; shift
cd "$(dirname "$0")/../transflow_amd/csrc"
out=../../build_abl
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result -DTF_EXPERIMENT"
mkdir -p $out/_common $out/$name
COMMON="runtime remap remap_step flowops batch"
for f in $COMMON; do
  if [ ! -f $out/_common/$f.o ] || [ $f.hip -nt $out/_common/$f.o ] || [ common.h -nt $out/_common/$f.o ] || [ remap_common.h -nt $out/_common/$f.o ]; then $CC -c $f.hip -o $out/_common/$f.o & fi
done
FB="farneback fb_level_image fb_pyramid fb_matrices fb_iterate fb_exact fb_postprocess fb_stages"
for f in $FB; do $CC "$@" -c $f.hip -o $out/$name/$f.o & done
wait
objs=""; for f in $FB; do objs="$objs $out/$name/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libtfhip_$name.so $(for f in $COMMON; do echo $out/_common/$f.o; done) $objs -ldl
echo built $out/libtfhip_$name.so
