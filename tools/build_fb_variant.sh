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
pids=""
for f in $COMMON; do
  if [ ! -f $out/_common/$f.o ] || [ $f.hip -nt $out/_common/$f.o ] || [ common.h -nt $out/_common/$f.o ] || [ remap_common.h -nt $out/_common/$f.o ]; then
    rm -f $out/_common/$f.o; $CC -c $f.hip -o $out/_common/$f.o & pids="$pids $!"
  fi
done
FB="farneback fb_level_image fb_pyramid fb_matrices fb_iterate fb_exact fb_postprocess fb_stages"
# a unit that fails to compile must fail the build: old objects go first, and every job is waited for by pid
# (a bare `wait` returns 0 whatever the jobs did)
# (fb_iterate.hip is built without the SLP vectoriser, as transflow_amd/csrc/Makefile builds it)
for f in $FB; do extra=""; [ $f = fb_iterate ] && extra="-fno-slp-vectorize"; rm -f $out/$name/$f.o; $CC $extra "$@" -c $f.hip -o $out/$name/$f.o & pids="$pids $!"; done
for p in $pids; do wait $p || { echo "build_fb_variant: a compile job failed" >&2; exit 1; }; done
objs=""; for f in $FB; do objs="$objs $out/$name/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libtfhip_$name.so $(for f in $COMMON; do echo $out/_common/$f.o; done) $objs -ldl
echo built $out/libtfhip_$name.so
