#!/bin/bash
# Measurement aid: build tools/libvgpmp_<name>.so from the sources of a git revision (default HEAD) with extra compiler
# flags, for interleaved comparisons by tools/ab_multi.sh.   tools/build_variant.sh head HEAD   |   ... noxcd HEAD -DVG_XCD_PATHS=0
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:-HEAD}; shift; shift || true
src=/tmp/vgpmp_variant_$name; rm -rf $src; mkdir -p $src/obj
if [ "$rev" = WORK ]; then mkdir -p $src/vgpmp_amd; cp -r vgpmp_amd/csrc $src/vgpmp_amd/; cp -r include $src/; else git archive $rev vgpmp_amd/csrc include | tar -x -C $src; fi
for f in fk_sdf gp_path mesh_sdf deriv_kernels plan inducing comm capi; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -fno-vectorize "$@" -I$src/include -I$src/vgpmp_amd/csrc -c $src/vgpmp_amd/csrc/$f.hip -o $src/obj/$f.o &
done
wait
hipcc --offload-arch=gfx950 -fPIC -shared $src/obj/*.o -ldl -o tools/libvgpmp_$name.so
echo tools/libvgpmp_$name.so
