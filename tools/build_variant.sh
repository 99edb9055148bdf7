#!/bin/bash
# Build a variant of libgsplat_hip.so with extra compiler flags into gpurun_ab/lib_<name>.so (A/B experiments).
# usage: tools/build_variant.sh <name> [extra hipcc flags...]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${GS_VARIANT_SRC:-$ROOT/gaussiansplats3d_amd/csrc}     # (GS_VARIANT_SRC: another source tree, e.g. a checkout of an older commit)
OUT=$ROOT/gpurun_ab; OBJ=/tmp/gsvar_$NAME
mkdir -p $OUT $OBJ
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*"
pids=()
for f in context selftest sorter mesh project tile_bin tile_blend tree assets group; do
  extra=""; [ $f = sorter -o $f = project -o $f = tree -o $f = assets ] && extra="-ffp-contract=off"
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c $SRC/$f.hip -o $OBJ/$f.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJ/*.o -ldl -o $OUT/lib_$NAME.so
echo built $OUT/lib_$NAME.so
