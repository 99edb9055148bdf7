#!/bin/bash
# The asset readers and the host octree builder (host code) under AddressSanitizer: builds the library with the HOST side instrumented
# (-fsanitize=address -fno-gpu-sanitize) into gpurun_ab/lib_asan.so and runs tests/tools/soak_assets.py (oracle comparison + damaged
# files) on it with the sanitizer's runtime preloaded into python.  No GPU needed.   usage: tests/tools/asan_assets.sh [asset iterations] [seed] [tree iterations] [seed]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
OBJ=/tmp/gsvar_asan; SRC=gaussiansplats3d_amd/csrc; mkdir -p $OBJ gpurun_ab
FL="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fsanitize=address -fno-gpu-sanitize -shared-libasan"
pids=()
for f in context selftest sorter mesh project tile_bin tile_blend tree assets group; do
  extra=""; [ $f = sorter -o $f = project -o $f = tree -o $f = assets ] && extra="-ffp-contract=off"
  ( /opt/rocm/bin/hipcc $FL $extra -c $SRC/$f.hip -o $OBJ/$f.o ) & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -shared-libasan $OBJ/*.o -ldl -o gpurun_ab/lib_asan.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
echo "__asan_report entry points the library imports (it is instrumented): $(nm -D gpurun_ab/lib_asan.so | grep -c __asan_report)"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 GSPLAT_HIP_LIB=$ROOT/gpurun_ab/lib_asan.so python tests/tools/soak_assets.py ${1:-300} ${2:-100} | grep -v "^ok"
# ... and the host octree builder (the other piece of host code that walks caller data)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 GSPLAT_HIP_LIB=$ROOT/gpurun_ab/lib_asan.so python tests/tools/soak_tree_host.py ${3:-120} ${4:-500} | grep -v "^ok"
