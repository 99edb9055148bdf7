#!/bin/bash
# The soaks again under the environment switches that select the ALTERNATIVE code paths (A/B switches, fallbacks): one summary
# line per (switch, soak).  (Not GSPLAT_POOL_SLOTS: with a pool of a few slots a dense frame runs out of chunk partials, the draw
# says so - GS_DRAW_POOL_EXHAUSTED - and its strips then differ from the full frame in the last bits by design.)   usage: tests/tools/soak_paths.sh > gpurun_out/soak_paths.txt
cd "$(dirname "$0")/../.."
run() {  # <env assignments or -> <soak script> <args...>
  local e=$1; shift
  local cmd="env"; [ "$e" != "-" ] && cmd="env $e"
  printf "%-44s %-18s " "$e" "$1"
  $cmd timeout 600 python tests/tools/"$@" 2>&1 | grep -E "^(FAIL|soak)" | tail -3 | tr '\n' ' '; echo
}
for E in - GSPLAT_NO_LDS_ATOMIC_RANK=1 GSPLAT_NO_SORT_PACK=1 GSPLAT_NO_SORT_CHUNK=1 GSPLAT_ONE_RECORD_SET=1 GSPLAT_SERIAL=1 \
         GSPLAT_NO_BLOCK_CULL=1 GSPLAT_NO_REORDER=1 GSPLAT_NO_DEEP=1 GSPLAT_NO_COARSE_VIS=1 GSPLAT_NO_BLEND_ORDER=1 GSPLAT_WIDE_ENTRY_KEYS=1 \
         GSPLAT_LIST_SHIFT=1 GSPLAT_LIST_SHIFT=5 GSPLAT_VIS_FRONT=stream GSPLAT_VIS_FRONT=compact GSPLAT_NO_BLOCK_LIST=1 GSPLAT_BLOCK_TEST_ALWAYS=1 \
         GSPLAT_BIN_FUSED=1 GSPLAT_BLEND_ORDER_STALE=1 GSPLAT_NO_LAZY_MASK=1 GSPLAT_NO_ASYNC_LIST_BINS=1 GSPLAT_ORDER_MOTION=0 GSPLAT_ORDER_MOTION=100 GSPLAT_NO_STAT_SHIFT=1 GSPLAT_DEEP_UNITS_LAST=1 GSPLAT_DEEP_ROW_MAJOR_ON_MOTION=1 GSPLAT_KEY_HIST_FUSED=1; do
  run "$E" soak.py 30 41000 120000
  run "$E" soak_stateful.py 3 42000 16 40000
done
for E in - GSPLAT_NO_LDS_ATOMIC_RANK=1 GSPLAT_NO_SORT_PACK=1 GSPLAT_NO_SORT_CHUNK=1 GSPLAT_KEY_HIST_FUSED=1; do
  run "$E" soak_sort.py 150 43000 300000
  run "$E" soak_sort.py sizes 4095,4096,4097,12287,12288,12289,24577,6291457,16777215,16777217
done
for E in - GSPLAT_TREE_NO_DEFER=1 GSPLAT_TREE_NO_FUSE=1 GSPLAT_TREE_HOST_BUILD=1 GSPLAT_NO_LDS_ATOMIC_RANK=1; do
  run "$E" soak_tree.py 40 44000 120000
done
