#!/bin/bash
# rocprofv3 kernel table of the depth sort alone for one library: tools/sort_prof.sh <tag> <cfg> <lib.so> [sorts]
# (kernel-trace + stats only; per-kernel averages over `sorts` + 3 sorts)
TAG=$1; CFG=$2; LIB=$3; SORTS=${4:-20}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python /root/repo/tools/sort_ab.py "$CFG" /root/repo/$LIB --rounds 1 --sorts $SORTS > $OUT/prof.log 2>&1
cd /root/repo
STATS=$(ls $OUT/prof/*/*_kernel_stats.csv | head -1)
cp $STATS $OUT/kernel_stats.csv
python tools/kstats.py $OUT/kernel_stats.csv $((SORTS + 3)) | grep -E "radix|depth_key|total" > $OUT/kstats.txt
echo "== $TAG ($CFG, $LIB)"; cat $OUT/kstats.txt
rm -rf $OUT/prof
