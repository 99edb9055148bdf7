#!/bin/bash
# rocprofv3 kernel table of ONE rank's frame of an N-rank strip-sharded run, on one GPU.  usage: tools/rank_prof.sh <tag> <cfg> <N:r>
TAG=$1; CFG=$2; NR=$3
OUT=/root/repo/gpurun_out/$TAG/rank_${CFG}_${NR/:/of}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python /root/repo/tools/strip_scaling.py $CFG 30 $NR > $OUT/log.txt 2>&1
cd /root/repo
STATS=$(ls $OUT/prof/*/*_kernel_stats.csv | head -1)
cp $STATS $OUT/kernel_stats.csv
rm -rf $OUT/prof
tail -1 $OUT/log.txt
python tools/kstats.py $OUT/kernel_stats.csv 33 | tee $OUT/kstats.txt
