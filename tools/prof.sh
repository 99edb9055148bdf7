#!/bin/bash
# rocprofv3 kernel-trace + stats over a short bench; summary printed and left under gpurun_out/<tag>/.
# usage: tools/prof.sh <tag> [bench args...]    (env such as GSPLAT_SERIAL=1 is passed through)
TAG=${1:-x}; shift
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu "$@" > $OUT/prof.log 2>&1
grep '"metric"' $OUT/prof.log > $OUT/bench_under_rocprof.json
cd /root/repo
STATS=$(ls $OUT/prof/*/*_kernel_stats.csv | head -1)
cp $STATS $OUT/kernel_stats.csv
python tools/kstats.py $OUT/kernel_stats.csv 46
