#!/bin/bash
# rocprofv3 kernel-trace + stats over the headline frames of bench.py (--only-headline: probe + warm-up + timed frames and
# nothing else, so every frame kernel runs once per frame); summary printed and left under gpurun_out/<tag>/.
# usage: tools/prof.sh <tag> [bench args...]    (env such as GSPLAT_SERIAL=1 is passed through)
TAG=${1:-x}; shift
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python /root/repo/bench.py --steps 30 --warmup 5 --only-headline "$@" > $OUT/prof.log 2>&1
grep '"metric"' $OUT/prof.log > $OUT/bench_under_rocprof.json
cd /root/repo
STATS=$(ls $OUT/prof/*/*_kernel_stats.csv | head -1)
cp $STATS $OUT/kernel_stats.csv
FRAMES=$(python -c "import json;print(json.load(open('$OUT/bench_under_rocprof.json'))['frames_drawn_before_timing_ended'])")
CFG=$(python -c "import json;print(json.load(open('$OUT/bench_under_rocprof.json'))['config']['workload'].split(':')[0])")
python tools/kstats.py $OUT/kernel_stats.csv $FRAMES $OUT/kstats.json $CFG
