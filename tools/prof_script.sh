#!/bin/bash
# rocprofv3 kernel-trace + stats over an arbitrary python script; usage: tools/prof_script.sh <tag> <frames> <script.py> [args]
TAG=$1; FRAMES=$2; shift 2
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python "$@" > $OUT/log.txt 2>&1
cd /root/repo
python tools/kstats.py $(ls $OUT/prof/*/*_kernel_stats.csv | head -1) $FRAMES
