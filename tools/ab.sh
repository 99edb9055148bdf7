#!/bin/bash
# A/B builds of libgsplat_hip.so on the same box, interleaved (box-to-box variance is ~10 %).
# usage: tools/ab.sh "<libA.so> <libB.so> ..." [bench args...]
LIBS=$1; shift
for round in 1 2; do
  for L in $LIBS; do
    GSPLAT_HIP_LIB=$(realpath $L) timeout 300 python bench.py --no-cpu "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['frame']['stage_ms_isolated_frame']
print('%-22s %8.1f Msplats/s  %.4f ms lat %.3f | ' % ('$L'.split('/')[-1], d['value'], d['ms_per_step'], d['frame_latency_ms']) + ' '.join('%s=%.3f' % (k, v) for k, v in s.items()) + ' | k_project %.4f ms frac %.3f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
  done
done
