#!/bin/bash
# A/B builds of libgsplat_hip.so on the same box, interleaved (box-to-box variance is ~10 %).
# usage: tools/ab.sh "<libA.so> <libB.so> ..." [bench args...]
LIBS=$1; shift
for round in 1 2; do
  for L in $LIBS; do
    GSPLAT_HIP_LIB=$(realpath $L) timeout 300 python bench.py --no-cpu "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline']['stages']
print('%-28s %8.1f Msplats/s  %.4f ms | ' % ('$L'.split('/')[-1], d['value'], d['ms_per_step']) + ' '.join('%s=%.3f' % (k, v['ms']) for k, v in s.items()))"
  done
done
