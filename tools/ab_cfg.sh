#!/bin/bash
# A/B of library builds over several bench configs on one box: tools/ab_cfg.sh "<libs>" "<configs>" [steps]
LIBS=$1; CFGS=$2; STEPS=${3:-30}
for C in $CFGS; do for L in $LIBS; do
GSPLAT_HIP_LIB=$(realpath $L) timeout 300 python bench.py --no-cpu --no-cull --config $C --steps $STEPS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['frame']['stage_ms_isolated_frame']
print('$C %-14s %8.1f Msplats/s  %.4f ms | ' % ('$L'.split('/')[-1], d['value'], d['ms_per_step']) + ' '.join('%s=%.3f' % (k, v) for k, v in s.items()) + ' E=%d' % d['frame']['list_entries'])"
done; done
