#!/bin/bash
# A/B of (environment, library) combinations on one box, interleaved.
# usage: tools/ab_env.sh "<ENV1=..,ENV2=..|lib.so> <...>" [bench args...]   ("-" = no env / default library)
COMBOS=$1; shift
for round in 1 2; do
  for C in $COMBOS; do
    ENVS=${C%%|*}; L=${C##*|}
    [ "$L" = "$C" ] && L="-"
    CMD="env"
    [ "$ENVS" != "-" ] && CMD="env ${ENVS//,/ }"
    [ "$L" != "-" ] && CMD="$CMD GSPLAT_HIP_LIB=$(realpath $L)"
    $CMD timeout 300 python bench.py --no-cpu "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['frame']['stage_ms_isolated_frame']
print('%-44s %8.1f Msplats/s  %.4f ms lat %s | ' % ('$C'[-44:], d['value'], d['ms_per_step'], d.get('frame_latency_ms')) + ' '.join('%s=%.3f' % (k, v) for k, v in s.items()))"
  done
done
