#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only, never combined with other trace domains) over a
# short bench; results under gpurun_out/pmc_<tag>/.  usage: tools/pmc.sh <tag> [hbm|all] [bench args...]
TAG=${1:-x}; MODE=${2:-all}; shift 2
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("FETCH_SIZE" "WRITE_SIZE")
if [ "$MODE" = "all" ]; then
  SETS+=("TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS")
fi
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/p$i.log 2>&1
done
