#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only, never combined with other trace domains) over the
# headline frames of bench.py (--only-headline: probe + warm-up + timed frames, nothing else); results under
# gpurun_out/pmc_<tag>/.  usage: tools/pmc.sh <tag> [hbm|valu|all] [bench args...]
#   hbm   FETCH_SIZE, WRITE_SIZE                      -> tools/pmc_traffic.py
#   valu  SQ instruction / busy counters + GRBM clock -> tools/pmc_valu.py
TAG=${1:-x}; MODE=${2:-all}; shift 2
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=()
if [ "$MODE" = "hbm" -o "$MODE" = "all" ]; then SETS+=("FETCH_SIZE" "WRITE_SIZE"); fi
if [ "$MODE" = "valu" -o "$MODE" = "all" ]; then
  SETS+=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE")
fi
if [ "$MODE" = "all" ]; then
  SETS+=("TCC_HIT_sum TCC_MISS_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS")
fi
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/${MODE}_p$i -- python /root/repo/bench.py --steps 3 --warmup 1 --only-headline "$@" > $OUT/${MODE}_p$i.log 2>&1
done
