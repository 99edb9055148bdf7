#!/bin/bash
# PMC passes over the depth sort alone (tools/sort_ab.py, one library): each counter set in its own rocprofv3 run, kernel-trace
# only.  usage: tools/sort_pmc.sh <tag> <cfg> <lib.so>   -> gpurun_out/<tag>/pmc.txt (per kernel, mean per dispatch)
TAG=$1; CFG=$2; LIB=$3
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("FETCH_SIZE" "WRITE_SIZE" \
      "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
      "TCC_HIT_sum TCC_MISS_sum")
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python /root/repo/tools/sort_ab.py "$CFG" /root/repo/$LIB --rounds 1 --sorts 5 > $OUT/p$i.log 2>&1
done
cd /root/repo
python tools/pmcstats.py $OUT | grep -A1 -E "^k_radix|^k_depth_key" > $OUT/pmc.txt
rm -rf $OUT/p[0-9]
echo "== $TAG ($CFG, $LIB)"; cat $OUT/pmc.txt
