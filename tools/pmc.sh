#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only) over a short bench; results under gpurun_out/pmc_<tag>/
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_$TAG/p$i -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu "$@" > /root/repo/gpurun_out/pmc_$TAG/p$i.log 2>&1
done
