#!/bin/bash
# One-box evidence set for a round tag: GPU tests, bench line, pipelined + serial rocprofv3 kernel stats, HBM PMC passes,
# the other single-GPU configs and the per-bin blend timeline.  usage: tools/evidence.sh <tag>
TAG=${1:-rXX}
cd /root/repo; mkdir -p gpurun_out/$TAG
(timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -5) > gpurun_out/$TAG/pytest_gpu.txt
(timeout 400 python bench.py 2>/dev/null | tail -1) > gpurun_out/$TAG/bench.json
bash tools/prof.sh ${TAG}_pipe --no-cull > gpurun_out/$TAG/kstats.txt 2>&1
GSPLAT_SERIAL=1 bash tools/prof.sh ${TAG}_serial --no-cull > gpurun_out/$TAG/serial_kstats.txt 2>&1
bash tools/pmc.sh $TAG hbm; bash tools/pmc.sh $TAG valu
cd /root/repo
python tools/pmc_traffic.py gpurun_out/pmc_$TAG gpurun_out/$TAG/pmc_traffic.json > gpurun_out/$TAG/pmc_traffic.txt 2>&1
for C in C2 C4 C5; do (timeout 300 python bench.py --config $C --no-cpu --no-cull --steps 30 2>/dev/null | tail -1) > gpurun_out/$TAG/bench_$C.json; done
if [ -f gpurun_ab/lib_bprof.so ]; then GSPLAT_HIP_LIB=$(realpath gpurun_ab/lib_bprof.so) timeout 120 python tools/blend_profile.py C3 > gpurun_out/$TAG/blend_bins.txt 2>&1; fi
cp gpurun_out/${TAG}_pipe/kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv
cp gpurun_out/${TAG}_pipe/bench_under_rocprof.json gpurun_out/$TAG/bench_under_rocprof.json
cp gpurun_out/${TAG}_serial/kernel_stats.csv gpurun_out/$TAG/serial_kernel_stats.csv
cat gpurun_out/$TAG/pytest_gpu.txt; head -c 600 gpurun_out/$TAG/bench.json; echo; head -20 gpurun_out/$TAG/serial_kstats.txt; cat gpurun_out/$TAG/pmc_traffic.txt | head -20
