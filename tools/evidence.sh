#!/bin/bash
# One-box evidence set for a round tag: GPU tests, the bench line (and the lines of the other configs), rocprofv3 kernel
# stats of the headline frames, the HBM and VALU counter passes, the per-rank cost of strip-sharded frames and a 2-rank
# dry run.  Everything lands under gpurun_out/<tag>/; copy what should be judged into profiles/.   usage: tools/evidence.sh <tag>
TAG=${1:-rXX}
cd /root/repo; mkdir -p gpurun_out/$TAG
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6) > gpurun_out/$TAG/pytest_gpu.txt
bash tools/pmc.sh $TAG hbm > /dev/null 2>&1
bash tools/pmc.sh $TAG valu > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_$TAG gpurun_out/$TAG/pmc_traffic.json > gpurun_out/$TAG/pmc_traffic.txt 2>&1
python tools/pmc_valu.py gpurun_out/pmc_$TAG gpurun_out/$TAG/pmc_valu.json > gpurun_out/$TAG/pmc_valu.txt 2>&1
# the bench reads the counter summaries from profiles/: stage this run's so that the line it prints quotes them
cp gpurun_out/$TAG/pmc_traffic.json profiles/${TAG}_pmc_traffic.json; cp gpurun_out/$TAG/pmc_valu.json profiles/${TAG}_pmc_valu.json
(timeout 500 python bench.py 2>/dev/null | tail -1) > gpurun_out/$TAG/bench.json
bash tools/prof.sh ${TAG}_kstats > gpurun_out/$TAG/kstats.txt 2>&1
cp gpurun_out/${TAG}_kstats/kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv
cp gpurun_out/${TAG}_kstats/bench_under_rocprof.json gpurun_out/$TAG/bench_under_rocprof.json
for C in C2 C4 C5; do (timeout 300 python bench.py --config $C --no-cpu --no-cull --steps 30 2>/dev/null | tail -1) > gpurun_out/$TAG/bench_$C.json; done
(timeout 200 python bench.py --config C1 --steps 30 2>/dev/null | tail -1) > gpurun_out/$TAG/bench_C1.json
(python tools/strip_scaling.py C3 20; python tools/strip_scaling.py C5 15) 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/strip_scaling.txt
(timeout 300 python bench.py --gpus 2 --steps 20 --no-cpu --no-cull 2>/dev/null | tail -1) > gpurun_out/$TAG/bench_2ranks_dry_run.json
cp gpurun_out/crops_C*.json gpurun_out/$TAG/ 2>/dev/null
cat gpurun_out/$TAG/pytest_gpu.txt; head -c 700 gpurun_out/$TAG/bench.json; echo; head -14 gpurun_out/$TAG/kstats.txt; cat gpurun_out/$TAG/pmc_traffic.txt | tail -3; cat gpurun_out/$TAG/strip_scaling.txt
