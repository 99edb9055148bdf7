cd /root/repo; mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
(timeout 600 python -m pytest tests/test_gpu_vis_cull.py tests/test_gpu_deep.py tests/test_gpu_soak.py tests/test_gpu_bench_ranks.py tests/test_gpu_sort.py -q -m gpu -x 2>&1) > $O/pytest_gpu_full.txt; tail -6 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
(for E in GSPLAT_VIS_FRONT_R04=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; env $E python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
bash tools/rank_prof.sh r05k C3 8:4 > $O/rank_C3_kstats.txt 2>&1
grep -n "passed\|failed\|Fatal\|fault\|Error" $O/pytest_gpu_full.txt | head; cat $O/rank.txt; head -8 $O/rank_C3_kstats.txt
