cd /root/repo; mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
(timeout 900 python -m pytest tests/test_gpu_vis_cull.py tests/test_gpu_render.py tests/test_gpu_deep.py tests/test_gpu_bench_ranks.py tests/test_gpu_soak.py tests/test_gpu_advice_r04.py tests/test_gpu_depth.py -q -m gpu 2>&1 | tail -15) > $O/pytest_gpu.txt
(for E in GSPLAT_VIS_FRONT_R04=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E python tools/strip_scaling.py C3 20 1:0; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
(timeout 300 python tools/ab_libs.py "C3 C2 C4" gpurun_ab/lib_r04.so gpurun_ab/lib_this_tree.so --frames 30 --rounds 2 2>&1 | grep -v amdgpu.ids) > $O/ab.txt
bash tools/rank_prof.sh r05d C5 8:4 > $O/rank_C5_kstats.txt 2>&1
bash tools/rank_prof.sh r05d C3 8:4 > $O/rank_C3_kstats.txt 2>&1
bash tools/rank_prof.sh r05d C3 1:0 > $O/rank_C3_n1_kstats.txt 2>&1
tail -6 $O/pytest_gpu.txt; cat $O/rank.txt $O/ab.txt; cat $O/rank_C5_kstats.txt; cat $O/rank_C3_kstats.txt; cat $O/rank_C3_n1_kstats.txt
