cd /root/repo; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
(timeout 1100 python -m pytest tests -q -m gpu -x 2>&1) > $O/pytest_gpu_full.txt; tail -6 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
(python tools/project_floor.py C3 2>&1 | grep k_project; GSPLAT_NO_BLOCK_LIST=1 python tools/project_floor.py C3 2>&1 | grep k_project) > $O/project_floor.txt
(for E in GSPLAT_VIS_FRONT_R04=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E python tools/strip_scaling.py C3 20 1:0; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
(timeout 300 python tools/ab_libs.py "C3 C2 C4" gpurun_ab/lib_r04.so gpurun_ab/lib_this_tree.so --frames 30 --rounds 2 2>&1 | grep -v amdgpu.ids) > $O/ab.txt
bash tools/rank_prof.sh r05h C3 1:0 > $O/rank_C3_n1_kstats.txt 2>&1
bash tools/rank_prof.sh r05h C5 8:4 > $O/rank_C5_kstats.txt 2>&1
grep -n "passed\|failed\|Fatal\|fault\|Error" $O/pytest_gpu_full.txt | head; cat $O/project_floor.txt $O/rank.txt $O/ab.txt; head -16 $O/rank_C3_n1_kstats.txt; head -18 $O/rank_C5_kstats.txt
