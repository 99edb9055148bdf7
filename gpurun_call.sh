cd /root/repo; mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
(timeout 1100 python -m pytest tests -q -m gpu 2>&1 | tail -25) > $O/pytest_gpu.txt
for L in gpurun_ab/lib_listv1.so gpurun_ab/lib_this_tree.so; do echo "== $L"; GSPLAT_HIP_LIB=$L python tools/project_floor.py C3 2>&1 | grep k_project; done > $O/project_floor.txt
(timeout 400 python tools/ab_libs.py "C3 C2 C4" gpurun_ab/lib_r04.so gpurun_ab/lib_listv1.so gpurun_ab/lib_this_tree.so --frames 30 --rounds 2 2>&1 | grep -v amdgpu.ids) > $O/ab.txt
(for E in GSPLAT_VIS_FRONT_R04=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E python tools/strip_scaling.py C3 20 1:0; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
tail -8 $O/pytest_gpu.txt; cat $O/project_floor.txt $O/ab.txt $O/rank.txt
