cd /root/repo; mkdir -p gpurun_out/r05m; O=gpurun_out/r05m
(timeout 500 python tools/ab_libs.py "C3 C2 C4" gpurun_ab/lib_r04.so gpurun_ab/lib_nocap.so gpurun_ab/lib_cap96.so gpurun_ab/lib_this_tree.so --frames 30 --rounds 3 2>&1 | grep -v amdgpu.ids) > $O/ab_sgpr.txt
for L in gpurun_ab/lib_nocap.so gpurun_ab/lib_cap96.so gpurun_ab/lib_this_tree.so; do echo "== $L"; GSPLAT_HIP_LIB=$L python tools/project_floor.py C3 2>&1 | grep k_project; done > $O/project_floor_sgpr.txt
cat $O/ab_sgpr.txt $O/project_floor_sgpr.txt
