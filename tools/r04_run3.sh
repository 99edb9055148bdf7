#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04d
L=gpurun_ab
timeout 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_frustum_cull.py tests/test_gpu_vis_cull.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04d/pytest_sort.txt
GSPLAT_NO_LDS_ATOMIC_RANK=1 timeout 600 python -m pytest tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -5 | sed 's/^/ballot: /' | tee -a gpurun_out/r04d/pytest_sort.txt
timeout 900 python tools/sort_ab.py "C3" $L/lib_r03.so $L/lib_t512_b512_occ5.so $L/lib_c3.so $L/lib_c2.so $L/lib_c3_t1024.so $L/lib_c3_occ6.so $L/lib_c3_h512.so --check --sorts 30 --rounds 2 2>&1 | tee gpurun_out/r04d/sort_ab_C3.txt
GSPLAT_NO_SORT_CHUNK=1 timeout 300 python tools/sort_ab.py "C3" $L/lib_c3.so --sorts 30 --rounds 2 2>&1 | sed 's/^/nochunk /' | tee -a gpurun_out/r04d/sort_ab_C3.txt
timeout 900 python tools/sort_ab.py "C4 C2" $L/lib_r03.so $L/lib_t512_b512_occ5.so $L/lib_c3.so $L/lib_c2.so --check --sorts 20 --rounds 2 2>&1 | tee gpurun_out/r04d/sort_ab_C4.txt
for v in c3 c2; do timeout 300 tools/sort_prof.sh r04d_k_$v C3 $L/lib_$v.so 20; done 2>&1 | tee gpurun_out/r04d/kstats_C3.txt
timeout 300 tools/sort_prof.sh r04d_k4_c3 C4 $L/lib_c3.so 10 2>&1 | tee gpurun_out/r04d/kstats_C4.txt
bash tools/sort_pmc.sh r04d_pmc_c3 C3 $L/lib_c3.so 2>&1 | tee gpurun_out/r04d/pmc_c3.txt
timeout 600 python tools/ab_libs.py "C3" $L/lib_r03.so $L/lib_t512_b512_occ5.so $L/lib_c3.so --frames 40 --rounds 2 2>&1 | tee gpurun_out/r04d/ab_frames_C3.txt
