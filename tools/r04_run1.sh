#!/bin/bash
# round 4, GPU run 1: KATs on the new default build, then the sort A/B matrix, then kernel tables
cd /root/repo
mkdir -p gpurun_out/r04a
timeout 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_frustum_cull.py tests/test_gpu_vis_cull.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04a/pytest_sort.txt
cat gpurun_out/r04a/pytest_sort.txt
L=gpurun_ab
timeout 900 python tools/sort_ab.py "C3" gaussiansplats3d_amd/csrc/libgsplat_hip.so $L/lib_r03.so $L/lib_t1024.so $L/lib_t1024_h512.so $L/lib_t1024_noxcd.so $L/lib_t512_occ6.so $L/lib_t512_occ8.so $L/lib_t512_b512.so $L/lib_t1024_b512.so $L/lib_t1024_b2048.so $L/lib_t8192.so $L/lib_t1024_key8.so --check --sorts 30 --rounds 2 2>&1 | tee gpurun_out/r04a/sort_ab_C3.txt
timeout 900 python tools/sort_ab.py "C4" $L/lib_r03.so $L/lib_t1024.so $L/lib_t1024_h512.so $L/lib_t1024_noxcd.so $L/lib_t512_occ6.so $L/lib_t1024_b2048.so $L/lib_t8192.so --check --sorts 20 --rounds 2 2>&1 | tee gpurun_out/r04a/sort_ab_C4.txt
GSPLAT_NO_SORT_PACK=1 timeout 300 python tools/sort_ab.py "C3" $L/lib_t1024.so --sorts 30 --rounds 2 2>&1 | sed 's/^/nopack /' | tee gpurun_out/r04a/sort_ab_C3_nopack.txt
for v in r03 t1024 t512_occ6 t8192; do timeout 300 tools/sort_prof.sh r04a_k_$v C3 $L/lib_$v.so 20; done 2>&1 | tee gpurun_out/r04a/kstats_C3.txt
for v in r03 t1024; do timeout 300 tools/sort_prof.sh r04a_k4_$v C4 $L/lib_$v.so 10; done 2>&1 | tee gpurun_out/r04a/kstats_C4.txt
