#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04b
L=gpurun_ab
timeout 900 python tools/sort_ab.py "C3" $L/lib_r03.so $L/lib_t512_b512.so $L/lib_t512_b512_occ5.so $L/lib_t512_b384.so $L/lib_t512_b256.so $L/lib_t1024_b256.so $L/lib_t1024_b384.so $L/lib_t1024_b512.so $L/lib_t2048.so $L/lib_t512_b512_key1.so $L/lib_t512_b512_h512.so $L/lib_t512_b512_hg8.so $L/lib_t512_b512_noxcd.so --check --sorts 30 --rounds 2 2>&1 | tee gpurun_out/r04b/sort_ab_C3.txt
timeout 900 python tools/sort_ab.py "C4" $L/lib_t1024.so $L/lib_t512_b512.so $L/lib_t512_b256.so $L/lib_t1024_b256.so $L/lib_t1024_b512.so --check --sorts 20 --rounds 2 2>&1 | tee gpurun_out/r04b/sort_ab_C4.txt
for v in t512_b512 t512_b256 t1024_b256 t512_b512_noxcd; do timeout 300 tools/sort_prof.sh r04b_k_$v C3 $L/lib_$v.so 20; done 2>&1 | tee gpurun_out/r04b/kstats_C3.txt
GSPLAT_HIP_LIB=$PWD/$L/lib_t512_b512_prof.so timeout 300 python tools/radix_profile.py C3 2>&1 | tee gpurun_out/r04b/radix_profile_C3.txt
timeout 900 python tools/ab_libs.py "C3" $L/lib_r03.so $L/lib_t512_b512.so $L/lib_t1024.so $L/lib_t1024_b256.so --frames 40 --rounds 2 2>&1 | tee gpurun_out/r04b/ab_frames_C3.txt
