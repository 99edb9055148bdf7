#!/bin/bash
mkdir -p gpurun_out/r05p
cp gaussiansplats3d_amd/csrc/libgsplat_hip.so gpurun_ab/lib_this_tree.so
(timeout 300 python tools/ab_libs.py "C3S C2 C4 C3" gpurun_ab/lib_r04.so gpurun_ab/lib_this_tree.so --frames 30 --rounds 2 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05p/ab.txt
(timeout 300 python -m pytest tests/test_gpu_render.py tests/test_gpu_depth.py -q -m gpu 2>&1 | tail -3) > gpurun_out/r05p/pytest_subset.txt
(timeout 420 python tests/tools/soak.py 80 51000 90000 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05p/soak_destination.txt
cat gpurun_out/r05p/ab.txt | cut -c1-170; cat gpurun_out/r05p/pytest_subset.txt; grep -c "^ok" gpurun_out/r05p/soak_destination.txt; grep -c destination gpurun_out/r05p/soak_destination.txt; grep "^FAIL\|^soak" gpurun_out/r05p/soak_destination.txt | head
