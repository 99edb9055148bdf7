cd /root/repo; mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
(timeout 300 python -m pytest tests/test_gpu_vis_cull.py tests/test_gpu_deep.py tests/test_gpu_soak.py -q -m gpu 2>&1 | tail -5) > $O/pytest_gpu.txt
(timeout 300 python tools/strip_blend.py 2>&1 | grep -v amdgpu.ids) > $O/strip_blend.txt
(GSPLAT_HIP_LIB=gaussiansplats3d_amd/csrc/libgsplat_hip_blendprof.so timeout 200 python tools/blend_profile.py C5 132:164 2>&1 | grep -v amdgpu.ids | head -40) > $O/blend_profile_C5_strip.txt
(GSPLAT_HIP_LIB=gaussiansplats3d_amd/csrc/libgsplat_hip_blendprof.so timeout 200 python tools/blend_profile.py C5 2>&1 | grep -v amdgpu.ids | head -40) > $O/blend_profile_C5_full.txt
(for E in GSPLAT_VIS_FRONT_R04=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E python tools/strip_scaling.py C3 20 1:0; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
bash tools/rank_prof.sh r05g C3 1:0 > $O/rank_C3_n1_kstats.txt 2>&1
cat $O/pytest_gpu.txt $O/strip_blend.txt $O/blend_profile_C5_strip.txt $O/blend_profile_C5_full.txt $O/rank.txt; head -12 $O/rank_C3_n1_kstats.txt
