cd /root/repo; mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
(timeout 1100 python -m pytest tests -q -m gpu 2>&1 | tail -25) > $O/pytest_gpu.txt
(for E in GSPLAT_VIS_FRONT_R04=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E python tools/strip_scaling.py C3 20 1:0; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
(python tools/project_floor.py C3 2>&1 | grep k_project) > $O/project_floor.txt
(timeout 300 python tools/strip_blend.py 2>&1 | grep -v amdgpu.ids) > $O/strip_blend.txt
(GSPLAT_HIP_LIB=gaussiansplats3d_amd/csrc/libgsplat_hip_blendprof.so timeout 200 python tools/blend_profile.py C5 132:164 2>&1 | grep -v amdgpu.ids) > $O/blend_profile_C5_strip.txt
bash tools/rank_prof.sh r05e C5 8:4 > $O/rank_C5_kstats.txt 2>&1
tail -8 $O/pytest_gpu.txt; cat $O/rank.txt $O/project_floor.txt $O/strip_blend.txt $O/blend_profile_C5_strip.txt; head -22 $O/rank_C5_kstats.txt
