cd /root/repo; mkdir -p gpurun_out/r05i; O=gpurun_out/r05i
(timeout 1100 python -m pytest tests -q -m gpu -x 2>&1) > $O/pytest_gpu_full.txt; tail -6 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
(for E in GSPLAT_ONE_DRAW_SET=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
(python tools/strip_scaling.py C3 20 1:0; GSPLAT_VIS_FRONT_R04=1 python tools/strip_scaling.py C3 20 1:0) 2>&1 | grep -v amdgpu.ids >> $O/rank.txt
bash tools/rank_prof.sh r05i C3 1:0 > $O/rank_C3_n1_kstats.txt 2>&1
(GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20; GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15) 2>&1 | grep -v amdgpu.ids > $O/strip_scaling_streams.txt
grep -n "passed\|failed\|Fatal\|fault\|Error" $O/pytest_gpu_full.txt | head; cat $O/rank.txt; head -8 $O/rank_C3_n1_kstats.txt; cat $O/strip_scaling_streams.txt
