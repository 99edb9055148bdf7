cd /root/repo; mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
(timeout 60 tools/probes/lookback_probe.bin 5800000 20; timeout 60 tools/probes/lookback_probe.bin 1440000 20; timeout 60 tools/probes/lookback_probe.bin 270000 20; timeout 60 tools/probes/lookback_probe.bin 8388608 10) > $O/lookback_probe.txt 2>&1
(timeout 1100 python -m pytest tests -q -m gpu -x 2>&1) > $O/pytest_gpu_full.txt; tail -25 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
(python tools/project_floor.py C3 2>&1 | grep k_project) > $O/project_floor.txt
(for E in GSPLAT_VIS_FRONT_R04=1 GSPLAT_X=1; do echo "== $E"; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15 8:4; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 8:4; env $E python tools/strip_scaling.py C3 20 1:0; env $E GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20 1:0; done) 2>&1 | grep -v amdgpu.ids > $O/rank.txt
cat $O/lookback_probe.txt; grep -n "passed\|failed\|Fatal\|fault\|Error" $O/pytest_gpu_full.txt | head -20; cat $O/project_floor.txt $O/rank.txt
