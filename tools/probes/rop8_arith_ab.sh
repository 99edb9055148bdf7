for l in rop8f0 rop8f1 rop8f2; do echo "== GS_ROP8_FUSED=${l#rop8f}"; GSPLAT_HIP_LIB=gpurun_ab/lib_$l.so timeout 300 python tools/rop8_ab.py "C3 C2" 2>&1 | grep " order" ; GSPLAT_HIP_LIB=gpurun_ab/lib_$l.so timeout 600 python -m pytest tests/test_gpu_crops.py tests/test_gpu_rop8_mode.py -q -m gpu -k "c3_garden_1080p or c2_truck_1080p or c3t_translucent_1080p or rop8" 2>&1 | tail -1; python - <<PY
import json
for c in ("C3","C2","C3T"):
    d=json.load(open("gpurun_out/crops_%s.json"%c)); print(c, json.dumps(d["rop8_mode"]))
PY
done
