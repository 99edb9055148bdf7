cd /root/repo; mkdir -p gpurun_out/r05o; O=gpurun_out/r05o
(timeout 900 python bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1) > $O/bench.json
tail -5 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05o/bench.json"))
for k in ("value","ms_per_step","median_ms_per_step","frame_latency_ms"): print(k, d.get(k))
print("config", {k:d["config"].get(k) for k in ("V_over_N","D_over_R")})
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("cull_on", json.dumps(d.get("cull_on"))[:900])
print("vis", json.dumps(d.get("visibility_cull_fused"))[:400])
print("fused", json.dumps(d.get("frustum_cull_fused"))[:300])
print("pipelined", json.dumps(d.get("pipelined"))[:500])
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
PY
