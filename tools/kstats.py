"""Print a compact per-kernel table from a rocprofv3 kernel_stats.csv (per-frame average = total / frames); with a third
argument also write it as JSON ({"config", "frames", "kernels": {name: {us_per_frame, calls, avg_us}}}) for bench.py's
roofline_sort object.  usage: python tools/kstats.py kernel_stats.csv <frames> [out.json [config]]"""
import csv, json, sys, re
path, frames = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = 0.0
doc = {"config": sys.argv[4] if len(sys.argv) > 4 else "C3", "frames": frames, "source": path, "kernels": {}}
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    name = re.sub(r"HIP_vector_type<([a-z ]+), (\d)u>", r"\1\2", name)
    per_frame = float(r["TotalDurationNs"]) / frames / 1e3
    if per_frame < 0.5: continue
    tot += per_frame
    doc["kernels"][name] = {"us_per_frame": round(per_frame, 2), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
    print(f"{per_frame:9.1f} us/frame  calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:8.1f} us  {name[:110]}")
print(f"{tot:9.1f} us/frame total")
if len(sys.argv) > 3:
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
