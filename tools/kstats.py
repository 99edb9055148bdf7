"""Print a compact per-kernel table from a rocprofv3 kernel_stats.csv (per-frame average = total / frames)."""
import csv, sys, re
path, frames = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = 0.0
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    name = re.sub(r"HIP_vector_type<([a-z ]+), (\d)u>", r"\1\2", name)
    per_frame = float(r["TotalDurationNs"]) / frames / 1e3
    if per_frame < 0.5: continue
    tot += per_frame
    print(f"{per_frame:9.1f} us/frame  calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:8.1f} us  {name[:110]}")
print(f"{tot:9.1f} us/frame total")
