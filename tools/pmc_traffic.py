"""Per-kernel HBM traffic from rocprofv3 PMC passes (tools/pmc.sh): mean FETCH_SIZE / WRITE_SIZE per dispatch.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-byte read requests as 64 bytes, so a wide
coalesced stream reads 2x what it reports; WRITE_SIZE matched the known write volume of k_project 1:1 (calibration
on this repo's own access pattern: 76.7 MB reported vs 76 MB of records + rects written, profiles/README.md).
hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024.

usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag> profiles/<tag>_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for name, cs in sorted(acc.items()):
    if name.startswith(("__amd", "at::")) or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
    write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    kernels[name] = {"FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1), "dispatches": len(cs["FETCH_SIZE"]),
                     "hbm_bytes_per_launch": int((2.0 * fetch + write) * 1024)}
json.dump({"source": root, "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B "
           "requests at 64 B; WRITE_SIZE calibrated 1:1)", "kernels": kernels}, open(out, "w"), indent=1)
for k, v in kernels.items():
    print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  {k[:100]}")
