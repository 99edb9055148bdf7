"""Per-kernel and per-frame HBM traffic from rocprofv3 PMC passes (tools/pmc.sh <tag> hbm): mean FETCH_SIZE / WRITE_SIZE
per dispatch, and the sum over one frame's kernels.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-byte read requests as 64 bytes, so a wide
coalesced stream reads 2x what it reports; WRITE_SIZE matched the known write volume of k_project 1:1 (calibration
on this repo's own access pattern: 76.7 MB reported vs 76 MB of records + rects written, profiles/README.md).
hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024.

The profiled command is `bench.py --only-headline`: every sort + draw it runs is a headline frame, and its JSON line
(found in the pass's log) says how many there were, so frame bytes = (bytes of all dispatches of the frame's kernels) /
frames.  Upload-time kernels (AoS -> SoA splits, the Morton re-ordering of an upload and its 32-bit-key radix passes, the
LDS self-test) are not part of a frame.

usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag> profiles/<tag>_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys

UPLOAD = ("k_split", "k_aos4", "k_morton", "k_perm_from", "k_scatter_u32", "k_selftest", "k_debug", "k_unmap", "k_block_boxes", "k_iota2")

root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
bench = None
for log in sorted(glob.glob(root + "/*.log")):
    for line in open(log, errors="replace"):
        if line.startswith('{"metric"'):
            bench = json.loads(line)
frames = int(bench["frames_drawn_before_timing_ended"]) if bench else None
config = bench["config"]["workload"].split(":")[0] if bench else "C3"
wide_entry_keys = False      # > 65536 list bins (32-bit entry keys): none of the bench configs
kernels, frame_bytes, frame_kernels = {}, 0.0, {}
for name, cs in sorted(acc.items()):
    if name.startswith(("__amd", "at::")) or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
    write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    per_launch = (2.0 * fetch + write) * 1024
    kernels[name] = {"FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1), "dispatches": len(cs["FETCH_SIZE"]),
                     "hbm_bytes_per_launch": int(per_launch)}
    upload = name.startswith(UPLOAD) or ("ArrayLoader<unsigned int>" in name and not wide_entry_keys)
    if frames and not upload:
        per_frame = per_launch * len(cs["FETCH_SIZE"]) / frames
        frame_kernels[name] = int(per_frame)
        frame_bytes += per_frame
doc = {"source": root, "config": config, "frames": frames,
       "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B; "
                     "WRITE_SIZE calibrated 1:1)", "kernels": kernels}
if frames:
    doc["frame_hbm_bytes"] = int(frame_bytes)
    doc["frame_kernels_bytes_per_frame"] = frame_kernels
json.dump(doc, open(out, "w"), indent=1)
for k, v in kernels.items():
    print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  {k[:100]}")
if frames:
    print(f"{frame_bytes / 1e6:9.1f} MB per frame over {frames} frames of {config}")
