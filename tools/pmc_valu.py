"""VALU utilisation per kernel from the SQ counter pass of tools/pmc.sh <tag> valu.

SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave's VALU instruction is executing, summed over the chip's 1024 SIMDs
(MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles); GRBM_GUI_ACTIVE is the kernel's
duration in shader clocks summed over the 8 XCDs (k_tile_blend: 2.24 M = 8 x 280 k cycles = 0.13 ms under the profiler), so
    valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)          VALU busy over the whole launch
    valu_busy_resident = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 8)                         ... while 8 waves share a SIMD
                         (the fraction of a SIMD's time its VALU is busy if all 8 wave slots were filled for the waves' lifetime)
SQ_INSTS_VALU is wave-level VALU instructions issued; cycles_per_valu_inst = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU.

usage: python tools/pmc_valu.py gpurun_out/pmc_<tag> profiles/<tag>_pmc_valu.json
"""
import collections
import csv
import glob
import json
import re
import statistics
import sys

root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/valu_*/**/*counter_collection.csv", recursive=True) + glob.glob(root + "/all_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
bench = None
for log in sorted(glob.glob(root + "/*.log")):
    for line in open(log, errors="replace"):
        if line.startswith('{"metric"'):
            bench = json.loads(line)
config = bench["config"]["workload"].split(":")[0] if bench else "C3"
kernels = {}
for name, cs in sorted(acc.items()):
    if name.startswith(("__amd", "at::", "k_split", "k_aos4", "k_morton", "k_perm_from", "k_scatter_u32", "k_selftest")):
        continue
    if "SQ_ACTIVE_INST_VALU" not in cs:
        continue
    # the MEDIAN over the dispatches: a launch or two of a pass carry the module load / a clock ramp (r06zz: one k_project dispatch of 56
    # with GRBM_GUI_ACTIVE = 41 M cycles against 1.0 M, the mean said 0.36 busy where every other dispatch says 0.63)
    m = {c: statistics.median(v) for c, v in cs.items()}
    clocks = (m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) or (m.get("SQ_BUSY_CYCLES", 0.0) / 32.0)
    k = {c: round(v, 1) for c, v in sorted(m.items())}
    k["dispatches"] = len(cs["SQ_ACTIVE_INST_VALU"])
    if clocks:
        k["valu_busy_frac"] = round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * clocks), 4)
    if m.get("SQ_WAVE_CYCLES"):
        k["valu_busy_resident"] = round(8.0 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], 4)
    if m.get("SQ_INSTS_VALU"):
        k["cycles_per_valu_inst"] = round(4.0 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"], 3)
    kernels[name] = k
json.dump({"source": root, "config": config,
           "formula": "valu_busy_frac = 4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs)", "kernels": kernels},
          open(out, "w"), indent=1)
for n, k in kernels.items():
    print(f"{k.get('valu_busy_frac', float('nan')):7.3f} busy  {k.get('cycles_per_valu_inst', float('nan')):6.2f} cyc/inst  {n[:90]}")
