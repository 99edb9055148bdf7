#!/bin/bash
# tools/kres.sh <file.hip> [extra hipcc flags]: per-kernel VGPRs / spills / occupancy / LDS of one translation unit (CPU box: hipcc
# cross-compiles gfx950).  Run from anywhere.
set -e
here="$(cd "$(dirname "$0")/.." && pwd)"
src="$1"; shift
nofma=""
case "$(basename "$src")" in sorter.hip|project.hip|tree.hip|assets.hip) nofma="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $nofma "$@" -Rpass-analysis=kernel-resource-usage \
    -c "$here/gaussiansplats3d_amd/csrc/$(basename "$src")" -o /tmp/kres_$$.o 2>&1 | python3 -c '
import sys, re, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-90s %5s %5s %5s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "spill", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)[:90]
    print("%-90s %5d %5d %5d %6d %4d %7d" % (n, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
'
rm -f /tmp/kres_$$.o
