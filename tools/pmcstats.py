"""Aggregate rocprofv3 counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, re, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
for name, cs in sorted(acc.items()):
    if name.startswith(("k_split", "k_aos", "__amd", "at::")): continue
    print(name[:100])
    print("   " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
