"""oracle/make_golden_tree.py — records the REFERENCE's own octree for tests/tree_cases.py into
tests/golden/tree_kat.json by running createSplatTreeWorker (cut out of /root/reference/src/splattree/SplatTree.js
as text, oracle/tree_ref.js) under Node.  Runs only where /root/reference exists.
usage: python -m oracle.make_golden_tree"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tree_cases  # noqa: E402

SRC = "/root/reference/src/splattree/SplatTree.js"


def digest_leaves(leaves):
    """sha256 over (min, max, center as IEEE-double hex, depth, indexes) of every leaf in order."""
    h = hashlib.sha256()
    for lf in leaves:
        h.update(("".join(lf["min"]) + "".join(lf["max"]) + "".join(lf["center"]) + str(lf["depth"])).encode())
        h.update(np.asarray(lf["indexes"], dtype=np.uint32).tobytes())
    return h.hexdigest()


def main():
    out = {}
    for name in tree_cases.CASES:
        case = tree_cases.make_case(name)
        c = case["centers"]
        n = c.shape[0]
        c4 = np.zeros((n, 4), np.float32)
        c4[:, :3] = c
        c4[:, 3] = np.arange(n)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "in.bin"), "wb") as f:
                f.write(np.array([n, case["max_depth"], case["max_centers"], 0], np.uint32).tobytes() + c4.tobytes())
            subprocess.check_call(["node", os.path.join(ROOT, "oracle", "tree_ref.js"), SRC, os.path.join(d, "in.bin"),
                                   os.path.join(d, "out.json")], stdout=subprocess.DEVNULL)
            ref = json.load(open(os.path.join(d, "out.json")))
        entry = dict(n=n, leaves=len(ref["leaves"]), all_leaves=ref["leafCount"],
                     splats=sum(len(lf["indexes"]) for lf in ref["leaves"]), sha256=digest_leaves(ref["leaves"]),
                     inputs=hashlib.sha256(c.tobytes()).hexdigest())
        if n <= 100:
            entry["full"] = ref["leaves"]
        out[name] = entry
        print(f"{name:24s} n={n:6d} leaves={entry['leaves']:5d}/{entry['all_leaves']:5d} splats={entry['splats']:6d} {entry['sha256'][:16]}")
    with open(os.path.join(ROOT, "tests", "golden", "tree_kat.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
