"""oracle/make_golden.py — records the REFERENCE's answers for tests/kat_cases.py into
tests/golden/sort_kat.json (+ sort_kat_small.npz).  Runs only where /root/reference exists.

Two independent executions of the reference's own code are recorded and must agree:
  wasm : /root/reference/src/worker/sorter_no_simd_non_shared.wasm under Node (oracle/wasm_ref.js)
  ref  : /root/reference/src/worker/sorter_no_simd.cpp compiled natively (oracle/_ref)
Small cases keep the full output array; large ones keep a sha256 of it.
usage: python -m oracle.make_golden
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import kat_cases  # noqa: E402
import oracle  # noqa: E402

WASM = "/root/reference/src/worker/sorter_no_simd_non_shared.wasm"


def run_wasm(args):
    n = args["centers4"].shape[0]
    hdr = np.array([n, args["render_count"], args["sort_count"], 1 << args["precision"], int(args["use_int"]),
                    int(args["dynamic"]), int(args["precomputed"] is not None), 0], dtype=np.uint32)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "in.bin"), "wb") as f:
            f.write(hdr.tobytes())
            f.write(args["indexes"].tobytes())
            f.write(np.ascontiguousarray(args["centers4"]).tobytes())
            f.write(np.asarray(args["mvp"], dtype=np.float64).astype(np.float32).tobytes())
            if args["dynamic"]:
                f.write(args["scene_indexes"].tobytes())
                f.write(args["transforms"].tobytes())
            if args["precomputed"] is not None:
                f.write(args["precomputed"].tobytes())
        subprocess.check_call(["node", os.path.join(ROOT, "oracle", "wasm_ref.js"), WASM,
                               os.path.join(d, "in.bin"), os.path.join(d, "out.bin")], stdout=subprocess.DEVNULL)
        return np.fromfile(os.path.join(d, "out.bin"), dtype=np.uint32)


def main():
    meta, small = {}, {}
    for case in kat_cases.CASES:
        args = kat_cases.make_case(case)
        w = run_wasm(args)
        kw = {k: args[k] for k in ("sort_count", "render_count", "precision", "use_int", "dynamic", "precomputed",
                                   "scene_indexes", "transforms")}
        r = oracle.ref_sort_indexes(args["indexes"], args["centers4"], args["mvp"], **kw)
        assert np.array_equal(w, r), f"{case['name']}: wasm and native builds of the reference disagree"
        meta[case["name"]] = dict(inputs=kat_cases.input_digest(args), output=kat_cases.digest(w),
                                  render=int(args["render_count"]), sort=int(args["sort_count"]))
        if w.size <= 5000:
            small[case["name"]] = w
        print(f"{case['name']:18s} n={case['n']:8d} ok  sha256={meta[case['name']]['output'][:16]}")
    # degenerate single splat: native build of the reference segfaults, WASM only
    one = dict(indexes=np.zeros(1, np.uint32), centers4=np.array([[1, 2, 3, 1000]], np.int32),
               mvp=np.arange(16, dtype=np.float64), sort_count=1, render_count=1, precision=16, use_int=True,
               dynamic=False, precomputed=None, scene_indexes=None, transforms=None)
    small["single"] = run_wasm(one)
    same = dict(one, indexes=np.arange(8, dtype=np.uint32), centers4=np.tile(one["centers4"], (8, 1)),
                sort_count=8, render_count=8)
    small["all_equal"] = run_wasm(same)
    print("single ->", small["single"], " all_equal ->", small["all_equal"])
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "sort_kat.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(out, "sort_kat_small.npz"), **small)


if __name__ == "__main__":
    main()
