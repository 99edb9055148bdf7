"""oracle/make_golden_ply.py — records what the REFERENCE's own PlyParserUtils.js derives from the headers (and first
rows) of tests/asset_cases.py into tests/golden/ply_header_kat.json (oracle/ply_ref.mjs under Node).
usage: python -m oracle.make_golden_ply"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import asset_cases  # noqa: E402

UTILS = "/root/reference/src/loaders/ply/PlyParserUtils.js"


def main():
    out = {}
    for name in asset_cases.CASES:
        data, _ = asset_cases.make_case(name)
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "a.ply"), "wb").write(data)
            subprocess.check_call(["node", os.path.join(ROOT, "oracle", "ply_ref.mjs"), UTILS, os.path.join(d, "a.ply"),
                                   os.path.join(d, "o.json"), "4"], stdout=subprocess.DEVNULL)
            out[name] = json.load(open(os.path.join(d, "o.json")))
        print(name, {k: out[name][k] for k in ("vertexCount", "bytesPerVertex", "headerSizeBytes", "sphericalHarmonicsDegree",
                                               "coefficientsPerChannel")})
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ply_header_kat.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
