"""oracle/make_golden_gather.py — records the REFERENCE's own frustum cull for seeded trees and cameras into
tests/golden/gather_kat.json: Viewer.gatherSceneNodesForSort (text cut out of /root/reference/src/Viewer.js:1969-2077) over
trees built by the reference's own worker (src/splattree/SplatTree.js), run under Node by oracle/gather_ref.mjs with THREE =
oracle/three_min.mjs.  Runs only where /root/reference exists.   usage: python -m oracle.make_golden_gather"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import tree_cases  # noqa: E402
from gaussiansplats3d_amd import camera  # noqa: E402

REF = "/root/reference/src"


def cameras_for(name):
    """Camera set of one case: matrixWorld / fov / render size / mesh transform / gatherAllNodes."""
    eye = np.eye(4).T.reshape(16).tolist()
    out = []
    poses = [((0.5, 0.3, 6.0), (0, 0, 0), (0, 1, 0)), ((-3.0, 1.0, 0.2), (2, 0, -1), (0, 1, 0)), ((0.1, 0.0, 0.05), (1, 0.2, 0.3), (0, -1, -0.5)),
             ((9.0, 7.0, -8.0), (0, 0, 0), (0, 0, 1)), ((0.0, 0.0, -2.5), (0, 0, 4), (0.2, 1, 0))]
    for k, (pos, look, up) in enumerate(poses):
        w, h = [(1920, 1080), (640, 480), (300, 900)][k % 3]
        cam = camera.PerspectiveCamera(w, h, pos, look, up, fov=[50.0, 75.0, 30.0][k % 3])
        out.append(dict(matrixWorld=np.asarray(cam.matrix_world, np.float64).reshape(16).tolist(), fov=[50.0, 75.0, 30.0][k % 3],
                        width=w, height=h, meshWorld=eye, gatherAll=False))
    out.append(dict(out[0], gatherAll=True))
    # a transformed mesh: rotation about y by 0.7 rad, then a translation (column-major)
    c, s = np.cos(0.7), np.sin(0.7)
    mesh = np.array([[c, 0, s, 1.5], [0, 1, 0, -0.5], [-s, 0, c, 0.25], [0, 0, 0, 1]]).T.reshape(16).tolist()
    out.append(dict(out[1], meshWorld=mesh))
    return out


def main():
    golden = {}
    for name in ("clusters40k", "gauss5k", "grid_on_split_planes"):
        case = tree_cases.make_case(name)
        c = case["centers"]
        n = c.shape[0]
        c4 = np.zeros((n, 4), np.float32)
        c4[:, :3] = c
        c4[:, 3] = np.arange(n)
        cams = cameras_for(name)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "in.bin"), "wb") as f:
                f.write(np.array([n, case["max_depth"], case["max_centers"], 0], np.uint32).tobytes() + c4.tobytes())
            json.dump(cams, open(os.path.join(d, "cams.json"), "w"))
            subprocess.check_call(["node", "--no-warnings", os.path.join(ROOT, "oracle", "gather_ref.mjs"), f"{REF}/splattree/SplatTree.js",
                                   f"{REF}/Viewer.js", f"{REF}/Constants.js", os.path.join(d, "in.bin"),
                                   os.path.join(d, "cams.json"), os.path.join(d, "out.json")], cwd=os.path.join(ROOT, "oracle"))
            ref = json.load(open(os.path.join(d, "out.json")))
        for cam, r in zip(cams, ref["cameras"]):
            cam.update(splatRenderCount=r["splatRenderCount"], sha256=r["sha256"], modelView=r["modelView"],
                       indexes=r["indexes"] if name != "clusters40k" else None)
        golden[name] = dict(n=n, leaves=ref["leaves"], cameras=cams)
        print(name, ref["leaves"], [cam["splatRenderCount"] for cam in cams])
    with open(os.path.join(ROOT, "tests", "golden", "gather_kat.json"), "w") as f:
        json.dump(golden, f)


if __name__ == "__main__":
    main()
