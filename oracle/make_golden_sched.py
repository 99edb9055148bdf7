"""oracle/make_golden_sched.py — records the REFERENCE's own sort trigger / partial-sort queue into tests/golden/sched_kat.json:
Viewer.runSplatSort (text cut out of /root/reference/src/Viewer.js:1833-1964) driven through scripted camera paths by
oracle/sched_ref.mjs under Node with THREE = oracle/three_min.mjs.  Runs only where /root/reference exists.
usage: python -m oracle.make_golden_sched"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussiansplats3d_amd import camera  # noqa: E402

REF = "/root/reference/src"
EYE = np.eye(4).T.reshape(16).tolist()


def cam_step(pos, look, up=(0, 1, 0), n=100000, sort_all=False, **kw):
    cam = camera.PerspectiveCamera(640, 360, pos, look, up)
    return dict(matrixWorld=np.asarray(cam.matrix_world, np.float64).reshape(16).tolist(),
                projection=np.asarray(cam.projection, np.float64).reshape(16).tolist(), splatRenderCount=n, shouldSortAll=sort_all, **kw)


DONE = dict(sortDone=True)


def rot_y(a, r=5.0):
    return (r * np.sin(a), 0.5, r * np.cos(a))


def scripts():
    out = []
    # 1. first sort from the initial state (lastSortViewDir = (0,0,-1)): camera looking along +x -> angleDiff 0 -> three partial
    #    sorts then the full one; the queue drains one sort per call; nothing more while the camera rests
    s = [cam_step((0, 0, 0), (1, 0, 0))]
    for _ in range(5):
        s += [DONE, cam_step((0, 0, 0), (1, 0, 0))]
    out.append(dict(name="first_sort_partial_queue", steps=s))
    # 2. a call while a sort is running posts nothing
    out.append(dict(name="sort_running", steps=[cam_step((0, 0, 5), (0, 0, 0)), cam_step((0, 0, 5), (0, 0, 0)), DONE, cam_step((0, 0, 5), (0, 0, 0))]))
    # 3. rotation thresholds: after a completed schedule, turn by growing angles (dot 0.995 no sort; 0.98 -> full only;
    #    0.75 -> one partial; 0.6 -> two; 0.3 -> three)
    s = [cam_step((0, 0, 5), (0, 0, 0)), DONE]
    ang = 0.0
    for d in (0.995, 0.98, 0.75, 0.6, 0.3):
        ang += float(np.arccos(d))
        look = (5 * np.sin(ang), 0.0, 5 - 5 * np.cos(ang))
        for _ in range(5):
            s += [cam_step((0, 0, 5), look), DONE]
    out.append(dict(name="rotation_thresholds", steps=s))
    # 4. position threshold: 0.9 units no sort, 1.0 and more sort; direction unchanged
    s = [cam_step((0, 0, 5), (0, 0, -100)), DONE]
    for z in (5.9, 6.0, 6.5, 7.49, 7.51):
        for _ in range(3):
            s += [cam_step((0, 0, z), (0, 0, -100)), DONE]
    out.append(dict(name="position_threshold", steps=s))
    # 5. shouldSortAll from the cull, force, forceSortAll, a render count that shrinks while partial sorts are queued
    s = [cam_step((0, 0, 5), (0, 0, 0), sort_all=True), DONE, cam_step((0, 0, 5), (0, 0, 0)), cam_step((0, 0, 5), (0, 0, 0), force=True), DONE,
         cam_step((3, 0, 5), (0, 3, 0), force=True, forceSortAll=True), DONE,
         cam_step((5, 0, 0), (0, 0, 0), n=100000), DONE, cam_step((5, 0, 0), (0, 0, 0), n=20000), DONE, cam_step((5, 0, 0), (0, 0, 0), n=20000), DONE,
         cam_step((5, 0, 0), (0, 0, 0), n=20000), DONE, cam_step((5, 0, 0), (0, 0, 0), n=20000)]
    out.append(dict(name="sort_all_force_shrinking_count", steps=s))
    # 6. dynamic mode: every call sorts everything; the mesh transform stays out of modelViewProj
    c, sn = np.cos(0.4), np.sin(0.4)
    mesh = np.array([[c, 0, sn, 0.5], [0, 1, 0, -1.0], [-sn, 0, c, 2.0], [0, 0, 0, 1]]).T.reshape(16).tolist()
    s = []
    for k in range(4):
        s += [cam_step(rot_y(0.1 * k), (0, 0, 0)), DONE]
    out.append(dict(name="dynamic_mode", dynamicMode=True, meshWorld=mesh, steps=s))
    # 7. static mode with a mesh transform: modelViewProj = proj * view * meshWorld
    s = []
    for k in range(3):
        s += [cam_step(rot_y(0.9 * k), (0, 0.2, 0), up=(0, -1, -0.3)), DONE]
    out.append(dict(name="static_mesh_transform", meshWorld=mesh, steps=s))
    for sc in out:
        sc.setdefault("dynamicMode", False)
        sc.setdefault("meshWorld", EYE)
        sc.setdefault("splatCount", 100000)
    return out


def main():
    sc = scripts()
    with tempfile.TemporaryDirectory() as d:
        json.dump(sc, open(os.path.join(d, "script.json"), "w"))
        subprocess.check_call(["node", "--no-warnings", os.path.join(ROOT, "oracle", "sched_ref.mjs"), f"{REF}/Viewer.js",
                               os.path.join(d, "script.json"), os.path.join(d, "out.json")], cwd=os.path.join(ROOT, "oracle"))
        ref = json.load(open(os.path.join(d, "out.json")))
    for s, r in zip(sc, ref):
        assert s["name"] == r["name"] and len(s["steps"]) == len(r["steps"])
        for st, rs in zip(s["steps"], r["steps"]):
            st["ref"] = rs
        print(s["name"], [rs.get("splatSortCount") for rs in r["steps"] if not rs.get("sortDone")])
    with open(os.path.join(ROOT, "tests", "golden", "sched_kat.json"), "w") as f:
        json.dump(sc, f)


if __name__ == "__main__":
    main()
