"""oracle/make_golden_raster.py — records what the REFERENCE's own shader text computes into tests/golden/raster_ref.npz.
Runs only where /root/reference exists (with Node and g++).

1. oracle/shader_dump.mjs imports /root/reference/src/splatmesh/SplatMaterial3D.js in place ('three' -> oracle/three_min.mjs)
   and calls SplatMaterial3D.build(...) for every shader permutation below: the reference's own builder returns the GLSL.
2. The GLSL becomes C++ by TOKEN REWRITES ONLY (rewrite() below): storage / precision qualifiers dropped, `in` / `out`
   parameter qualifiers mapped to by-value / by-reference, `float[](...)` -> `{...}`, an `f` suffix on float literals (GLSL
   literals are fp32, C++'s are double).  No statement of the shaders is re-typed.
3. oracle/shader_harness.cpp includes the rewritten text, compiled against oracle/glsl_shim.hpp with
   g++ -O1 -ffp-contract=off (IEEE fp32, no fusing) into oracle/_ref/ (scratch, git-ignored).
4. Seeded scenes (tests/raster_cases.py) go through the vertex shader for all four quad corners, sample fragments through
   the fragment shader; outputs are stored.  tests/test_raster_ref.py compares the C raster oracle (CPU tier) and the HIP
   vertex stage (GPU tier) with them.
usage: python -m oracle.make_golden_raster"""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import raster_cases  # noqa: E402

REF_SRC = "/root/reference/src"


def rewrite(glsl):
    """GLSL ES 3.00 -> C++ against glsl_shim.hpp: token rewrites only."""
    s = glsl
    s = re.sub(r"precision\s+highp\s+float\s*;", "", s)
    s = re.sub(r"#include\s*<common>", "", s)
    s = re.sub(r"\b(attribute|uniform|varying|highp)\s+", "", s)
    s = re.sub(r"\bin\s+(int|uint|float|vec2|vec3|vec4)\b", r"\1", s)                 # `in T x`  -> by value
    s = re.sub(r"\bout\s+(vec2|vec3|vec4|float)\s+(\w+)", r"\1& \2", s)               # `out T x` -> by reference
    s = re.sub(r"const\s+float\[(\d+)\]\s+(\w+)\s*=\s*float\[\]\(([^;]*)\)\s*;", r"const float \2[\1] = {\3};", s)
    # float literals are fp32 in GLSL
    s = re.sub(r"(?<![A-Za-z_0-9.])(\d+\.\d*|\.\d+)(?![0-9A-Za-z_.])", r"\1f", s)
    return s


def build_shader_lib(name, vert, frag, defines, out_dir):
    v, f = os.path.join(out_dir, name + ".vert.inc"), os.path.join(out_dir, name + ".frag.inc")
    open(v, "w").write(rewrite(vert))
    open(f, "w").write(rewrite(frag))
    so = os.path.join(out_dir, f"libshader_{name}.so")
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-w",
           f'-DSHADER_VERT="{v}"', f'-DSHADER_FRAG="{f}"'] + [f"-D{d}" for d in defines] + \
          [os.path.join(ROOT, "oracle", "shader_harness.cpp"), "-o", so]
    subprocess.check_call(cmd, cwd=os.path.join(ROOT, "oracle"))
    return C.CDLL(so)


class Scene(C.Structure):
    _fields_ = [("count", C.c_uint32), ("sh_degree_stored", C.c_uint32), ("cov_half", C.c_uint32), ("sh_u8", C.c_uint32),
                ("centers", C.c_void_p), ("rgba", C.c_void_p), ("cov", C.c_void_p), ("cov16", C.c_void_p), ("sh", C.c_void_p),
                ("scene_idx", C.c_void_p)]


class Uniforms(C.Structure):
    _fields_ = [("model_view", C.c_float * 16), ("projection", C.c_float * 16), ("view_matrix", C.c_float * 16),
                ("camera_position", C.c_float * 3), ("focal", C.c_float * 2), ("viewport", C.c_float * 2),
                ("ortho_zoom", C.c_float), ("inverse_focal_adjustment", C.c_float), ("splat_scale", C.c_float),
                ("orthographic", C.c_int32), ("point_cloud", C.c_int32), ("sh_degree", C.c_int32), ("sh_8bit", C.c_int32),
                ("fade_in_complete", C.c_int32), ("scene_count", C.c_int32), ("scene_center", C.c_float * 3),
                ("fade_start_radius", C.c_float), ("transforms", (C.c_float * 16) * 32), ("scene_opacity", C.c_float * 32),
                ("sh8_min", C.c_float * 32), ("sh8_max", C.c_float * 32), ("scene_visibility", C.c_int32 * 32)]


def run_vertex(lib, case):
    """case: what tests/raster_cases.py builds.  Returns float32 [n, 4 corners, 10]."""
    sc, u = Scene(), Uniforms()
    keep = []

    def ptr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    n = case["centers"].shape[0]
    sc.count, sc.sh_degree_stored, sc.cov_half, sc.sh_u8 = n, case["sh_stored"], int(case["cov_half"]), int(case["sh8"])
    sc.centers, sc.rgba = ptr(case["centers"], np.float32), ptr(case["rgba"], np.uint8)
    sc.cov = ptr(case["cov"], np.float32)
    sc.cov16 = ptr(case["cov16"], np.uint16) if case["cov_half"] else None
    sc.sh = ptr(case["sh_sampled"], np.float32) if case["sh_stored"] else None
    sc.scene_idx = ptr(case["scene_idx"], np.uint32) if case["scene_idx"] is not None else None
    un = case["uniforms"]
    for k in ("model_view", "projection", "view_matrix"):
        getattr(u, k)[:] = np.asarray(un[k], np.float64).astype(np.float32).reshape(16).tolist()
    u.camera_position[:] = np.asarray(un["camera_position"], np.float32).tolist()
    u.focal[:] = [float(np.float32(v)) for v in un["focal"]]
    u.viewport[:] = [float(v) for v in un["viewport"]]
    u.ortho_zoom, u.inverse_focal_adjustment, u.splat_scale = un["ortho_zoom"], un["inverse_focal_adjustment"], un["splat_scale"]
    u.orthographic, u.point_cloud, u.sh_degree, u.sh_8bit = un["orthographic"], un["point_cloud"], un["sh_degree"], int(case["sh8"])
    u.fade_in_complete, u.scene_count = un["fade_in_complete"], un["scene_count"]
    u.scene_center[:] = un["scene_center"]
    u.fade_start_radius = un["fade_start_radius"]
    for s_ in range(32):
        t = un["transforms"][s_] if s_ < len(un["transforms"]) else np.eye(4).T.reshape(16)
        u.transforms[s_][:] = np.asarray(t, np.float64).astype(np.float32).tolist()
        u.scene_opacity[s_] = un["scene_opacity"][s_] if s_ < len(un["scene_opacity"]) else 1.0
        u.scene_visibility[s_] = un["scene_visibility"][s_] if s_ < len(un["scene_visibility"]) else 1
        u.sh8_min[s_], u.sh8_max[s_] = un["sh8_range"]
    out = np.zeros((n, 4, 10), dtype=np.float32)
    lib.harness_run_vertex(C.byref(sc), C.byref(u), out.ctypes.data_as(C.c_void_p))
    return out


def run_fragment(lib, v_position, v_color):
    n = v_position.shape[0]
    col = np.zeros((n, 4), np.float32)
    disc = np.zeros(n, np.uint8)
    vp, vc = np.ascontiguousarray(v_position, np.float32), np.ascontiguousarray(v_color, np.float32)
    lib.harness_run_fragment(C.c_uint32(n), vp.ctypes.data_as(C.c_void_p), vc.ctypes.data_as(C.c_void_p),
                             col.ctypes.data_as(C.c_void_p), disc.ctypes.data_as(C.c_void_p))
    return col, disc


def main():
    assert os.path.isdir(REF_SRC), "reference not present"
    scratch = os.path.join(ROOT, "oracle", "_ref", "shaders")
    os.makedirs(scratch, exist_ok=True)
    builds = raster_cases.shader_builds()
    json.dump([dict(name=k, **v) for k, v in builds.items()], open(os.path.join(scratch, "perms.json"), "w"))
    subprocess.check_call(["node", "--no-warnings", "--experimental-loader", os.path.join(ROOT, "oracle", "three_loader.mjs"),
                           os.path.join(ROOT, "oracle", "shader_dump.mjs"), REF_SRC, scratch, os.path.join(scratch, "perms.json")],
                          cwd=os.path.join(ROOT, "oracle"), stdout=subprocess.DEVNULL)
    libs, meta = {}, {}
    for name, b in builds.items():
        vert, frag = open(os.path.join(scratch, name + ".vert")).read(), open(os.path.join(scratch, name + ".frag")).read()
        defines = (["SHADER_DYNAMIC"] if b.get("dynamicMode") else []) + (["SHADER_EFFECTS"] if b.get("enableOptionalEffects") else [])
        libs[name] = build_shader_lib(name, vert, frag, defines, scratch)
        meta[name] = dict(vert_sha256=hashlib.sha256(vert.encode()).hexdigest(), frag_sha256=hashlib.sha256(frag.encode()).hexdigest(),
                          state=json.load(open(os.path.join(scratch, name + ".state.json"))))
    out = {}
    for cname in raster_cases.CASES:
        case = raster_cases.make_case(cname)
        res = run_vertex(libs[case["build"]], case)
        out["vs_" + cname] = res
        drawn = np.isfinite(res[:, 0, 0]) & ~((res[:, 0, 2] == 2.0) & (res[:, 0, 3] == 1.0))
        print(f"{cname:14s} build={case['build']:10s} splats={res.shape[0]} drawn={int(drawn.sum())}")
    vp, vc = raster_cases.fragment_samples()
    col, disc = run_fragment(libs["base0"], vp, vc)
    out["fs_color"], out["fs_discard"] = col, disc
    print("fragments", len(disc), "discarded", int(disc.sum()))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "raster_ref.npz"), **out)


if __name__ == "__main__":
    main()
