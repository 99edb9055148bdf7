"""Deterministic synthetic stand-ins for the benchmark scenes (SURVEY.md §8d "Inputs").

None of bonsai.ksplat / truck.ply / garden.ply ship with the container, so every BASELINE.json config
gets a seeded generator producing the arrays the reference hands its GPU path
(src/splatmesh/SplatMesh.js:637-898): fp32 centres, fp32 (or fp16) covariances R*S^2*R^T
(src/loaders/SplatBuffer.js:440-486), RGBA8 colour and fp16 SH as coefficient-major RGB triples.
"""
import os
from dataclasses import dataclass

import numpy as np

SEED_BASE = 20260921


@dataclass
class SplatScene:
    centers: np.ndarray      # float32 [n,3]
    cov: np.ndarray          # float32 [n,6]  (m00,m01,m02,m11,m12,m22); fp16-rounded if cov_half
    rgba: np.ndarray         # uint8   [n,4]
    sh: np.ndarray           # float16 [n, 0|9|24]  coefficient-major RGB triples
    sh_degree: int
    cov_half: bool = False
    name: str = ""

    @property
    def count(self):
        return self.centers.shape[0]


def _covariances(rng, n, chunk=1 << 20):
    """R*S^2*R^T in fp64 -> fp32; log-scales ~ N(ln 0.015, 0.7^2) clipped to [ln 1e-3, ln 0.5];
    rotations = normalised N(0,1)^4 with w >= 0."""
    out = np.empty((n, 6), dtype=np.float32)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        ls = np.clip(rng.normal(np.log(0.015), 0.7, size=(m, 3)), np.log(1e-3), np.log(0.5))
        sc = np.exp(ls)
        q = rng.normal(size=(m, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        q[q[:, 0] < 0] *= -1.0
        w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.empty((m, 3, 3))
        R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
        R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
        R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
        M = R * sc[:, None, :]
        S = M @ np.transpose(M, (0, 2, 1))
        out[s:s + m] = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]],
                                axis=1).astype(np.float32)
    return out


def _appearance(rng, n, sh_degree):
    opacity = rng.normal(0.5, 2.0, size=n)
    a8 = np.clip(np.round(255.0 / (1.0 + np.exp(-opacity))), 1, 255).astype(np.uint8)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, a8[:, None]], axis=1)
    ncoef = {0: 0, 1: 9, 2: 24}[sh_degree]
    sh = rng.normal(0.0, 0.1, size=(n, ncoef)).astype(np.float16) if ncoef else np.zeros((n, 0), np.float16)
    return rgba, sh


def scene_like(n, sh_degree, seed, name=""):
    """'Scene-like' generator: 80 % of centres in 4096 Gaussian clusters (centres U([-4,4]^3),
    sigma ~ logU(0.02,0.4)), 20 % U([-8,8]^3) background."""
    rng = np.random.default_rng(seed)
    n_cl = int(round(0.8 * n))
    k = 4096
    cl_c = rng.uniform(-4.0, 4.0, size=(k, 3))
    cl_s = np.exp(rng.uniform(np.log(0.02), np.log(0.4), size=k))
    which = rng.integers(0, k, size=n_cl)
    centers = np.empty((n, 3), dtype=np.float32)
    centers[:n_cl] = (cl_c[which] + rng.normal(size=(n_cl, 3)) * cl_s[which, None]).astype(np.float32)
    centers[n_cl:] = rng.uniform(-8.0, 8.0, size=(n - n_cl, 3)).astype(np.float32)
    perm = rng.permutation(n)                       # file order carries no spatial structure
    centers = centers[perm]
    cov = _covariances(rng, n)
    rgba, sh = _appearance(rng, n, sh_degree)
    return SplatScene(centers, cov, rgba, sh, sh_degree, False, name)


def _frame_from_normal(nrm):
    """Two unit tangents (t1, t2) completing each unit normal to a right-handed frame."""
    helper = np.where(np.abs(nrm[:, :1]) < 0.9, np.array([[1.0, 0.0, 0.0]]), np.array([[0.0, 1.0, 0.0]]))
    t1 = np.cross(nrm, helper)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    return t1, np.cross(nrm, t1)


def _surfel_covariances(rng, nrm, spacing, chunk=1 << 20):
    """Flat, anisotropic splats lying in their surface, like a trained capture: in-plane scales ~ spacing * logN(ln 1.4,
    0.5^2) (two independent axes, a random in-plane rotation), the axis along the normal 5..20 times smaller than the
    smaller in-plane one; the normal itself is jittered by a few degrees.  R*S^2*R^T in fp64 -> fp32."""
    n = nrm.shape[0]
    out = np.empty((n, 6), dtype=np.float32)
    spacing = np.broadcast_to(np.asarray(spacing, dtype=np.float64), (n,))
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        nn = nrm[s:s + m] + rng.normal(size=(m, 3)) * 0.06
        nn /= np.linalg.norm(nn, axis=1, keepdims=True)
        t1, t2 = _frame_from_normal(nn)
        a = rng.uniform(0.0, 2.0 * np.pi, size=(m, 1))
        u1 = np.cos(a) * t1 + np.sin(a) * t2
        u2 = np.cross(nn, u1)
        s12 = spacing[s:s + m, None] * np.exp(rng.normal(np.log(1.4), 0.5, size=(m, 2)))
        s3 = s12.min(axis=1) / rng.uniform(5.0, 20.0, size=m)
        sc = np.clip(np.concatenate([s12, s3[:, None]], axis=1), 2e-4, 1.5)
        M = np.stack([u1, u2, nn], axis=2) * sc[:, None, :]            # columns = axes * scales
        S = M @ np.transpose(M, (0, 2, 1))
        out[s:s + m] = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]],
                                axis=1).astype(np.float32)
    return out


def capture_like(n, sh_degree, seed, pose="garden", name=""):
    """A stand-in that resembles a trained capture (the "C3S" scene): splats lie ON 2-D manifolds instead of filling a
    volume, they are flat, and their opacities are bimodal.
      * 35 %: a ground disc of radius 6 under the demo's look-at point (normal = the demo's up vector), half of it
              uniform, half a 2-D Gaussian about the object (splat sizes follow the local spacing);
      * 35 %: the object: a table-like box (2.0 x 0.9 x 1.2) standing on the ground at the look-at point with three balls on
              its top;
      * 30 %: 96 planar background patches (3..8 units across) tangent to spheres of radius 7..15 about the look-at point,
              upper hemisphere - outside the radius the demo camera (and its orbit, 5.7 units) moves on, so the camera
              stands outside the object and inside the backdrop, like demo/garden.html:38-43;
      * scales: see _surfel_covariances (in-plane ~ 1.4 x the surface's splat spacing, normal axis 5..20 x smaller);
      * opacity: 40 % sigmoid(N(-3.5, 0.8^2)) (alpha < 0.1 for ~95 % of them), 60 % sigmoid(N(3, 1.5^2));
      * colour: one base colour per surface + noise; SH ~ N(0, 0.05^2) as fp16.
    File order is a random permutation (a .ply carries no spatial order)."""
    from .camera import DEMO_POSES
    rng = np.random.default_rng(seed)
    up, _, look = (np.asarray(v, dtype=np.float64) for v in DEMO_POSES[pose])
    u = up / np.linalg.norm(up)
    e1, e2 = _frame_from_normal(u[None, :])
    e1, e2 = e1[0], e2[0]
    n_ground = int(round(0.35 * n))
    n_object = int(round(0.35 * n))
    n_back = n - n_ground - n_object
    pts, nrm, spc, col = [], [], [], []

    def add(p, nn, spacing, base_rgb):
        pts.append(p); nrm.append(nn); spc.append(np.broadcast_to(spacing, (p.shape[0],)).astype(np.float64))
        col.append(np.clip(np.asarray(base_rgb)[None, :] + rng.normal(0.0, 18.0, size=(p.shape[0], 3)), 0, 255))

    # ground: half uniform over the disc, half a 2-D Gaussian (sigma 1.6) about the object - denser where a capture has more views
    G = look - 0.6 * u
    R_g, sig = 6.0, 1.6
    h = n_ground // 2
    r_c = sig * np.sqrt(-2.0 * np.log(1.0 - rng.uniform(size=n_ground - h) * (1.0 - np.exp(-0.5 * (R_g / sig) ** 2))))   # Rayleigh, cut at R_g
    r = np.concatenate([R_g * np.sqrt(rng.uniform(size=h)), r_c])
    th = rng.uniform(0.0, 2.0 * np.pi, size=n_ground)
    p = G + (r * np.cos(th))[:, None] * e1 + (r * np.sin(th))[:, None] * e2 + rng.normal(0.0, 0.01, size=(n_ground, 1)) * u
    # local spacing from the local areal density of the two populations
    dens = h / (np.pi * R_g ** 2) + (n_ground - h) * np.exp(-0.5 * (r / sig) ** 2) / (2.0 * np.pi * sig ** 2)
    add(p, np.broadcast_to(u, p.shape).copy(), 1.0 / np.sqrt(dens), (96, 120, 64))

    # object: box faces by area + three balls
    bx = np.array([2.0, 0.9, 1.2])                    # extents along (e1, u, e2)
    C = look - 0.6 * u + 0.45 * u                     # the box stands on the ground
    n_balls = n_object // 4
    n_box = n_object - n_balls
    areas = np.array([bx[1] * bx[2], bx[1] * bx[2], bx[0] * bx[2], bx[0] * bx[2], bx[0] * bx[1], bx[0] * bx[1]])
    face = rng.choice(6, size=n_box, p=areas / areas.sum())
    axes = np.stack([e1, u, e2])                      # rows
    uv = rng.uniform(-0.5, 0.5, size=(n_box, 2))
    p = np.empty((n_box, 3)); nn = np.empty((n_box, 3))
    for f in range(6):
        k = f // 2; sgn = 1.0 if f % 2 == 0 else -1.0
        a_, b_ = [(1, 2), (0, 2), (0, 1)][k]
        sel = face == f
        p[sel] = C + sgn * 0.5 * bx[k] * axes[k] + (uv[sel, :1] * bx[a_]) * axes[a_] + (uv[sel, 1:] * bx[b_]) * axes[b_]
        nn[sel] = sgn * axes[k]
    p += rng.normal(0.0, 0.004, size=(n_box, 1)) * nn
    add(p, nn, np.sqrt(2.0 * areas.sum() / 2.0 / n_box), (150, 110, 80))
    top = C + 0.5 * bx[1] * u
    per_ball = [n_balls // 3, n_balls // 3, n_balls - 2 * (n_balls // 3)]
    for k, (ox, oz, rad, rgb) in enumerate([(-0.55, 0.1, 0.30, (200, 60, 50)), (0.15, -0.2, 0.38, (60, 90, 200)), (0.65, 0.25, 0.24, (220, 200, 70))]):
        d = rng.normal(size=(per_ball[k], 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        cb = top + ox * e1 + oz * e2 + rad * u
        add(cb + rad * d, d, np.sqrt(4.0 * np.pi * rad * rad / per_ball[k]), rgb)

    # backdrop patches
    n_patch = 96
    dirs = rng.normal(size=(4 * n_patch, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = dirs[dirs @ u > -0.1][:n_patch]
    radius = rng.uniform(7.0, 15.0, size=n_patch)
    ext = rng.uniform(3.0, 8.0, size=(n_patch, 2))
    area = ext[:, 0] * ext[:, 1]
    which = rng.choice(n_patch, size=n_back, p=area / area.sum())
    pn = -dirs + rng.normal(0.0, 0.25, size=dirs.shape)
    pn /= np.linalg.norm(pn, axis=1, keepdims=True)
    pt1, pt2 = _frame_from_normal(pn)
    xy = rng.uniform(-0.5, 0.5, size=(n_back, 2)) * ext[which]
    p = look + (radius[:, None] * dirs)[which] + xy[:, :1] * pt1[which] + xy[:, 1:] * pt2[which]
    p += rng.normal(0.0, 0.02, size=(n_back, 1)) * pn[which]
    base = rng.integers(40, 220, size=(n_patch, 3)).astype(np.float64)
    pts.append(p); nrm.append(pn[which]); spc.append(np.full(n_back, np.sqrt(area.sum() / n_back)))
    col.append(np.clip(base[which] + rng.normal(0.0, 18.0, size=(n_back, 3)), 0, 255))

    centers = np.concatenate(pts).astype(np.float32)
    normals = np.concatenate(nrm)
    spacing = np.concatenate(spc)
    rgb = np.concatenate(col)
    perm = rng.permutation(n)
    centers, normals, spacing, rgb = centers[perm], normals[perm], spacing[perm], rgb[perm]
    cov = _surfel_covariances(rng, normals, spacing)
    low = rng.uniform(size=n) < 0.4
    logit = np.where(low, rng.normal(-3.5, 0.8, size=n), rng.normal(3.0, 1.5, size=n))
    a8 = np.clip(np.round(255.0 / (1.0 + np.exp(-logit))), 1, 255).astype(np.uint8)
    rgba = np.concatenate([np.round(rgb).astype(np.uint8), a8[:, None]], axis=1)
    ncoef = {0: 0, 1: 9, 2: 24}[sh_degree]
    sh = rng.normal(0.0, 0.05, size=(n, ncoef)).astype(np.float16) if ncoef else np.zeros((n, 0), np.float16)
    return SplatScene(centers, cov, rgba, sh, sh_degree, False, name)


def uniform_box(n, seed, half=10.0, cov_half=True, name=""):
    """C4: centres ~ U([-10,10]^3), SH0, covariance stored as fp16 (the reference forces half precision
    above 4096^2 texels, src/splatmesh/SplatMesh.js:667-670)."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-half, half, size=(n, 3)).astype(np.float32)
    cov = _covariances(rng, n)
    if cov_half:
        cov = cov.astype(np.float16).astype(np.float32)
    rgba, sh = _appearance(rng, n, 0)
    return SplatScene(centers, cov, rgba, sh, 0, cov_half, name)


# BASELINE.json configs -> (generator, camera pose name, W, H)
CONFIGS = {
    "C1": dict(n=1_200_000, sh=0, pose="bonsai", width=1920, height=1080, label="bonsai.ksplat stand-in, SH0"),
    "C2": dict(n=2_500_000, sh=0, pose="truck", width=1920, height=1080, label="truck.ply stand-in, SH0, 1920x1080"),
    "C3": dict(n=5_800_000, sh=2, pose="garden", width=1920, height=1080, label="garden.ply stand-in, SH2, 1920x1080"),
    "C4": dict(n=16_000_000, sh=0, pose="synthetic16m", width=3840, height=2160,
               label="16M uniform-random Gaussians, SH0, 3840x2160"),
    "C5": dict(n=5_800_000, sh=2, pose="garden", width=7680, height=4320, label="garden.ply stand-in, SH2, 7680x4320"),
    # not a BASELINE.json config: the C3 geometry with translucent splats (opacity ~ sigmoid(N(-2, 1)) instead of
    # sigmoid(N(0.5, 4))), so that pixels do not saturate after a few splats and the blend really walks its lists
    "C3T": dict(n=5_800_000, sh=2, pose="garden", width=1920, height=1080,
                label="garden.ply stand-in with translucent splats, SH2, 1920x1080"),
    # not a BASELINE.json config either: a stand-in that resembles a trained capture (surfaces, flat splats, bimodal opacity,
    # the camera outside the object): brackets the headline from the honest side (capture_like)
    "C3S": dict(n=5_800_000, sh=2, pose="garden", width=1920, height=1080,
                label="capture-like stand-in (surfels on ground / object / backdrop), SH2, 1920x1080"),
}


# Real captures, when the caller supplies them: $GS_DATA_DIR/<file>; none ship with this repository.
REAL_FILES = {"C1": "bonsai.ksplat", "C2": "truck.ply", "C3": "garden.ply", "C5": "garden.ply"}


def load_real_scene(cfg):
    """The config's real capture from $GS_DATA_DIR through the native asset reader, or None if it is not there.
    .ksplat files with 8-bit SH are not supported here (the bench draws fp16 SH)."""
    root = os.environ.get("GS_DATA_DIR")
    name = REAL_FILES.get(cfg)
    if not root or not name:
        return None
    path = os.path.join(root, name)
    if not os.path.isfile(path):
        return None
    from . import assets
    want = CONFIGS[cfg]["sh"]
    arr = assets.load(path, spherical_harmonics_degree=want)
    deg = int(arr["sh_degree"])
    if deg and arr["sh_f16"] is None:
        raise ValueError(f"{path}: 8-bit spherical harmonics are not supported by the benchmark scene loader")
    sh = arr["sh_f16"].view(np.float16) if deg else np.zeros((arr["centers"].shape[0], 0), np.float16)
    return SplatScene(arr["centers"], arr["cov"], arr["rgba"], sh, deg, False, name)


def make_config_scene(cfg, n_override=None):
    if not n_override:
        real = load_real_scene(cfg)
        if real is not None:
            return real
    c = CONFIGS[cfg]
    n = int(n_override) if n_override else c["n"]
    if cfg == "C3S":
        return capture_like(n, c["sh"], SEED_BASE + 30, c["pose"], name=cfg)
    num = 3 if cfg in ("C5", "C3T") else int(cfg[1:])  # C5 is the C3 scene at 8K, C3T the C3 scene made translucent
    seed = SEED_BASE + num
    if cfg == "C4":
        return uniform_box(n, seed, name=cfg)
    scene = scene_like(n, c["sh"], seed, name=cfg)
    if cfg == "C3T":
        rng = np.random.default_rng(seed + 1000)
        scene.rgba[:, 3] = np.clip(np.round(255.0 / (1.0 + np.exp(-rng.normal(-2.0, 1.0, size=n)))), 1, 255).astype(np.uint8)
    return scene
