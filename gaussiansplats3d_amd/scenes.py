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
    num = 3 if cfg in ("C5", "C3T") else int(cfg[1:])  # C5 is the C3 scene at 8K, C3T the C3 scene made translucent
    seed = SEED_BASE + num
    if cfg == "C4":
        return uniform_box(n, seed, name=cfg)
    scene = scene_like(n, c["sh"], seed, name=cfg)
    if cfg == "C3T":
        rng = np.random.default_rng(seed + 1000)
        scene.rgba[:, 3] = np.clip(np.round(255.0 / (1.0 + np.exp(-rng.normal(-2.0, 1.0, size=n)))), 1, 255).astype(np.uint8)
    return scene
