"""Shared builders for the parity tests."""
import numpy as np

from gaussiansplats3d_amd import camera, scenes


def small_scene(n, sh_degree, seed, scale=0.06, cov_half=False):
    """Splats in a slab in front of the garden camera so most of them land on screen with visible extent."""
    rng = np.random.default_rng(seed)
    cam_pos = np.array(camera.DEMO_POSES["garden"][1])
    look = np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - cam_pos) / np.linalg.norm(look - cam_pos)
    centers = (cam_pos + fwd * rng.uniform(0.3, 9.0, size=(n, 1)) + rng.normal(size=(n, 3)) * 1.2).astype(np.float32)
    ls = rng.normal(np.log(scale), 0.8, size=(n, 3))
    sc = np.exp(ls)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    M = R * sc[:, None, :]
    S = M @ np.transpose(M, (0, 2, 1))
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1).astype(np.float32)
    rgba = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    rgba[:, 3] = np.clip(rgba[:, 3], 1, 255)
    ncoef = {0: 0, 1: 9, 2: 24}[sh_degree]
    sh = rng.normal(0, 0.25, size=(n, ncoef)).astype(np.float16)
    return scenes.SplatScene(centers, cov, rgba, sh, sh_degree, cov_half, "small")


def oracle_inputs(scene):
    """What the reference's shader sees after its storage formats: fp16 SH -> float, fp16 cov -> float."""
    from gaussiansplats3d_amd.util import to_half_three
    cov = scene.cov
    if scene.cov_half:
        cov = to_half_three(cov).view(np.float16).astype(np.float32)
    sh = scene.sh.astype(np.float32) if scene.sh_degree else None
    return scene.centers, cov, scene.rgba, sh


def compare_frames(got_u8, fb_f32, ambig, label="", strict=False):
    """The stated framebuffer tolerance:
       * >= 99.9 % of channel values within 1/255 of the fp32 oracle,
       * every pixel within 2/255 - except the pixels the oracle flags as discard-ambiguous (some splat had
         |A - 8| <= 1e-3 there: the hard `A > 8 -> discard` edge may flip under fp32 reassociation and jump by up to
         exp(-4)*alpha = 4.67/255).  Rounds 1-3 allowed those pixels 6/255; no frame of any test or crop ever used the slack
         (carve_out_pixels = 0 in every report), and round 4 first removed it.  tests/tools/soak.py then drew 1000 scenes on
         seeds nobody had looked at: 998 within 2/255 everywhere, and two frames with ONE ambiguous pixel each at 2.53 and 2.66 of
         1/255 (profiles/r04z_soak_long.txt, seeds 5222 and 5595) - the flip the analysis predicts.  So the ambiguous pixels get
         the ANALYTICAL bound, 255 * exp(-4) = 4.67 (not the old round number), every other pixel stays at 2/255, and the message
         reports how many pixels needed more than 2/255 (carve_out_pixels).
    strict=True (the full-size crop tests): every channel of every pixel within 1/255, ambiguous or not."""
    ref = np.clip(fb_f32, 0.0, 1.0) * 255.0
    err = np.abs(got_u8.astype(np.float32) - ref)           # in 1/255 units, vs the unquantised oracle
    frac_1 = float((err <= 1.0 + 0.5).mean())                # +0.5: our own final rounding to unorm8
    amb = ambig.astype(bool)[..., None]
    worst_clear = float(np.where(amb, 0.0, err).max())
    worst_amb = float(np.where(amb, err, 0.0).max())
    mse = float(((got_u8.astype(np.float64) - ref) ** 2).mean())
    psnr = 10.0 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    carve = int((np.where(amb, err, 0.0).max(axis=-1) > 2.0 + 0.5).sum())    # pixels that NEED the ambiguity carve-out
    msg = (f"{label}: within1={frac_1:.5f} worst_clear={worst_clear:.3f} worst_amb={worst_amb:.3f} psnr={psnr:.1f} "
           f"carve_out_pixels={carve}")
    if strict:       # what the full-size crops actually achieve: every channel within 1/255 (+ our final rounding), no carve-out
        assert max(worst_clear, worst_amb) <= 1.0 + 0.5 and carve == 0, msg
    assert frac_1 >= 0.999, msg
    assert worst_clear <= 2.0 + 0.5, msg
    assert worst_amb <= 255.0 * float(np.exp(-4.0)) + 0.5, msg
    assert psnr >= 50.0, msg
    return msg
