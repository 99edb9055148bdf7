"""three.js r160 camera math restated in fp64 (host side of the render/sort seams).

The reference builds its matrices with three.js (an un-vendored peer dependency, pinned 0.160.0 in
/root/reference/package-lock.json:5426-5430) at these call sites:
  * PerspectiveCamera(fov=50, aspect, near=0.1, far=1000)      src/Viewer.js:30,338
  * camera.up / position / lookAt                              src/Viewer.js:339-344
  * sort MVP = proj * inverse(camera.matrixWorld) * meshWorld  src/Viewer.js:1888-1891
  * focal lengths = proj[0]*0.5*dpr*W, proj[5]*0.5*dpr*H       src/Viewer.js:662-665
All matrices are column-major 16-vectors (``m[4*col + row]``) exactly like ``Matrix4.elements``.
"""
import numpy as np

THREE_FOV_DEG = 50.0     # src/Viewer.js:30
THREE_NEAR = 0.1         # src/Viewer.js:338
THREE_FAR = 1000.0


def _mat(cols16):
    """column-major 16-vector -> 4x4 math matrix."""
    return np.asarray(cols16, dtype=np.float64).reshape(4, 4).T


def _elements(m):
    """4x4 math matrix -> column-major 16-vector."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T).reshape(16)


def make_perspective(fov_deg, aspect, near, far):
    """PerspectiveCamera.updateProjectionMatrix + Matrix4.makePerspective (WebGL clip space)."""
    top = near * np.tan(np.deg2rad(0.5 * fov_deg))
    height = 2.0 * top
    width = aspect * height
    left = -0.5 * width
    right = left + width
    bottom = top - height
    p = np.zeros((4, 4))
    p[0, 0] = 2.0 * near / (right - left)
    p[1, 1] = 2.0 * near / (top - bottom)
    p[0, 2] = (right + left) / (right - left)
    p[1, 2] = (top + bottom) / (top - bottom)
    p[2, 2] = -(far + near) / (far - near)
    p[2, 3] = -2.0 * far * near / (far - near)
    p[3, 2] = -1.0
    return _elements(p)


def look_at_world(position, target, up):
    """Object3D.lookAt for a camera: matrixWorld with -Z pointing at the target.  `up` is normalised
    first (src/Viewer.js:343)."""
    position = np.asarray(position, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    up = np.asarray(up, dtype=np.float64)
    up = up / np.linalg.norm(up)
    z = position - target
    if np.dot(z, z) == 0.0:
        z = np.array([0.0, 0.0, 1.0])
    z = z / np.linalg.norm(z)
    x = np.cross(up, z)
    if np.dot(x, x) == 0.0:                       # up parallel to z: three nudges z and retries
        z = z.copy()
        if abs(up[2]) == 1.0:
            z[0] += 1e-4
        else:
            z[2] += 1e-4
        z = z / np.linalg.norm(z)
        x = np.cross(up, z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, position
    return _elements(m)


def invert(m16):
    return _elements(np.linalg.inv(_mat(m16)))


def multiply(a16, b16):
    return _elements(_mat(a16) @ _mat(b16))


class PerspectiveCamera:
    """Minimal stand-in for the three.js camera the Viewer owns."""

    def __init__(self, width, height, position, look_at, up, fov=THREE_FOV_DEG, near=THREE_NEAR, far=THREE_FAR):
        self.width, self.height = int(width), int(height)
        self.position = np.asarray(position, dtype=np.float64)
        self.projection = make_perspective(fov, self.width / self.height, near, far)
        self.matrix_world = look_at_world(position, look_at, up)
        self.view = invert(self.matrix_world)                    # matrixWorldInverse

    def model_view(self, mesh_world=None):
        """modelViewMatrix = view * meshWorld (identity mesh transform by default)."""
        return self.view if mesh_world is None else multiply(self.view, mesh_world)

    def sort_mvp(self, mesh_world=None):
        """Viewer.runSplatSort: mvp = proj * inverse(matrixWorld) * meshWorld (fp64; narrowed to fp32
        when written into the sorter's memory, src/worker/SortWorker.js:54)."""
        pv = multiply(self.projection, self.view)             # the reference's association: (proj * view) * meshWorld
        return pv if mesh_world is None else multiply(pv, mesh_world)

    def focal(self, focal_adjustment=1.0, dpr=1.0):
        return (self.projection[0] * 0.5 * dpr * self.width * focal_adjustment,
                self.projection[5] * 0.5 * dpr * self.height * focal_adjustment)


def make_orthographic(left, right, top, bottom, near, far, zoom=1.0):
    """OrthographicCamera.updateProjectionMatrix + Matrix4.makeOrthographic (three r160, WebGL clip space)."""
    dx = (right - left) / (2.0 * zoom)
    dy = (top - bottom) / (2.0 * zoom)
    cx = (right + left) / 2.0
    cy = (top + bottom) / 2.0
    l, r, t, b = cx - dx, cx + dx, cy + dy, cy - dy
    w, h, p = 1.0 / (r - l), 1.0 / (t - b), 1.0 / (far - near)
    m = np.zeros((4, 4))
    m[0, 0] = 2 * w; m[0, 3] = -((r + l) * w)
    m[1, 1] = 2 * h; m[1, 3] = -((t + b) * h)
    m[2, 2] = -2 * p; m[2, 3] = -((far + near) * p)
    m[3, 3] = 1.0
    return _elements(m)


class OrthographicCamera(PerspectiveCamera):
    """The Viewer's orthographic camera (src/Viewer.js:1488-1530): frustum = render dimensions in pixels / 2, `zoom`
    set from the look-at distance (setCameraZoomFromPosition, :640-648)."""
    is_orthographic = True

    def __init__(self, width, height, position, look_at, up, zoom=1.0, near=THREE_NEAR, far=THREE_FAR):
        super().__init__(width, height, position, look_at, up)
        self.zoom = float(zoom)
        self.projection = make_orthographic(width / -2.0, width / 2.0, height / 2.0, height / -2.0, near, far, zoom)


# Camera poses of the reference's demo pages (up, position, lookAt): demo/{bonsai,truck,garden}.html
DEMO_POSES = {
    "bonsai": ((0.01933, -0.75830, -0.65161), (1.54163, 2.68515, -6.37228), (0.45622, 1.95338, 1.51278)),
    "truck": ((0.0, -1.0, -0.17), (-5.0, -1.0, -1.0), (-1.72477, 0.05395, -0.00147)),
    "garden": ((0.0, -1.0, -0.54), (-3.15634, -0.16946, -0.51552), (1.52976, 2.27776, 1.65898)),
    "synthetic16m": ((0.0, 1.0, 0.0), (0.0, 0.0, 30.0), (0.0, 0.0, 0.0)),
}


def demo_camera(name, width, height):
    up, pos, look = DEMO_POSES[name]
    return PerspectiveCamera(width, height, pos, look, up)


def orbit_cameras(name, width, height, poses=60):
    """a `poses`-step orbit of the demo camera about its look-at point (rotation about the demo's up
    axis, same distance and height), for per-pose medians.  Pose 0 is the demo pose itself."""
    up, pos, look = (np.asarray(v, dtype=np.float64) for v in DEMO_POSES[name])
    axis = up / np.linalg.norm(up)
    rel = pos - look
    out = []
    for k in range(poses):
        a = 2.0 * np.pi * k / poses
        # Rodrigues rotation of the eye offset about the up axis
        r = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
        out.append(PerspectiveCamera(width, height, tuple(look + r), tuple(look), tuple(up)))
    return out
