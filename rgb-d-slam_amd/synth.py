"""Synthetic RGB-D inputs shaped like the reference's datasets.

Scene: the inside of a room corner -- three mutually orthogonal planes 1.0-2.2 m from the first
camera, oblique to the view so all six degrees of freedom are observable, textured with a
deterministic sin-product albedo (non-zero Sobel response for the photometric term).

Camera model of the reference (D1/Kernel.cu:42-44): looks down -z, pixel (i, j) of the *flipped*
image is the ray ((j-cx)/fx, (i-cy)/fy, -1); depth files are stored top-down, hence row n-1-i.
Poses are camera-to-world (R, t), as `_poseR/_poset` in D1/RG.h:55-58.
"""
import numpy as np

TUM_CALIB = np.array([525.0, 525.0, 319.5, 239.5, 0.2624, -0.9531, -0.0054, 0.0026, 1.1633, 1.0, 5000.0],
                     np.float32)  # D1/GP-demo.cpp:404 with ds=1.0 for synthetic data
PLANE_DIST = np.array([1.2, 1.0, 2.2])


def calib_for(n, m):
    """TUM-like intrinsics scaled to an n x m image (config 4 uses 1080 x 1920)."""
    c = TUM_CALIB.copy()
    s = m / 640.0
    c[0] = 525.0 * s
    c[1] = 525.0 * s
    c[2] = (m - 1) / 2.0
    c[3] = (n - 1) / 2.0
    return c


def _rot_to_z(u):
    """Rotation taking unit vector u to +z (Rodrigues)."""
    z = np.array([0.0, 0.0, 1.0])
    v = np.cross(u, z)
    s, c = np.linalg.norm(v), float(u @ z)
    if s < 1e-12:
        return np.eye(3)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + K + K @ K * ((1 - c) / (s * s))


def corner_scene():
    """Returns (normals[3,3] rows, d[3]) with n_k . p = d_k, normals facing the camera."""
    u = PLANE_DIST / np.linalg.norm(PLANE_DIST)
    Q = _rot_to_z(u)
    normals = Q.T.copy()  # row k = Q e_k
    c0 = -(normals.T @ PLANE_DIST)
    d = normals @ c0
    return normals, d


def euler_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def plane_equations():
    """Reference layout (n, e1, e2, d) x 3 (D1/RG.cu:300-303) and a centre point per plane."""
    normals, d = corner_scene()
    eq = np.zeros((3, 10), np.float32)
    centers = np.zeros((3, 3), np.float32)
    for k in range(3):
        e1, e2 = normals[(k + 1) % 3], normals[(k + 2) % 3]
        eq[k, 0:3], eq[k, 3:6], eq[k, 6:9], eq[k, 9] = normals[k], e1, e2, d[k]
        # centre: the point of the plane closest to the view axis hit at ~2 m
        p = np.array([0.0, 0.0, -2.0])
        centers[k] = p - (normals[k] @ p - d[k]) * normals[k]
    return eq, centers


def render(R, t, n=480, m=640, calib=None, noise_sigma=0.0, rng=None):
    """Depth (u16, file orientation) and BGR8 colour of the scene seen from camera-to-world (R, t)."""
    calib = TUM_CALIB if calib is None else calib
    fx, fy, cx, cy, ds, factor = calib[0], calib[1], calib[2], calib[3], calib[9], calib[10]
    normals, d = corner_scene()
    ii, jj = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(m, dtype=np.float64), indexing="ij")
    dirs = np.stack([(jj - cx) / fx, (ii - cy) / fy, -np.ones_like(ii)], -1)  # camera frame
    dw = dirs @ np.asarray(R, np.float64).T
    o = np.asarray(t, np.float64)
    best = np.full((n, m), np.inf)
    which = np.full((n, m), -1)
    for k in range(3):
        den = dw @ normals[k]
        num = d[k] - normals[k] @ o
        with np.errstate(divide="ignore", invalid="ignore"):
            tk = np.where(den < -1e-9, num / den, np.inf)
        tk = np.where(tk > 0, tk, np.inf)
        upd = tk < best
        best = np.where(upd, tk, best)
        which = np.where(upd, k, which)
    valid = np.isfinite(best) & (best < 9.0)
    depth_m = np.where(valid, best, 0.0)
    if noise_sigma > 0:
        rng = rng or np.random.default_rng(0)
        depth_m = np.where(valid, depth_m + rng.normal(0.0, noise_sigma, depth_m.shape), 0.0)
    pts = o + dw * depth_m[..., None]
    col = np.zeros((n, m, 3))
    for k in range(3):
        e1, e2 = normals[(k + 1) % 3], normals[(k + 2) % 3]
        a, b = pts @ e1, pts @ e2
        base = 0.5 + 0.22 * np.sin(9.0 * a + k) * np.sin(7.0 * b + 0.5 * k)
        ck = np.stack([base + 0.18 * np.sin(23.0 * a), base + 0.18 * np.sin(19.0 * b + 1.0),
                       base + 0.18 * np.sin(13.0 * (a + b))], -1)
        col = np.where((which == k)[..., None], ck, col)
    col = np.where(valid[..., None], np.clip(col, 0.0, 1.0), 0.0)
    rgb8 = np.rint(col * 255.0).astype(np.uint8)
    depth16 = np.rint(depth_m * factor / ds).astype(np.uint16)
    # file orientation: row r of the file is row n-1-r of the flipped working image; colour is BGR
    return np.ascontiguousarray(depth16[::-1]), np.ascontiguousarray(rgb8[::-1, :, ::-1])


def pair_offset_pose():
    """Config 1: rot (0.5, -0.3, 0.2) deg XYZ-Euler, t (5, -3, 4) mm."""
    R = euler_xyz(*np.deg2rad([0.5, -0.3, 0.2]))
    return R, np.array([0.005, -0.003, 0.004])


def trajectory(n_frames, seed=0):
    """Config 2: TUM fr1/xyz-like +-5 cm xyz sweep with small rotations; steps <= 1 cm, <= 1 deg."""
    poses = []
    for f in range(n_frames):
        ph = 2.0 * np.pi * f / 60.0
        t = np.array([0.05 * np.sin(ph), 0.04 * np.sin(1.3 * ph + 0.4 + seed), 0.05 * np.sin(0.7 * ph + 1.0)])
        t -= np.array([0.0, 0.04 * np.sin(0.4 + seed), 0.05 * np.sin(1.0)])  # frame 0 at the origin
        r = np.deg2rad([1.5 * np.sin(0.9 * ph), 1.2 * np.sin(1.1 * ph + seed), 1.0 * np.sin(0.8 * ph)])
        poses.append((euler_xyz(*r), t))
    R0, t0 = poses[0]
    return [(R0.T @ R, R0.T @ (t - t0)) for R, t in poses]


def sequence(n_frames, n=480, m=640, calib=None, noise_sigma=0.0, seed=0):
    rng = np.random.default_rng(seed)
    calib = calib_for(n, m) if calib is None else calib
    out = []
    for R, t in trajectory(n_frames, seed):
        out.append(render(R, t, n, m, calib, noise_sigma, rng))
    return out


def tiled_planes(nb=256, N=512, res=0.004):
    """BASELINE configs[2] "large planar scene": the three planes of the room corner, each tiled into square patches of
    N x N Bump cells (N * res = 2.048 m at N = 512), nb patches in all (spread evenly over the three planes, each plane's
    patches on a square grid centred on the plane's centre point, ordered from the rim inwards).  Returns the 14-float parameter blocks [nb, 14] in
    the layout of Primitive::SetParam (D1/Primitive.h:180-190): n(3) d scale(2) shift(2) e1(3) e2(3), scale = 2 / res
    (RG.cpp:706), shift = lower-left corner of the patch in plane coordinates (RG.cpp:720-721)."""
    eq, centers = plane_equations()
    params = np.zeros((nb, 14), np.float32)
    side = np.float32(N * res)
    k = 0
    for p in range(3):
        cnt = nb // 3 + (1 if p < nb % 3 else 0)
        g = int(np.ceil(np.sqrt(cnt)))
        E, c = eq[p], centers[p]
        c0, c1 = np.float32(c @ E[3:6]), np.float32(c @ E[6:9])
        # outermost patches first, the patch under the plane's centre last: the visible patches of the last plane then
        # carry the highest labels (nb itself among them -- beyond the u8 range when nb > 255)
        tiles = sorted(((t // g, t % g) for t in range(cnt)),
                       key=lambda ab: -((ab[0] + 0.5 - g / 2.0) ** 2 + (ab[1] + 0.5 - g / 2.0) ** 2))
        for a, b in tiles:
            sh0 = np.float32(c0 + (a - g / 2.0) * side)
            sh1 = np.float32(c1 + (b - g / 2.0) * side)
            params[k] = [E[0], E[1], E[2], E[9], 2.0 / res, 2.0 / res, sh0, sh1, E[3], E[4], E[5], E[6], E[7], E[8]]
            k += 1
    return params
