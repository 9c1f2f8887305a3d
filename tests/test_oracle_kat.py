"""CPU tests of the oracle: analytic known answers and the committed golden vectors.

The reference holds no tests or fixtures for this path ("parity unpinned"), so the oracle is pinned
by (i) closed-form answers that follow from the reference's formulas and (ii) the golden file
tests/golden/slam_96x128.npz written by tests/golden/make_golden.py."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIN20 = float(np.sin(np.deg2rad(20.0)))


def flat_depth(n, m, z_m, factor=5000.0):
    return np.full((n, m), int(round(z_m * factor)), np.uint16)


# ---- stage 1 -------------------------------------------------------------------------------------
def test_vertex_map_formula(orc, synth):
    """D1/Kernel.cu:42-44: x = s*d*((j-cx)/fx), y = s*d*((i-cy)/fy), z = -s*d; rows flipped; d==0 -> 0."""
    n, m = 12, 20
    calib = synth.calib_for(n, m)
    rng = np.random.default_rng(1)
    d = rng.integers(0, 20000, (n, m)).astype(np.uint16)
    d[3, 4] = 0
    v = orc.vertex_map_kinect(d, calib)
    s = np.float32(calib[9]) / np.float32(calib[10])
    for (i, j) in [(0, 0), (5, 7), (n - 1, m - 1), (n - 1 - 3, 4)]:
        dd = np.float32(d[n - 1 - i, j])
        if dd == 0:
            assert tuple(v[i, j]) == (0.0, 0.0, 0.0)
            continue
        x = (s * dd) * ((np.float32(j) - calib[2]) / calib[0])
        y = (s * dd) * ((np.float32(i) - calib[3]) / calib[1])
        assert v[i, j, 0] == np.float32(x) and v[i, j, 1] == np.float32(y) and v[i, j, 2] == np.float32(-s) * dd


def test_cpy_char3(orc):
    """D1/Kernel.cu:749-751: BGR8 -> RGB float /255.0 with the row flip."""
    n, m = 4, 5
    bgr = np.arange(n * m * 3, dtype=np.uint8).reshape(n, m, 3)
    rgb = orc.cpy_char3(bgr)
    exp = (bgr[::-1, :, ::-1].astype(np.float64) / 255.0).astype(np.float32)
    assert np.array_equal(rgb, exp)


def test_normals_fronto_parallel_plane(orc, synth):
    """A fronto-parallel plane gives normals exactly (0,0,+1) after inverse=true (Kernel.cu:175-177,
    288-291); the one-pixel border and holes are zero (Kernel.cu:134-137)."""
    n, m = 24, 32
    calib = synth.calib_for(n, m)
    d = flat_depth(n, m, 1.5)
    d[10, 10] = 0
    v = orc.vertex_map_kinect(d, calib)
    nm = orc.compute_normal(v, True)
    inner = nm[1:-1, 1:-1]
    hole_i = n - 1 - 10
    ok = np.ones((n - 2, m - 2), bool)
    ok[hole_i - 1, 10 - 1] = False
    assert np.array_equal(inner[ok], np.tile(np.float32([0, 0, 1]), (ok.sum(), 1)))
    assert not nm[0].any() and not nm[-1].any() and not nm[:, 0].any() and not nm[:, -1].any()
    assert not nm[hole_i, 10].any()
    assert np.array_equal(orc.compute_normal(v, False)[1:-1, 1:-1][ok], np.tile(np.float32([0, 0, -1]), (ok.sum(), 1)))


def test_sobel_ramp_and_border(orc):
    """3x3 Sobel, scale 1, reflect-101 (OpenCV BORDER_DEFAULT): a ramp a*j + b*i gives gx = 8a,
    gy = 8b in the interior and 0 across a reflected border."""
    n, m = 9, 11
    ii, jj = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(m, dtype=np.float32), indexing="ij")
    img = 0.5 * jj + 0.25 * ii
    rgb = np.stack([img, img, img], -1).astype(np.float32)
    out, gx, gy = orc.intensity_sobel(rgb)
    assert np.allclose(out, img, atol=1e-6)
    assert np.allclose(gx[1:-1, 1:-1], 4.0, atol=1e-5) and np.allclose(gy[1:-1, 1:-1], 2.0, atol=1e-5)
    assert np.allclose(gx[:, 0], 0.0, atol=1e-5) and np.allclose(gx[:, -1], 0.0, atol=1e-5)
    assert np.allclose(gy[0, :], 0.0, atol=1e-5) and np.allclose(gy[-1, :], 0.0, atol=1e-5)


# ---- stage 2/3 ---------------------------------------------------------------------------------
def test_icp_grid_divup(orc):
    """divUp(n/fact, 8) (D1/ICP.cu:260-261, cudaTypes.cuh:53)."""
    assert orc.icp_grid(480, 640, 1) == (60, 80)
    assert orc.icp_grid(480, 640, 2) == (30, 40)
    assert orc.icp_grid(480, 640, 4) == (15, 20)
    assert orc.icp_grid(1080, 1920, 4) == (34, 60)
    assert orc.icp_grid(37, 53, 2) == (3, 4)


def test_identical_pair_gives_zero_rhs_and_identity(orc, pair):
    """Model == current frame: every residual is exactly 0, so b = 0, the increment is the identity
    and the pose does not move (SURVEY.md 8c KAT)."""
    c = pair["calib"]
    I, z = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    img, gx, gy = orc.intensity_sobel(pair["rgb0"])
    A, b, sums = orc.estimate_system(c[:4], I, z, pair["v0"], pair["n0"], pair["rgb0"], gx, gy, pair["v0"], pair["n0"],
                                     pair["rgbd0"], 0.05, SIN20, 2)
    assert not b.any()
    assert np.array_equal(A, A.T) and np.all(np.diag(A) > 0)
    ok, R, t, trace = orc.align_gauss_newton(orc.icp_params(), c[:4], pair["v0"], pair["n0"], pair["rgb0"], pair["v0"],
                                             pair["n0"], pair["rgbd0"], I, z, I, z)
    assert ok and np.array_equal(R, I) and not t.any()
    assert len(trace) == 3          # converged at iteration 0 of each level (RG.cpp:889)


def test_known_offset_is_recovered(orc, pair):
    """Config 1: rot (0.5,-0.3,0.2) deg, t (5,-3,4) mm recovered to depth-quantisation accuracy."""
    c = pair["calib"]
    I, z = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    ok, R, t, trace = orc.align_gauss_newton(orc.icp_params(), c[:4], pair["v1"], pair["n1"], pair["rgb1"], pair["v0"],
                                             pair["n0"], pair["rgbd0"], I, z, I, z)
    assert ok and 3 <= len(trace) <= 12
    assert np.abs(R.reshape(3, 3).T - pair["R1"]).max() < 2e-4      # R is column-major
    assert np.abs(t - pair["t1"]).max() < 3e-4
    # rows of the trace: 27 sums + 6-vector solution; the first solution is the biggest step
    assert np.abs(trace[0, 27:]).max() > np.abs(trace[-1, 27:]).max()


def test_system_matches_blockwise_float64_sum(orc, pair):
    """The two-stage reduction (8x8 tree + 512-strided chains, ICP.cu:28-70) agrees with a float64
    sum of the same per-block partials to float32 accuracy."""
    c = pair["calib"]
    I, z = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    buf, corr = orc.icp_match(c[:4], I, z, pair["v1"], pair["n1"], pair["rgb1"], pair["gx1"], pair["gy1"], pair["v0"],
                              pair["n0"], pair["rgbd0"], 0.05, SIN20, 1)
    sums = orc.icp_reduce(buf)
    ref = buf.astype(np.float64).sum(axis=1)
    assert np.allclose(sums, ref, rtol=2e-5, atol=1e-7)
    assert corr.min() >= -1 and corr.max() < 480 * 640 and (corr >= 0).sum() > 100000


def test_dense_helpers(orc):
    rng = np.random.default_rng(3)
    Mx = rng.normal(size=(6, 6))
    A = (Mx @ Mx.T + 6 * np.eye(6)).astype(np.float32)
    b = rng.normal(size=6).astype(np.float32)
    rc, x = orc.llt_solve6(A, b)
    assert rc == 0 and np.allclose(x, np.linalg.solve(A.astype(np.float64), b), rtol=1e-4, atol=1e-5)
    assert np.isclose(orc.det6(A), np.linalg.det(A.astype(np.float64)), rtol=1e-4)
    assert orc.llt_solve6(-A, b)[0] == 1                            # not positive definite
    R = np.asarray(importlib.import_module("rgb-d-slam_amd.synth").euler_xyz(0.1, -0.2, 0.3), np.float32)
    inv = orc.inverse3(R.T.reshape(-1)).reshape(3, 3).T
    assert np.allclose(inv, R.T, atol=1e-6)


# ---- stage 4 -----------------------------------------------------------------------------------
def test_merge_depth_rules(orc, synth):
    """The five branches of MergeDepthOffKernel (D1/Kernel.cu:468-523) on hand-made pixels."""
    n, m = 1, 6
    calib = synth.calib_for(n, m)
    f = 5000.0
    #            no input  empty model  occluder   violation  average   saturate
    dv = np.array([0.0,      1.0,         1.0,       2.0,       1.004,    1.0])
    dp = np.array([1.0,      0.0,         1.5,       1.0,       1.0,      1.0])
    mk = np.array([15,       10,          20,        20,        13,       255], np.uint8)
    depth = np.rint(dv * f).astype(np.uint16).reshape(1, m)
    rgbd = np.zeros((n, m, 4), np.float32)
    rgbd[0, :, 3] = (dp / 100.0).astype(np.float32)
    rgb = np.full((n, m, 3), 0.25, np.float32)
    out, mask = orc.merge_depth_off(rgbd, depth, rgb, mk.reshape(1, m), calib)
    assert np.all(out[..., :3] == 0.25)                            # colour always replaced (Kernel.cu:463-465)
    w = out[0, :, 3] * 100.0
    assert np.isclose(w[0], 1.0) and mask[0, 0] == 15              # dv < 0.1: untouched
    assert np.isclose(w[1], 1.0) and mask[0, 1] == 11              # empty model: take the input
    assert np.isclose(w[2], 1.0) and mask[0, 2] == 11              # dv < dp - 0.2: occluder replaces
    assert np.isclose(w[3], 1.0) and mask[0, 3] == 10              # dv > dp + max(1,dv)*0.2: visibility violation
    assert np.isclose(w[4], (1.004 + 3 * 1.0) / 4.0, atol=1e-6) and mask[0, 4] == 14   # running average, count+1
    assert mask[0, 5] == 255                                        # saturates at 255


def plane_param(n_prim):
    # plane z = -2 (normal +z towards the camera), e1 = x, e2 = y, image centred on the view axis
    scale = np.float32(2.0 / 0.004)
    half = np.float32(n_prim * 0.004 / 2.0)
    return np.array([0, 0, 1, -2.0, scale, scale, -half, -half, 1, 0, 0, 0, 1, 0], np.float32)


def test_bump_codec_round_trip(orc, synth):
    """F4 -> R7: fusing a frame into a Bump image and decoding it returns the fused points to within
    half a quantisation step: 4 mm/60000/2 in plane, 0.25 mm along the normal."""
    n, m, n_prim = 48, 64, 256
    calib = synth.calib_for(n, m)
    rng = np.random.default_rng(5)
    depth = np.rint((2.0 + 0.01 * rng.standard_normal((n, m))) * 5000).astype(np.uint16)
    v = orc.vertex_map_kinect(depth, calib)
    rgbd = np.zeros((n, m, 4), np.float32)
    rgbd[..., :3] = rng.random((n, m, 3), dtype=np.float32)
    mask = np.full((n, m), 11, np.uint8)
    label = np.ones((n, m), np.uint8)
    bump = [np.zeros((n_prim, n_prim, 3), np.uint16)]
    rgbb = [np.zeros((n_prim, n_prim, 3), np.uint8)]
    maskb = [np.zeros((n_prim, n_prim), np.uint8)]
    claim = [np.zeros((n_prim, n_prim), np.uint64)]
    param = [plane_param(n_prim)]
    pose = np.eye(4, dtype=np.float32).T.reshape(-1)
    orc.update(bump, rgbb, maskb, param, claim, 1, v, rgbd, mask, label, pose)
    vd, cd, md = orc.vertex_from_bump(bump[0], rgbb[0], maskb[0], param[0])
    filled = maskb[0] == 11
    assert filled.sum() > 500
    # every fused point must be reproduced by the cell it fell in
    a = ((v[..., 0] - param[0][6]) * param[0][4] / 2.0)
    b = ((v[..., 1] - param[0][7]) * param[0][5] / 2.0)
    ci, cj = np.floor(a).astype(int), np.floor(b).astype(int)
    inside = (ci >= 0) & (ci < n_prim) & (cj >= 0) & (cj < n_prim)
    winners = 0
    for i, j in zip(*np.nonzero(inside)):
        p = vd[ci[i, j], cj[i, j]]
        err = np.abs(p - v[i, j])
        if err[0] <= 0.004 / 60000 / 2 + 1e-6 and err[1] <= 0.004 / 60000 / 2 + 1e-6 and err[2] <= 0.00025 + 1e-6:
            winners += 1
            assert np.abs(cd[ci[i, j], cj[i, j], :3] - rgbd[i, j, :3]).max() <= 0.5 / 255 + 1e-6
    assert winners == filled.sum()          # exactly one frame pixel wins each filled cell
    assert not vd[~filled].any() and not md[~filled].any()
    # a second, identical update changes nothing (gate mask_curr > cell mask, Primitive.cu:341)
    before = (bump[0].copy(), maskb[0].copy())
    orc.update(bump, rgbb, maskb, param, claim, 2, v, rgbd, mask, label, pose)
    assert np.array_equal(before[0], bump[0]) and np.array_equal(before[1], maskb[0])
    # out-of-view pixels (mask 0) erode the cell: 11 -> cleared to 10 (Primitive.cu:313-326)
    orc.update(bump, rgbb, maskb, param, claim, 3, v, rgbd, np.zeros_like(mask), label, pose)
    assert set(np.unique(maskb[0])) <= {0, 10} and not bump[0][maskb[0] == 10].any()


def test_update_plane_partition_is_a_partition(orc, synth):
    """Plane-partitioned fusion: shards [0,1) + [1,3) write exactly what [0,3) writes."""
    n, m, n_prim = 60, 80, 128
    calib = synth.calib_for(n, m)
    d, c = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    v = orc.vertex_map_kinect(d, calib)
    nm = orc.compute_normal(v)
    rgbd = orc.add_zbuff_off(d, orc.cpy_char3(c), calib)
    eq, centers = synth.plane_equations()
    pose = np.eye(4, dtype=np.float32).T.reshape(-1)
    label = orc.seed_labels(v, nm, pose, eq)
    mask = orc.init_mask(rgbd)

    def planes():
        P = []
        for k in range(3):
            p = np.zeros(14, np.float32)
            p[0:3], p[3], p[4:6] = eq[k, 0:3], eq[k, 9], 500.0
            p[6] = centers[k] @ eq[k, 3:6] - n_prim * 0.004 / 2
            p[7] = centers[k] @ eq[k, 6:9] - n_prim * 0.004 / 2
            p[8:11], p[11:14] = eq[k, 3:6], eq[k, 6:9]
            P.append(p)
        return ([np.zeros((n_prim, n_prim, 3), np.uint16) for _ in range(3)], [np.zeros((n_prim, n_prim, 3), np.uint8) for _ in range(3)],
                [np.zeros((n_prim, n_prim), np.uint8) for _ in range(3)], P, [np.zeros((n_prim, n_prim), np.uint64) for _ in range(3)])

    full = planes()
    orc.update(*full, 1, v, rgbd, mask, label, pose)
    part = planes()
    orc.update(*part, 1, v, rgbd, mask, label, pose, 0, 1)
    orc.update(*part, 1, v, rgbd, mask, label, pose, 1, 3)
    assert sum(int((mk > 10).sum()) for mk in full[2]) > 200
    for k in range(3):
        assert np.array_equal(full[0][k], part[0][k]) and np.array_equal(full[1][k], part[1][k]) and np.array_equal(full[2][k], part[2][k])


# ---- stage 5 -----------------------------------------------------------------------------------
def test_prediction_identity_warp_reproduces_model(orc, pair):
    """Rasterising the un-moved model mesh returns the model depth at every pixel whose quads
    survive trimming: the rendered value is the ray/mesh intersection depth."""
    c, v, rgbd = pair["calib"], pair["v0"], pair["rgbd0"]
    tex, zbuf = orc.raster(v, rgbd, c)
    v2, rgbd2 = orc.read_frame(tex, c)
    drawn = tex[..., 3] > 0
    assert np.array_equal(drawn, zbuf != np.uint64(0xFFFFFFFFFFFFFFFF))   # z-buffer keys: (depth bits, ~triangle id)
    assert drawn.mean() > 0.9
    assert np.abs(tex[..., 3][drawn] - rgbd[..., 3][drawn]).max() < 2e-6     # depth/100, f32 rounding only
    assert np.abs(tex[..., :3][drawn] - rgbd[..., :3][drawn]).max() < 1e-5
    assert np.abs(v2[drawn] - v[drawn]).max() < 2e-4
    assert not v2[~drawn].any() and not rgbd2[~drawn].any()
    # index-buffer path (QuadTrimFrame output) == inline trimming
    idx = orc.quad_trim_frame(v, 0.1)
    tex_i, _ = orc.raster(v, rgbd, c, index=idx)
    assert np.array_equal(tex, tex_i)
    kept = idx.any(axis=1)
    assert kept.mean() > 0.9                                           # quads across the plane creases are kept
    vh = v.copy()
    vh[100:120, 100:120] *= 2.0                                        # a block twice as far: its rim quads are trimmed
    cut = ~orc.quad_trim_frame(vh, 0.1).any(axis=1)
    assert 70 < cut.sum() < 400 and cut.reshape(479, 639)[99, 99:120].all() and not cut.reshape(479, 639)[105:115, 105:115].any()


def test_near_far_clipping(orc, synth):
    """Znear 0.5 / Zfar 10 (GP-demo.cpp:18-19) remove geometry from the prediction."""
    n, m = 16, 16
    calib = synth.calib_for(n, m)
    for z, seen in ((0.3, False), (1.0, True), (12.0, False)):
        d = flat_depth(n, m, z)
        v = orc.vertex_map_kinect(d, calib)
        rgbd = orc.add_zbuff_off(d, np.ones((n, m, 3), np.float32), calib)
        tex, _ = orc.raster(v, rgbd, calib, thresh=10.0)             # no trimming: clipping alone decides
        assert bool(tex[..., 3].any()) == seen


def test_transform_and_project_mask_round_trip(orc, synth, pair):
    """ProjectMask with the inverse of the warp brings a mask back onto itself (Kernel.cu:634-667)."""
    c, v, rgbd = pair["calib"], pair["v0"], pair["rgbd0"]
    mask = (np.arange(480 * 640).reshape(480, 640) % 251 + 1).astype(np.uint8)
    ident = np.eye(4, dtype=np.float32).T.reshape(-1)
    out = orc.project_mask(ident, c, mask, v)
    valid = v[..., 2] != 0
    assert np.array_equal(out[valid], mask[valid]) and not out[~valid].any()
    # a pure depth shift keeps pixels on their rays... a lateral shift moves them off the image edge -> 0
    P = np.eye(4, dtype=np.float32)
    P[0, 3] = 5.0
    out = orc.project_mask(P.T.reshape(-1), c, mask, v)
    assert not out[valid][-1000:].all()
    v2, r2 = orc.transform_vertex_map(P.T.reshape(-1), v, rgbd)
    assert np.allclose(v2[valid][:, 0], v[valid][:, 0] + 5.0) and np.array_equal(v2[~valid], v[~valid])
    assert np.allclose(r2[..., 3][valid], -v2[..., 2][valid] / 100.0)


def test_segment_grows_labels(orc, synth):
    """3 passes of SegmentKernel grow a seeded label by exactly 3 pixels on a plane (RG.cu:252-320)."""
    n, m = 40, 40
    calib = synth.calib_for(n, m)
    d = flat_depth(n, m, 2.0)
    v = orc.vertex_map_kinect(d, calib)
    nm = orc.compute_normal(v)
    eq = np.array([[0, 0, 1, 1, 0, 0, 0, 1, 0, -2.0]], np.float32)
    pose = np.eye(4, dtype=np.float32).T.reshape(-1)
    lab = np.zeros((n, m), np.uint8)
    lab[20, 20] = 1
    out = orc.segment(lab, v, nm, pose, eq)
    exp = np.zeros((n, m), np.uint8)
    exp[17:24, 17:24] = 1
    assert np.array_equal(out, exp)
    # a plane the pixels do not lie on is never adopted
    eq_far = eq.copy()
    eq_far[0, 9] = -3.0
    assert orc.segment(lab, v, nm, pose, eq_far).sum() == 1


# ---- golden vectors ---------------------------------------------------------------------------
def test_oracle_reproduces_golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from oracle import oracle as orc
    g = np.load(os.path.join(HERE, "golden", "slam_96x128.npz"))
    out = mg.run_oracle(orc, g["calib"], g["depth"], g["bgr"], g["equations"], g["centers"])
    for k, val in out.items():
        a, b = np.asarray(val), g[k]
        if a.dtype.kind == "f":
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
        else:
            assert np.array_equal(a, b), k
    assert list(g["status"]) == [2, 1, 1, 1, 1] and int(g["align_ok"][0]) == 1


def test_oracle_reproduces_detection_golden(orc):
    """oracle (plane detection) == tests/golden/detect_96x128.npz, recomputed from the slam fixture's inputs."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_detect", os.path.join(here, "make_golden_detect.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = np.load(os.path.join(here, "detect_96x128.npz"))
    got = mod.run_oracle(orc, np.load(os.path.join(here, "slam_96x128.npz")))
    assert set(got) == set(want.files)
    for k in want.files:
        assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k
    assert want["equations"].shape == (3, 10)


# ---- plane detection: closed-form answers ---------------------------------------------
def test_detect_canny_step_and_flat(orc):
    img = np.full((20, 30), 40, np.uint8)
    assert not orc.canny(img).any()                                  # no gradient, no edge
    img[:, 15:] = 200                                                # |dx| = 4 * 160 = 640 > 120 on columns 14 and 15
    e = orc.canny(img)
    cols = np.nonzero(e.any(axis=0))[0]
    # equal magnitudes on both sides of the step: "m > left && m >= right" keeps exactly the left one
    assert list(cols) == [14] and (e[:, 14] == 255).all()
    weak = np.full((20, 30), 40, np.uint8)
    weak[:, 15:] = 50                                                # |dx| = 40: above low (30), below high (120), no strong seed
    assert not orc.canny(weak).any()
    same = orc.canny(img, 30.0, 120.0)
    assert np.array_equal(same, e)                                   # RG.cpp passes (120, 30): OpenCV swaps them


def test_detect_blobs_and_erode(orc):
    b = np.zeros((6, 8), np.uint8)
    b[0, 0:3] = 1; b[1, 2] = 1                                       # blob A: 4 pixels, first pixel index 0
    b[3:5, 5:7] = 1                                                  # blob B: 4 pixels, first pixel (3,5) = 29
    b[5, 7] = 1                                                      # touches B only diagonally: its own blob (4-connectivity)
    blob, sizes = orc.find_blobs(b)
    assert sizes[0, 0] == 4 and sizes[3, 5] == 4 and sizes[5, 7] == 1 and sizes.sum() == 9
    assert (blob[b > 0] >= 0).all() and (blob[b == 0] == -1).all()
    assert blob[1, 2] == 0 and blob[4, 6] == 29 and blob[5, 7] == 47
    full = np.full((7, 7), 255, np.uint8)
    assert (orc.erode_ellipse5(full) == 255).all()                   # pixels outside the image do not constrain
    full[0, 0] = 0
    er = orc.erode_ellipse5(full)
    assert er[0, 0] == 0 and er[0, 2] == 0 and er[2, 0] == 0 and er[1, 2] == 0 and er[1, 1] == 0
    assert er[2, 1] == 255 and er[2, 2] == 255 and er[0, 3] == 255   # rows +-2 of the ellipse hold the centre column only


def test_detect_plane_fit_closed_form(orc):
    """A fronto-parallel wall at z = -2 (camera looks down -z), unit normals (0,0,1), identity pose: the median fit
    must return exactly n = (0,0,1), d = -2, the basis the reference's rule picks, and the mean point."""
    n, m = 40, 50
    v = np.zeros((n, m, 3), np.float32)
    v[..., 0] = np.linspace(-1, 1, m, dtype=np.float32)[None, :]
    v[..., 1] = np.linspace(-0.8, 0.8, n, dtype=np.float32)[:, None]
    v[..., 2] = -2.0
    nm = np.zeros((n, m, 3), np.float32)
    nm[..., 2] = 1.0
    nm[0, :] = 0.0                                                   # a row without normals: its zeros stay in the median lists
    blob = np.zeros((n, m), np.int32)
    sizes = np.zeros((n, m), np.int32)
    sizes[0, 0] = n * m
    I16 = np.eye(4, dtype=np.float32).reshape(-1)
    nb, eq, ce, lab = orc.fit_planes(blob, sizes, v, nm, I16, np.zeros((n, m), np.uint8), min_size=100, min_normals=100)
    assert nb == 1 and (lab == 1).all()
    assert np.array_equal(eq[0], np.float32([0, 0, 1, 1, 0, 0, 0, 1, 0, -2]))   # e1 = (n2, 0, -n0), e2 = n x e1
    assert np.allclose(ce[0], v.reshape(-1, 3).mean(0), atol=1e-4)
    # too small / too few normals: no plane
    assert orc.fit_planes(blob, sizes, v, nm, I16, np.zeros((n, m), np.uint8), min_size=n * m, min_normals=100)[0] == 0
    assert orc.fit_planes(blob, sizes, v, nm, I16, np.zeros((n, m), np.uint8), min_size=100, min_normals=n * m)[0] == 0


def test_icp_system_matches_literal_jacobian_form_in_float64(orc, pair):
    """Independent restatement of D1/ICP.cu:164-241 against the oracle: the oracle (and the HIP kernel) drop the products
    with the literal 0 / 1 entries of JD, JRot and JProj and keep the remaining operations in order; here the three
    Jacobians are built as the full matrices the reference writes down (JD, JRot: 3 x 6, column k = entries k*3..k*3+2;
    JProj: entries (0,2,4) and (1,3,5) pair with rgb_val[0] and rgb_val[1]) and multiplied out in float64.  A sign, index
    or dropped-term error in the simplified form would show as an O(1) difference; float32 rounding is ~1e-6."""
    c = pair["calib"].astype(np.float64)
    fx, fy, cx, cy = c[0], c[1], c[2], c[3]
    R, t = np.eye(3), np.zeros(3)
    I32, z32 = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    dist_thres, angle_thres = 0.05, SIN20
    buf, corr = orc.icp_match(c[:4].astype(np.float32), I32, z32, pair["v1"], pair["n1"], pair["rgb1"], pair["gx1"], pair["gy1"],
                              pair["v0"], pair["n0"], pair["rgbd0"], dist_thres, angle_thres, 1)
    n, m = pair["v0"].shape[:2]
    vp = pair["v0"].reshape(-1, 3).astype(np.float64); npv = pair["n0"].reshape(-1, 3).astype(np.float64)
    rgbp = pair["rgbd0"].reshape(-1, 4).astype(np.float64)
    vc = pair["v1"].reshape(-1, 3).astype(np.float64); nc = pair["n1"].reshape(-1, 3).astype(np.float64)
    rgbc = pair["rgb1"].reshape(-1, 3).astype(np.float64)
    gx = pair["gx1"].reshape(-1).astype(np.float64); gy = pair["gy1"].reshape(-1).astype(np.float64)
    # searchGauss (ICP.cu:72-146), every test evaluated independently of the oracle
    d = vp @ R.T + t
    nn = npv @ R.T
    with np.errstate(divide="ignore", invalid="ignore"):
        pu = np.clip(np.rint(np.nan_to_num(d[:, 0] / np.abs(d[:, 2])) * fx + cx), 0, m - 1).astype(np.int64)
        pv = np.clip(np.rint(np.nan_to_num(d[:, 1] / np.abs(d[:, 2])) * fy + cy), 0, n - 1).astype(np.int64)
    pidx = pv * m + pu
    s = vc[pidx]; ncur = nc[pidx]
    ok = npv.any(axis=1) & ncur.any(axis=1)
    ok &= ~(np.linalg.norm(d - s, axis=1) > dist_thres)
    ok &= ~(np.linalg.norm(np.cross(ncur, nn), axis=1) > angle_thres)
    mine = np.where(ok, pidx, -1)
    agree = (mine == corr.reshape(-1)).mean()
    assert agree > 0.9999, agree               # float64 vs float32 can only differ on rounding ties / threshold edges
    sel = ok & (corr.reshape(-1) == mine)
    d, nn, s = d[sel], nn[sel], s[sel]
    g = np.stack([gx[pidx[sel]] / 8.0, gy[pidx[sel]] / 8.0], 1)
    g2 = rgbc[pidx[sel]].sum(1) / 3.0 - rgbp[sel, :3].sum(1) / 3.0
    w = 0.0012 / (0.0012 + 0.0019 * (-s[:, 2] - 0.4) ** 2)      # s[3] = -vcurr.z
    K = d.shape[0]
    zero, one = np.zeros(K), np.ones(K)
    def skew2(a):      # columns 3..5 of JD / JRot as the reference writes them: 2 * [[0, a2, -a1], [-a2, 0, a0], [a1, -a0, 0]]
        return 2.0 * np.stack([np.stack([zero, a[:, 2], -a[:, 1]], 1), np.stack([-a[:, 2], zero, a[:, 0]], 1),
                               np.stack([a[:, 1], -a[:, 0], zero], 1)], 1)          # [K, row, col]
    JD = np.concatenate([np.broadcast_to(np.eye(3), (K, 3, 3)), skew2(d)], 2)        # [K, 3, 6]
    JRot = np.concatenate([np.zeros((K, 3, 3)), skew2(nn)], 2)
    e = s - d
    row = np.empty((K, 7)); row_rgb = np.empty((K, 7))
    row[:, :6] = w[:, None] * (-np.einsum("kr,krc->kc", nn, JD) + np.einsum("krc,kr->kc", JRot, e))
    row[:, 6] = -w * np.einsum("kr,kr->k", nn, e)
    az = np.abs(d[:, 2])
    JProj = np.zeros((K, 2, 3))
    JProj[:, 0, 0] = fx / az; JProj[:, 1, 1] = fy / az
    JProj[:, 0, 2] = -d[:, 0] * fx / d[:, 2] ** 2; JProj[:, 1, 2] = -d[:, 1] * fy / d[:, 2] ** 2
    gp = np.einsum("ka,kab->kb", g, JProj)                                           # (rgb_val[0:2]) * JProj
    lam = 0.03
    row_rgb[:, :6] = (lam * w)[:, None] * np.einsum("kr,krc->kc", gp, JD)
    row_rgb[:, 6] = -lam * w * g2
    want, scale = [], []
    for k in range(6):
        for l in range(k, 7):
            terms = row[:, k] * row[:, l] + row_rgb[:, k] * row_rgb[:, l]
            want.append(terms.sum()); scale.append(np.abs(terms).sum())
    got = buf.astype(np.float64).sum(axis=1)
    # pixels on which the float64 tests disagree with the float32 ones are left out on both sides only approximately:
    # their share is < 5e-4 of the terms
    err = np.abs(got - np.array(want)) / (np.array(scale) + 1e-300)
    print("literal-Jacobian KAT: correspondence agreement %.6f, max |diff| / sum|terms| = %.3g" % (agree, err.max()))
    assert err.max() <= 1e-6


def test_normals_match_float64_four_triangle_average(orc):
    """Independent restatement of NormalProcess (D1/Kernel.cu:122-300) in vectorised float64 on a rough surface with
    holes: the four triangles (down, right), (right, up), (up, left), (left, down) around a pixel, each normalised when
    both of its neighbours have z != 0 and its cross product is not zero, averaged over the triangles that counted,
    normalised again, negated for inverse = true; border pixels and pixels with z == 0 get zeros."""
    rng = np.random.default_rng(11)
    n, m = 40, 56
    ii, jj = np.meshgrid(np.arange(n), np.arange(m), indexing="ij")
    z = -(1.5 + 0.2 * np.sin(ii / 5.0) * np.cos(jj / 7.0) + 0.01 * rng.standard_normal((n, m)))
    v = np.stack([(jj - 28.0) / 60.0 * -z, (ii - 20.0) / 60.0 * -z, z], -1).astype(np.float32)
    holes = rng.random((n, m)) < 0.08
    v[holes] = 0.0
    for inverse in (True, False):
        got = orc.compute_normal(v, inverse)
        V = v.astype(np.float64)
        want = np.zeros((n, m, 3))
        c = V[1:-1, 1:-1]
        dn, up, rt, lf = V[2:, 1:-1], V[:-2, 1:-1], V[1:-1, 2:], V[1:-1, :-2]
        acc = np.zeros_like(c); cnt = np.zeros(c.shape[:2])
        for p2, p3 in ((dn, rt), (rt, up), (up, lf), (lf, dn)):
            cr = np.cross(p2 - c, p3 - c)
            nn = (cr * cr).sum(-1)
            use = (p2[..., 2] != 0) & (p3[..., 2] != 0) & (nn != 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                acc += np.where(use[..., None], cr / np.sqrt(nn)[..., None], 0.0)
            cnt += use
        with np.errstate(divide="ignore", invalid="ignore"):
            avg = acc / cnt[..., None]
            nrm = np.sqrt((avg * avg).sum(-1))
            res = np.where(((cnt > 0) & (nrm != 0) & (c[..., 2] != 0))[..., None], avg / nrm[..., None], 0.0)
        want[1:-1, 1:-1] = -res if inverse else res
        # a float32 cross product of nearly parallel edges can round to exactly zero where the float64 one does not (and
        # the triangle then counts or not): compare where both agree on "has a normal", which must be nearly everywhere
        has_g, has_w = np.abs(got).sum(-1) > 0, np.abs(want).sum(-1) > 0
        assert (has_g == has_w).mean() > 0.999
        both = has_g & has_w
        assert both.sum() > 0.6 * n * m
        assert np.abs(got[both] - want[both]).max() < 2e-3     # cancellation in the cross products of a 1 cm-rough surface
        assert np.abs(got[both] - want[both]).mean() < 2e-5
        assert not got[0].any() and not got[-1].any() and not got[:, 0].any() and not got[:, -1].any()


def test_bump_encode_and_decode_match_float64_formulas(orc, synth):
    """Independent float64 restatement of UpdateKernel's write (D1/Primitive.cu:285-349) and of VertexFromBumpKernel
    (Primitive.cu:31-93) on a sparse frame (one pixel per cell, so the winner rule is not involved): cell index, the three
    Bump words, the colour bytes, the mask; and the decoded vertex / colour / packed mask of every written cell."""
    rng = np.random.default_rng(5)
    N = 96
    eq, centers = synth.plane_equations()
    E, ctr = eq[1].astype(np.float64), centers[1].astype(np.float64)
    res = 0.004
    scale = 2.0 / res
    shift = np.array([ctr @ E[3:6] - N * res / 2.0, ctr @ E[6:9] - N * res / 2.0])
    param = np.array([E[0], E[1], E[2], E[9], scale, scale, shift[0], shift[1], E[3], E[4], E[5], E[6], E[7], E[8]], np.float32)
    P64 = param.astype(np.float64)
    # one point per chosen cell, well inside it, a few mm off the plane; world -> camera by a small rigid motion
    n, m = 24, 32
    cells = rng.choice(N * N, n * m, replace=False)
    ci, cj = cells // N, cells % N
    a = (ci + rng.uniform(0.15, 0.85, n * m)) * 2.0 / P64[4] + P64[6]
    b = (cj + rng.uniform(0.15, 0.85, n * m)) * 2.0 / P64[5] + P64[7]
    off = rng.uniform(-0.02, 0.02, n * m)
    world = a[:, None] * P64[8:11] + b[:, None] * P64[11:14] + (P64[3] + off)[:, None] * P64[0:3]
    Rw = synth.euler_xyz(0.02, -0.015, 0.01); tw = np.array([0.01, -0.02, 0.015])      # camera-to-world
    cam = (world - tw) @ Rw                                                              # = R^T (p - t)
    pose = np.eye(4); pose[:3, :3] = Rw; pose[:3, 3] = tw
    pose_cm = pose.T.reshape(-1).astype(np.float32)                                      # column-major, as pose[0..15] is read
    vmap = cam.reshape(n, m, 3).astype(np.float32)
    rgbd = rng.random((n, m, 4)).astype(np.float32)
    mask = np.full((n, m), 50, np.uint8)
    label = np.ones((n, m), np.uint8)
    bumps = [np.zeros((N, N, 3), np.uint16)]; rgbs = [np.zeros((N, N, 3), np.uint8)]; masks = [np.zeros((N, N), np.uint8)]
    claims = [np.zeros((N, N), np.uint64)]
    orc.update(bumps, rgbs, masks, [param], claims, 1, vmap, rgbd, mask, label, pose_cm)
    # ---- encode, float64, from the float32 inputs the oracle saw ----
    Pm = pose_cm.astype(np.float64).reshape(4, 4).T
    pt = vmap.reshape(-1, 3).astype(np.float64) @ Pm[:3, :3].T + Pm[:3, 3]
    err = pt @ P64[0:3] - P64[3]
    pa = (pt @ P64[8:11] - P64[6]) * P64[4] / 2.0
    pb = (pt @ P64[11:14] - P64[7]) * P64[5] / 2.0
    ii, jj = np.floor(pa).astype(np.int64), np.floor(pb).astype(np.int64)
    assert np.array_equal(ii, ci) and np.array_equal(jj, cj)
    assert int((masks[0] == 50).sum()) == n * m and np.array_equal(np.flatnonzero(masks[0].reshape(-1) == 50), np.sort(cells))
    want_b = np.stack([np.rint((pa - ii) * 60000.0), np.rint((pb - jj) * 60000.0), np.rint((err + 15.0) * 2000.0)], 1)
    got_b = bumps[0][ii, jj].astype(np.float64)
    # float32: the in-plane coordinate (~1 m, a three-term dot product: a few ulps of 1.2e-7) times scale / 2 = 250 cells per
    # metre times 60000 counts per cell is ~2 counts per ulp; the distance word sees 2000 counts per metre: < 0.01 count
    diff = np.abs(got_b - want_b)
    assert diff[:, :2].max() <= 8 and diff[:, :2].mean() < 1.5 and diff[:, 2].max() <= 1 and (diff[:, 2] == 0).mean() > 0.98
    want_c = np.rint(rgbd.reshape(-1, 4)[:, :3].astype(np.float64) * 255.0)
    assert np.abs(rgbs[0][ii, jj].astype(np.float64) - want_c).max() <= 1 and (rgbs[0][ii, jj] == want_c).mean() > 0.99
    # ---- decode, float64, from the stored words ----
    v, c, mk = orc.vertex_from_bump(bumps[0], rgbs[0], masks[0], param)
    B = bumps[0].astype(np.float64)
    gi, gj = np.meshgrid(np.arange(N, dtype=np.float64), np.arange(N, dtype=np.float64), indexing="ij")
    x = (gi + B[..., 0] / 60000.0) * 2.0 / P64[4] + P64[6]
    y = (gj + B[..., 1] / 60000.0) * 2.0 / P64[5] + P64[7]
    d = P64[3] + (B[..., 2] / 2000.0 - 15.0)
    V = x[..., None] * P64[8:11] + y[..., None] * P64[11:14] + d[..., None] * P64[0:3]
    w = masks[0] == 50
    assert np.abs(v[w] - V[w]).max() < 2e-6 * 4                       # float32 evaluation of numbers of size ~2
    assert np.abs(c[w][:, :3] - rgbs[0][w] / 255.0).max() < 1e-7 and np.all(c[w][:, 3] == 1.0)
    assert np.abs(mk[w][:, :3] - (10.0 + V[w]) / 20.0).max() < 1e-6 and np.allclose(mk[w][:, 3], 50 / 255.0, atol=1e-7)
    assert not v[~w].any() and not c[~w].any() and not mk[~w].any()   # empty cells decode to zeros (Primitive.cu:47-60)
    # and the decoded points are the fused points to within the quantisation of the code: 1/60000 of a cell in the plane,
    # half of 1/2000 m = 0.25 mm along the normal
    assert np.abs(V[ii, jj] - pt).max() < 2.6e-4


def test_demostep2_decode_matches_float64_formula_and_d1():
    """D2/Primitive.cu:31-179 restated (oracle.vertex_from_bump_d2) against the formula written out in float64 and against
    the D1 decode on cells both call non-empty."""
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    n, m, lvl = 48, 40, 2
    bump = rng.integers(0, 65536, (n, m, 3)).astype(np.uint16)
    rgbb = rng.integers(0, 256, (n, m, 3)).astype(np.uint8)
    mask = rng.integers(0, 30, (n, m)).astype(np.uint16)
    param = np.array([0.0, 0.6, 0.8, -1.7, 500.0, 480.0, -0.9, -1.1, 1.0, 0.0, 0.0, 0.0, 0.8, -0.6], np.float32)
    v = orc.vertex_from_bump_d2(bump, rgbb, mask, param, lvl, 0)
    p = param.astype(np.float64)
    ii, jj = np.meshgrid(np.arange(0, n, lvl)[:n // lvl], np.arange(0, m, lvl)[:m // lvl], indexing="ij")
    b = bump[ii, jj].astype(np.float64)
    x = (ii + b[..., 0] / 60000.0) * 2.0 / p[4] + p[6]
    y = (jj + b[..., 1] / 60000.0) * 2.0 / p[5] + p[7]
    d = p[3] + (b[..., 2] / 2000.0 - 15.0)
    ref = np.stack([x * p[8] + y * p[11] + d * p[0], x * p[9] + y * p[12] + d * p[1], x * p[10] + y * p[13] + d * p[2]], axis=-1)
    ref[mask[ii, jj] < 11] = 0.0
    assert np.abs(v - ref).max() < 2e-5
    assert ((v == 0).all(axis=2) == (mask[ii, jj] < 11)).all()
    v1, c1 = orc.vertex_from_bump_d2(bump, rgbb, np.minimum(mask, 255).astype(np.uint8), param, 1, 1)
    d1v, d1c, _ = orc.vertex_from_bump(bump, rgbb, mask.astype(np.uint8), param, 1)
    keep = mask >= 11
    assert np.array_equal(v1[keep], d1v[keep]) and np.array_equal(c1[keep], d1c[keep])
    assert (c1[~keep] == 0).all() and (v1[~keep] == 0).all()


def test_solve_is_insensitive_to_the_cholesky_operation_order():
    """Eigen's llt().solve() is third-party code that is not in the reference tree; the oracle restates it as a row
    Cholesky.  Whatever order Eigen's unblocked kernel really uses (its published llt_inplace::unblocked goes column by
    column), the 1e-5 pose bar does not depend on it: on the normal equations of a real 640x480 pair (condition number
    ~150) the float32 solutions in row order (the oracle's), in column order and the float64 solution agree to a few 1e-8
    per increment, three orders of magnitude below the bar."""
    import importlib
    from oracle import oracle as orc
    synth = importlib.import_module("rgb-d-slam_amd.synth")
    calib = synth.TUM_CALIB
    d0, c0 = synth.render(np.eye(3), np.zeros(3), noise_sigma=0.001, rng=np.random.default_rng(1))
    R1, t1 = synth.pair_offset_pose()
    d1, c1 = synth.render(R1, t1, noise_sigma=0.001, rng=np.random.default_rng(2))
    v0 = orc.vertex_map_kinect(d0, calib); n0 = orc.compute_normal(v0); rgbd0 = orc.add_zbuff_off(d0, orc.cpy_char3(c0), calib)
    v1 = orc.vertex_map_kinect(d1, calib); n1 = orc.compute_normal(v1); rgb1 = orc.cpy_char3(c1)
    _, gx1, gy1 = orc.intensity_sobel(rgb1)
    I, z = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    prm = orc.icp_params()
    f32 = np.float32

    def dot(a, b):
        s = f32(0)
        for u, v in zip(a, b):
            s = f32(s + f32(u * v))
        return s
    for fact in (4, 2, 1):
        A, b, _ = orc.estimate_system(calib[:4], I, z, v1, n1, rgb1, gx1, gy1, v0, n0, rgbd0, prm.distThres, prm.angleThres, fact)
        A = np.array(A, f32).reshape(6, 6); b = np.array(b, f32)
        _, x_row = orc.llt_solve6(A.reshape(-1), b)
        L = A.copy()
        for k in range(6):                                                   # diagonal entry, then the column below it
            L[k, k] = np.sqrt(f32(L[k, k] - dot(L[k, :k], L[k, :k])))
            for i in range(k + 1, 6):
                L[i, k] = f32(f32(L[i, k] - dot(L[i, :k], L[k, :k])) / L[k, k])
        y = np.zeros(6, f32)
        for i in range(6):
            y[i] = f32(f32(b[i] - dot(L[i, :i], y[:i])) / L[i, i])
        x_col = np.zeros(6, f32)
        for i in range(5, -1, -1):
            x_col[i] = f32(f32(y[i] - dot(L[i + 1:, i], x_col[i + 1:])) / L[i, i])
        x64 = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
        assert np.linalg.cond(A.astype(np.float64)) < 1e4
        assert 1e-3 < np.abs(x64).max() < 0.1                                # a real increment, not a converged one
        assert np.abs(x_row - x64).max() < 1e-7 and np.abs(x_col - x64).max() < 1e-7 and np.abs(x_row - x_col).max() < 1e-7


def test_u16_label_images_give_the_labels_of_the_u8_loop(orc, synth):
    """The oracle's 16-bit label form (the build's extension for models of more than 255 planes; the reference keeps unsigned
    char, Frame.h:197) is the same statements on a wider element: on a model the u8 image can hold, both widths leave the same
    labels, model maps and Bump images -- and a model whose visible planes are numbered beyond 255 keeps them."""
    n, m = 96, 128
    calib = synth.calib_for(n, m)
    seq = synth.sequence(5, n, m, calib, noise_sigma=0.001)
    eq3, c3 = synth.plane_equations()

    def run(wide, eq, ce):
        s = orc.Slam(n, m, calib, N_prim=128, M_prim=128, max_planes=len(eq))
        s.set_finish(False)
        if wide:
            s.set_label16(True)
        for f, (d, c) in enumerate(seq):
            s.step(d, c)
            if f == 1:
                s.set_planes(eq, ce)
        nb = len(eq)
        out = (s.buffer("Label").copy(), s.buffer("Mask").copy(), s.buffer("Maskb%d" % (nb - 1)).copy(), s.buffer("Bump%d" % (nb - 1)).copy())
        s.close()
        return out
    l8, m8, mb8, b8 = run(False, eq3, c3)
    l16, m16, mb16, b16 = run(True, eq3, c3)
    assert l8.dtype == np.uint8 and l16.dtype == np.uint16
    assert np.array_equal(l8.astype(np.int32), l16.astype(np.int32)) and (l8 > 0).mean() > 0.5
    assert np.array_equal(m8, m16) and np.array_equal(mb8, mb16) and np.array_equal(b8, b16) and (mb8 > 10).sum() > 50
    # 300 planes: 297 that no pixel satisfies (pushed 0.4 m+ behind the walls), then the three walls
    eq = np.zeros((300, 10), np.float32)
    ce = np.zeros((300, 3), np.float32)
    for k in range(297):
        off = np.float32(0.4 + 0.01 * (k // 3))
        eq[k] = eq3[k % 3]
        eq[k, 9] += off
        ce[k] = c3[k % 3] + off * eq3[k % 3][:3]
    eq[297:], ce[297:] = eq3, c3
    lw, mw, mbw, bw = run(True, eq, ce)
    assert np.array_equal(np.where(lw > 0, lw.astype(np.int32) - 297, 0), l8.astype(np.int32))
    assert np.array_equal(mw, m8) and np.array_equal(mbw, mb8) and np.array_equal(bw, b8)
