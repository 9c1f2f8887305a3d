"""GPU parity: every C-ABI entry point against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): correspondence / mask / label / index work bit-exact; here the
floating-point maps are ALSO required bit-exact, because the HIP kernels and the oracle evaluate
the same IEEE operations in the same order (no FMA contraction on either side)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    if a.dtype.kind == "f":
        ok = (a == b) | (np.isnan(a) & np.isnan(b))
        assert ok.all(), "mismatch at %d of %d elements, max abs diff %g" % ((~ok).sum(), ok.size, np.nanmax(np.abs(a - b)))
    else:
        assert np.array_equal(a, b), "mismatch at %d of %d elements" % ((a != b).sum(), a.size)


@pytest.fixture(scope="module")
def ctx(pkg, synth):
    c = pkg.Context(0)
    c.set_calibration_matrix(synth.TUM_CALIB)
    yield c
    c.close()


def pose16(R, t):
    P = np.eye(4, dtype=np.float32)
    P[:3, :3] = R
    P[:3, 3] = t
    return P.T.reshape(-1).copy()  # column-major


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(480, 640), (37, 53), (5, 130)])
def test_preprocess(ctx, orc, synth, shape):
    n, m = shape
    calib = synth.calib_for(n, m)
    ctx.set_calibration_matrix(calib)
    d, c = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    d[::7, ::5] = 0  # holes
    vmap = torch.empty(n, m, 3, device="cuda")
    rgb = torch.empty(n, m, 3, device="cuda")
    nmap = torch.empty(n, m, 3, device="cuda")
    ctx.vertex_map_kinect(vmap, dev(d), 2 * m, n, m)
    ctx.gpu_cpy_char3(dev(c), 3 * m, rgb, n, m)
    ctx.compute_normal(nmap, vmap, n, m, True)
    ctx.synchronize()
    v_ref = orc.vertex_map_kinect(d, calib)
    same(host(vmap), v_ref)
    same(host(rgb), orc.cpy_char3(c))
    same(host(nmap), orc.compute_normal(v_ref))
    ctx.compute_normal(nmap, vmap, n, m, False)
    same(host(nmap), orc.compute_normal(v_ref, False))
    img = torch.empty(n, m, device="cuda")
    gx = torch.empty(n, m, device="cuda")
    gy = torch.empty(n, m, device="cuda")
    ctx.intensity_sobel(rgb, img, gx, gy, n, m)
    ri, rx, ry = orc.intensity_sobel(host(rgb))
    same(host(img), ri)
    same(host(gx), rx)
    same(host(gy), ry)
    ctx.set_calibration_matrix(synth.TUM_CALIB)


def test_pitched_inputs(ctx, orc, synth):
    n, m = 48, 70
    calib = synth.calib_for(n, m)
    ctx.set_calibration_matrix(calib)
    d, c = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    dp = np.zeros((n, 96), np.uint16)
    dp[:, :m] = d
    cp = np.zeros((n, 256), np.uint8)
    cp[:, :3 * m] = c.reshape(n, 3 * m)
    vmap = torch.empty(n, m, 3, device="cuda")
    rgb = torch.empty(n, m, 3, device="cuda")
    ctx.vertex_map_kinect(vmap, dev(dp), 2 * 96, n, m)
    ctx.gpu_cpy_char3(dev(cp), 256, rgb, n, m)
    same(host(vmap), orc.vertex_map_kinect(d, calib))
    same(host(rgb), orc.cpy_char3(c))
    ctx.set_calibration_matrix(synth.TUM_CALIB)


@pytest.mark.parametrize("fact", [4, 2, 1])
def test_icp_system_bit_exact(ctx, orc, pair, fact):
    n, m = 480, 640
    calib = pair["calib"]
    R = np.eye(3, dtype=np.float32).T.reshape(-1)
    t = np.zeros(3, np.float32)
    gr, gc = orc.icp_grid(n, m, 1)
    ctx.alloc_buffers(gr, gc)
    corr = torch.empty(n, m, dtype=torch.int32, device="cuda")
    A, b = ctx.estimate_system(dev(R), dev(t), dev(pair["v1"]), dev(pair["n1"]), dev(pair["rgb1"]), dev(pair["gx1"]),
                               dev(pair["gy1"]), dev(pair["v0"]), dev(pair["n0"]), dev(pair["rgbd0"]), dev(calib[:4]),
                               0.05, float(np.sin(np.deg2rad(20.0))), n, m, fact, corr)
    A_ref, b_ref, _ = orc.estimate_system(calib[:4], R, t, pair["v1"], pair["n1"], pair["rgb1"], pair["gx1"], pair["gy1"],
                                          pair["v0"], pair["n0"], pair["rgbd0"], 0.05, float(np.sin(np.deg2rad(20.0))), fact)
    _, corr_ref = orc.icp_match(calib[:4], R, t, pair["v1"], pair["n1"], pair["rgb1"], pair["gx1"], pair["gy1"],
                                pair["v0"], pair["n0"], pair["rgbd0"], 0.05, float(np.sin(np.deg2rad(20.0))), fact)
    same(host(corr), corr_ref)                      # correspondence indices bit-exact
    assert (corr_ref >= 0).sum() > 1000
    same(np.array(A, np.float32).reshape(6, 6), A_ref)   # 27 sums bit-identical (reference order)
    same(np.array(b, np.float32), b_ref)


def test_icp_system_rotated_estimate(ctx, orc, pair):
    """A non-identity Rcurr/tcurr exercises the column-major indexing of ICP.cu:98-104."""
    n, m = 480, 640
    calib = pair["calib"]
    Rm = np.linalg.inv(pair["R1"]).astype(np.float32)
    R = Rm.T.reshape(-1).copy()
    t = (-Rm @ pair["t1"]).astype(np.float32)
    gr, gc = orc.icp_grid(n, m, 1)
    ctx.alloc_buffers(gr, gc)
    A, b = ctx.estimate_system(dev(R), dev(t), dev(pair["v1"]), dev(pair["n1"]), dev(pair["rgb1"]), dev(pair["gx1"]),
                               dev(pair["gy1"]), dev(pair["v0"]), dev(pair["n0"]), dev(pair["rgbd0"]), dev(calib[:4]),
                               0.05, 0.342, n, m, 2)
    A_ref, b_ref, _ = orc.estimate_system(calib[:4], R, t, pair["v1"], pair["n1"], pair["rgb1"], pair["gx1"], pair["gy1"],
                                          pair["v0"], pair["n0"], pair["rgbd0"], 0.05, 0.342, 2)
    same(np.array(A, np.float32).reshape(6, 6), A_ref)
    same(np.array(b, np.float32), b_ref)


def test_align_device_loop(ctx, pkg, orc, pair):
    """Whole Gauss-Newton loop on the device vs RG::AlignFrameGaussNewton restated on the CPU."""
    n, m = 480, 640
    calib = pair["calib"]
    gr, gc = orc.icp_grid(n, m, 1)
    ctx.alloc_buffers(gr, gc)
    I = np.eye(3, dtype=np.float32).reshape(-1)
    z = np.zeros(3, np.float32)
    state = dev(np.concatenate([I, z, I, z]).astype(np.float32))
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    trace = torch.zeros(16, 33, device="cuda")
    prm = pkg.icp_params()
    ctx.align_gauss_newton(prm, dev(pair["v1"]), dev(pair["n1"]), dev(pair["rgb1"]), dev(pair["gx1"]), dev(pair["gy1"]),
                           dev(pair["v0"]), dev(pair["n0"]), dev(pair["rgbd0"]), n, m, state, status, trace)
    ctx.synchronize()
    ok, R, t, tr = orc.align_gauss_newton(orc.icp_params(), calib[:4], pair["v1"], pair["n1"], pair["rgb1"], pair["v0"],
                                          pair["n0"], pair["rgbd0"], I, z, I, z)
    st = host(status)
    assert ok and st[0] == 1
    assert st[1] == len(tr)
    same(host(trace)[:len(tr)], tr)                 # every iteration's sums and solution
    s = host(state)
    same(s[:9], R)
    same(s[9:12], t)
    # and the pose is the known offset up to depth quantisation
    assert np.linalg.norm(s[:9].reshape(3, 3).T - pair["R1"]) < 5e-5
    assert np.linalg.norm(s[9:12] - pair["t1"]) < 5e-5


def test_align_track_loss(ctx, pkg, orc, pair):
    n, m = 480, 640
    gr, gc = orc.icp_grid(n, m, 1)
    ctx.alloc_buffers(gr, gc)
    I = np.eye(3, dtype=np.float32).reshape(-1)
    z = np.zeros(3, np.float32)
    state = dev(np.concatenate([I, z, I, z]).astype(np.float32))
    status = torch.ones(4, dtype=torch.int32, device="cuda")
    empty = torch.zeros(n, m, 3, device="cuda")   # no normals at all -> singular system
    ctx.align_gauss_newton(pkg.icp_params(), empty, empty, empty, dev(pair["gx1"]), dev(pair["gy1"]),
                           dev(pair["v0"]), dev(pair["n0"]), dev(pair["rgbd0"]), n, m, state, status)
    ctx.synchronize()
    assert host(status)[0] == 0
    same(host(state)[:12], np.concatenate([I, z]))   # pose kept (GP-demo.cpp:189-196)
    ok, _, _, _ = orc.align_gauss_newton(orc.icp_params(), pair["calib"][:4], host(empty), host(empty), host(empty),
                                         pair["v0"], pair["n0"], pair["rgbd0"], I, z, I, z)
    assert not ok


def _pose_block(torch, I, z):
    """rgbdslam_pose_block as 76 float words: pose_state(24) status(4, int32 bits) P_rel P_rel_inv P_glob."""
    pb = torch.zeros(24 + 4 + 48, dtype=torch.float32, device="cuda")
    pb[:24] = torch.from_numpy(np.concatenate([I, z, I, z]).astype(np.float32)).cuda()
    return pb


@pytest.mark.parametrize("case", ["default", "two_levels", "one_level", "no_extra_iterations", "many_iterations", "never_converges",
                                  "singular", "two_frames"])
def test_track_frame_persistent_loop_equals_multi_launch(ctx, pkg, orc, pair, case):
    """rgbdslam_set_icp_loop: one persistent launch per frame must leave the pose block and the history row of the
    one-launch-per-iteration path bit for bit -- level counts below 3, max_iter 0 (two steps per level), many steps,
    thresholds that never converge (every step executes), a singular system (the solve fails at the first step and
    the loop ends from there), and two frames in a row (epoch base moves on, nothing is cleared in between).  The
    default case is also checked against the oracle's AlignFrameGaussNewton."""
    n, m = 480, 640
    gr, gc = orc.icp_grid(n, m, 1)
    ctx.alloc_buffers(gr, gc)
    I = np.eye(3, dtype=np.float32).reshape(-1)
    z = np.zeros(3, np.float32)
    prm = pkg.icp_params()
    if case == "two_levels":
        prm.lvl = 2
    elif case == "one_level":
        prm.lvl = 1
    elif case == "no_extra_iterations":
        prm.max_iter[0] = prm.max_iter[1] = prm.max_iter[2] = 0
    elif case == "many_iterations":
        prm.max_iter[0], prm.max_iter[1], prm.max_iter[2] = 9, 7, 8
    elif case == "never_converges":
        prm.thres_cvg_alpha = prm.thres_cvg_tr = 0.0
    cur = [dev(pair[k]) for k in ("v1", "n1", "rgb1", "gx1", "gy1")]
    if case == "singular":
        empty = torch.zeros(n, m, 3, device="cuda")
        cur[0] = cur[1] = cur[2] = empty
    prev = [dev(pair[k]) for k in ("v0", "n0", "rgbd0")]
    frames = 2 if case == "two_frames" else 1
    out = []
    for loop in (False, True):
        ctx.set_icp_loop(loop)
        pb = _pose_block(torch, I, z)
        hist = torch.zeros(4, 16, device="cuda")
        for f in range(frames):
            ctx.track_frame(prm, *cur, *prev, n, m, pb, hist, f)
        ctx.synchronize()
        if loop:
            assert ctx.icp_loop_faults() == 0
        out.append((host(pb).copy(), host(hist).copy()))
    ctx.set_icp_loop(False)
    same(out[1][0].view(np.int32), out[0][0].view(np.int32))
    same(out[1][1].view(np.int32), out[0][1].view(np.int32))
    status = out[1][0].view(np.int32)[24:28]
    if case == "singular":
        assert status[0] == 0 and status[1] == 0
    else:
        assert status[0] == 1 and status[3] == frames
    if case == "never_converges":
        assert status[1] == sum(prm.max_iter[k] + 2 for k in range(3))
    if case == "default":
        ok, R, t, tr = orc.align_gauss_newton(orc.icp_params(), pair["calib"][:4], pair["v1"], pair["n1"], pair["rgb1"], pair["v0"],
                                              pair["n0"], pair["rgbd0"], I, z, I, z)
        assert ok and status[1] == len(tr)
        same(out[1][0][:9], R)
        same(out[1][0][9:12], t)


def test_merge_and_model_maps(ctx, orc, pair, synth):
    n, m = 480, 640
    calib = pair["calib"]
    rng = np.random.default_rng(1)
    rgbd = pair["rgbd0"].copy()
    # perturb the predicted depth to hit every branch: equal, occluded, violated, empty
    w = rgbd[..., 3]
    w[100:200] *= 1.5
    w[200:300] *= 0.5
    w[300:350] = 0
    w[350:400] += rng.normal(0, 1e-5, w[350:400].shape).astype(np.float32)
    mask = rng.integers(9, 256, (n, m)).astype(np.uint8)
    mask[::3] = 11
    mask[1::9] = 1
    mask[2::9] = 255
    g_rgbd, g_mask = dev(rgbd), dev(mask)
    ctx.merge_depth_off(g_rgbd, dev(pair["d1"]), 2 * m, dev(pair["rgb1"]), g_mask, n, m)
    r_rgbd, r_mask = orc.merge_depth_off(rgbd, pair["d1"], pair["rgb1"], mask, calib)
    same(host(g_mask), r_mask)
    same(host(g_rgbd), r_rgbd)
    assert len(np.unique(r_mask)) > 5
    vmap = torch.empty(n, m, 3, device="cuda")
    ctx.vertex_map_gl(vmap, g_rgbd, n, m)
    same(host(vmap), orc.vertex_map_gl(r_rgbd, calib))
    out = torch.empty(n, m, 4, device="cuda")
    ctx.gpu_add_zbuff_off(dev(pair["d1"]), 2 * m, dev(pair["rgb1"]), out, n, m)
    ref = orc.add_zbuff_off(pair["d1"], pair["rgb1"], calib)
    same(host(out), ref)
    mk = torch.empty(n, m, dtype=torch.uint8, device="cuda")
    ctx.init_mask(mk, out, n, m)
    same(host(mk), orc.init_mask(ref))


def _rel_pose(pair):
    Rinv = np.linalg.inv(pair["R1"])
    return pose16(Rinv, -Rinv @ pair["t1"])


def test_transform_trim_raster_read(ctx, orc, pair):
    n, m = 480, 640
    calib = pair["calib"]
    P = _rel_pose(pair)
    v = pair["v0"].copy()
    v[50:60, 70:90] = 0            # a hole
    v[200:260, 300:380, 2] += 0.5  # a foreground box -> discontinuities + occlusion after the warp
    v[200:260, 300:380, :2] *= (v[200:260, 300:380, 2:3] / (v[200:260, 300:380, 2:3] - 0.5))
    g_v, g_rgbd = dev(v), dev(pair["rgbd0"])
    ctx.set_pose_matrix(P)
    ctx.transform_vertex_map(g_v, g_rgbd, n, m)
    r_v, r_rgbd = orc.transform_vertex_map(P, v, pair["rgbd0"])
    same(host(g_v), r_v)
    same(host(g_rgbd), r_rgbd)
    idx = torch.empty((n - 1) * (m - 1), 4, dtype=torch.int32, device="cuda")
    ctx.quad_trim_frame(g_v, idx, n, m, 0.1)
    r_idx = orc.quad_trim_frame(r_v, 0.1)
    same(host(idx).view(np.uint32), r_idx)
    assert (r_idx.sum(1) == 0).sum() > 100
    zbuf = torch.empty(n, m, dtype=torch.int64, device="cuda")
    tex = torch.empty(n, m, 4, device="cuda")
    ctx.raster_clear(zbuf, n, m)
    # inline trimming
    ctx.predict_raster(tex, zbuf, g_v, g_rgbd, None, n, m)
    r_tex, _ = orc.raster(r_v, r_rgbd, calib)
    same(host(tex), r_tex)
    assert (host(zbuf) == -1).all()                 # z-buffer left clean for the next frame
    assert (r_tex[..., 3] > 0).mean() > 0.9
    # explicit GL_QUADS index buffer
    tex2 = torch.empty(n, m, 4, device="cuda")
    ctx.predict_raster(tex2, zbuf, g_v, g_rgbd, idx, n, m)
    r_tex2, _ = orc.raster(r_v, r_rgbd, calib, index=r_idx)
    same(host(tex2), r_tex2)
    same(r_tex2, r_tex)
    ctx.read_frame(g_v, g_rgbd, tex, n, m)
    rv2, rr2 = orc.read_frame(r_tex, calib)
    same(host(g_v), rv2)
    same(host(g_rgbd), rr2)


def test_raster_large_and_small_triangles(ctx, orc, pair):
    """The z-buffer pass steps its edge functions in 32 bits when a triangle's bounding box spans at most 64 pixels (every
    triangle of a depth image's mesh) and in 64 bits otherwise: a surface close to the camera (quads of 30 .. 150 pixels,
    both sides of the switch, many crossing the image border) must give the oracle's texture bit for bit."""
    n, m = 480, 640
    calib = pair["calib"]
    rng = np.random.default_rng(11)
    ii, jj = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(m, dtype=np.float32), indexing="ij")
    # a slanted plane 0.6 .. 0.9 m away whose vertex (i, j) is the ray of pixel (40 i, 40 j) scaled up to 150 j / row: the
    # mesh of the first rows / columns covers the whole image with quads that grow from ~30 to ~150 pixels
    step = (30.0 + 0.25 * jj).astype(np.float32)
    z = (0.6 + 0.0004 * jj + 0.0002 * ii).astype(np.float32)
    u = (np.cumsum(step, axis=1) - step[:, :1] - 40.0).astype(np.float32)
    vstep = (20.0 + 6.0 * ii).astype(np.float32)     # rows 20, 26, 32, ... pixels apart: the first few quads stay below 64 pixels
    w = (np.cumsum(vstep, axis=0) - vstep[:1] - 30.0).astype(np.float32)
    v = np.zeros((n, m, 3), np.float32)
    v[..., 0] = z * (u - calib[2]) / calib[0]
    v[..., 1] = z * (w - calib[3]) / calib[1]
    v[..., 2] = -z
    rgbd = rng.random((n, m, 4)).astype(np.float32)
    rgbd[..., 3] = z / 100.0
    g_v, g_rgbd = dev(v), dev(rgbd)
    zbuf = torch.empty(n, m, dtype=torch.int64, device="cuda")
    tex = torch.empty(n, m, 4, device="cuda")
    ctx.raster_clear(zbuf, n, m)
    ctx.predict_raster(tex, zbuf, g_v, g_rgbd, None, n, m, thresh=10.0)
    r_tex, _ = orc.raster(v, rgbd, calib, thresh=10.0)
    same(host(tex), r_tex)
    assert (r_tex[..., 3] > 0).mean() > 0.8
    assert (host(zbuf) == -1).all()


def test_project_mask_and_segment(ctx, orc, pair, synth):
    n, m = 480, 640
    calib = pair["calib"]
    rng = np.random.default_rng(2)
    mask = rng.integers(0, 256, (n, m)).astype(np.uint8)
    P = pose16(pair["R1"], pair["t1"] * 20)   # large enough shift to push projections off-image
    ctx.set_pose_matrix(P)
    out = torch.empty(n, m, dtype=torch.uint8, device="cuda")
    ctx.project_mask(dev(mask), out, dev(pair["v0"]), n, m)
    ref = orc.project_mask(P, calib, mask, pair["v0"])
    same(host(out), ref)
    assert (ref == 0).sum() > 500
    eq, _ = synth.plane_equations()
    I16 = pose16(np.eye(3), np.zeros(3))
    lab = torch.empty(n, m, dtype=torch.uint8, device="cuda")
    ctx.seed_labels(lab, dev(pair["v0"]), dev(pair["n0"]), I16, dev(eq), 3, n, m)
    lab_ref = orc.seed_labels(pair["v0"], pair["n0"], I16, eq)
    same(host(lab), lab_ref)
    assert len(np.unique(lab_ref)) == 4
    holes = lab_ref.copy()
    holes[rng.random((n, m)) < 0.4] = 0
    holes[100:140, 100:160] = 0
    g_lab, swap = dev(holes), torch.empty(n, m, dtype=torch.uint8, device="cuda")
    ctx.segment(g_lab, swap, dev(pair["v0"]), dev(pair["n0"]), I16, dev(eq), 3, n, m, 0.1, 0.9)
    seg_ref = orc.segment(holes, pair["v0"], pair["n0"], I16, eq, 0.1, 0.9)
    same(host(g_lab), seg_ref)
    assert (seg_ref > 0).sum() > (holes > 0).sum()


def _plane_setup(synth, N_prim, nb=3):
    eq, cen = synth.plane_equations()
    params = []
    for k in range(nb):
        E, c = eq[k], cen[k]
        scale = np.float32(2.0 / 0.004)
        pt0 = np.float32(c @ E[3:6])
        pt1 = np.float32(c @ E[6:9])
        sh0 = np.float32(pt0 - N_prim * 0.004 / 2.0)
        sh1 = np.float32(pt1 - N_prim * 0.004 / 2.0)
        params.append(np.array([E[0], E[1], E[2], E[9], scale, scale, sh0, sh1, E[3], E[4], E[5], E[6], E[7], E[8]], np.float32))
    return eq, params


def _ptr_table(tensors):
    return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device="cuda")


def test_update_and_decode(ctx, orc, pair, synth):
    n, m = 480, 640
    N_prim = 1024
    eq, params = _plane_setup(synth, N_prim)
    I16 = pose16(np.eye(3), np.zeros(3))
    label = orc.seed_labels(pair["v0"], pair["n0"], I16, eq)
    rng = np.random.default_rng(3)
    rgbd = pair["rgbd0"]
    bumps = [np.zeros((N_prim, N_prim, 3), np.uint16) for _ in range(3)]
    rgbs = [np.zeros((N_prim, N_prim, 3), np.uint8) for _ in range(3)]
    masks = [np.zeros((N_prim, N_prim), np.uint8) for _ in range(3)]
    claims = [np.zeros((N_prim, N_prim), np.uint64) for _ in range(3)]
    g_b, g_c, g_m = [dev(x) for x in bumps], [dev(x) for x in rgbs], [dev(x) for x in masks]
    g_k = [torch.zeros(N_prim, N_prim, dtype=torch.int64, device="cuda") for _ in range(3)]
    g_p = [dev(p) for p in params]
    tabs = [_ptr_table(x) for x in (g_b, g_c, g_m, g_p, g_k)]
    g_v, g_rgbd, g_lab = dev(pair["v0"]), dev(rgbd), dev(label)
    # three rounds: fresh write (mask 11..14), higher-confidence overwrite, then erosion (mask 0)
    for stamp, maskgen in enumerate([lambda: rng.integers(11, 15, (n, m)), lambda: rng.integers(10, 40, (n, m)),
                                     lambda: np.where(rng.random((n, m)) < 0.5, 0, rng.integers(10, 60, (n, m)))], start=1):
        mk = maskgen().astype(np.uint8)
        ctx.update(tabs[0], tabs[1], tabs[2], tabs[3], tabs[4], stamp, g_v, g_rgbd, dev(mk), g_lab, I16, N_prim, N_prim, n, m)
        orc.update(bumps, rgbs, masks, params, claims, stamp, pair["v0"], rgbd, mk, label, I16)
        for k in range(3):
            same(host(g_m[k]), masks[k])
            same(host(g_b[k]), bumps[k])
            same(host(g_c[k]), rgbs[k])
    assert sum((mk_ > 10).sum() for mk_ in masks) > 50000
    # decode (VertexFromBump) incl. the special mask value 1 and sub-sampling
    masks[0][10:20, 10:20] = 1
    g_m[0].copy_(dev(masks[0]))
    for lvl in (1, 2):
        no = N_prim // lvl
        v = torch.empty(no, no, 3, device="cuda")
        c = torch.empty(no, no, 4, device="cuda")
        k4 = torch.empty(no, no, 4, device="cuda")
        ctx.vertex_from_bump(v, c, k4, g_b[0], g_c[0], g_m[0], g_p[0], N_prim, N_prim, lvl)
        rv, rc, rk = orc.vertex_from_bump(bumps[0], rgbs[0], masks[0], params[0], lvl)
        same(host(v), rv)
        same(host(c), rc)
        same(host(k4), rk)
    # whole-model decode in one launch (planes 1..2 of 3), 4 cells per lane
    outs = [[torch.full((N_prim, N_prim, ch), -7.0, device="cuda") for _ in range(3)] for ch in (3, 4, 4)]
    otabs = [_ptr_table(x) for x in outs]
    ctx.vertex_from_bump_batch(otabs[0], otabs[1], otabs[2], tabs[0], tabs[1], tabs[2], tabs[3], 1, 3, N_prim, N_prim)
    assert float(outs[0][0].min()) == -7.0 and float(outs[0][0].max()) == -7.0      # plane 0 is outside [1, 3)
    for k in (1, 2):
        rv, rc, rk = orc.vertex_from_bump(bumps[k], rgbs[k], masks[k], params[k], 1)
        same(host(outs[0][k]), rv)
        same(host(outs[1][k]), rc)
        same(host(outs[2][k]), rk)


_DECODE_PARAMS = {
    "typical": [0.3, -0.5, 0.81, -1.7, 500.0, 500.0, -0.51, 0.27, 0.9, 0.1, -0.42, 0.2, 0.84, 0.5],
    "odd scales": [0.3, -0.5, 0.81, 2.25, 333.3, 0.73, 7.0, -3.0, 0.9, 0.1, -0.42, 0.2, 0.84, 0.5],
    "negative scales": [1.0, 0.0, 0.0, -14.9, -1023.9999, -57.0, 0.0, 0.0, 0.6, 0.8, 0.0, -0.8, 0.6, 0.0],
    # outside the range in which the device forms its quotients from reciprocals (|p| <= 2^60, |p[4]|, |p[5]| >= 2^-40):
    # the plain divisions run, and the results -- infinities and NaN included -- are still the oracle's
    "scale below 2^-40": [0.3, -0.5, 0.81, -1.7, 1e-20, 500.0, -0.51, 0.27, 0.9, 0.1, -0.42, 0.2, 0.84, 0.5],
    "scale zero": [0.3, -0.5, 0.81, -1.7, 500.0, 0.0, -0.51, 0.27, 0.9, 0.1, -0.42, 0.2, 0.84, 0.5],
    "huge axis": [0.3, -0.5, 0.81, -1.7, 500.0, 500.0, -0.51, 0.27, 3e30, 0.1, -0.42, 0.2, 0.84, 0.5],
    "infinite offset": [0.3, -0.5, 0.81, np.inf, 500.0, 500.0, -0.51, 0.27, 0.9, 0.1, -0.42, 0.2, 0.84, 0.5],
    "nan": [0.3, np.nan, 0.81, -1.7, 500.0, 500.0, -0.51, 0.27, 0.9, 0.1, -0.42, 0.2, 0.84, 0.5],
}


@pytest.mark.parametrize("which", list(_DECODE_PARAMS))
def test_decode_all_byte_and_word_values(ctx, orc, which):
    """Bump codec decode over every u8 colour/mask value and EVERY u16 value of each of the three channels: pins the claim
    that the quotients the device forms -- from correctly rounded reciprocals and an exact FMA residual where the plane's
    parameters allow it (device_fns.hpp: quot_rcp), by plain division otherwise -- are the reference's divisions bit for bit."""
    N = 256
    rng = np.random.default_rng(9)
    bump = rng.integers(0, 65536, (N, N, 3)).astype(np.uint16)
    sweep = (np.arange(N * N).reshape(N, N) % 65536).astype(np.uint16)
    bump[:, :, 0] = sweep                                                     # all 65536 values of every channel
    bump[:, :, 1] = sweep.T
    bump[:, :, 2] = sweep[::-1, ::-1]
    rgbb = rng.integers(0, 256, (N, N, 3)).astype(np.uint8)
    rgbb[0, :, 0] = np.arange(256)
    mask = (np.arange(N * N).reshape(N, N) % 256).astype(np.uint8)
    param = np.asarray(_DECODE_PARAMS[which], np.float32)
    g = [dev(bump), dev(rgbb), dev(mask), dev(param)]
    outs = [[torch.empty(N, N, ch, device="cuda")] for ch in (3, 4, 4)]
    ctx.vertex_from_bump_batch(_ptr_table(outs[0]), _ptr_table(outs[1]), _ptr_table(outs[2]), _ptr_table([g[0]]), _ptr_table([g[1]]),
                               _ptr_table([g[2]]), _ptr_table([g[3]]), 0, 1, N, N)
    with np.errstate(all="ignore"):
        rv, rc, rk = orc.vertex_from_bump(bump, rgbb, mask, param, 1)
    same(host(outs[0][0]), rv)
    same(host(outs[1][0]), rc)
    same(host(outs[2][0]), rk)
    # the per-plane kernel (one cell per thread) shares the device function
    v, c, k4 = (torch.empty(N, N, ch, device="cuda") for ch in (3, 4, 4))
    ctx.vertex_from_bump(v, c, k4, g[0], g[1], g[2], g[3], N, N, 1)
    same(host(v), rv)
    same(host(k4), rk)


def test_preprocess_fused_all_byte_values(ctx, orc, synth):
    """rgbdslam_frame_preprocess (one kernel) == cpy_char3 + vertex_map_kinect + compute_normal +
    intensity + Sobel of the oracle, on an image holding every byte value and holes, odd size."""
    n, m = 61, 150
    calib = synth.calib_for(n, m)
    ctx.set_calibration_matrix(calib)
    rng = np.random.default_rng(4)
    d, _ = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    d[rng.random((n, m)) < 0.05] = 0
    c = rng.integers(0, 256, (n, m, 3)).astype(np.uint8)
    c.reshape(-1)[:256] = np.arange(256)
    v = torch.empty(n, m, 3, device="cuda"); nm = torch.empty(n, m, 3, device="cuda"); rgb = torch.empty(n, m, 3, device="cuda")
    gx = torch.empty(n, m, device="cuda"); gy = torch.empty(n, m, device="cuda")
    import ctypes as C
    gd, gcol = dev(d), dev(c)
    rc = ctx.lib.rgbdslam_frame_preprocess(ctx._h, C.c_void_p(gd.data_ptr()), C.c_size_t(2 * m), C.c_void_p(gcol.data_ptr()),
                                           C.c_size_t(3 * m), C.c_void_p(v.data_ptr()), C.c_void_p(nm.data_ptr()),
                                           C.c_void_p(rgb.data_ptr()), C.c_void_p(gx.data_ptr()), C.c_size_t(4 * m),
                                           C.c_void_p(gy.data_ptr()), C.c_size_t(4 * m), n, m)
    assert rc == 0
    ctx.synchronize()
    rv = orc.vertex_map_kinect(d, calib)
    rrgb = orc.cpy_char3(c)
    _, rgx, rgy = orc.intensity_sobel(rrgb)
    same(host(v), rv)
    same(host(rgb), rrgb)
    same(host(nm), orc.compute_normal(rv))
    same(host(gx), rgx)
    same(host(gy), rgy)
    ctx.set_calibration_matrix(synth.TUM_CALIB)


def _calib_variants(synth, n, m):
    base = np.asarray(synth.calib_for(n, m), np.float32)
    out = {}
    c = base.copy(); c[0] = 333.3; c[1] = 291.7; c[2] = 0.0; c[3] = 17.25
    out["odd focal lengths, zero centre (reciprocal path)"] = c
    c = base.copy(); c[0] = np.frombuffer(np.uint32(0x43FFFFFF).tobytes(), np.float32)[0]       # 511.99997: significand all ones
    out["focal length with an all-ones significand (plain divisions)"] = c
    c = base.copy(); c[2] = 1e-30
    out["centre below 2^-60 (plain divisions)"] = c
    c = base.copy(); c[9] = 1e-20; c[10] = 1.0
    out["depth scale below 2^-40 (plain divisions)"] = c
    c = base.copy(); c[1] = 0.0
    out["zero focal length (plain divisions, infinite results)"] = c
    return out


@pytest.mark.parametrize("which", [0, 1, 2, 3, 4])
def test_calibrations_on_either_side_of_the_reciprocal_path(ctx, orc, synth, which):
    """The per-pixel quotients by fx, fy and the depth scale come from host-side reciprocals where the calibration allows it
    (CalibV::fast) and from plain divisions otherwise: preprocessing, merge, model vertex map and seeding against the oracle for
    calibrations on either side of that switch, bit for bit (infinities and NaN included)."""
    import ctypes as C
    n, m = 61, 150
    name, calib = list(_calib_variants(synth, n, m).items())[which]
    ctx.set_calibration_matrix(calib)
    try:
        rng = np.random.default_rng(40 + which)
        d, _ = synth.render(np.eye(3), np.zeros(3), n, m, synth.calib_for(n, m))
        d[rng.random((n, m)) < 0.05] = 0
        c = rng.integers(0, 256, (n, m, 3)).astype(np.uint8)
        v = torch.empty(n, m, 3, device="cuda"); nm = torch.empty(n, m, 3, device="cuda"); rgb = torch.empty(n, m, 3, device="cuda")
        gx = torch.empty(n, m, device="cuda"); gy = torch.empty(n, m, device="cuda")
        gd, gcol = dev(d), dev(c)
        assert ctx.lib.rgbdslam_frame_preprocess(ctx._h, C.c_void_p(gd.data_ptr()), C.c_size_t(2 * m), C.c_void_p(gcol.data_ptr()),
                                                 C.c_size_t(3 * m), C.c_void_p(v.data_ptr()), C.c_void_p(nm.data_ptr()),
                                                 C.c_void_p(rgb.data_ptr()), C.c_void_p(gx.data_ptr()), C.c_size_t(4 * m),
                                                 C.c_void_p(gy.data_ptr()), C.c_size_t(4 * m), n, m) == 0
        ctx.synchronize()
        with np.errstate(all="ignore"):
            rv = orc.vertex_map_kinect(d, calib)
            rn_ = orc.compute_normal(rv)
        same(host(v), rv)
        same(host(nm), rn_)
        same(host(rgb), orc.cpy_char3(c))
        # seeding, merge (every branch) and the model's vertex map
        rgbf = orc.cpy_char3(c)
        out = torch.empty(n, m, 4, device="cuda")
        ctx.gpu_add_zbuff_off(gd, 2 * m, dev(rgbf), out, n, m)
        with np.errstate(all="ignore"):
            ref = orc.add_zbuff_off(d, rgbf, calib)
        same(host(out), ref)
        rgbd = ref.copy()
        w = rgbd[..., 3]
        w[10:20] *= 1.5; w[20:30] *= 0.5; w[30:40] = 0
        w[40:50] += rng.normal(0, 1e-5, w[40:50].shape).astype(np.float32)
        mask = rng.integers(9, 256, (n, m)).astype(np.uint8)
        mask[::3] = 11; mask[1::9] = 1; mask[2::9] = 255
        g_rgbd, g_mask = dev(rgbd), dev(mask)
        ctx.merge_depth_off(g_rgbd, gd, 2 * m, dev(rgbf), g_mask, n, m)
        with np.errstate(all="ignore"):
            r_rgbd, r_mask = orc.merge_depth_off(rgbd, d, rgbf, mask, calib)
        same(host(g_mask), r_mask)
        same(host(g_rgbd), r_rgbd)
        vmap = torch.empty(n, m, 3, device="cuda")
        ctx.vertex_map_gl(vmap, g_rgbd, n, m)
        with np.errstate(all="ignore"):
            same(host(vmap), orc.vertex_map_gl(r_rgbd, calib))
    finally:
        ctx.set_calibration_matrix(synth.TUM_CALIB)


def test_icp_row_bands_and_gathered_solve(ctx, pkg, orc, pair):
    """Row-sharded ICP on one GPU: the two 'ranks' are two band launches.  Each band's 27 sums equal
    the oracle's two-stage reduction restricted to the band's blocks; the rank-ordered total drives
    the solve; the loop converges to the single-GPU pose within 1e-5 (north_star tolerance)."""
    import ctypes as C
    n, m = 480, 640
    calib = pair["calib"]
    ctx.set_calibration_matrix(calib)
    ctx.alloc_buffers(60, 80)
    prm = pkg.icp_params()
    I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
    g = {k: dev(pair[k]) for k in ("v0", "n0", "rgbd0", "v1", "n1", "rgb1", "gx1", "gy1")}
    p = lambda t: C.c_void_p(t.data_ptr())
    lib = ctx.lib
    state = dev(np.concatenate([I, z, I, z]).astype(np.float32)); status = torch.zeros(4, dtype=torch.int32, device="cuda")
    gathered = torch.zeros(2, 27, device="cuda")
    sin20 = float(np.sin(np.deg2rad(20.0)))
    assert lib.rgbdslam_align_begin(ctx._h, p(state), p(status)) == 0
    checked = False
    for lvl in (2, 1, 0):
        fact = 1 << lvl
        gr, gc = orc.icp_grid(n, m, fact)
        split = gr // 2 + 1                                        # uneven bands
        for it in range(prm.max_iter[lvl] + 2):
            for r, (lo, hi) in enumerate(((0, split), (split, gr))):
                rc = lib.rgbdslam_align_band_system(ctx._h, C.byref(prm), p(g["v1"]), p(g["n1"]), p(g["rgb1"]), p(g["gx1"]),
                                                    C.c_size_t(4 * m), p(g["gy1"]), C.c_size_t(4 * m), p(g["v0"]), p(g["n0"]),
                                                    p(g["rgbd0"]), n, m, lvl, it, lo, hi, C.c_void_p(gathered[r].data_ptr()))
                assert rc == 0
            if not checked:                                        # first iteration: identity estimate -> compare with the oracle
                buf, _ = orc.icp_match(calib[:4], I, z, pair["v1"], pair["n1"], pair["rgb1"], pair["gx1"], pair["gy1"], pair["v0"],
                                       pair["n0"], pair["rgbd0"], 0.05, sin20, fact)
                b3 = buf.reshape(27, gc, gr)
                for r, (lo, hi) in enumerate(((0, split), (split, gr))):
                    local = np.ascontiguousarray(b3[:, :, lo:hi]).reshape(27, -1)
                    same(host(gathered[r]), orc.icp_reduce(local))
                checked = True
            assert lib.rgbdslam_align_solve_gathered(ctx._h, C.byref(prm), p(gathered), 2, p(state), lvl, it, None) == 0
    assert lib.rgbdslam_align_end(ctx._h, p(state), p(status)) == 0
    ctx.synchronize()
    ok, R, t, trace = orc.align_gauss_newton(orc.icp_params(), calib[:4], pair["v1"], pair["n1"], pair["rgb1"], pair["v0"], pair["n0"],
                                             pair["rgbd0"], I, z, I, z)
    s = host(state)
    assert ok and int(status[0]) == 1
    assert np.abs(s[:9] - R).max() <= 1e-5 and np.abs(s[9:12] - t).max() <= 1e-5


def _pack27(A, b):
    """(A symmetric 6x6, b) -> the 27 sums as the reduction leaves them: row i = A[i][i..5], b[i] (ICP.cu:178-191)."""
    s = []
    for i in range(6):
        s.extend(A[i, i:])
        s.append(b[i])
    return np.asarray(s, np.float32)


def _solve_cases():
    rng = np.random.default_rng(77)
    cases = []
    for k in range(24):                                           # well-posed: J^T J of a random Jacobian
        J = rng.standard_normal((40 + k, 6)).astype(np.float32) * np.float32(10.0 ** rng.integers(-1, 3))
        A = (J.T @ J).astype(np.float32)
        A = np.triu(A) + np.triu(A, 1).T                          # exactly symmetric
        xs = (rng.standard_normal(6) * 1e-2).astype(np.float32)
        cases.append(("spd%d" % k, A, (A @ xs).astype(np.float32)))
    b = (rng.standard_normal(6) * 1e-3).astype(np.float32)
    d = lambda *v: np.diag(np.asarray(v, np.float32))
    cases.append(("zero", np.zeros((6, 6), np.float32), b))
    cases.append(("first pivot negative, det +1", d(-1, -1, 1, 1, 1, 1), b))
    cases.append(("last pivot negative", d(1, 1, 1, 1, 1, -1), b))
    cases.append(("middle pivot zero", d(1, 1, 0, 1, 1, 1), b))
    cases.append(("positive definite, |det| < 1e-15", d(*[1e-3] * 6), b * np.float32(1e-3)))
    cases.append(("positive definite, det just above", d(*[4e-3] * 6), b * np.float32(1e-3)))
    A = d(2, 2, 2, 2, 2, 2); A[2, 3] = A[3, 2] = np.nan
    cases.append(("nan off the diagonal", A, b))
    A = d(2, 2, 2, 2, 2, 2); A[4, 4] = np.nan
    cases.append(("nan on the diagonal", A, b))
    A = d(2, 2, 2, 2, 2, 2); A[0, 0] = np.inf
    cases.append(("inf pivot", A, b))
    A = d(1, 1, 1, 1, 1, 1); A[0, 1] = A[1, 0] = 2.0                # indefinite through an off-diagonal, det = -3
    cases.append(("indefinite off-diagonal", A, b))
    J = rng.standard_normal((5, 6)).astype(np.float32)              # rank 5: the last pivot is rounding noise of either sign
    A = (J.T @ J).astype(np.float32); A = np.triu(A) + np.triu(A, 1).T
    cases.append(("rank deficient", A, b))
    cases.append(("nan right-hand side", d(2, 2, 2, 2, 2, 2), np.asarray([0, 0, np.nan, 0, 0, 0], np.float32)))
    # tiny and denormal positive pivots with a determinant that passes
    cases.append(("pivot below 2^-96, determinant 1", d(1e-30, 1e30, 1, 1, 1, 1), b * np.float32(1e-30)))
    cases.append(("denormal pivot", d(1e-40, 1e20, 1e20, 1, 1, 1), b * np.float32(1e-42)))
    return cases


def test_icp_solve_step_on_crafted_systems(ctx, pkg, orc, pair):
    """ONE solve step (k_icp_solve: wave_llt_solve6 + wave_det6 + the pose update, the code k_icp_loop's tail runs) on
    crafted 6x6 systems, against the oracle's orc_llt_solve6 / orc_det6 and the lost-frame rule of RG.cpp:862-873:
    the increment x bit-exact where the step is accepted; where it is not -- non-positive pivot at any position, NaN /
    Inf entries, |det| < 1e-15 although the factorisation succeeds -- no iteration is counted, the frame is reported
    lost and the caller's pose is untouched.  Pins the branch-free forms of the solve, the determinant and the update
    (a failed factorisation's update runs on garbage and must be dropped)."""
    import ctypes as C
    ctx.set_calibration_matrix(pair["calib"])
    ctx.alloc_buffers(60, 80)
    prm = pkg.icp_params()
    lib = ctx.lib
    p = lambda t: C.c_void_p(t.data_ptr())
    I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
    pose0 = np.concatenate([I, z, I, z]).astype(np.float32)
    accepted = rejected = 0
    for name, A, b in _solve_cases():
        rc_ref, x_ref = orc.llt_solve6(A, b)
        det = np.float32(orc.det6(A))
        lost = bool(rc_ref != 0 or np.isnan(det) or float(abs(det)) < 1e-15)
        state = dev(pose0); status = torch.full((4,), -7, dtype=torch.int32, device="cuda")
        sums = dev(_pack27(A, b)); trace = torch.full((2, 33), -1.0, device="cuda")
        assert lib.rgbdslam_align_begin(ctx._h, p(state), p(status)) == 0
        assert lib.rgbdslam_align_solve_gathered(ctx._h, C.byref(prm), p(sums), 1, p(state), 2, 0, p(trace)) == 0
        assert lib.rgbdslam_align_end(ctx._h, p(state), p(status)) == 0
        ctx.synchronize()
        st, tr, s = host(status), host(trace), host(state)
        if lost:
            rejected += 1
            assert st[0] == 0 and st[1] == 0 and st[2] == 0, (name, st)
            same(s, pose0)                                           # the pose the caller holds is untouched
            assert (tr == -1.0).all(), name                          # and nothing was traced
        else:
            accepted += 1
            assert st[1] == 1 and st[2] == 1 << 16, (name, st)
            same(tr[0, :27], _pack27(A, b))
            same(tr[0, 27:], x_ref)                                  # bit-exact increment
            if not np.isnan(x_ref).any():                            # (a NaN right-hand side passes RG.cpp:862-873 and, NaN
                                                                     # comparing false, RG.cpp:910 too: only x is checked)
                assert st[0] == 1, (name, st)
                # identity start: tinv = x[0..2] and Rinv = Rinc exactly, so the pose is inverse(Rinc), -inverse(Rinc) x[0..2]
                q = np.asarray([0.0, x_ref[3], x_ref[4], x_ref[5]], np.float64)
                q[0] = np.sqrt(1.0 - float(np.float32(np.float32(x_ref[3] * x_ref[3] + x_ref[4] * x_ref[4]) + x_ref[5] * x_ref[5])))
                Rinc = np.asarray([[q[0]**2 + q[1]**2 - q[2]**2 - q[3]**2, 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2])],
                                   [2 * (q[1] * q[2] + q[0] * q[3]), q[0]**2 - q[1]**2 + q[2]**2 - q[3]**2, 2 * (q[2] * q[3] - q[0] * q[1])],
                                   [2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), q[0]**2 - q[1]**2 - q[2]**2 + q[3]**2]])
                R = np.linalg.inv(Rinc)
                assert np.abs(s[:9].reshape(3, 3).T - R).max() <= 1e-5, name
                assert np.abs(s[9:12] + R @ x_ref[:3].astype(np.float64)).max() <= 1e-5, name
    assert accepted >= 25 and rejected >= 9, (accepted, rejected)

    # a failed step after an accepted one: the count stays at one, the frame is lost, the pose untouched
    name, A, b = _solve_cases()[0]
    state = dev(pose0); status = torch.zeros(4, dtype=torch.int32, device="cuda")
    good = dev(_pack27(A, b)); bad = dev(_pack27(-A, b))
    assert lib.rgbdslam_align_begin(ctx._h, p(state), p(status)) == 0
    assert lib.rgbdslam_align_solve_gathered(ctx._h, C.byref(prm), p(good), 1, p(state), 2, 0, None) == 0
    assert lib.rgbdslam_align_solve_gathered(ctx._h, C.byref(prm), p(bad), 1, p(state), 2, 1, None) == 0
    assert lib.rgbdslam_align_end(ctx._h, p(state), p(status)) == 0
    ctx.synchronize()
    st = host(status)
    assert st[0] == 0 and st[1] == 1, st
    same(host(state), pose0)


def test_update_plane_partition(ctx, orc, pair, synth):
    """Plane-partitioned fusion: shards [0,1) and [1,3) together equal the unpartitioned result."""
    n, m = 480, 640
    N_prim = 512
    eq, params = _plane_setup(synth, N_prim)
    I16 = pose16(np.eye(3), np.zeros(3))
    label = orc.seed_labels(pair["v0"], pair["n0"], I16, eq)
    mk = np.full((n, m), 12, np.uint8)

    def run(parts):
        g_b = [torch.zeros(N_prim, N_prim, 3, dtype=torch.int16, device="cuda") for _ in range(3)]
        g_c = [torch.zeros(N_prim, N_prim, 3, dtype=torch.uint8, device="cuda") for _ in range(3)]
        g_m = [torch.zeros(N_prim, N_prim, dtype=torch.uint8, device="cuda") for _ in range(3)]
        g_k = [torch.zeros(N_prim, N_prim, dtype=torch.int64, device="cuda") for _ in range(3)]
        g_p = [dev(p) for p in params]
        tabs = [_ptr_table(x) for x in (g_b, g_c, g_m, g_p, g_k)]
        for lo, hi in parts:
            ctx.update(tabs[0], tabs[1], tabs[2], tabs[3], tabs[4], 1, dev(pair["v0"]), dev(pair["rgbd0"]), dev(mk),
                       dev(label), I16, N_prim, N_prim, n, m, lo, hi)
        return [host(x) for x in g_b + g_c + g_m]

    full = run([(0, 2 ** 31 - 1)])
    parts = run([(0, 1), (1, 3)])
    for a, b in zip(full, parts):
        same(a, b)


def test_copy_segments_and_polled_synchronize(ctx, pkg):
    """rgbdslam_copy_segments: up to four device ranges stored into pinned host memory by one launch (what RG::Sync reads the
    pose block and the history ring segments with), 16-byte and 4-byte aligned, ragged sizes; rgbdslam_synchronize with and
    without the polling phase returns with the data on the host."""
    import ctypes as C
    lib = ctx.lib
    rng = np.random.default_rng(5)
    for spin in (0, 5000):
        assert lib.rgbdslam_set_sync_spin(ctx._h, spin) == 0
        words = [33, 4096 * 16, 7, 1]                                  # pose-block-sized, a long ring segment, ragged tails
        offs = [0, 1, 3, 0]                                            # in words: the second and third are only 4-byte aligned
        src = [torch.from_numpy(rng.integers(0, 2 ** 31, w + 8, dtype=np.int32)).cuda() for w in words]
        dst = [torch.full((w + 8,), -1, dtype=torch.int32).pin_memory() for w in words]
        for n in (4, 3, 1, 0):
            for d in dst:
                d.fill_(-1)
            sp = (C.c_void_p * 4)(*[s_.data_ptr() + 4 * o for s_, o in zip(src, offs)])
            dp = (C.c_void_p * 4)(*[d_.data_ptr() + 4 * o for d_, o in zip(dst, offs)])
            by = (C.c_size_t * 4)(*[4 * w for w in words])
            assert lib.rgbdslam_copy_segments(ctx._h, sp, dp, by, n) == 0
            assert lib.rgbdslam_synchronize(ctx._h) == 0
            for k in range(4):
                got, want = dst[k].numpy(), host(src[k])
                o, w = offs[k], words[k]
                if k < n:
                    same(got[o:o + w], want[o:o + w])
                    assert (got[:o] == -1).all() and (got[o + w:] == -1).all()      # nothing outside the range
                else:
                    assert (got == -1).all()
        # argument errors (RGBDSLAM_E_ARG = 1): too many segments, a size that is no multiple of 4, a null range
        assert lib.rgbdslam_copy_segments(ctx._h, sp, dp, by, 5) == 1
        bad = (C.c_size_t * 4)(6, 4, 4, 4)
        assert lib.rgbdslam_copy_segments(ctx._h, sp, dp, bad, 1) == 1
        nul = (C.c_void_p * 4)(None, None, None, None)
        assert lib.rgbdslam_copy_segments(ctx._h, nul, dp, by, 1) == 1
    assert lib.rgbdslam_set_sync_spin(ctx._h, -1) == 1


def test_argument_errors(ctx, pkg):
    lib = pkg.load()
    assert lib.rgbdslam_compute_normal(ctx._h, None, None, 480, 640, 1) == 1
    assert lib.rgbdslam_vertex_map_gl(ctx._h, None, None, 0, 0) == 1
    assert lib.rgbdslam_create(None, 0, None) == 1


def test_sharded_aligner_over_rccl_single_rank(pkg, orc, pair):
    """ShardedAligner (row-sharded ICP driver) over a real RCCL process group of world size 1: with one
    band and one gathered part the result must be bit-identical to the single-GPU loop."""
    import importlib
    import os
    import socket
    import torch.distributed as dist
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    started = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        started = True
    try:
        n, m = 480, 640
        calib = pair["calib"]
        c2 = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        c2.set_calibration_matrix(calib)
        c2.alloc_buffers(60, 80)
        g = {k: dev(pair[k]) for k in ("v0", "n0", "rgbd0", "v1", "n1", "rgb1", "gx1", "gy1")}
        I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
        state = dev(np.concatenate([I, z, I, z]).astype(np.float32)); status = torch.zeros(4, dtype=torch.int32, device="cuda")
        al = mg.ShardedAligner(c2, pkg.icp_params(), n, m)
        assert al.world == 1 and al.rank == 0
        al.align(g["v1"], g["n1"], g["rgb1"], g["gx1"], g["gy1"], g["v0"], g["n0"], g["rgbd0"], state, status)
        torch.cuda.synchronize()
        ok, R, t, _ = orc.align_gauss_newton(orc.icp_params(), calib[:4], pair["v1"], pair["n1"], pair["rgb1"], pair["v0"], pair["n0"],
                                             pair["rgbd0"], I, z, I, z)
        sres = host(state)
        assert ok and int(status[0]) == 1
        same(sres[:9], R)
        same(sres[9:12], t)
        # the collective helper itself on the GPU backend
        gathered = mg.allgather_sums(torch.arange(27, dtype=torch.float32, device="cuda"))
        assert gathered.shape == (1, 27) and float(mg.ordered_total(gathered)[26]) == 26.0
        c2.close()
    finally:
        if started:
            dist.destroy_process_group()


def test_icp_1080p_four_bands_and_persistent_loop(ctx, pkg, orc, synth):
    """BASELINE configs[3] shape: a 1920x1080 pair.  (a) the full 27 sums of the first iteration at every level equal the
    oracle's two-stage reduction; (b) four row bands (the 4-GPU split) give the oracle's banded sums; (c) the whole
    Gauss-Newton loop through rgbdslam_track_frame -- 63 blocks per chain, 13 rounds -- leaves the same pose block with
    one launch per iteration and with the persistent loop kernel, and that pose is the oracle's."""
    import ctypes as C
    n, m = 1080, 1920
    calib = synth.calib_for(n, m)
    d0, c0 = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    R1, t1 = synth.pair_offset_pose()
    d1, c1 = synth.render(R1, t1, n, m, calib)
    pr = {}
    for k, (d, c) in enumerate([(d0, c0), (d1, c1)]):
        v = orc.vertex_map_kinect(d, calib)
        pr["v%d" % k] = v
        pr["n%d" % k] = orc.compute_normal(v)
        pr["rgb%d" % k] = orc.cpy_char3(c)
    pr["rgbd0"] = orc.add_zbuff_off(d0, pr["rgb0"], calib)
    _, pr["gx1"], pr["gy1"] = orc.intensity_sobel(pr["rgb1"])
    ctx.set_calibration_matrix(calib)
    gr, gc = orc.icp_grid(n, m, 1)
    ctx.alloc_buffers(gr, gc)
    g = {k: dev(pr[k]) for k in ("v0", "n0", "rgbd0", "v1", "n1", "rgb1", "gx1", "gy1")}
    I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
    prm = pkg.icp_params()
    sin20 = float(np.sin(np.deg2rad(20.0)))
    p = lambda t: C.c_void_p(t.data_ptr())
    lib = ctx.lib
    state = dev(np.concatenate([I, z, I, z]).astype(np.float32)); status = torch.zeros(4, dtype=torch.int32, device="cuda")
    assert lib.rgbdslam_align_begin(ctx._h, p(state), p(status)) == 0
    gathered = torch.zeros(4, 27, device="cuda")
    Rd, td, intr = dev(I), dev(z), dev(calib[:4])
    for lvl in (2, 1, 0):
        fact = 1 << lvl
        lgr, lgc = orc.icp_grid(n, m, fact)
        buf, _ = orc.icp_match(calib[:4], I, z, pr["v1"], pr["n1"], pr["rgb1"], pr["gx1"], pr["gy1"], pr["v0"], pr["n0"], pr["rgbd0"],
                               0.05, sin20, fact)
        A, b = ctx.estimate_system(Rd, td, g["v1"], g["n1"], g["rgb1"], g["gx1"], g["gy1"], g["v0"], g["n0"], g["rgbd0"], intr,
                                   0.05, sin20, n, m, fact)
        want = orc.icp_reduce(buf)
        got = np.array([A[i * 6 + j] if j < 6 else b[i] for i in range(6) for j in range(i, 7)], np.float32)
        same(got, want)
        bounds = [lgr * k // 4 for k in range(5)]
        b3 = buf.reshape(27, lgc, lgr)
        for r in range(4):
            lo, hi = bounds[r], bounds[r + 1]
            assert lib.rgbdslam_align_band_system(ctx._h, C.byref(prm), p(g["v1"]), p(g["n1"]), p(g["rgb1"]), p(g["gx1"]),
                                                  C.c_size_t(4 * m), p(g["gy1"]), C.c_size_t(4 * m), p(g["v0"]), p(g["n0"]),
                                                  p(g["rgbd0"]), n, m, lvl, 0, lo, hi, C.c_void_p(gathered[r].data_ptr())) == 0
            same(host(gathered[r]), orc.icp_reduce(np.ascontiguousarray(b3[:, :, lo:hi]).reshape(27, -1)))
    out = []
    for loop in (False, True):
        ctx.set_icp_loop(loop)
        pb = _pose_block(torch, I, z)
        hist = torch.zeros(2, 16, device="cuda")
        ctx.track_frame(prm, g["v1"], g["n1"], g["rgb1"], g["gx1"], g["gy1"], g["v0"], g["n0"], g["rgbd0"], n, m, pb, hist, 0)
        ctx.synchronize()
        if loop:
            assert ctx.icp_loop_faults() == 0
        out.append(host(pb).copy())
    ctx.set_icp_loop(False)
    same(out[1].view(np.int32), out[0].view(np.int32))
    ok, R, t, tr = orc.align_gauss_newton(orc.icp_params(), calib[:4], pr["v1"], pr["n1"], pr["rgb1"], pr["v0"], pr["n0"], pr["rgbd0"],
                                          I, z, I, z)
    assert ok and out[1].view(np.int32)[24] == 1 and out[1].view(np.int32)[25] == len(tr)
    same(out[1][:9], R)
    same(out[1][9:12], t)


@pytest.mark.parametrize("shape", [(480, 640), (61, 150)])
def test_frame_aux_records(ctx, pkg, orc, synth, shape):
    """rgbdslam_set_frame_aux: (a) the records the fused preprocessing leaves are the reference's four per-pixel expressions
    (ICP.cu:167-173) on the stored maps, bit for bit -- Gx / 8 and Gy / 8 in double, (r + g + b) / 3, the depth weight in double;
    (b) an announcement holds for ONE launch: the next preprocessing without one writes nothing; (c) the persistent ICP
    kernel fed with the records leaves the pose block of the run without them and of the multi-launch path, word for word,
    and the oracle's pose; (d) an announcement for track_frame is consumed by it."""
    import ctypes as C
    n, m = shape
    lib = ctx.lib
    calib = synth.calib_for(n, m)
    ctx.set_calibration_matrix(calib)
    ctx.alloc_buffers(60, 80)
    p = lambda t: C.c_void_p(t.data_ptr())
    try:
        rng = np.random.default_rng(12)
        R1, t1 = synth.pair_offset_pose()
        maps = []
        for k, (R, t) in enumerate(((np.eye(3), np.zeros(3)), (R1, t1))):
            d, c = synth.render(R, t, n, m, calib)
            d[rng.random((n, m)) < 0.02] = 0
            c = c.copy(); c.reshape(-1)[:256] = np.arange(256)                      # every byte value
            v, nm, rgb = (torch.empty(n, m, 3, device="cuda") for _ in range(3))
            gx, gy = torch.empty(n, m, device="cuda"), torch.empty(n, m, device="cuda")
            aux = torch.full((n, m, 4), -7.0, device="cuda")
            gd, gc = dev(d), dev(c)
            call = lambda: lib.rgbdslam_frame_preprocess(ctx._h, p(gd), C.c_size_t(2 * m), p(gc), C.c_size_t(3 * m), p(v), p(nm), p(rgb),
                                                         p(gx), C.c_size_t(4 * m), p(gy), C.c_size_t(4 * m), n, m)
            assert lib.rgbdslam_set_frame_aux(ctx._h, p(aux), None) == 0
            assert call() == 0
            ctx.synchronize()
            rv = orc.vertex_map_kinect(d, calib); rrgb = orc.cpy_char3(c)
            _, rgx, rgy = orc.intensity_sobel(rrgb)
            same(host(v), rv); same(host(gx), rgx); same(host(gy), rgy); same(host(rgb), rrgb)
            want = np.empty((n, m, 4), np.float32)
            want[..., 0] = (rgx.astype(np.float64) / 8.0).astype(np.float32)
            want[..., 1] = (rgy.astype(np.float64) / 8.0).astype(np.float32)
            want[..., 2] = ((rrgb[..., 0] + rrgb[..., 1]) + rrgb[..., 2]) / np.float32(3.0)
            sd = (-rv[..., 2]).astype(np.float64)
            want[..., 3] = (0.0012 / (0.0012 + 0.0019 * (sd - 0.4) * (sd - 0.4))).astype(np.float32)
            same(host(aux), want)
            alone = torch.full((n, m, 4), -7.0, device="cuda")          # the stand-alone form on the stored maps: the same records
            ctx.frame_aux(v, rgb, gx, gy, alone, n, m)
            same(host(alone), want)
            keep = aux.clone()
            aux.fill_(-7.0)
            assert call() == 0                                                      # (b) not announced again: nothing written
            ctx.synchronize()
            assert (host(aux) == -7.0).all()
            maps.append(dict(v=v, n=nm, rgb=rgb, gx=gx, gy=gy, aux=keep, hv=rv, hn=orc.compute_normal(rv), hrgb=rrgb))
        # model = frame 0 as the seeded model (VMap, NMap, RGBD with .w), current = frame 1
        f0, f1 = maps
        rgbd0 = np.concatenate([f0["hrgb"], np.zeros((n, m, 1), np.float32)], axis=2)
        g_rgbd0 = dev(rgbd0)
        prm = pkg.icp_params()
        I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
        out = {}
        for name, loop, use_aux in (("iter", False, False), ("iter+aux", False, True), ("loop", True, False), ("loop+aux", True, True),
                                    ("loop after", True, False)):
            ctx.set_icp_loop(loop)
            pb = _pose_block(torch, I, z)
            hist = torch.zeros(2, 16, device="cuda")
            if use_aux:
                assert lib.rgbdslam_set_frame_aux(ctx._h, None, p(f1["aux"])) == 0
            elif name == "loop after":
                f1["aux"].fill_(float("nan"))                                        # (d) were it still announced, the pose would be NaN
            ctx.track_frame(prm, f1["v"], f1["n"], f1["rgb"], f1["gx"], f1["gy"], f0["v"], f0["n"], g_rgbd0, n, m, pb, hist, 0)
            ctx.synchronize()
            assert ctx.icp_loop_faults() == 0
            out[name] = host(pb).copy().view(np.int32)
        ctx.set_icp_loop(False)
        for name in ("iter+aux", "loop", "loop+aux", "loop after"):
            same(out[name], out["iter"])
        ok, R, t, tr = orc.align_gauss_newton(orc.icp_params(), calib[:4], f1["hv"], f1["hn"], f1["hrgb"], f0["hv"], f0["hn"], rgbd0,
                                              I, z, I, z)
        assert out["loop+aux"][24] == int(ok) and out["loop+aux"][25] == len(tr)
        if ok:
            same(out["loop+aux"][:12].view(np.float32), np.concatenate([R, t]).astype(np.float32))
        # misaligned records are refused
        assert lib.rgbdslam_set_frame_aux(ctx._h, C.c_void_p(f1["aux"].data_ptr() + 4), None) == 1
    finally:
        ctx.set_icp_loop(False)
        ctx.set_calibration_matrix(synth.TUM_CALIB)


def test_decode_and_mesh_256_planes_full_size(ctx, orc):
    """BASELINE configs[2] at its full size: 256 planes x 512x512 Bump cells, all byte values in the masks.
    Size-independent properties: (a) eight plane ranges (the 8-GPU partition) reproduce the single launch bit for
    bit and leave the planes outside their range untouched; (b) the per-plane entry point equals the batch;
    (c) sampled planes equal the oracle; (d) the fused mesh launch equals decode + the oracle's normals and trim
    on a sampled plane."""
    planes, N = 256, 512
    g = torch.Generator(device="cuda").manual_seed(11)
    bump = torch.randint(-32768, 32767, (planes, N, N, 3), dtype=torch.int16, device="cuda", generator=g)
    rgbb = torch.randint(0, 256, (planes, N, N, 3), dtype=torch.uint8, device="cuda", generator=g)
    mask = torch.randint(0, 256, (planes, N, N), dtype=torch.uint8, device="cuda", generator=g)
    rng = np.random.default_rng(5)
    prm = np.zeros((planes, 14), np.float32)
    for k in range(planes):
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        e1 = np.cross(nrm, [0.3, -0.5, 0.8]); e1 /= np.linalg.norm(e1)
        e2 = np.cross(nrm, e1)
        prm[k] = [*nrm, rng.uniform(-3, 3), 500.0, 500.0, rng.uniform(-2, 0), rng.uniform(-2, 0), *e1, *e2]
    param = dev(prm)
    tab = lambda t: torch.tensor([t[k].data_ptr() for k in range(planes)], dtype=torch.int64, device="cuda")
    itabs = [tab(x) for x in (bump, rgbb, mask, param)]

    def decode(ranges):
        outs = [torch.full((planes, N, N, ch), -7.0, device="cuda") for ch in (3, 4, 4)]
        for lo, hi in ranges:
            ctx.vertex_from_bump_batch(tab(outs[0]), tab(outs[1]), tab(outs[2]), *itabs, lo, hi, N, N)
        ctx.synchronize()
        return outs
    full = decode([(0, planes)])
    parts = decode([(planes * r // 8, planes * (r + 1) // 8) for r in range(8)])
    for a, b in zip(full, parts):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    only = decode([(32, 64)])
    assert float(only[0][:32].min()) == -7.0 and float(only[0][64:].max()) == -7.0
    assert torch.equal(only[0][32:64].view(torch.int32), full[0][32:64].view(torch.int32))
    for k in (0, 77, 255):
        v = torch.empty(N, N, 3, device="cuda"); c = torch.empty(N, N, 4, device="cuda"); k4 = torch.empty(N, N, 4, device="cuda")
        ctx.vertex_from_bump(v, c, k4, bump[k], rgbb[k], mask[k], param[k], N, N, 1)
        rv, rc, rk = orc.vertex_from_bump(host(bump[k]).view(np.uint16), host(rgbb[k]), host(mask[k]), prm[k], 1)
        for got, batch, want in ((v, full[0][k], rv), (c, full[1][k], rc), (k4, full[2][k], rk)):
            same(host(got), want)
            same(host(batch), want)
    # mesh of the whole model in one launch vs the oracle on one plane
    nout = torch.empty(planes, N, N, 3, device="cuda")
    iout = torch.empty(planes, (N - 1) * (N - 1), 4, dtype=torch.int32, device="cuda")
    outs = [torch.empty(planes, N, N, ch, device="cuda") for ch in (3, 4, 4)]
    ctx.bump_mesh_batch(tab(outs[0]), tab(outs[1]), tab(outs[2]), tab(nout), tab(iout), *itabs, 0, planes, N, N, 0, 0.1, False, 0)
    ctx.synchronize()
    for a, b in zip(full, outs):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    k = 131
    rv = host(full[0][k])
    same(host(nout[k]), orc.compute_normal(rv, False))
    same(host(iout[k]).view(np.uint32), orc.quad_trim(rv, host(mask[k]), 0, 0.1, 1, 0))


def test_update_many_planes_eight_way_partition(ctx, orc, pair, synth):
    """Fusion into a model of 120 planes (labels over the whole uint8 range that is in use), two frames with
    different confidences: eight plane ranges (the 8-GPU partition of configs[2]) == one launch == the oracle."""
    n, m = 480, 640
    nb, N_prim = 120, 256
    rng = np.random.default_rng(17)
    v = pair["v0"]
    params = []
    for k in range(nb):
        # planes through random pixels of the frame, random in-plane axes: every plane receives observations
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        e1 = np.cross(nrm, [0.2, 0.9, -0.4]); e1 /= np.linalg.norm(e1)
        e2 = np.cross(nrm, e1)
        c = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(1, 3)])
        scale = np.float32(2.0 / 0.004)
        sh0 = np.float32(c @ e1 - N_prim * 0.004 / 2.0); sh1 = np.float32(c @ e2 - N_prim * 0.004 / 2.0)
        params.append(np.array([*nrm, -(nrm @ c), scale, scale, sh0, sh1, *e1, *e2], np.float32))
    label = rng.integers(0, nb + 1, (n, m)).astype(np.uint8)
    I16 = pose16(np.eye(3), np.zeros(3))
    frames = [rng.integers(0, 40, (n, m)).astype(np.uint8), rng.integers(0, 80, (n, m)).astype(np.uint8)]

    def run(parts):
        g_b = [torch.zeros(N_prim, N_prim, 3, dtype=torch.int16, device="cuda") for _ in range(nb)]
        g_c = [torch.zeros(N_prim, N_prim, 3, dtype=torch.uint8, device="cuda") for _ in range(nb)]
        g_m = [torch.zeros(N_prim, N_prim, dtype=torch.uint8, device="cuda") for _ in range(nb)]
        g_k = [torch.zeros(N_prim, N_prim, dtype=torch.int64, device="cuda") for _ in range(nb)]
        g_p = [dev(p) for p in params]
        tabs = [_ptr_table(x) for x in (g_b, g_c, g_m, g_p, g_k)]
        for stamp, mk in enumerate(frames, start=1):
            for lo, hi in parts:
                ctx.update(tabs[0], tabs[1], tabs[2], tabs[3], tabs[4], stamp, dev(v), dev(pair["rgbd0"]), dev(mk), dev(label), I16,
                           N_prim, N_prim, n, m, lo, hi)
        return [host(x) for x in g_b + g_c + g_m]

    full = run([(0, 2 ** 31 - 1)])
    parts = run([(nb * r // 8, nb * (r + 1) // 8) for r in range(8)])
    for a, b in zip(full, parts):
        same(a, b)
    bumps = [np.zeros((N_prim, N_prim, 3), np.uint16) for _ in range(nb)]
    rgbs = [np.zeros((N_prim, N_prim, 3), np.uint8) for _ in range(nb)]
    masks = [np.zeros((N_prim, N_prim), np.uint8) for _ in range(nb)]
    claims = [np.zeros((N_prim, N_prim), np.uint64) for _ in range(nb)]
    for stamp, mk in enumerate(frames, start=1):
        orc.update(bumps, rgbs, masks, params, claims, stamp, v, pair["rgbd0"], mk, label, I16)
    for k in range(nb):
        same(full[k].view(np.uint16), bumps[k])
        same(full[nb + k], rgbs[k])
        same(full[2 * nb + k], masks[k])
    assert sum(int((mk_ > 10).sum()) for mk_ in masks) > 1000


def test_hbm_stream_probe_moves_what_it_counts(ctx, pkg):
    """The bandwidth probe behind bench.py's `hbm_stream_probe`: copy mode copies every 16-byte word (ragged tail of
    < 16 bytes left alone), fill mode writes zeros, read mode writes nothing; bad arguments are refused."""
    import ctypes as C
    lib = pkg.load()
    nbytes = (1 << 20) * 16 + 16 * 7                       # not a multiple of the grid size
    src = torch.arange(nbytes // 4, dtype=torch.float32, device="cuda")
    dst = torch.full((nbytes // 4,), -1.0, device="cuda")
    rate = C.c_double()
    call = lambda mode, nb=nbytes, reps=2: lib.rgbdslam_profile_hbm_stream(ctx._h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                                                           C.c_size_t(nb), mode, reps, C.byref(rate))
    assert call(2) == 0 and rate.value > 0
    assert (dst == -1.0).all()                             # read mode leaves dst alone
    assert call(0, nbytes - 20) == 0 and rate.value > 0    # 20 bytes short: the last whole 16-byte word is not touched
    n_copied = (nbytes - 20) // 16 * 4
    assert torch.equal(dst[:n_copied], src[:n_copied]) and (dst[n_copied:] == -1.0).all()
    assert call(1) == 0 and (dst == 0.0).all()
    assert call(3) != 0 and call(0, 8) != 0 and call(0, nbytes, 0) != 0
    assert lib.rgbdslam_profile_hbm_stream(ctx._h, C.c_void_p(dst.data_ptr() + 4), C.c_void_p(src.data_ptr()), C.c_size_t(64), 0, 1,
                                           C.byref(rate)) != 0   # misaligned


@pytest.mark.parametrize("shape", [(37, 53), (45, 91), (64, 64), (33, 1024)])
@pytest.mark.parametrize("skew", [(0, 0, 0), (1, 1, 3), (2, 2, 2), (3, 0, 1)])
def test_decode_batch_ragged_sizes_and_unaligned_planes(ctx, orc, shape, skew):
    """The whole-model decode fetches a workgroup's inputs with full-width loads when its 1024 cells are all there and
    the plane's pointers are word-aligned, and cell by cell otherwise: both ways, and their mix inside one launch
    (full workgroups followed by a ragged one; aligned and skewed planes side by side), must give the same bits.
    skew = element offsets of the (Bump u16, RGB u8, Mask u8) planes inside their allocations."""
    n, m = shape
    rng = np.random.default_rng(n * 1000 + m + sum(skew))
    param = np.array([0.3, -0.5, 0.81, -1.7, 500.0, 500.0, -0.51, 0.27, 0.9, 0.1, -0.42, 0.2, 0.84, 0.5], np.float32)
    planes = []
    for k, (sb, sc, sm) in enumerate(((0, 0, 0), skew)):
        bump = rng.integers(0, 65536, (n, m, 3)).astype(np.uint16)
        rgbb = rng.integers(0, 256, (n, m, 3)).astype(np.uint8)
        mask = rng.integers(0, 60, (n, m)).astype(np.uint8)
        store_b = torch.zeros(bump.size + 8, dtype=torch.int16, device="cuda")
        store_c = torch.zeros(rgbb.size + 8, dtype=torch.uint8, device="cuda")
        store_m = torch.zeros(mask.size + 8, dtype=torch.uint8, device="cuda")
        gb = store_b[sb:sb + bump.size]; gb.copy_(dev(bump.view(np.int16).reshape(-1)))
        gc = store_c[sc:sc + rgbb.size]; gc.copy_(dev(rgbb.reshape(-1)))
        gm = store_m[sm:sm + mask.size]; gm.copy_(dev(mask.reshape(-1)))
        assert gb.data_ptr() % 8 == (2 * sb) % 8 and gc.data_ptr() % 4 == sc % 4 and gm.data_ptr() % 4 == sm % 4
        planes.append((bump, rgbb, mask, gb, gc, gm, (store_b, store_c, store_m)))
    gp = dev(param)
    outs = [[torch.full((n, m, ch), -7.0, device="cuda") for _ in planes] for ch in (3, 4, 4)]
    ctx.vertex_from_bump_batch(_ptr_table(outs[0]), _ptr_table(outs[1]), _ptr_table(outs[2]), _ptr_table([p[3] for p in planes]),
                               _ptr_table([p[4] for p in planes]), _ptr_table([p[5] for p in planes]), _ptr_table([gp, gp]), 0, 2, n, m)
    for k, p in enumerate(planes):
        rv, rc, rk = orc.vertex_from_bump(p[0], p[1], p[2], param, 1)
        same(host(outs[0][k]), rv)
        same(host(outs[1][k]), rc)
        same(host(outs[2][k]), rk)


@pytest.mark.parametrize("shape,nb,density", [((480, 640), 3, 0.6), ((123, 150), 100, 0.1), ((61, 64), 255, 0.02), ((9, 700), 30, 0.5)])
def test_segment_merge_claim_one_launch(ctx, pkg, orc, pair, synth, shape, nb, density):
    """rgbdslam_model_segment_merge_claim -- Merge, the three growth passes out of LDS and the claim pass in ONE launch --
    against the operator sequence rgbdslam_model_normal_segment + rgbdslam_model_merge_claim (four launches) and the
    oracle's Segment: labels, merged maps and claim words bit for bit.  Cases: the steady state (few pixels without a
    label), sparse labels (more queued pixels per tile than lanes), more planes than the LDS table holds (64) and than
    one staging round holds (25), ragged sizes, a strip narrower than the label halo."""
    import ctypes as C
    n, m = shape
    lib = ctx.lib
    rng = np.random.default_rng(n * m + nb)
    crop = lambda a: np.ascontiguousarray(np.tile(a, (1, 2) + (1,) * (a.ndim - 2))[:n, :m])
    v = crop(pair["v0"]).copy()
    v[5:8, 20:40] = 0                                          # no vertex: such pixels never get a label
    eq3, _ = synth.plane_equations()
    eq = np.zeros((nb, 10), np.float32)
    for k in range(nb):                                        # the scene's three planes, pushed away a little more each time
        eq[k] = eq3[k % 3]
        eq[k, 9] += np.float32(0.02 * (k // 3))
    Rg = synth.euler_xyz(0.004, -0.003, 0.002)
    Pg = pose16(Rg, np.array([0.002, -0.001, 0.003]))
    nrm = orc.compute_normal(v)
    truth = orc.seed_labels(v, nrm, Pg, eq[:3])
    lab0 = np.where(rng.random((n, m)) < density, truth, 0).astype(np.uint8)
    if nb > 3:                                                 # the same plane under other labels: 3, 6, ... steps further away
        lab0 = np.where(lab0 > 0, lab0 + 3 * rng.integers(0, (nb - lab0.max()) // 3 + 1, (n, m)), 0).astype(np.uint8)
    lab0[n // 3:n // 3 + 12, m // 4:m // 4 + 90] = 0           # a block without labels (every pixel of some tiles is queued)
    lab0[n // 2:n // 2 + 30, m // 8:m // 8 + 220] = 0          # a larger one: tiles whose 3 rings hold no label at all are skipped
    rgbd = crop(pair["rgbd0"]).copy()
    rgbd[..., 3][n // 2:] *= 1.02                               # a few merge branches
    mask = rng.integers(9, 60, (n, m)).astype(np.uint8)
    depth, rgb = crop(pair["d1"]), crop(pair["rgb1"])
    N_prim = 128
    params = []
    for k in range(nb):
        E = eq[k]
        cpt = np.array([0.0, 0.0, -2.0]); cpt = cpt - (E[:3] @ cpt - E[9]) * E[:3]
        sh0 = np.float32(cpt @ E[3:6] - N_prim * 0.02 / 2.0); sh1 = np.float32(cpt @ E[6:9] - N_prim * 0.02 / 2.0)
        params.append(np.array([E[0], E[1], E[2], E[9], 100.0, 100.0, sh0, sh1, *E[3:9]], np.float32))
    cellmask = [rng.integers(0, 40, (N_prim, N_prim)).astype(np.uint8) for _ in range(nb)]
    pb = torch.zeros(76, dtype=torch.float32, device="cuda")
    pb_h = np.zeros(76, np.float32)
    pb_h[24:28] = np.array([1, 5, 0, 1], np.int32).view(np.float32)
    pb_h[60:76] = Pg
    pb.copy_(dev(pb_h))
    g_v, g_eq, g_rgbd, g_mask, g_d, g_rgb = dev(v), dev(eq), dev(rgbd), dev(mask), dev(depth), dev(rgb)
    g_m = [dev(x) for x in cellmask]
    g_p = [dev(x) for x in params]
    P = lambda t: C.c_void_p(t.data_ptr())
    f = C.c_float
    results = []
    for fused in (False, True):
        lab, swap = dev(lab0), torch.zeros(n, m, dtype=torch.uint8, device="cuda")
        nseg = torch.zeros(n, m, 3, device="cuda")
        o_rgbd, o_mask = torch.zeros(n, m, 4, device="cuda"), torch.zeros(n, m, dtype=torch.uint8, device="cuda")
        o_v, o_n = torch.zeros(n, m, 3, device="cuda"), torch.zeros(n, m, 3, device="cuda")
        g_k = [torch.zeros(N_prim, N_prim, dtype=torch.int64, device="cuda") for _ in range(nb)]
        t_m, t_p, t_k = _ptr_table(g_m), _ptr_table(g_p), _ptr_table(g_k)
        if fused:
            rc = lib.rgbdslam_model_segment_merge_claim(ctx._h, P(pb), P(nseg), P(g_v), P(lab), P(swap), P(g_eq), nb, f(0.1), f(0.9),
                                                        P(g_rgbd), P(g_mask), P(g_d), C.c_size_t(2 * m), P(g_rgb), P(o_rgbd), P(o_mask),
                                                        P(o_v), P(o_n), P(t_m), P(t_p), P(t_k), C.c_uint32(7), N_prim, N_prim, n, m, 0, nb)
            assert rc == 0
        else:
            rc = lib.rgbdslam_model_normal_segment(ctx._h, P(pb), P(nseg), P(g_v), P(lab), P(swap), P(g_eq), nb, f(0.1), f(0.9), n, m)
            assert rc == 0
            rc = lib.rgbdslam_model_merge_claim(ctx._h, P(pb), P(g_rgbd), P(g_mask), P(g_d), C.c_size_t(2 * m), P(g_rgb), P(o_rgbd),
                                                P(o_mask), P(o_v), P(o_n), P(swap), P(t_m), P(t_p), P(t_k), C.c_uint32(7), N_prim, N_prim,
                                                n, m, 0, nb)
            assert rc == 0
        ctx.synchronize()
        results.append([host(x) for x in (swap, o_rgbd, o_mask, o_v, o_n)] + [np.stack([host(k) for k in g_k])])
    for a, b in zip(results[1], results[0]):
        same(a, b)
    seg_ref = orc.segment(lab0, v, nrm, Pg, eq, 0.1, 0.9)
    same(results[1][0], seg_ref)
    grown = (seg_ref > 0).sum() - (lab0 > 0).sum()
    assert grown > 0 and (seg_ref == 0).sum() > 0
    assert (results[1][5] != 0).sum() > 0                      # claims were made


@pytest.mark.parametrize("shape,lvl", [((256, 256), 1), ((200, 136), 2), ((67, 130), 3)])
def test_vertex_from_bump_demostep2_forms(ctx, orc, shape, lvl):
    """SURVEY 8f N4, the decode as DemoStep2 calls it (D2/Primitive.cu:31-179): 16-bit mask + vertices only on the
    sub-sampled grid (variant 0), 8-bit mask + vertices and RGBA on the full grid (variant 1); empty = mask < 11."""
    n, m = shape
    rng = np.random.default_rng(n + lvl)
    bump = rng.integers(0, 65536, (n, m, 3)).astype(np.uint16)
    rgbb = rng.integers(0, 256, (n, m, 3)).astype(np.uint8)
    param = np.array([0.0, 0.6, 0.8, -1.7, 500.0, 480.0, -0.9, -1.1, 1.0, 0.0, 0.0, 0.0, 0.8, -0.6], np.float32)
    mask16 = rng.integers(0, 40, (n, m)).astype(np.uint16) * 200 // 7            # values around the threshold and far above 255
    mask16[::5, ::3] = 10
    mask16[1::5, ::3] = 11
    mask16[2::5, ::3] = 1                                                        # D1's special value: empty here
    no, mo = n // lvl, m // lvl
    v = torch.full((no, mo, 3), -7.0, device="cuda")
    ctx.vertex_from_bump_d2(v, None, dev(bump), None, dev(mask16), dev(param), n, m, lvl, 0)
    want = orc.vertex_from_bump_d2(bump, rgbb, mask16, param, lvl, 0)
    same(host(v), want)
    sub = mask16[:no * lvl:lvl, :mo * lvl:lvl]
    assert ((want == 0).all(axis=2) == (sub < 11)).all() and (sub < 11).any() and (sub >= 11).any()
    mask8 = np.minimum(mask16, 255).astype(np.uint8)
    v1 = torch.full((n, m, 3), -7.0, device="cuda")
    c1 = torch.full((n, m, 4), -7.0, device="cuda")
    ctx.vertex_from_bump_d2(v1, c1, dev(bump), dev(rgbb), dev(mask8), dev(param), n, m, lvl, 1)
    wv, wc = orc.vertex_from_bump_d2(bump, rgbb, mask8, param, lvl, 1)
    same(host(v1), wv)
    same(host(c1), wc)
    # where D1's decode (mask >= 11) and DemoStep2's agree on emptiness they give the same vertex
    d1v, _, _ = orc.vertex_from_bump(bump, rgbb, mask8, param, lvl)
    keep = mask8[:no * lvl:lvl, :mo * lvl:lvl] >= 11
    same(want[keep & (sub >= 11)], d1v[keep & (sub >= 11)])


@pytest.mark.parametrize("tracked", [1, 0])
def test_rider_entry_points_equal_separate_calls(ctx, pkg, orc, pair, synth, tracked):
    """The two entry points that let an independent job ride on a frame launch, against the same jobs as separate calls,
    bit for bit: rgbdslam_model_segment_merge_claim_preprocess (next frame's LoadFrame on the merge launch) and
    rgbdslam_model_predict_apply (previous frame's apply pass on the z-buffer launch, posed by its history entry) --
    for a tracked block and for a lost one (the predict then re-seeds the model and nothing else of the frame runs; the
    riders do not look at this frame's block)."""
    import ctypes as C
    n, m, nb, N_prim = 480, 640, 3, 256
    lib = ctx.lib
    ctx.set_calibration_matrix(synth.TUM_CALIB)
    rng = np.random.default_rng(11 + tracked)
    eq, _ = synth.plane_equations()
    Pg = pose16(synth.euler_xyz(0.003, -0.002, 0.001), np.array([0.001, -0.002, 0.002]))
    Prel = pose16(synth.euler_xyz(0.002, 0.001, -0.001), np.array([0.003, 0.001, -0.002]))
    Prel_inv = np.linalg.inv(Prel.reshape(4, 4).T).T.reshape(-1).astype(np.float32)
    v0, nrm0 = pair["v0"], pair["n0"]
    lab0 = orc.seed_labels(v0, nrm0, Pg, eq)
    lab0[rng.random((n, m)) < 0.05] = 0
    mask0 = rng.integers(9, 60, (n, m)).astype(np.uint8)
    params = _plane_setup(synth, N_prim)[1]
    pb_h = np.zeros(76, np.float32)
    pb_h[24:28] = np.array([tracked, 5, 0, 1], np.int32).view(np.float32)
    pb_h[28:44], pb_h[44:60], pb_h[60:76] = Prel, Prel_inv, Pg
    hist_h = np.zeros(16, np.float32)                           # the PREVIOUS frame's entry: poseR (col-major 3x3), poset, tracked
    hist_h[:9] = Pg.reshape(4, 4)[:3, :3].reshape(-1)
    hist_h[9:12] = Pg[12:15]
    hist_h[12] = 1.0
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    f = C.c_float
    g_eq = dev(eq)
    g_p = [dev(x) for x in params]
    g_d0, g_d1, g_c1 = dev(pair["d0"]), dev(pair["d1"]), dev(pair["c1"])
    g_fv, g_fn, g_frgb = dev(pair["v1"]), dev(pair["n1"]), dev(pair["rgb1"])
    out = []
    for riders in (False, True):
        pb = dev(pb_h); hist = dev(hist_h)
        V, R, M, L = dev(v0), dev(pair["rgbd0"]), dev(mask0), dev(lab0)       # model set "in" (re-seeded when lost)
        Nin = dev(nrm0)
        Va, Ra = torch.zeros(n, m, 3, device="cuda"), torch.zeros(n, m, 4, device="cuda")
        Ma, La = torch.zeros(n, m, dtype=torch.uint8, device="cuda"), torch.zeros(n, m, dtype=torch.uint8, device="cuda")
        zbuf = torch.full((n, m), -1, dtype=torch.int64, device="cuda")
        vs = torch.zeros(5 * n * m, dtype=torch.float32, device="cuda")          # RGBDSLAM_PREDICT_SCRATCH_BYTES
        g_b = [torch.zeros(N_prim, N_prim, 3, dtype=torch.int16, device="cuda") for _ in range(nb)]
        g_c = [torch.zeros(N_prim, N_prim, 3, dtype=torch.uint8, device="cuda") for _ in range(nb)]
        g_m = [torch.full((N_prim, N_prim), 10, dtype=torch.uint8, device="cuda") for _ in range(nb)]
        g_k = [torch.zeros(N_prim, N_prim, dtype=torch.int64, device="cuda") for _ in range(nb)]
        t_b, t_c, t_m, t_p, t_k = (_ptr_table(x) for x in (g_b, g_c, g_m, g_p, g_k))
        # the previous frame's claim pass (stamp 3) on the model it left
        assert lib.rgbdslam_model_merge_claim(ctx._h, P(dev(np.concatenate([pb_h[:24], np.array([1, 5, 0, 1], np.int32).view(np.float32), pb_h[28:]]))),
                                              P(R), P(M), P(g_d0), C.c_size_t(2 * m), P(dev(pair["rgb0"])), P(Ra), P(Ma), P(Va),
                                              P(torch.zeros(n, m, 3, device="cuda")), P(L), P(t_m), P(t_p), P(t_k), C.c_uint32(3), N_prim,
                                              N_prim, n, m, 0, nb) == 0
        R.copy_(Ra); M.copy_(Ma); V.copy_(Va)                                  # = the merged model of the previous frame
        pred = (P(pb), P(V), P(R), P(M), P(L), P(zbuf), P(Va), P(Ra), P(Ma), P(La), f(0.1), f(0.5), f(10.0), n, m, P(g_fv), P(g_fn),
                P(g_frgb), P(g_d1), C.c_size_t(2 * m), P(Nin), P(vs) if riders else P(None))   # the scratch form must give the same bits
        apply_tabs = (P(t_b), P(t_c), P(t_m), P(t_p), P(t_k), C.c_uint32(3))
        if riders:
            assert lib.rgbdslam_model_predict_apply(ctx._h, *pred, P(hist), *apply_tabs, N_prim, N_prim, 0, nb) == 0
        else:
            assert lib.rgbdslam_update_apply_from_history(ctx._h, P(hist), *apply_tabs, P(V), P(R), P(M), P(L), N_prim, N_prim, n, m, 0, nb) == 0
            assert lib.rgbdslam_model_predict(ctx._h, *pred) == 0
        # this frame: growth passes + merge + claim (stamp 4) (+ the next frame's preprocessing)
        nseg = torch.zeros(n, m, 3, device="cuda")
        fV, fN, fRGB = (torch.full((n, m, 3), -3.0, device="cuda") for _ in range(3))
        gx, gy = torch.full((n, m), -3.0, device="cuda"), torch.full((n, m), -3.0, device="cuda")
        smc = (P(pb), P(nseg), P(Va), P(La), P(L), P(g_eq), nb, f(0.1), f(0.9), P(Ra), P(Ma), P(g_d1), C.c_size_t(2 * m), P(g_frgb),
               P(R), P(M), P(V), P(Nin), P(t_m), P(t_p), P(t_k), C.c_uint32(4), N_prim, N_prim, n, m, 0, nb)
        nxt = (P(g_d1), C.c_size_t(2 * m), P(g_c1), C.c_size_t(3 * m), P(fV), P(fN), P(fRGB), P(gx), C.c_size_t(4 * m), P(gy), C.c_size_t(4 * m))
        if riders:
            assert lib.rgbdslam_model_segment_merge_claim_preprocess(ctx._h, *smc, *nxt) == 0
        else:
            assert lib.rgbdslam_model_segment_merge_claim(ctx._h, *smc) == 0
            assert lib.rgbdslam_frame_preprocess(ctx._h, *nxt, n, m) == 0
        ctx.synchronize()
        out.append([host(x) for x in (V, R, M, L, Nin, Va, Ra, Ma, La, fV, fN, fRGB, gx, gy, zbuf)] +
                   [np.stack([host(x) for x in tab]) for tab in (g_b, g_c, g_m, g_k)])
    for a, b in zip(out[1], out[0]):
        same(a, b)
    bumps, cellmasks = out[1][15], out[1][17]
    assert (cellmasks > 10).sum() > 1000 and (bumps != 0).sum() > 1000        # the deferred apply pass wrote cells
    same(out[1][9], pair["v1"])                                                # the next frame's maps are the oracle's
    same(out[1][10], pair["n1"])
    if not tracked:
        same(out[1][0], pair["v1"])                                            # lost: model := input frame
        assert (out[1][14] == -1).all()
