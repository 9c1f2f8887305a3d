"""GPU parity of the remaining operator functions the reference's host code calls (csrc/compat.hip): Kinect-layout merge /
model seed, stand-alone grey copy, live-Kinect frame mapping, label vote / remap (RG::ReorganizeEqua), plane overlap,
pointer-table store, and the reference-shaped Update_cu / Segment_cu (library-owned claim slabs / swap image).
All bit-exact against the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_gpu_kernels import dev, host, same, pose16, _plane_setup, _ptr_table  # noqa: E402


@pytest.fixture(scope="module")
def ctx(pkg, synth):
    c = pkg.Context(0)
    c.set_calibration_matrix(synth.TUM_CALIB)
    yield c
    c.close()


def test_kinect_layout_merge_and_seed(ctx, orc, pair):
    n, m = 480, 640
    calib = pair["calib"]
    rng = np.random.default_rng(11)
    dk = np.ascontiguousarray(pair["d1"][::-1, ::-1])        # live-Kinect layout of the same depth image, shifted a little
    dk = dk.copy()
    dk[rng.random((n, m)) < 0.02] = 0
    rgbd = pair["rgbd0"].copy()
    rgbd[100:140, 200:260, 3] += 0.004                        # occlusion / visibility-violation branches
    rgbd[300:330, 100:160, 3] -= 0.004
    rgbd[50:60, 50:70, 3] = 0.0
    mask = rng.integers(9, 256, (n, m)).astype(np.uint8)
    mask[rng.random((n, m)) < 0.05] = 1
    g_rgbd, g_mask = dev(rgbd), dev(mask)
    ctx.merge_depth_kinect(g_rgbd, dev(dk), dev(pair["rgb1"]), g_mask, n, m)
    r_rgbd, r_mask = orc.merge_depth_kinect(rgbd, dk, pair["rgb1"], mask, calib)
    same(host(g_rgbd), r_rgbd)
    same(host(g_mask), r_mask)
    assert len(np.unique(r_mask)) > 5 and (r_mask == 10).sum() > 100
    # the Off and Kinect kernels are the same function of the same physical depth sample
    o_rgbd, o_mask = orc.merge_depth_off(rgbd, np.ascontiguousarray(dk[:, ::-1]), pair["rgb1"], mask, calib)
    same(o_rgbd, r_rgbd)
    same(o_mask, r_mask)
    out = torch.empty(n, m, 4, device="cuda")
    ctx.gpu_add_zbuff_kinect(dev(dk), dev(pair["rgb1"]), out, n, m)
    same(host(out), orc.add_zbuff_kinect(dk, pair["rgb1"], calib))


def test_gpu_cpy_pitched(ctx, orc, pair):
    n, m = 480, 640
    pitch = 4 * 704
    img = torch.full((n, 704), -3.0, device="cuda")
    ctx.gpu_cpy(img, pitch, dev(pair["rgb1"]), n, m)
    got = host(img)
    same(got[:, :m], orc.gpu_cpy(pair["rgb1"]))
    assert (got[:, m:] == -3.0).all()
    ri, _, _ = orc.intensity_sobel(pair["rgb1"])              # the fused kernel's grey image is the same image
    same(got[:, :m], ri)


def test_map_data_2_vbo(ctx, orc, synth):
    n, m = 97, 131
    calib = synth.calib_for(n, m)
    ctx.set_calibration_matrix(calib)
    rng = np.random.default_rng(5)
    depth = rng.integers(0, 9000, (n, m)).astype(np.uint16)
    color = rng.integers(0, 256, (n, m, 4)).astype(np.uint8)
    jj, ii = np.meshgrid(np.arange(m), np.arange(n))
    coord = np.stack([jj + rng.integers(-6, 7, (n, m)), ii + rng.integers(-6, 7, (n, m))], -1).astype(np.int32)   # some fall outside
    v = torch.zeros(n, m, 3, device="cuda")
    rgb = torch.zeros(n, m, 3, device="cuda")
    ctx.map_data_2_vbo(dev(color), dev(depth), dev(coord), v, rgb, n, m)
    rv, rrgb = orc.map_data_2_vbo(color, depth, coord, calib)
    same(host(v), rv)
    same(host(rgb), rrgb)
    assert (rv[..., 2] != 0).mean() > 0.5 and (rv[..., 2] == 0).sum() > 100
    ctx.set_calibration_matrix(synth.TUM_CALIB)


def test_vote_and_affect_plan(ctx, orc, pair, synth):
    n, m = 480, 640
    eq, _ = synth.plane_equations()
    I16 = pose16(np.eye(3), np.zeros(3))
    label = orc.seed_labels(pair["v0"], pair["n0"], I16, eq)
    label[200:230, 300:360] = 7                                # a label with no plane behind it, few pixels
    P = pose16(pair["R1"], pair["t1"])
    nb = 9
    centers = torch.zeros(3 * nb, device="cuda")
    count = torch.zeros(nb, device="cuda")
    ctx.vote_plan(dev(label), dev(pair["v0"]), centers, count, P, nb, n, m)
    rc, rn = orc.vote_plan(label, pair["v0"], P, nb)
    same(host(centers), rc)
    same(host(count), rn)
    assert rn[6] == 30 * 60 and (rn[:3] > 5000).all() and rn[3] == 0
    # against a float64 sum: the defined order is an ordinary float32 summation of the reference's terms
    for L in range(3):
        sel = label == L + 1
        v = pair["v0"][sel].astype(np.float64)
        Pm = np.array(P, np.float64).reshape(4, 4).T
        want = (v @ Pm[:3, :3].T + Pm[:3, 3]).sum(0)
        assert np.allclose(rc[3 * L:3 * L + 3], want, rtol=2e-4, atol=1e-2)
    # a second call adds to the buffers (the reference's atomics accumulate into whatever is there)
    ctx.vote_plan(dev(label), dev(pair["v0"]), centers, count, P, nb, n, m)
    rc2, rn2 = orc.vote_plan(label, pair["v0"], P, nb, rc, rn)
    same(host(centers), rc2)
    same(host(count), rn2)
    # ReorganizeEqua's remap (RG.cpp:953-965): planes with more than MIN_SIZE_PLAN votes are renumbered densely
    assoc = np.zeros(nb, np.int32)
    k = 0
    for i in range(nb):
        if int(rn[i]) > 5000:
            k += 1
            assoc[i] = k
    g_lab = dev(label)
    ctx.affect_plan(g_lab, dev(assoc), n, m)
    want = orc.affect_plan(label, assoc)
    same(host(g_lab), want)
    assert (want[200:230, 300:360] == 0).all() and set(np.unique(want)) == {0, 1, 2, 3}


def test_overlap(ctx, orc, pair, synth):
    n, m = 480, 640
    N_prim = 512
    eq, params = _plane_setup(synth, N_prim)
    I16 = pose16(np.eye(3), np.zeros(3))
    label = orc.seed_labels(pair["v0"], pair["n0"], I16, eq)
    bumps = [np.zeros((N_prim, N_prim, 3), np.uint16) for _ in range(3)]
    rgbs = [np.zeros((N_prim, N_prim, 3), np.uint8) for _ in range(3)]
    masks = [np.zeros((N_prim, N_prim), np.uint8) for _ in range(3)]
    claims = [np.zeros((N_prim, N_prim), np.uint64) for _ in range(3)]
    orc.update(bumps, rgbs, masks, params, claims, 1, pair["v0"], pair["rgbd0"], np.full((n, m), 12, np.uint8), label, I16)
    assert (masks[0] > 10).sum() > 1000
    # the same plane shifted by 0.6 m against its own e1 direction: part of the occupied cells leave the other image
    p_in = params[0].copy()
    p_in[6] -= 0.6
    for q in (params[0], p_in, params[1]):
        ratio, cnt = ctx.overlap(dev(bumps[0]), dev(masks[0]), dev(params[0]), dev(q), N_prim, N_prim, N_prim, N_prim)
        r_ratio, r_cnt = orc.overlap(bumps[0], masks[0], params[0], q, N_prim, N_prim)
        assert cnt == (int(r_cnt[0]), int(r_cnt[1])) and ratio == np.float32(r_ratio)
    assert ctx.overlap(dev(bumps[0]), dev(masks[0]), dev(params[0]), dev(params[0]), N_prim, N_prim, N_prim, N_prim)[0] == 1.0
    r_shift, _ = ctx.overlap(dev(bumps[0]), dev(masks[0]), dev(params[0]), dev(p_in), N_prim, N_prim, N_prim, N_prim)
    assert 0.0 < r_shift < 1.0
    empty = np.zeros((N_prim, N_prim), np.uint8)
    assert ctx.overlap(dev(bumps[0]), dev(empty), dev(params[0]), dev(p_in), N_prim, N_prim, N_prim, N_prim)[0] == 1.0


def test_reference_shaped_update_and_segment(ctx, orc, pair, synth):
    """Update_cu / Segment_cu with the reference's argument lists: tables filled entry by entry (Affect*), claim slabs
    and swap image owned by the library.  Same bits as the explicit forms / the oracle, over three rounds."""
    n, m = 480, 640
    N_prim = 512
    eq, params = _plane_setup(synth, N_prim)
    I16 = pose16(np.eye(3), np.zeros(3))
    label = orc.seed_labels(pair["v0"], pair["n0"], I16, eq)
    rng = np.random.default_rng(4)
    bumps = [np.zeros((N_prim, N_prim, 3), np.uint16) for _ in range(3)]
    rgbs = [np.zeros((N_prim, N_prim, 3), np.uint8) for _ in range(3)]
    masks = [np.zeros((N_prim, N_prim), np.uint8) for _ in range(3)]
    claims = [np.zeros((N_prim, N_prim), np.uint64) for _ in range(3)]
    g_b, g_c, g_m, g_p = [dev(x) for x in bumps], [dev(x) for x in rgbs], [dev(x) for x in masks], [dev(p) for p in params]
    tabs = [torch.zeros(3, dtype=torch.int64, device="cuda") for _ in range(4)]
    for tab, bufs in zip(tabs, (g_b, g_c, g_m, g_p)):
        for k, b in enumerate(bufs):
            ctx.table_store(tab, b, k)                          # Graph.cpp:62-69
    assert [int(x) for x in tabs[0].cpu()] == [b.data_ptr() for b in g_b]
    g_v, g_rgbd, g_lab = dev(pair["v0"]), dev(pair["rgbd0"]), dev(label)
    for stamp, maskgen in enumerate([lambda: rng.integers(11, 15, (n, m)), lambda: rng.integers(10, 40, (n, m)),
                                     lambda: np.where(rng.random((n, m)) < 0.5, 0, rng.integers(10, 60, (n, m)))], start=1):
        mk = maskgen().astype(np.uint8)
        ctx.update_cu(tabs[0], tabs[1], tabs[2], tabs[3], g_v, g_rgbd, dev(mk), g_lab, I16, N_prim, N_prim, n, m, 3)
        orc.update(bumps, rgbs, masks, params, claims, stamp, pair["v0"], pair["rgbd0"], mk, label, I16)
        for k in range(3):
            same(host(g_m[k]), masks[k])
            same(host(g_b[k]), bumps[k])
            same(host(g_c[k]), rgbs[k])
    # nb_planes = 2: labels of plane 2 are left alone
    before = host(g_m[2]).copy()
    ctx.update_cu(tabs[0], tabs[1], tabs[2], tabs[3], g_v, g_rgbd, dev(np.full((n, m), 200, np.uint8)), g_lab, I16, N_prim, N_prim, n, m, 2)
    same(host(g_m[2]), before)
    assert (host(g_m[0]) == 200).sum() > 1000
    # Segment_cu
    holes = label.copy()
    holes[rng.random((n, m)) < 0.4] = 0
    g_h = dev(holes)
    ctx.segment_cu(g_h, g_v, dev(pair["n0"]), I16, dev(eq), 3, n, m)
    same(host(g_h), orc.segment(holes, pair["v0"], pair["n0"], I16, eq))
    # a larger image afterwards: the swap image grows
    n2, m2 = 37, 53
    calib = synth.calib_for(n2, m2)
    d, _ = synth.render(np.eye(3), np.zeros(3), n2, m2, calib)
    v2 = orc.vertex_map_kinect(d, calib)
    n2map = orc.compute_normal(v2)
    l2 = orc.seed_labels(v2, n2map, I16, eq)
    h2 = l2.copy()
    h2[::3, ::2] = 0
    g_h2 = dev(h2)
    ctx.segment_cu(g_h2, dev(v2), dev(n2map), I16, dev(eq), 3, n2, m2)
    same(host(g_h2), orc.segment(h2, v2, n2map, I16, eq))


# ---- the optional, default-off bilateral stage of OffLineFrame::LoadFrame (D1/Frame.cpp:50-54) ---------------------------
@pytest.mark.parametrize("shape,pitch_extra", [((480, 640), 0), ((37, 53), 22), ((5, 130), 2)])
def test_bilateral_depth_kernel(ctx, orc, synth, shape, pitch_extra):
    n, m = shape
    calib = synth.calib_for(n, m)
    rng = np.random.default_rng(6)
    d, _ = synth.render(np.eye(3), np.zeros(3), n, m, calib, noise_sigma=0.002, rng=rng)
    d[::9, ::7] = 0                                            # holes
    d[n // 2:, m // 2:] += 900                                 # a depth step of 18 cm: must survive the filter
    pitch = m + pitch_extra
    src = np.zeros((n, pitch), np.uint16)
    src[:, :m] = d
    dst = torch.full((n, pitch), 7, dtype=torch.int16, device="cuda")
    ctx.bilateral_depth(dst, 2 * pitch, dev(src), 2 * pitch, n, m)
    got = host(dst).view(np.uint16)
    want = orc.bilateral_depth(d)
    same(got[:, :m], want)
    assert (got[:, m:] == 7).all()
    if n >= 37:
        # edge preserving: the two sides of the step stay apart; smoothing: the noise on the planes shrinks
        valid = (d > 0)
        a, b = want[n // 2 - 1, 5:m // 2 - 5].astype(np.int64), want[n // 2 + 1, m // 2 + 5:m - 5].astype(np.int64)
        assert abs(int(np.median(b)) - int(np.median(a))) > 500
        clean, _ = synth.render(np.eye(3), np.zeros(3), n, m, calib)
        sel = valid.copy()
        sel[:2] = sel[-2:] = False; sel[:, :2] = sel[:, -2:] = False
        sel[n // 2 - 3:, m // 2 - 3:] = False                  # keep away from the synthetic step
        sel &= np.roll(valid, 1, 0) & np.roll(valid, -1, 0) & np.roll(valid, 1, 1) & np.roll(valid, -1, 1)
        e_in = np.abs(d.astype(np.int64) - clean)[sel].mean()
        e_out = np.abs(want.astype(np.int64) - clean)[sel].mean()
        if n >= 480:                                           # (at 37 x 53 the planes' own slope per pixel exceeds the noise)
            assert e_out < 0.8 * e_in, (e_in, e_out)


def test_bilateral_stage_in_the_loop_default_off(pkg, orc, synth):
    from test_gpu_slam_loop import same as same_named
    n, m = 120, 160
    calib = synth.calib_for(n, m)
    seq = synth.sequence(5, n, m, calib, noise_sigma=0.002)
    runs = {}
    for on in (False, True):
        ref = orc.Slam(n, m, calib, N_prim=256, M_prim=256)
        rg = pkg.RG(n, m, calib, N_prim=256, M_prim=256)
        if on:
            ref.set_bilateral(True)
            rg.set_bilateral(True)
        for f, (d, c) in enumerate(seq):
            st = ref.step(d, c)
            rg.load_frame(d, c)
            assert rg.step() == st
            R, t = rg.pose()
            Rr, tr = ref.pose()
            same_named(np.array(R, np.float32), Rr, "poseR %d" % f)
            same_named(np.array(t, np.float32), tr, "poset %d" % f)
        for name in ("VMap", "NMap", "RGBD", "Mask", "frame.VMap", "frame.NMap"):
            same_named(rg.buffer(name), ref.buffer(name), name)
        runs[on] = rg.buffer("frame.VMap").copy()
        rg.close()
        ref.close()
    assert not np.array_equal(runs[False], runs[True])         # the stage does something when asked for, nothing by default
