"""SURVEY 8f N1 -- plane detection front end on the GPU (RG::DetectPrimitives, D1/RG.cpp:277-338):
per-operator parity and the fused call against the oracle (oracle/detect_oracle.c).  All integer / byte
work: bit-exact.  Canny, erode and the blob labelling stand in for OpenCV 2.4 functions (parity unpinned)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_gpu_kernels import ctx, dev, host, same  # noqa: E402,F401


def scene(orc, synth, n, m, noise=0.002, seed=0):
    calib = synth.calib_for(n, m)
    d, c = synth.sequence(1, n, m, calib, noise_sigma=noise)[0]
    v = orc.vertex_map_kinect(d, calib)
    return v, orc.compute_normal(v, True)


def gpu_canny(ctx, img, low=120.0, high=30.0):
    n, m = img.shape
    edges = torch.full((n, m), 77, dtype=torch.uint8, device="cuda")
    wl = torch.empty(n * m, dtype=torch.int32, device="cuda")
    wb = torch.empty(2 * n * m, dtype=torch.uint8, device="cuda")
    ctx.canny(dev(img), edges, wl, wb, n, m, low, high)
    return host(edges)


@pytest.mark.parametrize("shape", [(480, 640), (93, 131)])
def test_reference_kernels(ctx, orc, synth, shape):
    n, m = shape
    v, nm = scene(orc, synth, n, m)
    g_nm, g_v = dev(nm), dev(v)
    chans = []
    for c in range(3):
        g = torch.empty(n, m, dtype=torch.uint8, device="cuda")
        ctx.gpu_cpy_n(g, g_nm, n, m, c)
        same(host(g), orc.cpy_n(nm, c))
        chans.append(g)
    rng = np.random.default_rng(1)
    e = [(rng.random((n, m)) < 0.3).astype(np.uint8) * 255 for _ in range(3)]
    out = torch.empty(n, m, dtype=torch.uint8, device="cuda")
    ctx.gpu_sum(out, dev(e[0]), dev(e[1]), dev(e[2]), n, m)
    ref = orc.edge_sum(*e)
    same(host(out), ref)
    assert set(np.unique(ref)) == {0, 85, 170, 255}
    ctx.gpu_discont(out, g_v, n, m, 0.03)
    ref = orc.discont(ref, v, 0.03)
    same(host(out), ref)
    label = (rng.random((n, m)) < 0.1).astype(np.uint8) * 3
    out2 = torch.empty(n, m, dtype=torch.uint8, device="cuda")
    ctx.gpu_thresh(out2, out, dev(label), n, m, 20)
    ref2 = orc.thresh(ref, label, 20)
    same(host(out2), ref2)
    out3 = torch.empty(n, m, dtype=torch.uint8, device="cuda")
    ctx.erode_ellipse5(out3, out2, n, m)
    same(host(out3), orc.erode_ellipse5(ref2))


def test_discont_visits_only_upper_left_neighbours(ctx, orc):
    """Kernel.cu:704-717: a depth step is seen from the far side only (the loops stop before i+1 / j+1)."""
    v = np.zeros((4, 5, 3), np.float32)
    v[:, :, 2] = -1.0
    v[2:, :, 2] = -1.5                                   # step between rows 1 and 2
    out = torch.zeros(4, 5, dtype=torch.uint8, device="cuda")
    ctx.gpu_discont(out, dev(v), 4, 5, 0.03)
    got = host(out)
    same(got, orc.discont(np.zeros((4, 5), np.uint8), v, 0.03))
    assert (got[2] == 255).all() and (got[1] == 0).all() and (got[3] == 0).all()


def test_erode_known_answer(ctx, orc):
    img = np.full((9, 9), 255, np.uint8)
    img[4, 4] = 0
    out = torch.empty(9, 9, dtype=torch.uint8, device="cuda")
    ctx.erode_ellipse5(out, dev(img), 9, 9)
    hole = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], bool)
    exp = np.full((9, 9), 255, np.uint8)
    exp[2:7, 2:7][hole] = 0                              # the zero spreads over the (symmetric) 5x5 ellipse
    same(host(out), exp)
    same(host(out), orc.erode_ellipse5(img))


@pytest.mark.parametrize("kind", ["normals", "noise", "blocks", "ramp"])
def test_canny(ctx, orc, synth, kind):
    rng = np.random.default_rng(4)
    if kind == "normals":
        v, nm = scene(orc, synth, 480, 640, noise=0.004)
        imgs = [orc.cpy_n(nm, c) for c in range(3)]
    elif kind == "noise":
        imgs = [rng.integers(0, 256, (97, 150)).astype(np.uint8), rng.integers(100, 140, (64, 64)).astype(np.uint8)]
    elif kind == "blocks":
        a = np.zeros((120, 200), np.uint8)
        a[20:70, 30:90] = 200
        a[60:110, 80:190] = 90
        a += rng.integers(0, 12, a.shape).astype(np.uint8)
        imgs = [a, a.T.copy()]
    else:
        imgs = [np.tile((np.arange(256) // 2).astype(np.uint8), (40, 1)), np.tile(np.arange(0, 250, 5, dtype=np.uint8)[:, None], (1, 70))]
    total = 0
    for img in imgs:
        ref = orc.canny(img, 120.0, 30.0)
        same(gpu_canny(ctx, img), ref)
        total += int((ref > 0).sum())
        same(gpu_canny(ctx, img, 30.0, 120.0), ref)      # the thresholds are swapped when low > high
    if kind in ("noise", "blocks", "normals"):
        assert total > 100


def test_canny_hysteresis_known_answer(ctx, orc):
    """A strong vertical step fading into a weak one: the weak part survives only where it is connected to the strong part."""
    img = np.full((40, 40), 50, np.uint8)
    img[:20, 20:] = 120                                  # upper half: step of 70 -> |dx| = 280 > 120: strong
    img[20:, 20:] = 62                                   # lower half: step of 12 -> |dx| = 48: weak (> 30)
    img[30:, 20:] = 50                                   # below row 30: no step
    img[34:38, 5:9] = 62                                 # isolated weak blob: removed by the hysteresis
    got = gpu_canny(ctx, img)
    same(got, orc.canny(img))
    assert got[5, 19] == 255 or got[5, 20] == 255
    assert (got[25, 18:22] == 255).any()                 # weak edge kept: connected to the strong one above
    assert not got[33:39, 3:11].any()


@pytest.mark.parametrize("p_fg", [0.45, 0.6, 0.75, 1.0])
def test_find_blobs_random(ctx, orc, p_fg):
    rng = np.random.default_rng(int(p_fg * 100))
    for (n, m) in ((240, 320), (53, 67)):
        binary = (rng.random((n, m)) < p_fg).astype(np.uint8)
        blob = torch.empty(n, m, dtype=torch.int32, device="cuda")
        sizes = torch.empty(n, m, dtype=torch.int32, device="cuda")
        ctx.find_blobs(dev(binary), blob, sizes, n, m)
        rb, rs = orc.find_blobs(binary)
        same(host(blob), rb)
        same(host(sizes), rs)
        assert rs.sum() == binary.sum()


def test_find_blobs_serpentine(ctx, orc):
    """One long snake (worst case for label propagation), plus 4- versus 8-connectivity: diagonal touches do not join."""
    n, m = 64, 65
    binary = np.zeros((n, m), np.uint8)
    binary[::2, :] = 1
    for r in range(1, n, 2):
        binary[r, (m - 1) if (r // 2) % 2 == 0 else 0] = 1
    binary[10, 30] = 0
    binary[11, 31] = 1                                   # touches row 10 / row 12 vertically -> joins the snake
    diag = np.zeros((8, 8), np.uint8)
    diag[np.arange(8), np.arange(8)] = 1
    for b in (binary, diag):
        nn, mm = b.shape
        blob = torch.empty(nn, mm, dtype=torch.int32, device="cuda")
        sizes = torch.empty(nn, mm, dtype=torch.int32, device="cuda")
        ctx.find_blobs(dev(b), blob, sizes, nn, mm)
        rb, rs = orc.find_blobs(b)
        same(host(blob), rb)
        same(host(sizes), rs)
    assert (rs > 0).sum() == 8                           # eight diagonal pixels = eight blobs


@pytest.mark.parametrize("shape,noise", [((480, 640), 0.0), ((480, 640), 0.0005), ((120, 160), 0.001), ((93, 131), 0.001),
                                         ((1080, 1920), 0.0005)])
def test_detect_regions_fused(ctx, orc, synth, shape, noise):
    n, m = shape
    v, nm = scene(orc, synth, n, m, noise)
    blob = torch.empty(n, m, dtype=torch.int32, device="cuda")
    sizes = torch.empty(n, m, dtype=torch.int32, device="cuda")
    wl = torch.empty(3 * n * m, dtype=torch.int32, device="cuda")
    wb = torch.empty(8 * n * m, dtype=torch.uint8, device="cuda")
    ctx.detect_regions(dev(nm), dev(v), None, blob, sizes, wl, wb, n, m)
    rb, rs, binary = orc.detect_regions(nm, v)
    same((host(wb)[7 * n * m:].reshape(n, m) > 0).astype(np.uint8), binary)      # the eroded image the blobs are cut from
    same(host(blob), rb)
    same(host(sizes), rs)
    if noise == 0.0:                                     # clean normals: the three walls of the corner scene come out whole
        assert (rs > (5000 if n >= 480 else 300)).sum() == 3
    else:                                                # noisy normals shatter the walls into many blobs: a harder labelling case
        # (at 1080p the same noise on finer pixels leaves only small shards: many blobs, none large)
        assert (rs > 0).sum() > 3 and rs.max() > (1000 if n < 1080 else 30)
    # a label image blocks its pixels (ThreshKernel): regions shrink accordingly
    label = np.zeros((n, m), np.uint8)
    label[: n // 2] = 1
    ctx.detect_regions(dev(nm), dev(v), dev(label), blob, sizes, wl, wb, n, m)
    rb2, rs2, _ = orc.detect_regions(nm, v, label)
    same(host(blob), rb2)
    same(host(sizes), rs2)
    assert (rb2[: n // 2] == -1).all()


@pytest.mark.parametrize("shape,grid,min_size,min_normals", [((480, 640), (3, 4), 5000, 1000), ((240, 320), (7, 8), 300, 200), ((96, 128), (1, 2), 100, 50)])
def test_device_median_fit_equals_the_sorted_lists(ctx, orc, synth, shape, grid, min_size, min_normals):
    """rgbdslam_fit_planes: the median plane fit of RG::DetectPrimitives (RG.cpp:469-646) by radix select over the image,
    against the oracle's restatement that builds and sorts the lists: equations, centres (a serial float sum in raster order)
    and labels bit-exact.  The frame is cut into a grid of blobs of different sizes (some below min_size), some pixels of a
    blob carry zero normals (they stay in the lists as zeros), one blob has too few normals to be accepted, blobs interleave in
    raster order, and a noisy pose rotates everything."""
    n, m = shape
    v, nm = scene(orc, synth, n, m, noise=0.003, seed=3)
    rng = np.random.default_rng(n + m)
    binary = np.zeros((n, m), np.uint8)
    gr, gc = grid
    hs, ws = n // gr, m // gc
    for a in range(gr):
        for b in range(gc):
            shrink = int(rng.integers(2, min(hs, ws) // 3))        # blobs of different sizes, 4-connected and separate
            binary[a * hs + 1:(a + 1) * hs - shrink, b * ws + 1:(b + 1) * ws - shrink] = 1
    nm = nm.copy()
    nm[rng.random((n, m)) < 0.05] = 0.0                             # holes in the normal map
    nm[1:hs - 2, 1:ws - 2] = 0.0                                    # the first blob: (almost) no normals at all -> not accepted
    blob, sizes = orc.find_blobs(binary)
    R = synth.euler_xyz(0.11, -0.23, 0.07).astype(np.float32)
    pose = np.eye(4, dtype=np.float32); pose[:3, :3] = R; pose[:3, 3] = [0.3, -0.2, 0.5]
    pose16 = np.ascontiguousarray(pose.T.reshape(-1))
    nb_want, eq_want, ce_want, lab_want = orc.fit_planes(blob, sizes, v, nm, pose16, np.zeros((n, m), np.uint8), min_size, min_normals, 255)
    lab = torch.full((n, m), 99, dtype=torch.uint8, device="cuda")
    found, eq, ce = ctx.fit_planes(dev(blob), dev(sizes), dev(v), dev(nm), pose16, lab, n, m, min_size, min_normals)
    assert found == int((sizes > min_size).sum()) and found > nb_want >= 1, (found, nb_want)
    same(eq, eq_want)
    same(ce, ce_want)
    same(host(lab), lab_want)


def test_device_median_fit_reports_more_blobs_than_it_holds(ctx, orc, synth):
    """More than RGBDSLAM_FIT_MAX_BLOBS blobs above min_size: the count comes back and the caller (RG::DetectPrimitives) fits on
    the host instead."""
    n, m = 240, 320
    v, nm = scene(orc, synth, n, m)
    binary = np.zeros((n, m), np.uint8)
    for a in range(10):
        for b in range(10):
            binary[a * 24 + 1:a * 24 + 20, b * 32 + 1:b * 32 + 28] = 1
    blob, sizes = orc.find_blobs(binary)
    lab = torch.zeros(n, m, dtype=torch.uint8, device="cuda")
    found, eq, ce = ctx.fit_planes(dev(blob), dev(sizes), dev(v), dev(nm), np.eye(4, dtype=np.float32).reshape(-1), lab, n, m, 100, 50)
    assert found == 100 and eq is None
