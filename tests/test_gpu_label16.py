"""16-bit labels through the TRACKER (rgbdslam_set_label_width(ctx, 2) / RG::SetLabel16): a model of more than 255 planes
tracked end to end -- ProjectLabel (Frame.cpp:385-405), the three SegmentKernel passes (RG.cu:252-320), Graph::Update
(Primitive.cu:267-357) -- against the oracle's display() loop with u16 label images.  The reference's label image is
unsigned char (Frame.h:197); the u16 form is the build's extension and runs the same statements on a wider element, so
(i) with at most 255 planes both widths must give the same labels, and (ii) beyond 255 the GPU must match the oracle bit
for bit.  Parity status as everywhere: against the build's own oracle (the reference holds no fixtures)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    ok = (a == b) | ((a != a) & (b != b)) if a.dtype.kind == "f" else a == b
    assert ok.all(), "%s: mismatch at %d of %d elements" % (what, (~ok).sum(), ok.size)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def many_planes(synth, nb, visible_last=3):
    """nb plane equations + centres: the last three are the walls of the room corner (so that the VISIBLE labels are
    nb-2 .. nb, beyond the u8 range for nb > 257); the others are copies of the walls pushed 0.4 m .. behind them, which
    no pixel of the scene satisfies (|n.p - d| > 0.1) -- membership seeding takes the first match in plane order."""
    eq3, c3 = synth.plane_equations()
    eq = np.zeros((nb, 10), np.float32)
    ce = np.zeros((nb, 3), np.float32)
    for k in range(nb - visible_last):
        w = k % 3
        off = np.float32(0.4 + 0.01 * (k // 3))
        eq[k] = eq3[w]
        eq[k, 9] = eq3[w][9] + off
        ce[k] = c3[w] + off * eq3[w][0:3]
    eq[nb - visible_last:] = eq3[3 - visible_last:] if visible_last < 3 else eq3
    ce[nb - visible_last:] = c3[3 - visible_last:] if visible_last < 3 else c3
    return eq, ce


@pytest.mark.parametrize("n,m", [(120, 160), (93, 131)])
def test_label_operators_u16_against_the_oracle(pkg, orc, synth, n, m):
    """project_label / segment / seed_labels at label width 2 on label values up to 300."""
    calib = synth.calib_for(n, m)
    d, c = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    v = orc.vertex_map_kinect(d, calib)
    nm = orc.compute_normal(v)
    nb = 300
    eq, _ = many_planes(synth, nb)
    pose = np.eye(4, dtype=np.float32).T.reshape(-1).copy()
    ctx = pkg.Context(0)
    ctx.set_calibration_matrix(calib)
    ctx.set_label_width(2)
    assert ctx.label_width() == 2
    V, N, EQ = dev(v), dev(nm), dev(eq)
    # seed: first plane in order whose equation fits -> labels 298..300
    lab = torch.zeros((n, m), dtype=torch.int16, device="cuda")
    ctx.seed_labels(lab, V, N, pose, EQ, nb, n, m)
    ctx.synchronize()
    want = orc.seed_labels(v, nm, pose, eq, wide=True)
    got = lab.cpu().numpy().view(np.uint16)
    same(got, want, "seed_labels u16")
    assert want.max() == nb and (want > 255).mean() > 0.5
    # growth: knock holes into the labels, three passes bring (most of) them back
    rng = np.random.default_rng(5)
    holes = want.copy()
    holes[rng.random((n, m)) < 0.3] = 0
    holes[n // 3: n // 3 + 9, m // 4: m // 4 + 40] = 0
    L, S = dev(holes.view(np.int16)), torch.zeros((n, m), dtype=torch.int16, device="cuda")
    ctx.segment(L, S, V, N, pose, EQ, nb, n, m, eps=0.1, alpha=0.9)
    ctx.synchronize()
    grown = orc.segment(holes, v, nm, pose, eq, eps=0.1, alpha=0.9)
    same(L.cpu().numpy().view(np.uint16), grown, "segment u16")
    assert (grown > 0).sum() > (holes > 0).sum()
    # gather through a small rigid motion
    R, t = synth.pair_offset_pose()
    P = np.eye(4, dtype=np.float32)
    P[:3, :3] = R
    P[:3, 3] = t
    Pcm = np.ascontiguousarray(P.T).reshape(-1)
    ctx.set_pose_matrix(Pcm)
    out = torch.zeros((n, m), dtype=torch.int16, device="cuda")
    ctx.project_label(dev(want.view(np.int16)), out, V, n, m)
    ctx.synchronize()
    same(out.cpu().numpy().view(np.uint16), orc.project_label_l16(Pcm, calib, want, v), "project_label u16")
    # width 1 again: the u8 forms are untouched
    ctx.set_label_width(1)
    lab8 = torch.zeros((n, m), dtype=torch.uint8, device="cuda")
    ctx.seed_labels(lab8, V, N, pose, dev(eq[-3:]), 3, n, m)
    ctx.synchronize()
    same(lab8.cpu().numpy(), orc.seed_labels(v, nm, pose, eq[-3:]), "seed_labels u8")
    ctx.close()


def run_loop(pkg, orc, synth, n, m, frames, N_prim, nb, planes_at, fused, icp_loop, noise=0.001, label16=True):
    calib = synth.calib_for(n, m)
    seq = synth.sequence(frames, n, m, calib, noise_sigma=noise)
    eq, centers = many_planes(synth, nb)
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim, max_planes=nb)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim, max_planes=nb)
    ref.set_finish(False)
    rg.set_finish(False)
    if label16:
        ref.set_label16(True)
        rg.set_label16(True)
    rg.set_fused(fused)
    if icp_loop:
        rg.set_icp_loop(True)
    for f, (d, c) in enumerate(seq):
        st_ref = ref.step(d, c)
        rg.load_frame(d, c)
        st = rg.step()
        assert st == st_ref, "frame %d: status %d vs oracle %d" % (f, st, st_ref)
        R, t = rg.pose()
        Rr, tr = ref.pose()
        same(np.array(R, np.float32), Rr, "frame %d poseR" % f)
        same(np.array(t, np.float32), tr, "frame %d poset" % f)
        if f == planes_at:
            ref.set_planes(eq, centers)
            rg.set_planes(eq, centers)
        if f in (planes_at, frames - 1):
            for name in ("VMap", "NMap", "RGBD", "Mask", "Label"):
                same(rg.buffer(name), ref.buffer(name), "frame %d %s" % (f, name))
    lab = ref.buffer("Label")
    out = {"labels": lab.copy(), "planes": []}
    for k in range(nb - 3, nb):
        mk = ref.buffer("Maskb%d" % k)
        assert (mk > 10).sum() > 100, "plane %d received no observations" % k
        same(rg.buffer("Maskb%d" % k), mk, "Maskb%d" % k)
        same(rg.buffer("Bump%d" % k), ref.buffer("Bump%d" % k), "Bump%d" % k)
        same(rg.buffer("RGBb%d" % k), ref.buffer("RGBb%d" % k), "RGBb%d" % k)
        out["planes"].append((mk.copy(), ref.buffer("Bump%d" % k).copy()))
    # a plane no pixel belongs to stays empty on both sides
    same(rg.buffer("Maskb0"), ref.buffer("Maskb0"), "Maskb0")
    assert (ref.buffer("Maskb0") > 10).sum() == 0
    if icp_loop:
        assert rg.icp_loop_faults() == 0
    rg.close()
    ref.close()
    return out


@pytest.mark.parametrize("fused,icp_loop", [(True, True), (True, False), (False, False)])
def test_loop_with_300_planes_and_u16_labels(pkg, orc, synth, fused, icp_loop):
    """300 injected planes, the visible ones carry labels 298..300; fused pipeline (one launch for merge +
    growth passes + claim), its multi-launch ICP form and the per-operator path, all bit-exact against the oracle."""
    out = run_loop(pkg, orc, synth, 120, 160, frames=8, N_prim=256, nb=300, planes_at=1, fused=fused, icp_loop=icp_loop)
    lab = out["labels"]
    assert lab.dtype == np.uint16 and lab.max() == 300 and (lab > 255).mean() > 0.5


def test_u16_and_u8_labels_agree_on_a_small_model(pkg, orc, synth):
    """With <= 255 planes the label width must not matter: same labels (as numbers), same Bump images."""
    a = run_loop(pkg, orc, synth, 120, 160, frames=6, N_prim=256, nb=12, planes_at=1, fused=True, icp_loop=False, label16=True)
    b = run_loop(pkg, orc, synth, 120, 160, frames=6, N_prim=256, nb=12, planes_at=1, fused=True, icp_loop=False, label16=False)
    assert a["labels"].dtype == np.uint16 and b["labels"].dtype == np.uint8
    same(a["labels"].astype(np.int32), b["labels"].astype(np.int32), "labels u16 vs u8")
    for (ma, ba), (mb, bb) in zip(a["planes"], b["planes"]):
        same(ma, mb, "plane mask u16 vs u8")
        same(ba, bb, "plane bump u16 vs u8")


def test_label_width_cannot_change_while_planes_exist(pkg, synth):
    n, m = 96, 128
    calib = synth.calib_for(n, m)
    rg = pkg.RG(n, m, calib, N_prim=128, M_prim=128)
    rg.set_finish(False)
    seq = synth.sequence(2, n, m, calib)
    for d, c in seq:
        rg.load_frame(d, c)
        rg.step()
    eq, ce = synth.plane_equations()
    rg.set_planes(eq, ce)
    with pytest.raises(pkg.SlamError):
        rg.set_label16(True)
    rg.close()


def test_config_pod_round_trip(pkg):
    """rgbdslam_configure / rgbdslam_get_config: calibration and switches of a context through one POD."""
    import ctypes as C
    lib = pkg.load()

    class IcpParams(C.Structure):
        _fields_ = [("lvl", C.c_int), ("max_iter", C.c_int * 3), ("distThres", C.c_float), ("angleThres", C.c_float),
                    ("thres_cvg_alpha", C.c_float), ("thres_cvg_tr", C.c_float)]

    class Config(C.Structure):
        _fields_ = [("calib", C.c_float * 11), ("icp", IcpParams), ("icp_loop", C.c_int), ("icp_line_check", C.c_int), ("label_width", C.c_int)]
    ctx = pkg.Context(0)
    cfg, back = Config(), Config()
    lib.rgbdslam_config_defaults.restype = None
    lib.rgbdslam_config_defaults(C.byref(cfg))
    cfg.label_width, cfg.icp_line_check = 2, 1
    assert lib.rgbdslam_configure(ctx._h, C.byref(cfg)) == 0
    assert lib.rgbdslam_get_config(ctx._h, C.byref(back)) == 0
    assert list(back.calib) == list(cfg.calib) and (back.label_width, back.icp_line_check, back.icp_loop) == (2, 1, 0)
    assert ctx.label_width() == 2
    cfg.label_width = 3
    assert lib.rgbdslam_configure(ctx._h, C.byref(cfg)) == 1      # RGBDSLAM_E_ARG
    ctx.close()
