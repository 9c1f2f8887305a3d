"""GPU parity of the whole display() loop (D1/GP-demo.cpp:138-296): the C++ host classes over the
HIP C ABI against the CPU oracle's restatement, frame by frame, on the same synthetic sequence.

Bar: poses within 1e-5 rad / 1e-5 m per frame (north_star) -- in fact required bit-identical here,
like every map, mask, label and Bump image, because both sides evaluate the same IEEE operations in
the same order."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    if a.dtype.kind == "f":
        ok = (a == b) | (np.isnan(a) & np.isnan(b))
    else:
        ok = a == b
    assert ok.all(), "%s: mismatch at %d of %d elements" % (what, (~ok).sum(), ok.size)


def run_pair(pkg, orc, synth, n, m, frames, N_prim, planes_at, device_loop=True, index_buffer=False, noise=0.0, fused=True, overlap=False,
             icp_loop=False):
    calib = synth.calib_for(n, m)
    seq = synth.sequence(frames, n, m, calib, noise_sigma=noise)
    eq, centers = synth.plane_equations()
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg.set_fused(fused)
    rg.set_overlap(overlap)
    rg.set_device_loop(device_loop)
    rg.set_index_buffer(index_buffer)
    if icp_loop:
        rg.set_icp_loop(True)
    statuses = []
    for f, (d, c) in enumerate(seq):
        st_ref = ref.step(d, c)
        rg.load_frame(d, c)
        st = rg.step()
        assert st == st_ref, "frame %d: status %d vs oracle %d" % (f, st, st_ref)
        statuses.append(st)
        R, t = rg.pose()
        Rr, tr = ref.pose()
        # north_star tolerance (1e-5) and the stronger bit-exact claim
        assert np.abs(np.array(R, np.float32) - Rr).max() <= 1e-5 and np.abs(np.array(t, np.float32) - tr).max() <= 1e-5
        same(np.array(R, np.float32), Rr, "frame %d poseR" % f)
        same(np.array(t, np.float32), tr, "frame %d poset" % f)
        if f == planes_at:
            ref.set_planes(eq, centers)
            rg.set_planes(eq, centers)
        if f in (0, planes_at, frames - 1):
            for name in ("VMap", "NMap", "RGBD", "Mask", "Label"):
                same(rg.buffer(name), ref.buffer(name), "frame %d %s" % (f, name))
    assert statuses[0] == 2 and all(s == 1 for s in statuses[1:]), statuses
    lab = ref.buffer("Label")
    assert (lab > 0).mean() > 0.5, "labels did not survive re-projection"
    for k in range(3):
        mk = ref.buffer("Maskb%d" % k)
        assert (mk > 10).sum() > 100, "plane %d received no observations" % k
        same(rg.buffer("Maskb%d" % k), mk, "Maskb%d" % k)
        same(rg.buffer("Bump%d" % k), ref.buffer("Bump%d" % k), "Bump%d" % k)
        same(rg.buffer("RGBb%d" % k), ref.buffer("RGBb%d" % k), "RGBb%d" % k)
    # tracked trajectory stays near the ground truth (camera-to-world of frame f relative to frame 0)
    Rg, tg = synth.trajectory(frames)[-1]
    R, t = rg.pose()
    assert np.abs(np.array(t) - tg).max() < 5e-3
    assert np.abs(np.array(R).reshape(3, 3).T - Rg).max() < 5e-3
    if icp_loop:
        assert rg.icp_loop_faults() == 0
    rg.close()
    ref.close()


@pytest.mark.parametrize("fused", [True, False])
def test_loop_small(pkg, orc, synth, fused):
    run_pair(pkg, orc, synth, 120, 160, frames=6, N_prim=512, planes_at=2, fused=fused)


def test_loop_multi_stream_overlap(pkg, orc, synth):
    """Opt-in stream overlap (loads and Bump updates on auxiliary streams, event-ordered): same bits."""
    run_pair(pkg, orc, synth, 120, 160, frames=8, N_prim=512, planes_at=2, overlap=True, noise=0.001)


def test_loop_odd_size_fused(pkg, orc, synth):
    """Image sides that are no multiple of the 64x4 tiles / 8x8 blocks (halo and edge handling)."""
    run_pair(pkg, orc, synth, 93, 131, frames=4, N_prim=256, planes_at=1)


def test_loop_small_host_solve_and_index_buffer(pkg, orc, synth):
    """Per-iteration host solve (RG.cpp:849-902 statement by statement) + explicit GL_QUADS index buffer."""
    run_pair(pkg, orc, synth, 96, 128, frames=4, N_prim=256, planes_at=1, device_loop=False, index_buffer=True)


@pytest.mark.parametrize("fused", [True, False])
def test_loop_640x480(pkg, orc, synth, fused):
    """Config 2 shape: TUM fr1/xyz-like 640x480 sequence, 3 planes of 1024x1024 Bump cells."""
    run_pair(pkg, orc, synth, 480, 640, frames=5, N_prim=1024, planes_at=1, fused=fused, noise=0.001 if fused else 0.0)


@pytest.mark.parametrize("shape,N_prim,noise", [((120, 160), 512, 0.001), ((93, 131), 256, 0.0), ((480, 640), 1024, 0.001)])
def test_loop_persistent_icp_kernel(pkg, orc, synth, shape, N_prim, noise):
    """rgbdslam_set_icp_loop: the whole coarse-to-fine Gauss-Newton loop in one persistent launch (epoch word
    between iterations instead of kernel boundaries) gives the oracle's bits, without a single give-up."""
    run_pair(pkg, orc, synth, shape[0], shape[1], frames=6, N_prim=N_prim, planes_at=1, noise=noise, icp_loop=True)


def test_loop_noisy(pkg, orc, synth):
    run_pair(pkg, orc, synth, 120, 160, frames=5, N_prim=512, planes_at=1, noise=0.001)


@pytest.mark.parametrize("fused,icp_loop", [(True, False), (False, False), (True, True)])
def test_loop_track_loss_reseeds_model(pkg, orc, synth, fused, icp_loop):
    """A frame from a far-away pose must report lost (RG.cpp:910-920), keep the pose and re-seed the
    model from the input frame (GP-demo.cpp:189-196).  In the fused pipeline that decision is taken
    on the device (kernels gated on the pose block)."""
    n, m = 120, 160
    calib = synth.calib_for(n, m)
    ref = orc.Slam(n, m, calib, N_prim=256, M_prim=256)
    rg = pkg.RG(n, m, calib, N_prim=256, M_prim=256)
    rg.set_fused(fused)
    if icp_loop:
        rg.set_icp_loop(True)
    far = synth.euler_xyz(*np.deg2rad([14.0, -12.0, 9.0]))
    frames = [synth.render(np.eye(3), np.zeros(3), n, m, calib), synth.render(far, np.array([0.35, -0.3, 0.25]), n, m, calib),
              synth.render(far, np.array([0.35, -0.3, 0.25]), n, m, calib)]
    out = []
    for d, c in frames:
        rg.load_frame(d, c)
        out.append((rg.step(), ref.step(d, c)))
    assert out[0] == (2, 2) and out[1] == (0, 0), out
    assert out[2][0] == out[2][1]
    R, t = rg.pose()
    Rr, tr = ref.pose()
    same(np.array(R, np.float32), Rr, "poseR")
    same(np.array(t, np.float32), tr, "poset")
    for name in ("VMap", "NMap", "RGBD", "Mask"):
        same(rg.buffer(name), ref.buffer(name), name)
    if icp_loop:
        assert rg.icp_loop_faults() == 0
    rg.close()
    ref.close()


def test_hip_path_reproduces_golden_fixture(pkg):
    """The HIP path against the committed golden vectors (no oracle library involved)."""
    import hashlib
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slam_96x128.npz"))
    n, m = g["depth"].shape[1:]
    n_prim = g["Maskb0"].shape[0]
    rg = pkg.RG(n, m, g["calib"], N_prim=n_prim, M_prim=n_prim)
    for f in range(g["depth"].shape[0]):
        rg.load_frame(np.ascontiguousarray(g["depth"][f]), np.ascontiguousarray(g["bgr"][f]))
        assert rg.step() == int(g["status"][f])
        R, t = rg.pose()
        same(np.array(R + t, np.float32), g["poses"][f], "frame %d pose" % f)
        if f == 1:
            rg.set_planes(g["equations"], g["centers"])
    same(rg.buffer("Label"), g["Label"], "Label")
    same(rg.buffer("Mask"), g["Mask"], "Mask")
    same(rg.buffer("RGBD")[..., 3], g["RGBD_w"], "RGBD.w")
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    for name in ("VMap", "NMap", "RGBD"):
        same(sha(rg.buffer(name)), g["sha_" + name], name)
    for k in range(g["equations"].shape[0]):
        same(rg.buffer("Maskb%d" % k), g["Maskb%d" % k], "Maskb%d" % k)
        same(sha(rg.buffer("Bump%d" % k)), g["sha_Bump%d" % k], "Bump%d" % k)
        same(sha(rg.buffer("RGBb%d" % k)), g["sha_RGBb%d" % k], "RGBb%d" % k)
    rg.close()


def test_hip_detection_reproduces_golden_fixture(pkg):
    """HIP plane detection against the committed golden vectors (no oracle library involved)."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, want = np.load(os.path.join(here, "slam_96x128.npz")), np.load(os.path.join(here, "detect_96x128.npz"))
    n, m = g["depth"].shape[1:]
    rg = pkg.RG(n, m, g["calib"], N_prim=256, M_prim=256)
    for f in range(3):
        rg.load_frame(np.ascontiguousarray(g["depth"][f]), np.ascontiguousarray(g["bgr"][f]))
        rg.step()
    eq, ce = rg.detect_planes(300, 200)
    same(eq, want["equations"], "equations")
    same(ce, want["centers"], "centres")
    same(rg.buffer("Label"), want["Label"], "Label")
    rg.close()


@pytest.mark.parametrize("overlap,icp_loop,shape", [(False, False, (120, 160)), (True, False, (120, 160)), (False, True, (120, 160)),
                                                     (False, True, (122, 150)), (True, False, (66, 70))])
def test_async_sequence_matches_stepwise_oracle(pkg, orc, synth, overlap, icp_loop, shape):
    """run_device_sequence enqueues every frame without a host sync (pose history on the device);
    the per-frame poses and the final model must equal the oracle's frame-by-frame loop, including
    a tracking loss in the middle of the sequence.  The ragged shapes leave partial tiles in both directions for the
    fused launches (riders, growth passes out of LDS, deferred apply pass)."""
    (n, m), N_prim = shape, 256
    calib = synth.calib_for(n, m)
    seq = synth.sequence(8, n, m, calib, noise_sigma=0.001)
    far = synth.euler_xyz(*np.deg2rad([14.0, -12.0, 9.0]))
    seq[4] = synth.render(far, np.array([0.35, -0.3, 0.25]), n, m, calib)      # lost here, re-seeded, then lost again going back
    eq, centers = synth.plane_equations()
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg.set_overlap(overlap)
    if icp_loop:
        rg.set_icp_loop(True)
    ref.step(*seq[0]); rg.load_frame(*seq[0]); rg.step()
    ref.set_planes(eq, centers); rg.set_planes(eq, centers)
    want = []
    for d, c in seq[1:]:
        st = ref.step(d, c)
        R, t = ref.pose()
        want.append((st, R.copy(), t.copy()))
    depth_t = [torch.from_numpy(d).cuda() for d, _ in seq[1:]]
    bgr_t = [torch.from_numpy(c).cuda() for _, c in seq[1:]]
    tracked = rg.run_device_sequence(depth_t, bgr_t)
    assert tracked == sum(1 for st, _, _ in want if st != 0) and 0 < tracked < len(want)
    hist = rg.history(1, len(want))
    for k, (st, R, t) in enumerate(want):
        assert int(hist[k, 12]) == (1 if st else 0), k
        same(hist[k, :9], R, "frame %d poseR" % (k + 1))
        same(hist[k, 9:12], t, "frame %d poset" % (k + 1))
    for name in ("VMap", "NMap", "RGBD", "Mask", "Label", "Maskb0", "Bump1", "RGBb2"):
        same(rg.buffer(name), ref.buffer(name), name)
    if icp_loop:
        assert rg.icp_loop_faults() == 0
    rg.close()
    ref.close()


@pytest.mark.parametrize("source", ["tum_fused", "tum_unfused", "numbered", "real_tiff"])
def test_loop_from_dataset_files_and_trajectory(pkg, orc, synth, tmp_path, source):
    """RG::LoadFrame() from the three file sources of RG.cpp:105-139 (images written to disk,
    decoded by the host library) gives the same poses as feeding the arrays, and SaveTrajectory writes the
    text the oracle's restatement predicts from those poses.  The reference's frame counter starts at 1, so
    the first listed frame is never loaded."""
    import os
    from oracle import dataset_io as ora
    n, m, frames = 96, 128, 6
    calib = synth.calib_for(n, m)
    seq = synth.sequence(frames, n, m, calib)
    d = str(tmp_path)
    stamps = ["1305031102.%06d" % (160407 + 33333 * k) for k in range(frames)]
    if source.startswith("tum"):
        os.makedirs(d + "/depth"), os.makedirs(d + "/rgb")
        with open(d + "/depth.txt", "w") as fd, open(d + "/rgb.txt", "w") as fc:
            fd.write("# depth maps\n# file\n# timestamp filename\n")
            fc.write("# color images\n# file\n# timestamp filename\n")
            for k, (dep, col) in enumerate(seq):
                pkg.write_image("%s/depth/%s.png" % (d, stamps[k]), dep)
                pkg.write_image("%s/rgb/%s.png" % (d, stamps[k]), col)
                fd.write("%s depth/%s.png\n" % (stamps[k], stamps[k]))
                fc.write("%s rgb/%s.png\n" % (stamps[k], stamps[k]))
            fc.write("1305031199.000000 rgb/none.png\n")   # unmatched tail entry: the one the iterator quirk skips
        kw = dict()
        expect = frames
    elif source == "numbered":
        os.makedirs(d + "/depth"), os.makedirs(d + "/color")
        for k, (dep, col) in enumerate(seq):
            pkg.write_image("%s/depth/%06d.png" % (d, k + 1), dep)
            pkg.write_image("%s/color/%06d.png" % (d, k + 1), col)
        kw = dict(size_data=frames, rgbd_rep=False)
        expect = frames
    else:
        for k, (dep, col) in enumerate(seq):
            pkg.write_image("%s/Depth_%d.tiff" % (d, k), dep)
            pkg.write_image("%s/RGB_%d.tiff" % (d, k), col)
        kw = dict(size_data=frames, real=True)
        expect = frames
    rg = pkg.RG(n, m, calib, N_prim=256, M_prim=256)
    rg.set_fused(source != "tum_unfused")
    assert rg.open_dataset(d, **kw) == expect
    ref = orc.Slam(n, m, calib, N_prim=256, M_prim=256)
    # the counter starts at 1: listing entry 1, file 000002.png (named count+1), Depth_1.tiff -- all frame 1 of `seq`
    first = 1
    Rs, ts, used = [], [], []
    k = first
    while rg.load_next_frame():
        st = rg.step()
        dep, col = seq[k]
        assert st == ref.step(dep, col)
        R, t = rg.pose()
        Rr, tr = ref.pose()
        same(np.array(R, np.float32), Rr, "frame %d poseR" % k)
        same(np.array(t, np.float32), tr, "frame %d poset" % k)
        Rs.append(np.array(R, np.float32).reshape(3, 3).T), ts.append(np.array(t, np.float32)), used.append(k)
        k += 1
    assert used == list(range(first, frames)) and not rg.load_next_frame()
    if source.startswith("tum"):
        out = d + "/trajectory.txt"
        rg.save_trajectory(out)
        names = ["depth/%s.png" % stamps[j] for j in used]
        text = open(out).read()
        assert text == ora.trajectory_text(Rs, ts, names)
        assert len(text.strip().split("\n")) == 1 + len(used)
    else:
        with pytest.raises(pkg.SlamError):
            rg.save_trajectory(d + "/trajectory.txt")
    rg.close()
    ref.close()


def test_save_model_dump_round_trip(pkg, orc, synth, tmp_path):
    """RG::SaveModel (saving half of RG::Finish + SaveThread, RG.cpp:16-59) writes the fused
    planes and the key frame in the reference's dump format; the DemoStep2-side loader reads back exactly the
    cleaned Bump images the oracle's restatement derives from the device buffers."""
    from oracle import model_io as ora
    n, m, frames, N_prim = 120, 160, 5, 512
    calib = synth.calib_for(n, m)
    seq = synth.sequence(frames, n, m, calib)
    eq, centers = synth.plane_equations()
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    for f, (d, c) in enumerate(seq):
        rg.load_frame(d, c)
        assert rg.step() in (1, 2)
        if f == 1:
            rg.set_planes(eq, centers)
    out = str(tmp_path)
    rg.save_model(out, 0)
    model = pkg.load_model(out, 0)
    assert model is not None and len(model["planes"]) == 3
    R, t = rg.pose()
    pose = np.zeros(16, np.float32)
    pose[[0, 1, 2, 4, 5, 6, 8, 9, 10]] = np.array(R, np.float32)
    pose[12:15] = np.array(t, np.float32)
    pose[15] = 1.0
    assert open(out + "/Pose1.txt").read() == ora.pose_text(pose)
    dimg, cimg = ora.frame_images(rg.buffer("RGBD"))
    assert np.array_equal(model["depth"], dimg) and np.array_equal(model["rgb"], cimg)
    assert (dimg > 0).mean() > 0.9
    for k in range(3):
        bump, rgbb, mask = rg.buffer("Bump%d" % k), rg.buffer("RGBb%d" % k), rg.buffer("Maskb%d" % k)
        cb, cr, cm, sh = ora.clean(bump, rgbb, mask, (0.0, 0.0))
        q = model["planes"][k]
        assert q["size"] == cm.shape and cm.shape[0] < N_prim and (cm > 10).sum() > 100
        same(q["bump"], cb, "Bump%d" % k)
        same(q["rgb"], cr, "RGBb%d" % k)
        same(q["mask"], cm.astype(np.uint16), "Maskb%d" % k)
        # the cropped plane decodes to the same surface as the live one: the shift moved with the crop
        ii, jj = np.nonzero(mask > 10)
        i0, j0 = int(ii.min()), int(jj.min())
        E = eq[k].astype(np.float32)
        prm = lambda s0, s1: np.array([E[0], E[1], E[2], E[9], q["scale"][0], q["scale"][1], s0, s1, E[3], E[4], E[5], E[6], E[7], E[8]],
                                      np.float32)
        live_shift = q["shift"] - np.float32([i0 * 0.004, j0 * 0.004])
        v_live, _, _ = orc.vertex_from_bump(bump, rgbb, mask, prm(*live_shift), 1)
        v_crop, _, _ = orc.vertex_from_bump(q["bump"], q["rgb"], q["mask"].astype(np.uint8), prm(*q["shift"]), 1)
        sel = cm > 10
        assert np.abs(v_live[i0:i0 + cm.shape[0], j0:j0 + cm.shape[1]][sel] - v_crop[sel]).max() < 2e-5
    rg.close()


@pytest.mark.parametrize("shape,min_size,fused", [((120, 160), 300, True), ((120, 160), 300, False), ((480, 640), 5000, True)])
def test_loop_with_detected_planes(pkg, orc, synth, shape, min_size, fused):
    """the planes come from RG::DetectPlanes (DetectPrimitives + CreatePrimitives, RG.cpp:154-175,
    277-793) instead of being injected: GPU image stages + host median fit against the oracle's restatement --
    equations, centres, grown labels and the Bump images fused afterwards, all bit-exact."""
    n, m = shape
    frames, N_prim = 6, 512 if n < 480 else 1024
    calib = synth.calib_for(n, m)
    seq = synth.sequence(frames, n, m, calib)
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg.set_fused(fused)
    for f, (d, c) in enumerate(seq):
        st_ref = ref.step(d, c)
        rg.load_frame(d, c)
        assert rg.step() == st_ref
        if f == 2:
            eq_r, ce_r = ref.detect_planes(min_size, 200 if n < 480 else 1000)
            eq, ce = rg.detect_planes(min_size, 200 if n < 480 else 1000)
            assert len(eq_r) == 3, "the corner scene has three walls"
            same(eq, eq_r, "equations")
            same(ce, ce_r, "centres")
            same(rg.buffer("Label"), ref.buffer("Label"), "labels after Segment_cu")
            # the fitted planes are the scene's planes (up to the sign / basis the median fit picks)
            eq0, _ = synth.plane_equations()
            for k in range(3):
                best = max(abs(float(np.dot(eq[k, :3], eq0[j, :3]))) for j in range(3))
                assert best > 0.999
                assert abs(float(np.dot(eq[k, :3], np.cross(eq[k, 3:6], eq[k, 6:9]))) - 1.0) < 1e-4   # right-handed orthonormal basis
    R, t = rg.pose()
    Rr, tr = ref.pose()
    same(np.array(R, np.float32), Rr, "poseR")
    same(np.array(t, np.float32), tr, "poset")
    for name in ("VMap", "RGBD", "Mask", "Label"):
        same(rg.buffer(name), ref.buffer(name), name)
    for k in range(3):
        mk = ref.buffer("Maskb%d" % k)
        assert (mk > 10).sum() > 100
        same(rg.buffer("Maskb%d" % k), mk, "Maskb%d" % k)
        same(rg.buffer("Bump%d" % k), ref.buffer("Bump%d" % k), "Bump%d" % k)
    rg.close()
    ref.close()


def test_host_fit_fallback_equals_device_fit(pkg, orc, synth, monkeypatch):
    """RG::DetectPrimitives fits on the device; RGBDSLAM_HOST_FIT=1 (and frames with more blobs than the device path holds)
    take the host fit on its bounded worker pool: same equations, centres and labels."""
    n, m, N_prim = 240, 320, 256
    calib = synth.calib_for(n, m)
    seq = synth.sequence(4, n, m, calib, noise_sigma=0.0005)
    out = []
    for host_fit in (False, True):
        if host_fit:
            monkeypatch.setenv("RGBDSLAM_HOST_FIT", "1")
        import subprocess, sys, json, os
        code = ("import importlib, numpy as np, json, sys; sys.path.insert(0, %r);"
                "pkg = importlib.import_module('rgb-d-slam_amd'); synth = importlib.import_module('rgb-d-slam_amd.synth');"
                "n, m = 240, 320; calib = synth.calib_for(n, m); seq = synth.sequence(4, n, m, calib, noise_sigma=0.0005);"
                "rg = pkg.RG(n, m, calib, N_prim=256, M_prim=256);"
                "[ (rg.load_frame(d, c), rg.step()) for d, c in seq ];"
                "eq, ce = rg.detect_planes(1000, 300);"
                "print(json.dumps({'eq': eq.view(np.uint32).tolist(), 'ce': ce.view(np.uint32).tolist(), 'lab': int(rg.buffer('Label').astype(np.int64).sum()),"
                " 'hist': np.bincount(rg.buffer('Label').reshape(-1), minlength=8).tolist()}))") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200, env=dict(os.environ))
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    assert out[0] == out[1] and len(out[0]["eq"]) == 3 and out[0]["lab"] > 0


def test_next_frame_hint_true_false_and_absent(pkg, orc, synth):
    """RG::HintNextDeviceFrame: the hinted frame's maps are computed in the Bump-update launch of the frame before it.
    A hint that comes true, a hint that names ANOTHER frame than the one loaded next, and no hint at all must all give
    the oracle's poses and model."""
    n, m, N_prim = 120, 160, 256
    calib = synth.calib_for(n, m)
    seq = synth.sequence(7, n, m, calib, noise_sigma=0.001)
    eq, centers = synth.plane_equations()
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    ref.step(*seq[0]); rg.load_frame(*seq[0]); rg.step()
    ref.set_planes(eq, centers); rg.set_planes(eq, centers)
    dt = [torch.from_numpy(d).cuda() for d, _ in seq]
    bt = [torch.from_numpy(c).cuda() for _, c in seq]
    hints = {1: 2, 2: 5, 3: None, 4: 5, 5: 6}         # frame -> frame hinted while it is processed (2: wrong, 3: none)
    for f in range(1, 7):
        st = ref.step(*seq[f])
        rg.load_frame_device(dt[f], bt[f])
        h = hints.get(f)
        if h is not None:
            rg.hint_next_device_frame(dt[h], bt[h])
        rg.step_async()
    rg.synchronize()
    hist = rg.history(1, 6)
    R, t = ref.pose()
    same(hist[5, :9], R, "poseR")
    same(hist[5, 9:12], t, "poset")
    for name in ("VMap", "NMap", "RGBD", "Mask", "Label", "Maskb0", "Bump1", "RGBb2"):
        same(rg.buffer(name), ref.buffer(name), name)
    rg.close()
    ref.close()


@pytest.mark.parametrize("fused", [True, False])
def test_finish_schedule_first_detection_and_periodic_cycle(pkg, orc, synth, tmp_path, fused):
    """RG::Finish (D1/RG.cpp:149-271, step 12 of display()): NO planes are injected -- Step() detects them itself once
    _count > 30, and when _count % 100 == 0 it dumps the model, resets the plane set and detects again.  132 frames at
    120 x 160 against the oracle with the same schedule: poses, model maps, labels and Bump images bit-exact right after the
    first detection, right before and after the periodic cycle, and at the end; the dump of the cycle is on disk."""
    n, m, frames, N_prim = 120, 160, 132, 256
    calib = synth.calib_for(n, m)
    seq = synth.sequence(60, n, m, calib)
    order, k, d = [], 0, 1
    for _ in range(frames):
        order.append(k)
        if k + d < 0 or k + d >= 60:
            d = -d
        k += d
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    ref.set_finish(True, 30, 100, 300, 200)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg.set_fused(fused)
    rg.set_finish(True, 30, 100, str(tmp_path), 300, 200)

    def compare(f):
        nb = ref.nb_planes()
        assert rg.finish_stats()["planes"] == nb
        for name in ("VMap", "NMap", "RGBD", "Mask", "Label"):
            same(rg.buffer(name), ref.buffer(name), "frame %d %s" % (f, name))
        for p in range(nb):
            same(rg.buffer("Maskb%d" % p), ref.buffer("Maskb%d" % p), "frame %d Maskb%d" % (f, p))
            same(rg.buffer("Bump%d" % p), ref.buffer("Bump%d" % p), "frame %d Bump%d" % (f, p))
            same(rg.buffer("RGBb%d" % p), ref.buffer("RGBb%d" % p), "frame %d RGBb%d" % (f, p))

    for f, i in enumerate(order):
        dd, cc = seq[i]
        st_ref = ref.step(dd, cc)
        rg.load_frame(dd, cc)
        assert rg.step() == st_ref, "frame %d" % f
        R, t = rg.pose()
        Rr, tr = ref.pose()
        same(np.array(R, np.float32), Rr, "frame %d poseR" % f)
        same(np.array(t, np.float32), tr, "frame %d poset" % f)
        # _count = f + 2 after frame f: first detection at f = 29 (_count 31), cycle at f = 98 (_count 100)
        if f == 28:
            assert rg.finish_stats()["detections"] == 0 and ref.nb_planes() == 0
        if f in (29, 30, 97, 98, 99, frames - 1):
            compare(f)
        if f == 29:
            assert rg.finish_stats() == {"detections": 1, "dumps": 0, "planes": 3, "indx": 0} and ref.finish_detections() == 1
        if f == 97:
            filled = sum(int((ref.buffer("Maskb%d" % p) > 10).sum()) for p in range(3))
            assert filled > 3000, "the planes detected at frame 29 were fused into"
        if f == 98:
            assert rg.finish_stats() == {"detections": 2, "dumps": 1, "planes": 3, "indx": 1} and ref.finish_detections() == 2
            assert all(int((ref.buffer("Maskb%d" % p) > 10).sum()) == 0 for p in range(3)), "the cycle starts the plane set afresh"
    # the dump of the cycle, in the reference's format (Graph::save / PredictedFrame::save)
    names = sorted(os.listdir(tmp_path))
    assert "Graph-eq0.txt" in names and "Pose1.txt" in names and any(x.startswith("Bump0-") for x in names), names
    model = pkg.load_model(str(tmp_path), 0)
    assert model is not None and len(model["planes"]) == 3
    rg.close()
    ref.close()


def test_reorganize_equa_drops_unseen_planes_and_renumbers(pkg, orc, synth):
    """RG::ReorganizeEqua (D1/RG.cpp:934-1006) as a host method over VotePlan_cu / AffectPlan_cu: a plane nobody is labelled
    with (injected FIRST, so that every other plane has to be renumbered) is dropped; labels, equations, centres and the Bump
    images follow the renumbering -- after which the tracker is in exactly the state of a run that never had the extra plane."""
    n, m, N_prim, frames = 120, 160, 256, 6
    calib = synth.calib_for(n, m)
    seq = synth.sequence(frames, n, m, calib, noise_sigma=0.0005)
    eq, centers = synth.plane_equations()
    bogus_eq = np.array([[0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 57.0]], np.float32)      # a plane 57 m away: no pixel lies on it
    eq4 = np.concatenate([bogus_eq, eq]).astype(np.float32)
    ce4 = np.concatenate([np.array([[0.0, 0.0, 57.0]], np.float32), centers]).astype(np.float32)
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg.set_finish(False, 30, 0, None, 300, 200)                  # MIN_SIZE_PLAN 300 at this image size
    ref.set_finish(False)
    for f, (d, c) in enumerate(seq):
        assert ref.step(d, c) == (rg.load_frame(d, c) or rg.step())
        if f == 1:
            ref.set_planes(eq, centers)
            rg.set_planes(eq4, ce4)
        if f == 3:
            lab_before = rg.buffer("Label")
            assert set(np.unique(lab_before).tolist()) <= {0, 2, 3, 4} and (lab_before > 0).mean() > 0.5
            v, pose = rg.buffer("VMap"), np.zeros(16, np.float32)
            R, t = rg.pose()
            pose[[0, 1, 2, 4, 5, 6, 8, 9, 10]] = np.array(R, np.float32); pose[12:15] = np.array(t, np.float32); pose[15] = 1.0
            want_c, want_n = orc.vote_plan(lab_before, v, pose, 4)
            eq_k, ce_k = rg.reorganize_equa()
            assert eq_k.shape == (3, 10) and rg.finish_stats()["planes"] == 3
            same(eq_k, eq, "equations compacted in order")
            same(ce_k, (want_c.reshape(4, 3)[1:] / want_n[1:, None]).astype(np.float32), "centres = mean labelled point")
            lab_after = rg.buffer("Label")
            same(lab_after, np.where(lab_before > 0, lab_before - 1, 0).astype(np.uint8), "labels renumbered")
    # two more frames were tracked and fused after the renumbering: same state as the run that only ever had three planes
    for name in ("VMap", "NMap", "RGBD", "Mask", "Label"):
        same(rg.buffer(name), ref.buffer(name), name)
    for k in range(3):
        assert int((ref.buffer("Maskb%d" % k) > 10).sum()) > 100
        for slab in ("Maskb", "Bump", "RGBb"):
            same(rg.buffer("%s%d" % (slab, k)), ref.buffer("%s%d" % (slab, k)), "%s%d" % (slab, k))
    rg.close()
    ref.close()


def test_stage_timers_cover_the_frame(pkg, synth):
    """RG::SetStageTiming (the reference's TIMING switch, Utilities.h:82 / GP-demo.cpp:153-277): HIP events at the stage boundaries;
    the four spans are positive, the ICP is the longest, and switching the timers on changes no pose."""
    n, m = 240, 320
    calib = synth.calib_for(n, m)
    seq = synth.sequence(12, n, m, calib, noise_sigma=0.001)
    dd = [torch.from_numpy(d).cuda() for d, _ in seq]
    bb = [torch.from_numpy(c).cuda() for _, c in seq]
    hist = []
    for timed in (False, True):
        rg = pkg.RG(n, m, calib, N_prim=256, M_prim=256)
        rg.set_finish(False)
        if timed:
            rg.set_stage_timing(True)
        assert rg.run_device_sequence(dd[:1], bb[:1]) == 1
        rg.set_planes(*synth.plane_equations())
        for k in range(1, 12):                                   # frame by frame: stage 1 is a launch of its own
            rg.load_frame_device(dd[k], bb[k])
            assert rg.step() == 1
        if timed:
            st, frames = rg.stage_times()
            assert frames == 12
            assert all(st[k] > 1.0 for k in ("stage1_preprocess", "stage2_3_icp", "stage5_prediction", "stage4_fusion")), st
            assert st["stage2_3_icp"] > st["stage5_prediction"] and st["stage2_3_icp"] > st["stage1_preprocess"]
            assert rg.stage_times()[1] == 0                      # read and reset
        hist.append(rg.history(0, 12).copy())
        rg.close()
    assert np.array_equal(hist[0].view(np.uint32), hist[1].view(np.uint32))
