"""The code path bench.py TIMES, pinned to the oracle at its own size.

bench.py runs `run_device_sequence` (riders: the next frame's preprocessing on the merge launch, the previous frame's
apply pass deferred onto the next draw launch) + the persistent ICP kernel `k_icp_loop` + planes DETECTED by RG::Finish
at 640 x 480 with 1024^2 slabs.  These tests replay exactly that set-up (same pool, same ping-pong order, same calls)
and compare every pose of the history and the whole model with the oracle's frame-by-frame display() loop
(D1/GP-demo.cpp:138-296, D1/RG.cpp:149-271), bit for bit."""
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


def ping_pong(count, pool):
    order, k, d = [], 0, 1
    for _ in range(count):
        order.append(k)
        if k + d < 0 or k + d >= pool:
            d = -d
        k += d
    return order


def compare_model(rg, ref, what):
    nb = ref.nb_planes()
    assert rg.finish_stats()["planes"] == nb, what
    for name in ("VMap", "NMap", "RGBD", "Mask", "Label"):
        same(rg.buffer(name), ref.buffer(name), "%s %s" % (what, name))
    for p in range(nb):
        for slab in ("Maskb", "Bump", "RGBb"):
            same(rg.buffer("%s%d" % (slab, p)), ref.buffer("%s%d" % (slab, p)), "%s %s%d" % (what, slab, p))
    return nb


def test_bench_path_640x480_matches_the_oracle(pkg, orc, synth):
    """bench.py's own set-up: 60-frame pool with 1 mm noise, set_finish(True, 30, 0), 31 set-up frames in one recorded
    sequence (seed frame + detection by RG::Finish after frame 30, nothing injected), then 45 frames through
    run_device_sequence with the persistent ICP kernel -- the timed call.  Whole pose history, model maps and all three
    1024^2 Bump / RGB / Mask slabs against orc.Slam."""
    n, m, SETUP, TIMED, POOL = 480, 640, 31, 45, 60
    calib = synth.calib_for(n, m)
    pool = synth.sequence(POOL, n, m, calib, noise_sigma=0.001, seed=0)
    order = ping_pong(SETUP + TIMED, POOL)
    depth_dev = [torch.from_numpy(p[0]).cuda() for p in pool]
    bgr_dev = [torch.from_numpy(p[1]).cuda() for p in pool]
    stream = torch.cuda.Stream()
    rg = pkg.RG(n, m, calib, stream=stream.cuda_stream)
    rg.set_icp_loop(True)
    rg.set_finish(True, 30, 0)
    run = lambda frames: rg.run_device_sequence([depth_dev[i] for i in frames], [bgr_dev[i] for i in frames])
    run(order[:SETUP])
    fs = rg.finish_stats()
    assert fs["detections"] == 1 and fs["planes"] >= 3
    tracked = run(order[SETUP:SETUP + TIMED])
    assert tracked == TIMED and rg.icp_loop_faults() == 0
    hist = rg.history(0, SETUP + TIMED)

    orc.use_openmp(True)            # the same sources with -fopenmp: bit-identical (tests/test_oracle_omp.py), 10x faster
    try:
        ref = orc.Slam(n, m, calib)
        ref.set_finish(True, 30, 0)
        for f, i in enumerate(order):
            st = ref.step(*pool[i])
            R, t = ref.pose()
            assert int(hist[f, 12]) == (1 if st else 0), "frame %d status" % f
            assert np.abs(hist[f, :9] - R).max() <= 1e-5 and np.abs(hist[f, 9:12] - t).max() <= 1e-5   # north_star's bar
            same(hist[f, :9], R, "frame %d poseR" % f)                                                  # ... and the bits
            same(hist[f, 9:12], t, "frame %d poset" % f)
        assert ref.finish_detections() == 1
        nb = compare_model(rg, ref, "after %d frames" % len(order))
        assert nb >= 3
        for p in range(3):
            assert int((ref.buffer("Maskb%d" % p) > 10).sum()) > 10000, "plane %d was fused into" % p
        # the iterations did real work: with 1 mm noise no level converges at once
        it = hist[SETUP:, 14].astype(np.int64)
        assert ((it & 255) >= 2).all() and (((it >> 8) & 255) >= 2).all()
        ref.close()
    finally:
        orc.use_openmp(False)
    rg.close()


@pytest.mark.parametrize("icp_loop", [True, False])
def test_fused_frame_kernels_1080p_match_the_oracle(pkg, orc, synth, icp_loop):
    """bench.py's `hbm_bound_kernels` workload at BASELINE configs[3]'s frame size: 1920 x 1080, 1 mm noise, seed frame, the
    three synthetic plane equations attached by set_planes, then the 8-frame order 1 2 3 2 1 0 1 2 in ONE recorded sequence
    (so k_segment3_merge_claim carries the next frame's preprocessing, k_predict_draw the deferred apply pass).  The DESIGN
    / bench figures for those kernels at this size are only worth quoting if the frames they produce are the reference's:
    every pose and the whole model against the oracle's display() loop (D1/GP-demo.cpp:138-296), bit for bit."""
    n, m = 1080, 1920
    calib = synth.calib_for(n, m)
    pool = synth.sequence(4, n, m, calib, noise_sigma=0.001)
    dd = [torch.from_numpy(p[0]).cuda() for p in pool]
    bb = [torch.from_numpy(p[1]).cuda() for p in pool]
    eq, centers = synth.plane_equations()
    order = [1, 2, 3, 2, 1, 0, 1, 2]
    rg = pkg.RG(n, m, calib)
    rg.set_icp_loop(icp_loop)
    rg.run_device_sequence(dd[:1], bb[:1])
    rg.set_planes(eq, centers)
    tracked = rg.run_device_sequence([dd[i] for i in order], [bb[i] for i in order])
    hist = rg.history(0, 1 + len(order))

    orc.use_openmp(True)
    try:
        ref = orc.Slam(n, m, calib)
        ref.step(*pool[0])
        ref.set_planes(eq, centers)
        want = 0
        for f, i in enumerate(order, start=1):
            st = ref.step(*pool[i])
            want += 1 if st else 0
            R, t = ref.pose()
            assert int(hist[f, 12]) == (1 if st else 0), "frame %d status" % f
            same(hist[f, :9], R, "frame %d poseR" % f)
            same(hist[f, 9:12], t, "frame %d poset" % f)
        assert tracked == want == len(order)
        nb = compare_model(rg, ref, "1080p after %d frames" % len(order))
        assert nb == 3
        assert sum(int((ref.buffer("Maskb%d" % p) > 10).sum()) for p in range(nb)) > 100000, "the planes were fused into"
        ref.close()
    finally:
        orc.use_openmp(False)
    if icp_loop:
        assert rg.icp_loop_faults() == 0
    rg.close()


@pytest.mark.parametrize("icp_loop", [True, False])
def test_finish_cycle_inside_a_recorded_sequence(pkg, orc, synth, tmp_path, icp_loop):
    """`--finish-period 100` inside run_device_sequence: RG::Finish's first detection (frame 29) and its periodic dump +
    plane reset + re-detection (frame 98; RG.cpp:187-270) land in the MIDDLE of one recorded sequence, where the previous
    frame's apply pass is still deferred onto the next draw launch and the next frame's preprocessing has already ridden
    on the merge launch.  144 frames at 120 x 160 in ONE call, fused, against the oracle's frame-by-frame loop; the
    dump of the cycle is readable by the DemoStep2-style loader."""
    n, m, frames, N_prim, POOL = 120, 160, 144, 256, 60
    calib = synth.calib_for(n, m)
    pool = synth.sequence(POOL, n, m, calib, noise_sigma=0.0005)
    order = ping_pong(frames, POOL)
    ref = orc.Slam(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    ref.set_finish(True, 30, 100, 300, 200)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    if icp_loop:
        rg.set_icp_loop(True)
    rg.set_finish(True, 30, 100, str(tmp_path), 300, 200)
    depth_dev = [torch.from_numpy(p[0]).cuda() for p in pool]
    bgr_dev = [torch.from_numpy(p[1]).cuda() for p in pool]
    tracked = rg.run_device_sequence([depth_dev[i] for i in order], [bgr_dev[i] for i in order])
    hist = rg.history(0, frames)
    want_tracked = 0
    filled_before_cycle = None
    for f, i in enumerate(order):
        if f == 98:
            filled_before_cycle = sum(int((ref.buffer("Maskb%d" % p) > 10).sum()) for p in range(ref.nb_planes()))
        st = ref.step(*pool[i])
        want_tracked += 1 if st else 0
        R, t = ref.pose()
        assert int(hist[f, 12]) == (1 if st else 0), "frame %d status" % f
        same(hist[f, :9], R, "frame %d poseR" % f)
        same(hist[f, 9:12], t, "frame %d poset" % f)
    assert tracked == want_tracked == frames
    assert filled_before_cycle > 3000, "the planes of the first detection were fused into before the cycle reset them"
    assert rg.finish_stats() == {"detections": 2, "dumps": 1, "planes": ref.nb_planes(), "indx": 1} and ref.finish_detections() == 2
    nb = compare_model(rg, ref, "after %d frames" % frames)
    assert nb == 3
    # 45 frames after the reset the new planes hold observations again
    assert all(int((ref.buffer("Maskb%d" % p) > 10).sum()) > 500 for p in range(nb))
    names = sorted(os.listdir(tmp_path))
    assert "Graph-eq0.txt" in names and "Pose1.txt" in names and "Dummy0.txt" in names, names
    model = pkg.load_model(str(tmp_path), 0)
    assert model is not None and len(model["planes"]) == 3
    assert sum(int((q["mask"] > 10).sum()) for q in model["planes"]) > 3000, "the dump holds what was fused before the reset"
    if icp_loop:
        assert rg.icp_loop_faults() == 0
    rg.close()
    ref.close()


def test_finish_cycle_is_skipped_while_dummy_txt_exists(pkg, orc, synth, tmp_path):
    """RG.cpp:177-187: `(_count % 100) == 0 && _finished && ok2Save` gates the WHOLE periodic block -- with a Dummy.txt in
    the data directory the reference neither dumps nor resets nor re-detects.  The planes and _indx must be untouched;
    once the file is gone the next period runs the cycle."""
    n, m, N_prim, POOL = 120, 160, 256, 60
    calib = synth.calib_for(n, m)
    pool = synth.sequence(POOL, n, m, calib)
    order = ping_pong(125, POOL)
    rg = pkg.RG(n, m, calib, N_prim=N_prim, M_prim=N_prim)
    rg.set_finish(True, 30, 40, str(tmp_path), 300, 200)          # period 40: _count % 40 == 0 at frames 38, 78, 118
    open(os.path.join(str(tmp_path), "Dummy.txt"), "w").close()
    depth_dev = [torch.from_numpy(p[0]).cuda() for p in pool]
    bgr_dev = [torch.from_numpy(p[1]).cuda() for p in pool]
    run = lambda frames: rg.run_device_sequence([depth_dev[i] for i in frames], [bgr_dev[i] for i in frames])
    run(order[:37])
    assert rg.finish_stats() == {"detections": 1, "dumps": 0, "planes": 3, "indx": 0}
    filled = [int((rg.buffer("Maskb%d" % p) > 10).sum()) for p in range(3)]
    run(order[37:60])                                             # frame 38 is due, but Dummy.txt is there
    assert rg.finish_stats() == {"detections": 1, "dumps": 0, "planes": 3, "indx": 0} and rg.finish_skipped() == 1
    after = [int((rg.buffer("Maskb%d" % p) > 10).sum()) for p in range(3)]
    assert all(a >= b > 0 for a, b in zip(after, filled)), "the fused planes survive a period that cannot be saved"
    assert not any(x.startswith("Graph-eq") for x in os.listdir(tmp_path))
    os.remove(os.path.join(str(tmp_path), "Dummy.txt"))
    run(order[60:85])                                             # frame 78: the cycle runs
    assert rg.finish_stats() == {"detections": 2, "dumps": 1, "planes": 3, "indx": 1}
    assert "Graph-eq0.txt" in os.listdir(tmp_path)
    # no directory at all: nothing can be saved, so nothing is reset either -- unless the caller opts in
    rg.set_finish(True, 30, 40, None, 300, 200)
    run(order[85:125])                                            # frame 118 is due
    assert rg.finish_stats() == {"detections": 2, "dumps": 1, "planes": 3, "indx": 1} and rg.finish_skipped() == 2
    # ... opted in: the reset runs, nothing is written
    rg.set_finish(True, 30, 40, None, 300, 200, reset_without_dump=True)
    run(ping_pong(165, POOL)[125:165])                            # frame 158 is due
    assert rg.finish_stats() == {"detections": 3, "dumps": 1, "planes": 3, "indx": 2}
    rg.close()
