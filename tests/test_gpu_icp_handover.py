"""The cross-workgroup hand-over of the ICP kernels: lines published with relaxed
agent-scope stores and accepted by their per-sector sequence words.

* soak: >= 2000 frames of the noisy 640x480 sequence through the persistent loop kernel WITH the integrity check of every
  accepted line switched on -- zero rejections (no torn or stale sector was ever accepted) and a pose history that is
  bit-identical to the multi-launch path's;
* the give-up paths (every wave must reach an exit): with the waits' time limit set to 0 the loop kernel's waiters leave at
  once -- the launch completes, the fault is reported, the frame counts as lost, the mode switches itself off and the next
  frames track on the multi-launch path."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

POOL = 40


def _frames(synth, n, m, count, noise=0.001):
    calib = synth.calib_for(n, m)
    pool = synth.sequence(POOL, n, m, calib, noise_sigma=noise)
    order, k, d = [], 0, 1
    for _ in range(count):
        order.append(k)
        if k + d < 0 or k + d >= POOL:
            d = -d
        k += d
    dd = [torch.from_numpy(p[0]).cuda() for p in pool]
    bb = [torch.from_numpy(p[1]).cuda() for p in pool]
    return calib, [dd[i] for i in order], [bb[i] for i in order]


def test_soak_2000_frames_loop_kernel_with_line_check(pkg, synth):
    n, m, frames = 480, 640, 2001
    calib, dd, bb = _frames(synth, n, m, frames)
    eq, centers = synth.plane_equations()
    hist = {}
    for mode in ("loop+check", "multi-launch"):
        rg = pkg.RG(n, m, calib)
        rg.set_finish(False)                                     # planes are injected once; no re-detection cycles in the soak
        if mode == "loop+check":
            rg.set_icp_loop(True)
            rg.set_icp_line_check(True)
        assert rg.run_device_sequence(dd[:1], bb[:1]) == 1
        rg.set_planes(eq, centers)
        tracked = 0
        for lo in range(1, frames, 500):                         # the pose ring holds 8192 frames; chunks keep the host lists short
            tracked += rg.run_device_sequence(dd[lo:lo + 500], bb[lo:lo + 500])
        assert tracked == frames - 1, "%s: %d of %d frames tracked" % (mode, tracked, frames - 1)
        hist[mode] = rg.history(0, frames).copy()
        if mode == "loop+check":
            assert rg.icp_loop_faults() == 0
            assert rg.icp_line_rejections() == 0, "a line with current sequence words failed its integrity word"
        rg.close()
    a, b = hist["loop+check"], hist["multi-launch"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "frame %d differs" % int(np.argwhere((a != b).any(1))[0])
    assert (a[:, 12] == 1.0).all() and a[1:, 13].min() >= 3


def test_soak_under_uneven_load_with_line_check(pkg, synth):
    """the hand-over soak again while a SECOND stream keeps streaming large device-to-device copies (512 MiB
    each, 2 GiB in flight per round of four) beside the loop kernel -- the CUs, the L2s and the fabric are busy with somebody
    else's traffic, workgroups of the persistent kernel become resident late and unevenly.  Every accepted line is verified:
    zero rejections, no give-up, and the pose history is the multi-launch path's (measured without the load), bit for bit."""
    import threading
    n, m, frames = 480, 640, 601
    calib, dd, bb = _frames(synth, n, m, frames)
    eq, centers = synth.plane_equations()
    hist = {}
    copies = [0]
    for mode in ("loop+check under load", "multi-launch"):
        rg = pkg.RG(n, m, calib)
        rg.set_finish(False)
        stop = threading.Event()
        worker = None
        if mode != "multi-launch":
            rg.set_icp_loop(True)
            rg.set_icp_line_check(True)
            src = torch.ones(128 << 20, dtype=torch.float32, device="cuda")          # 512 MiB
            dst = [torch.empty_like(src) for _ in range(4)]
            side = torch.cuda.Stream()

            def stream_copies():
                with torch.cuda.stream(side):
                    while not stop.is_set():
                        for d_ in dst:
                            d_.copy_(src, non_blocking=True)
                        side.synchronize()
                        copies[0] += len(dst)
            worker = threading.Thread(target=stream_copies)
        assert rg.run_device_sequence(dd[:1], bb[:1]) == 1
        rg.set_planes(eq, centers)
        if worker:
            worker.start()
        try:
            tracked = 0
            for lo in range(1, frames, 100):
                tracked += rg.run_device_sequence(dd[lo:lo + 100], bb[lo:lo + 100])
        finally:
            stop.set()
            if worker:
                worker.join()
        assert tracked == frames - 1, "%s: %d of %d frames tracked" % (mode, tracked, frames - 1)
        hist[mode] = rg.history(0, frames).copy()
        if worker:
            assert copies[0] >= 8, "the load ran beside the frames (%d copies)" % copies[0]
            assert rg.icp_loop_faults() == 0
            assert rg.icp_line_rejections() == 0, "a line with current sequence words failed its integrity word"
            del src, dst
            torch.cuda.empty_cache()
        rg.close()
    a, b = hist["loop+check under load"], hist["multi-launch"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "frame %d differs" % int(np.argwhere((a != b).any(1))[0])


def test_line_check_on_the_multi_launch_and_banded_paths(pkg, synth, orc):
    """k_icp_iter (one launch per iteration) uses the same chain lines: check on == check off == oracle."""
    n, m = 240, 320
    calib, dd, bb = _frames(synth, n, m, 12)
    poses = []
    for check in (False, True):
        rg = pkg.RG(n, m, calib)
        rg.set_icp_line_check(check)
        assert rg.run_device_sequence(dd, bb) == 12
        poses.append(rg.history(0, 12).copy())
        assert rg.icp_line_rejections() == 0
        rg.close()
    assert np.array_equal(poses[0].view(np.uint32), poses[1].view(np.uint32))


def test_give_up_paths_leave_no_wave_behind(pkg, synth):
    n, m = 240, 320
    calib, dd, bb = _frames(synth, n, m, 8)
    rg = pkg.RG(n, m, calib)
    rg.set_icp_loop(True)
    assert rg.run_device_sequence(dd[:3], bb[:3]) == 3
    rg.debug_set_icp_timeout(0)                                  # every waiter (and the tail's wait for the lines) gives up at once
    tracked = rg.run_device_sequence(dd[3:4], bb[3:4])           # must return: no workgroup may be left waiting
    assert tracked == 0                                          # the frame counts as lost ...
    assert rg.icp_loop_faults() >= 1                             # ... the fault is reported, and the mode is now off
    rg.debug_set_icp_timeout(5000000)
    assert rg.run_device_sequence(dd[4:8], bb[4:8]) == 4         # multi-launch path: tracking resumes (model re-seeded by the loss)
    assert rg.icp_loop_faults() == 0
    h = rg.history(0, 8)
    assert list(h[:, 12]) == [1, 1, 1, 0, 1, 1, 1, 1]
    # the same on the multi-launch path: the tail's wait gives up -> NaN sums -> frame lost, fault raised, nothing hangs
    rg.debug_set_icp_timeout(0)
    lost = rg.run_device_sequence(dd[3:4], bb[3:4]) == 0         # (a tail whose every first look finds all lines would still track)
    assert (rg.icp_loop_faults() >= 1) == lost
    rg.close()
