"""The OpenMP build of the oracle (bench.py's multi-core CPU baseline) against the default single-thread build:
the same sources, so every buffer of a tracked and fused sequence must be bit-identical."""
import numpy as np


def test_openmp_oracle_is_bit_identical(orc, synth):
    n, m = 120, 160
    calib = synth.calib_for(n, m)
    frames = synth.sequence(4, n, m, calib, noise_sigma=0.001)
    eq, centers = synth.plane_equations()
    runs = []
    for omp in (False, True):
        orc.use_openmp(omp)
        try:
            s = orc.Slam(n, m, calib)
            s.step(*frames[0])
            s.set_planes(eq, centers)
            ok = [s.step(*f) for f in frames[1:]]
            bufs = {k: np.array(s.buffer(k)) for k in ("VMap", "NMap", "RGBD", "Mask", "Label", "Bump0", "RGBb0", "Maskb0", "Bump2", "Maskb1")}
            runs.append((ok, s.pose(), bufs))
            s.close()
        finally:
            orc.use_openmp(False)
    (ok0, pose0, b0), (ok1, pose1, b1) = runs
    assert ok0 == ok1 and all(ok0)
    assert np.array_equal(pose0[0], pose1[0]) and np.array_equal(pose0[1], pose1[1])
    for k in b0:
        assert b0[k].shape == b1[k].shape and np.array_equal(b0[k].view(np.uint8), b1[k].view(np.uint8)), k
    assert (b0["Maskb0"] > 10).sum() > 100                  # something was fused
