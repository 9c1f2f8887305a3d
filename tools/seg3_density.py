"""Time of rgbdslam_model_segment_merge_claim as one launch (default) and as three (RGBDSLAM_SEG3=0) against the share of
pixels that carry a label before the growth passes (the one-launch form queues the others per tile).  640 x 480, 3 planes.
usage: python tools/seg3_density.py"""
import ctypes as C, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rgb-d-slam_amd")
synth = importlib.import_module("rgb-d-slam_amd.synth")
from oracle import oracle as orc

n, m, nb, N_prim = 480, 640, 3, 512
calib = synth.TUM_CALIB
d0, c0 = synth.render(np.eye(3), np.zeros(3))
v = orc.vertex_map_kinect(d0, calib); nrm = orc.compute_normal(v); rgb = orc.cpy_char3(c0)
rgbd = orc.add_zbuff_off(d0, rgb, calib)
eq, cen = synth.plane_equations()
I16 = np.eye(4, dtype=np.float32).reshape(-1)
truth = orc.seed_labels(v, nrm, I16, eq)
rng = np.random.default_rng(0)
params = []
for k in range(nb):
    E, c = eq[k], cen[k]
    sh0 = np.float32(c @ E[3:6] - N_prim * 0.004 / 2.0); sh1 = np.float32(c @ E[6:9] - N_prim * 0.004 / 2.0)
    params.append(np.array([E[0], E[1], E[2], E[9], 500.0, 500.0, sh0, sh1, *E[3:9]], np.float32))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P = lambda t: C.c_void_p(t.data_ptr())
f = C.c_float
pb_h = np.zeros(76, np.float32); pb_h[24:28] = np.array([1, 5, 0, 1], np.int32).view(np.float32); pb_h[28:44] = I16; pb_h[44:60] = I16; pb_h[60:76] = I16
for mode in ("1", "0"):
    os.environ["RGBDSLAM_SEG3"] = mode
    ctx = pkg.Context(0); ctx.set_calibration_matrix(calib)
    pb = dev(pb_h); g_v, g_eq, g_rgbd, g_d, g_rgb = dev(v), dev(eq), dev(rgbd), dev(d0), dev(rgb)
    g_mask = dev(rng.integers(11, 40, (n, m)).astype(np.uint8))
    g_m = [torch.full((N_prim, N_prim), 10, dtype=torch.uint8, device="cuda") for _ in range(nb)]
    g_p = [dev(x) for x in params]; g_k = [torch.zeros(N_prim, N_prim, dtype=torch.int64, device="cuda") for _ in range(nb)]
    tab = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device="cuda")
    t_m, t_p, t_k = tab(g_m), tab(g_p), tab(g_k)
    swap = torch.zeros(n, m, dtype=torch.uint8, device="cuda"); nseg = torch.zeros(n, m, 3, device="cuda")
    o_rgbd, o_mask = torch.zeros(n, m, 4, device="cuda"), torch.zeros(n, m, dtype=torch.uint8, device="cuda")
    o_v, o_n = torch.zeros(n, m, 3, device="cuda"), torch.zeros(n, m, 3, device="cuda")
    for dens in (1.0, 0.99, 0.9, 0.5, 0.1, 0.0):
        lab_h = np.where(rng.random((n, m)) < dens, truth, 0).astype(np.uint8)
        lab = dev(lab_h); lab0 = lab.clone()
        def call(stamp):
            rc = ctx.lib.rgbdslam_model_segment_merge_claim(ctx._h, P(pb), P(nseg), P(g_v), P(lab), P(swap), P(g_eq), nb, f(0.1), f(0.9),
                                                            P(g_rgbd), P(g_mask), P(g_d), C.c_size_t(2 * m), P(g_rgb), P(o_rgbd), P(o_mask),
                                                            P(o_v), P(o_n), P(t_m), P(t_p), P(t_k), C.c_uint32(stamp), N_prim, N_prim, n, m, 0, nb)
            assert rc == 0
        for w in range(3):
            lab.copy_(lab0); call(1 + w)
        ctx.synchronize()
        reps, tot = 50, 0.0
        for r in range(reps):
            lab.copy_(lab0); torch.cuda.synchronize()
            t0 = time.perf_counter(); call(10 + r); ctx.synchronize(); tot += time.perf_counter() - t0
        print("SEG3=%s labelled %.2f: %.1f us per call (host-timed, incl. ~launch + sync)" % (mode, dens, 1e6 * tot / reps), flush=True)
    ctx.close()
