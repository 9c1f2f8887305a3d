#!/usr/bin/env python3
"""Per-step phase stamps of the persistent ICP kernel (k_icp_loop).  Needs a library built with
-DICP_LOOP_TIMING (cd rgb-d-slam_amd/csrc && touch icp.hip && make EXTRA=-DICP_LOOP_TIMING); the normal
build carries no stamps.  Stamps per executed step, from the tail workgroup (the last workgroup of the step): start,
after its match rounds, after its line of sums is stored, when it turns to the other workgroups' lines, after all
lines have arrived and the reduce<512> tree is done, after solve + publication; slot 0 is workgroup 0's start."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd"); synth = importlib.import_module("rgb-d-slam_amd.synth")
n, m = 480, 640
calib = synth.calib_for(n, m)
seq = synth.sequence(14, n, m, calib, noise_sigma=0.001)
rg = pkg.RG(n, m, calib)
rg.set_icp_loop(True)
host = pkg.load_host(); lib = pkg.load()
host.slamhost_rg_ctx.restype = C.c_void_p
ctx = C.c_void_p(host.slamhost_rg_ctx(rg._h))
eq, centers = synth.plane_equations()
timing = torch.zeros(8 * 32, dtype=torch.int64, device="cuda")
acc = {}
for f, (d, c) in enumerate(seq):
    if f == 4:
        lib.rgbdslam_debug_icp_timing(ctx, C.c_void_p(timing.data_ptr()))
    timing.zero_()
    rg.load_frame(d, c); rg.step()
    if f == 0:
        rg.set_planes(eq, centers)
    if f >= 4:
        ts = timing.cpu().numpy().reshape(32, 8)
        for s in range(32):
            if ts[s, 6] == 0:
                continue
            acc.setdefault(s, []).append(ts[s].copy())
print("faults", rg.icp_loop_faults())
prev_end = None
for s in sorted(acc):
    a = np.array(acc[s], dtype=np.float64)
    ph = np.diff(a[:, 1:7], axis=1).mean(0) * 10.0          # 100 MHz ticks -> ns
    gap = ""
    if prev_end is not None and len(prev_end) == len(a):
        gap = "hand-over %5.0f ns (previous publication -> this tail workgroup starts)" % ((a[:, 1] - prev_end).mean() * 10.0)
    solve = ((a[:, 7] - a[:, 5]).mean() * 10.0) if (a[:, 7] > 0).all() else float("nan")   # barrier + det || solve + barrier
    print("step %2d (%2d frames): rounds %6.0f | store %5.0f | - %5.0f | lines+tree %5.0f | solve+pub %5.0f (det/solve %5.0f) | sum %6.0f ns   %s"
          % (s, len(a), *ph, solve, ph.sum(), gap))
    prev_end = a[:, 6]
rg.close()
