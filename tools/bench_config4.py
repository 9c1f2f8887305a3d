"""BASELINE configs[3] on ONE GPU: a 1920x1080 pair, row-sharded ICP.  Times what ONE rank computes per Gauss-Newton
iteration -- the normal equations of its band of block rows (rgbdslam_align_band_system) and the gathered solve
(rgbdslam_align_solve_gathered) -- for rank 0 of a 1-, 2-, 4- and 8-way split, level by level.  The all-gather of
27 floats per rank that sits between the two on a multi-GPU node is not in these numbers (one GPU here).
Usage: python tools/bench_config4.py [reps]"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")
synth = importlib.import_module("rgb-d-slam_amd.synth")
multigpu = importlib.import_module("rgb-d-slam_amd.multigpu")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    n, m = 1080, 1920
    calib = synth.calib_for(n, m)
    ctx = pkg.Context(0)
    ctx.set_calibration_matrix(calib)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    R1, t1 = synth.pair_offset_pose()
    g = {}
    for k, (R, t) in enumerate(((np.eye(3), np.zeros(3)), (R1, t1))):
        d, c = synth.render(R, t, n, m, calib)
        gd, gc = dev(d), dev(c)
        v = torch.empty(n, m, 3, device="cuda"); nm = torch.empty(n, m, 3, device="cuda"); rgb = torch.empty(n, m, 3, device="cuda")
        ctx.vertex_map_kinect(v, gd, 2 * m, n, m)
        ctx.compute_normal(nm, v, n, m, True)
        ctx.gpu_cpy_char3(gc, 3 * m, rgb, n, m)
        g["v%d" % k], g["n%d" % k], g["rgb%d" % k], g["d%d" % k] = v, nm, rgb, gd
    rgbd0 = torch.empty(n, m, 4, device="cuda")
    ctx.gpu_add_zbuff_off(g["d0"], 2 * m, g["rgb0"], rgbd0, n, m)
    img = torch.empty(n, m, device="cuda"); gx = torch.empty(n, m, device="cuda"); gy = torch.empty(n, m, device="cuda")
    ctx.intensity_sobel(g["rgb1"], img, gx, gy, n, m)
    ctx.alloc_buffers((n + 7) // 8, (m + 7) // 8)
    prm = pkg.icp_params()
    lib, h = ctx.lib, ctx._h
    p = lambda t: C.c_void_p(t.data_ptr())
    I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
    state = dev(np.concatenate([I, z, I, z]).astype(np.float32)); status = torch.zeros(4, dtype=torch.int32, device="cuda")
    out = {"workload": "1920x1080 pair, per-rank work of one Gauss-Newton iteration (band system + gathered solve), rank 0", "reps": reps,
           "us_per_iteration": {}}
    for world in (1, 2, 4, 8):
        gathered = torch.zeros(world, 27, device="cuda")
        per_level = {}
        for lvl in (2, 1, 0):
            rows = (n // (1 << lvl) + 7) // 8
            lo, hi = multigpu.band_rows(0, world, rows)

            def it():
                # iteration 0 of the level from the same state every time: begin resets the state
                assert lib.rgbdslam_align_begin(h, p(state), p(status)) == 0
                assert lib.rgbdslam_align_band_system(h, C.byref(prm), p(g["v1"]), p(g["n1"]), p(g["rgb1"]), p(gx), C.c_size_t(4 * m), p(gy),
                                                      C.c_size_t(4 * m), p(g["v0"]), p(g["n0"]), p(rgbd0), n, m, lvl, 0, lo, hi,
                                                      p(gathered[0])) == 0
                assert lib.rgbdslam_align_solve_gathered(h, C.byref(prm), p(gathered), world, p(state), lvl, 0, None) == 0
            for _ in range(3):
                it()
            ctx.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                it()
            e1.record()
            torch.cuda.synchronize()
            per_level["level %d (block rows [%d, %d) of %d)" % (lvl, lo, hi, rows)] = round(e0.elapsed_time(e1) / reps * 1e3, 1)
        out["us_per_iteration"]["world %d" % world] = per_level
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
