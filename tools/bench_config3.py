"""BASELINE configs[2] on ONE GPU (the fuse + decode phase of `bench.py --mode planes`, without tracking and broadcast):
a model of 256 planes x 512x512 Bump cells; a step = fusion of one 640x480 frame
(Graph::Update, claim + apply) into the planes a rank owns + decode (VertexFromBump) of those planes.  Timed for the
whole model (world 1) and for rank 0's share of an 8-way plane partition -- the per-GPU step of the 8-GPU run, which
has no data-path collective (every rank receives the same frame maps).  Usage: python tools/bench_config3.py [steps]"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")
multigpu = importlib.import_module("rgb-d-slam_amd.multigpu")


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    planes, N, n, m = 256, 512, 480, 640
    synth = importlib.import_module("rgb-d-slam_amd.synth")
    rng = np.random.default_rng(3)
    ctx = pkg.Context(0)
    calib = synth.calib_for(n, m)
    ctx.set_calibration_matrix(calib)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    # the planar-tiled scene of SURVEY 8d config 3: the room corner's three planes tiled into 256 patches of 2.048 m
    # (synth.tiled_planes); the frame = the synthetic 640x480 view of that corner, labelled (u16) by plane membership
    d0, c0 = synth.render(np.eye(3), np.zeros(3), n, m, calib, noise_sigma=0.001, rng=rng)
    g_v = torch.empty(n, m, 3, device="cuda"); g_n = torch.empty(n, m, 3, device="cuda"); g_rgb = torch.empty(n, m, 3, device="cuda")
    g_rgbd = torch.empty(n, m, 4, device="cuda")
    gd = dev(d0)
    ctx.vertex_map_kinect(g_v, gd, 2 * m, n, m)
    ctx.compute_normal(g_n, g_v, n, m, True)
    ctx.gpu_cpy_char3(dev(c0), 3 * m, g_rgb, n, m)
    ctx.gpu_add_zbuff_off(gd, 2 * m, g_rgb, g_rgbd, n, m)
    params = synth.tiled_planes(planes, N)
    g_param = dev(params)
    bump = torch.zeros(planes, N, N, 3, dtype=torch.int16, device="cuda")
    rgbb = torch.zeros(planes, N, N, 3, dtype=torch.uint8, device="cuda")
    mask = torch.full((planes, N, N), 10, dtype=torch.uint8, device="cuda")
    claim = torch.zeros(planes, N, N, dtype=torch.int64, device="cuda")
    vout = torch.zeros(planes, N, N, 3, device="cuda"); cout = torch.zeros(planes, N, N, 4, device="cuda"); kout = torch.zeros(planes, N, N, 4, device="cuda")
    tab = lambda t: torch.tensor([t[k].data_ptr() for k in range(planes)], dtype=torch.int64, device="cuda")
    t_b, t_c, t_m, t_k, t_p, t_v, t_co, t_ko = (tab(x) for x in (bump, rgbb, mask, claim, g_param, vout, cout, kout))
    I16 = np.eye(4, dtype=np.float32).T.reshape(-1).copy()
    g_label = torch.empty(n, m, dtype=torch.int16, device="cuda")
    ctx.label_planes_l16(g_label, g_v, g_n, I16, None, t_p, planes, N, N, n, m)
    labels_used = int(torch.unique(g_label).numel()) - 1
    confs = [dev(rng.integers(11, 200, (n, m)).astype(np.uint8)) for _ in range(4)]
    out = {"workload": "256 planes x 512x512 Bump cells tiled over the room corner (synth.tiled_planes), one 640x480 frame fused (u16 labels, "
                       "%d patches in view) + owned planes decoded per step" % labels_used, "steps": steps, "ranks": {}}
    stamp = 0
    for world in (1, 8):
        lo, hi = multigpu.plane_range(0, world, planes)

        def step():
            nonlocal stamp
            stamp += 1
            ctx.update_l16(t_b, t_c, t_m, t_p, t_k, stamp, g_v, g_rgbd, confs[stamp % 4], g_label, I16, N, N, n, m, lo, hi)
            ctx.vertex_from_bump_batch(t_v, t_co, t_ko, t_b, t_c, t_m, t_p, lo, hi, N, N)
        for _ in range(3):
            step()
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / steps * 1e3
        out["ranks"]["world %d, rank 0: planes [%d, %d)" % (world, lo, hi)] = {
            "us_per_step": round(us, 1), "steps_per_s": round(1e6 / us, 1),
            "decode_bytes_per_step": 54 * (hi - lo) * N * N, "decode_plus_fusion_GBps": round(54 * (hi - lo) * N * N / us / 1e3, 1)}
    fused = int((mask > 10).sum().item())
    out["cells_fused"] = fused
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
