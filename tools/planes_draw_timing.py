#!/usr/bin/env python3
"""k_planes_draw on the configs[2] model (256 planes x 512^2): how much of a launch is the mask scan of the empty tiles,
how much the drawing of the occupied ones.  Usage (GPU box): python tools/planes_draw_timing.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")
synth = importlib.import_module("rgb-d-slam_amd.synth")
multigpu = importlib.import_module("rgb-d-slam_amd.multigpu")

planes, N, n, m = 256, 512, 480, 640
calib = synth.calib_for(n, m)
lib = pkg.load()
ctx = pkg.Context(0)
ctx.set_calibration_matrix(calib)
shard = multigpu.PlaneShard(ctx, synth.tiled_planes(planes, N), N, 0, 1)
pack = multigpu.FramePack(n, m)
view = np.ascontiguousarray(np.eye(4, dtype=np.float32).reshape(-1))


def timed(kernel, run, reps=20):
    run()
    ctx.synchronize()
    lib.rgbdslam_profile_kernel(ctx._h, kernel.encode())
    for _ in range(reps):
        run()
    ctx.synchronize()
    tot, cnt = C.c_double(), C.c_int()
    lib.rgbdslam_profile_read(ctx._h, C.byref(tot), C.byref(cnt))
    lib.rgbdslam_profile_kernel(ctx._h, None)
    ev = C.c_double()
    lib.rgbdslam_profile_event_overhead(ctx._h, 100, C.byref(ev))
    return (tot.value / max(1, cnt.value) - ev.value) * 1e3


zbuf = torch.full((n, m), -1, dtype=torch.int64, device="cuda")
draw = lambda: ctx.planes_draw(zbuf, shard.t_bump, shard.t_rgb, shard.t_mask, shard.t_param, 0, planes, view, N, N, n, m, min_conf=1)
print("empty model: scan %.1f us, draw %.1f us" % (timed("k_planes_scan", draw), timed("k_planes_draw", draw)))
rg = pkg.RG(n, m, calib)
rg.set_finish(False)
pool = synth.sequence(6, n, m, calib, noise_sigma=0.001, seed=0)
for d, c in pool:
    rg.load_frame_device(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda())
    rg.step_async()
    rg.export_frame(pack.buf, pack.off_vmap, pack.off_rgbd, pack.off_mask, pack.off_hist)
    rg.synchronize()
    ctx.synchronize()
    shard.label(pack, rg.buffer_ptr("VMap")[0], rg.buffer_ptr("NMap")[0])
    shard.fuse_and_decode(pack)
    ctx.synchronize()
usable = int((shard.mask > 10).sum())
tiles = int((shard.mask.view(planes, N // 8, 8, N // 64, 64) > 10).any(dim=4).any(dim=2).sum())
per_plane = (shard.mask > 10).view(planes, -1).sum(dim=1).cpu().numpy()
print("usable cells %d, occupied 8x64 tiles ~%d, planes touched %d (indices %s)" % (usable, tiles, int((per_plane > 0).sum()),
                                                                                 np.nonzero(per_plane)[0].tolist()))
zbuf.fill_(-1)
print("fused model: scan %.1f us, draw %.1f us" % (timed("k_planes_scan", draw), timed("k_planes_draw", draw)))
zbuf.fill_(-1)
lbl = lambda: shard.label(pack, rg.buffer_ptr("VMap")[0], rg.buffer_ptr("NMap")[0])
print("k_label_planes_l16: %.1f us" % timed("k_label_planes_l16", lbl))
lab = pack.view("label").cpu().numpy().astype(np.uint16)
print("labels used:", np.unique(lab).tolist())
