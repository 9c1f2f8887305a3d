import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("rgb-d-slam_amd"); synth = importlib.import_module("rgb-d-slam_amd.synth")
n, m = 480, 640
calib = synth.calib_for(n, m)
pool = synth.sequence(40, n, m, calib, noise_sigma=0.001)
dd = [torch.from_numpy(p[0]).cuda() for p in pool]; bb = [torch.from_numpy(p[1]).cuda() for p in pool]
rg = pkg.RG(n, m, calib)
rg.set_icp_loop(True)
rg.set_finish(True, 30, 0)
rg.run_device_sequence(dd[:36], bb[:36])
for k in range(36, 40):
    rg.run_device_sequence(dd[k:k+1], bb[k:k+1])
    L = rg.buffer("Label"); N = rg.buffer("NMap"); V = rg.buffer("VMap")
    hasn = np.abs(N).sum(-1) > 0
    print("frame", k, "labels==0: %.4f" % (L == 0).mean(), " of which with a normal: %.4f" % ((L == 0) & hasn).mean(), " valid depth: %.4f" % (V[..., 2] != 0).mean(), "planes", rg.finish_stats())
