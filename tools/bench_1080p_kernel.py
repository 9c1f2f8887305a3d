"""Runs the fused frame pipeline at 1920x1080 for a few frames (for rocprofv3 / PMC passes on the large-size kernels).
Usage: python tools/bench_1080p_kernel.py [frames]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")
synth = importlib.import_module("rgb-d-slam_amd.synth")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n, m = 1080, 1920
    calib = synth.calib_for(n, m)
    pool = synth.sequence(4, n, m, calib, noise_sigma=0.001)
    dd = [torch.from_numpy(p[0]).cuda() for p in pool]
    bb = [torch.from_numpy(p[1]).cuda() for p in pool]
    rg = pkg.RG(n, m, calib)
    eq, centers = synth.plane_equations()
    rg.run_device_sequence(dd[:1], bb[:1])
    rg.set_planes(eq, centers)
    order = [1, 2, 3, 2, 1, 0, 1, 2]
    seq = [order[k % len(order)] for k in range(frames)]
    ok = rg.run_device_sequence([dd[i] for i in seq], [bb[i] for i in seq])
    print("tracked", ok, "of", frames)
    rg.close()


if __name__ == "__main__":
    main()
