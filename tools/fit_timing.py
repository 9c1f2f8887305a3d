#!/usr/bin/env python3
"""Kernels of RG::DetectPlanes (detect_regions + device median fit) on a 640x480 model frame; run under
rocprofv3 --kernel-trace --stats (tools/kstat.sh style) or alone for host-side stage timing (RGBDSLAM_FINISH_TIMING=1)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")
synth = importlib.import_module("rgb-d-slam_amd.synth")
n, m = 480, 640
calib = synth.calib_for(n, m)
seq = synth.sequence(4, n, m, calib, noise_sigma=0.001)
rg = pkg.RG(n, m, calib)
rg.set_finish(False)
for d, c in seq:
    rg.load_frame(d, c)
    rg.step()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    eq, ce = rg.detect_planes()
print("planes", len(eq))
