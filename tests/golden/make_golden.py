#!/usr/bin/env python3
"""Generates tests/golden/slam_96x128.npz with the CPU oracle in this container.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors and cannot be built or
run here, so these vectors pin the ORACLE (and through it the HIP path) against
drift; they are not outputs of the reference binary.  The fixture stores the INPUT frames too, so
the checks do not depend on libm/numpy reproducing the synthetic renderer bit for bit.

    python tests/golden/make_golden.py
"""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

N, M, FRAMES, N_PRIM, PLANES_AT = 96, 128, 5, 256, 1


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def run_oracle(orc, calib, depth, bgr, eq, centers):
    """The fixture's outputs, recomputed from its inputs (used by the generator AND the CPU test)."""
    out = {}
    ref = orc.Slam(N, M, calib, N_prim=N_PRIM, M_prim=N_PRIM)
    status, poses = [], []
    for f in range(depth.shape[0]):
        status.append(ref.step(depth[f], bgr[f]))
        R, t = ref.pose()
        poses.append(np.concatenate([R, t]))
        if f == PLANES_AT:
            ref.set_planes(eq, centers)
    out["status"] = np.array(status, np.int32)
    out["poses"] = np.array(poses, np.float32)
    out["Label"] = ref.buffer("Label")
    out["Mask"] = ref.buffer("Mask")
    out["RGBD_w"] = ref.buffer("RGBD")[..., 3].copy()
    for name in ("VMap", "NMap", "RGBD"):
        out["sha_" + name] = digest(ref.buffer(name))
    for k in range(eq.shape[0]):
        out["Maskb%d" % k] = ref.buffer("Maskb%d" % k)
        out["sha_Bump%d" % k] = digest(ref.buffer("Bump%d" % k))
        out["sha_RGBb%d" % k] = digest(ref.buffer("RGBb%d" % k))
    ref.close()
    # ICP normal equations of frame 0 -> frame 1 at the three levels (27 sums each) + correspondences
    v = [orc.vertex_map_kinect(depth[f], calib) for f in (0, 1)]
    nm = [orc.compute_normal(x) for x in v]
    rgb = [orc.cpy_char3(bgr[f]) for f in (0, 1)]
    rgbd0 = orc.add_zbuff_off(depth[0], rgb[0], calib)
    _, gx, gy = orc.intensity_sobel(rgb[1])
    I, z = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    for fact in (4, 2, 1):
        _, _, sums = orc.estimate_system(calib[:4], I, z, v[1], nm[1], rgb[1], gx, gy, v[0], nm[0], rgbd0, 0.05,
                                         float(np.sin(np.deg2rad(20.0))), fact)
        _, corr = orc.icp_match(calib[:4], I, z, v[1], nm[1], rgb[1], gx, gy, v[0], nm[0], rgbd0, 0.05,
                                float(np.sin(np.deg2rad(20.0))), fact)
        out["sums_f%d" % fact] = sums
        out["corr_f%d" % fact] = corr
    ok, R, t, trace = orc.align_gauss_newton(orc.icp_params(), calib[:4], v[1], nm[1], rgb[1], v[0], nm[0], rgbd0, I, z, I, z)
    out["align_ok"] = np.array([int(ok)], np.int32)
    out["align_pose"] = np.concatenate([R, t])
    out["align_trace"] = trace
    return out


def main():
    from oracle import oracle as orc
    synth = importlib.import_module("rgb-d-slam_amd.synth")
    calib = synth.calib_for(N, M)
    seq = synth.sequence(FRAMES, N, M, calib, noise_sigma=0.001, seed=0)
    depth = np.stack([s[0] for s in seq])
    bgr = np.stack([s[1] for s in seq])
    eq, centers = synth.plane_equations()
    out = run_oracle(orc, calib, depth, bgr, eq, centers)
    path = os.path.join(HERE, "slam_96x128.npz")
    np.savez_compressed(path, calib=calib, depth=depth, bgr=bgr, equations=eq, centers=centers, **out)
    print("wrote", path, os.path.getsize(path), "bytes; status", out["status"], "iters", len(out["align_trace"]))


if __name__ == "__main__":
    main()
