#!/usr/bin/env python3
"""Generates tests/golden/detect_96x128.npz with the CPU oracle in this container: plane detection
(RG::DetectPrimitives + CreatePrimitives) on the model frame reached after the first three input
frames of tests/golden/slam_96x128.npz (whose inputs it reuses).

PARITY UNPINNED, like the other fixture: the reference holds no vectors for this stage and cannot run here;
the file pins the oracle and, through it, the HIP path against drift.

    python tests/golden/make_golden_detect.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

FRAMES_BEFORE, MIN_SIZE, MIN_NORMALS, N_PRIM = 3, 300, 200, 256


def run_oracle(orc, g):
    """Outputs recomputed from the slam fixture's inputs (used by the generator AND the CPU test)."""
    n, m = g["depth"].shape[1:]
    ref = orc.Slam(n, m, g["calib"], N_prim=N_PRIM, M_prim=N_PRIM)
    for f in range(FRAMES_BEFORE):
        ref.step(np.ascontiguousarray(g["depth"][f]), np.ascontiguousarray(g["bgr"][f]))
    blob, sizes, binary = orc.detect_regions(ref.buffer("NMap"), ref.buffer("VMap"))
    eq, ce = ref.detect_planes(MIN_SIZE, MIN_NORMALS)
    out = {"equations": eq, "centers": ce, "Label": ref.buffer("Label"), "blob": blob, "binary": binary,
           "sizes_nonzero": np.stack([np.nonzero(sizes.ravel())[0], sizes.ravel()[np.nonzero(sizes.ravel())[0]]]).astype(np.int32)}
    ref.close()
    return out


def main():
    from oracle import oracle as orc
    g = np.load(os.path.join(HERE, "slam_96x128.npz"))
    out = run_oracle(orc, g)
    assert len(out["equations"]) == 3, "the corner scene must yield its three walls"
    np.savez_compressed(os.path.join(HERE, "detect_96x128.npz"), **out)
    print("planes", len(out["equations"]), "blobs", out["sizes_nonzero"].shape[1])


if __name__ == "__main__":
    main()
