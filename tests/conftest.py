import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("rgb-d-slam_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("rgb-d-slam_amd.synth")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pair(synth, orc):
    """Config 1: synthetic 640x480 pair with known 6-DoF offset, preprocessed by the oracle."""
    calib = synth.TUM_CALIB
    d0, c0 = synth.render(np.eye(3), np.zeros(3))
    R1, t1 = synth.pair_offset_pose()
    d1, c1 = synth.render(R1, t1)
    out = {"calib": calib, "d0": d0, "c0": c0, "d1": d1, "c1": c1, "R1": R1, "t1": t1}
    for k, (d, c) in enumerate([(d0, c0), (d1, c1)]):
        v = orc.vertex_map_kinect(d, calib)
        out["v%d" % k] = v
        out["n%d" % k] = orc.compute_normal(v)
        out["rgb%d" % k] = orc.cpy_char3(c)
    out["rgbd0"] = orc.add_zbuff_off(d0, out["rgb0"], calib)
    out["img1"], out["gx1"], out["gy1"] = orc.intensity_sobel(out["rgb1"])
    return out
