"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/rgbdslam.h declares; argument validation that needs no device; the host-side
algebra of librgbdslam_host.so (Eigen stand-ins of D1/RG.cpp:834-926) against the oracle's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rgbdslam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbdslam_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_operator_set():
    names = declared_symbols()
    # one entry point per free host function of Kernel.cuh / ICP.cuh / Primitive.cuh / RG.cuh on the hot path
    for want in ["set_calibration_matrix", "set_pose_matrix", "vertex_map_kinect", "gpu_cpy_char3", "compute_normal",
                 "intensity_sobel", "alloc_buffers", "free_buffers", "estimate_system_gauss_newton", "align_gauss_newton",
                 "transform_vertex_map", "quad_trim_frame", "predict_raster", "read_frame", "project_mask", "segment",
                 "merge_depth_off", "vertex_map_gl", "gpu_add_zbuff_off", "init_mask", "update", "vertex_from_bump"]:
        assert "rgbdslam_" + want in names, want


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert b"gfx950" in lib.rgbdslam_version()


def test_null_arguments_are_rejected_without_a_device(pkg):
    """Argument errors are reported as RGBDSLAM_E_ARG (1) before anything touches the GPU; nothing
    aborts (the reference's checkCudaErrors exits the process, cudaTypes.cuh:36)."""
    lib = pkg.load()
    null = C.c_void_p(0)
    assert lib.rgbdslam_destroy(null) == 1
    assert lib.rgbdslam_set_calibration_matrix(null, null) == 1
    assert lib.rgbdslam_compute_normal(null, null, null, 480, 640, 1) == 1
    assert lib.rgbdslam_update(null, null, null, null, null, null, C.c_uint32(1), null, null, null, null, null, 4, 4, 4, 4, 0, 1) == 1
    assert lib.rgbdslam_create(null, 0, null) == 1


def test_config_pod_defaults_are_the_reference_values(pkg):
    """rgbdslam_config (one POD for the reference's #defines + RG constructor arguments): the defaults are the TUM fr1/xyz
    values of D1/GP-demo.cpp:404-409 and D1/RG.h:111,124-126; the context calls reject NULL without a device."""
    lib = pkg.load()

    class IcpParams(C.Structure):
        _fields_ = [("lvl", C.c_int), ("max_iter", C.c_int * 3), ("distThres", C.c_float), ("angleThres", C.c_float),
                    ("thres_cvg_alpha", C.c_float), ("thres_cvg_tr", C.c_float)]

    class Config(C.Structure):
        _fields_ = [("calib", C.c_float * 11), ("icp", IcpParams), ("icp_loop", C.c_int), ("icp_line_check", C.c_int), ("label_width", C.c_int)]
    cfg = Config()
    lib.rgbdslam_config_defaults.restype = None
    lib.rgbdslam_config_defaults(C.byref(cfg))
    assert [round(x, 4) for x in cfg.calib] == [525.0, 525.0, 319.5, 239.5, 0.2624, -0.9531, -0.0054, 0.0026, 1.1633, 1.035, 5000.0]
    assert cfg.icp.lvl == 3 and list(cfg.icp.max_iter) == [3, 2, 1]
    assert abs(cfg.icp.distThres - 0.05) < 1e-7 and abs(cfg.icp.angleThres - np.sin(np.deg2rad(20.0))) < 1e-7
    assert cfg.icp.thres_cvg_alpha == np.float32(1e-6) and cfg.icp.thres_cvg_tr == np.float32(1e-6)
    assert (cfg.icp_loop, cfg.icp_line_check, cfg.label_width) == (0, 0, 1)
    null = C.c_void_p(0)
    assert lib.rgbdslam_configure(null, C.byref(cfg)) == 1 and lib.rgbdslam_get_config(null, C.byref(cfg)) == 1
    assert lib.rgbdslam_set_label_width(null, 2) == 1


def test_host_library_exports(pkg):
    host = pkg.load_host()
    for s in ["slamhost_rg_create", "slamhost_rg_destroy", "slamhost_rg_load_frame", "slamhost_rg_load_frame_device",
              "slamhost_rg_set_planes", "slamhost_rg_step", "slamhost_rg_run_device_sequence", "slamhost_rg_get_pose",
              "slamhost_rg_buffer", "slamhost_last_error"]:
        assert hasattr(host, s), s


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def test_host_algebra_matches_oracle(pkg, orc, synth):
    host = pkg.load_host()
    host.slamhost_test_det6.restype = C.c_float
    rng = np.random.default_rng(11)
    for _ in range(20):
        R, Rp = _f(synth.euler_xyz(*rng.normal(0, 0.2, 3)).T.reshape(-1))
        out = np.empty(9, np.float32)
        host.slamhost_test_inverse3(Rp, out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, orc.inverse3(R))
        Mx = rng.normal(size=(6, 6))
        A, Ap = _f(Mx @ Mx.T + np.eye(6))
        b, bp = _f(rng.normal(size=6))
        x = np.empty(6, np.float32)
        assert host.slamhost_test_llt_solve6(Ap, bp, x.ctypes.data_as(C.c_void_p)) == 0
        rc, xr = orc.llt_solve6(A, b)
        assert rc == 0 and np.array_equal(x, xr)
        assert np.float32(host.slamhost_test_det6(Ap)) == np.float32(orc.det6(A))
    # quaternion2matrix (D1/Utilities.cpp:1145-1167): identity and a 90 degree turn about z
    m = np.empty(9, np.float64)
    q = np.array([1.0, 0, 0, 0])
    host.slamhost_test_quaternion2matrix(q.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
    assert np.array_equal(m.reshape(3, 3), np.eye(3))
    q = np.array([np.sqrt(0.5), 0, 0, np.sqrt(0.5)])
    host.slamhost_test_quaternion2matrix(q.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
    assert np.allclose(m.reshape(3, 3), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)
    # pose packing: column-major 4x4 (GP-demo.cpp:206-211)
    R, Rp = _f(np.arange(9))
    t, tp = _f([10, 11, 12])
    P = np.empty(16, np.float32)
    host.slamhost_test_pack_pose(Rp, tp, P.ctypes.data_as(C.c_void_p))
    assert list(P) == [0, 1, 2, 0, 3, 4, 5, 0, 6, 7, 8, 0, 10, 11, 12, 1]


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    import subprocess
    for lib in ("librgbdslam_hip.so", "librgbdslam_host.so"):
        out = subprocess.run(["ldd", os.path.join(ROOT, "rgb-d-slam_amd", lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rgb-d-slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "rgbdslam_oracle" not in src, f
