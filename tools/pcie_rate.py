"""PCIe-inclusive frame rate: host-resident frames go through RG::LoadFrame (2 x hipMemcpyAsync H2D, 1.5 MB per
640x480 frame) every step, against the HBM-resident rate bench.py reports.  Never the bench `value`.
Usage: python tools/pcie_rate.py [frames]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")
synth = importlib.import_module("rgb-d-slam_amd.synth")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    n, m = 480, 640
    calib = synth.TUM_CALIB
    pool = synth.sequence(60, n, m, calib, noise_sigma=0.001)
    order = list(range(60)) + list(range(58, 0, -1))
    rg = pkg.RG(n, m, calib)
    eq, centers = synth.plane_equations()
    rg.load_frame(*pool[0]); rg.step()
    rg.set_planes(eq, centers)
    host = pkg.load_host()
    import ctypes as C
    def run(k):
        for f in range(k):
            d, c = pool[order[(f + 1) % len(order)]]
            host.slamhost_rg_load_frame(rg._h, C.c_void_p(d.ctypes.data), C.c_void_p(c.ctypes.data))
            host.slamhost_rg_step_async(rg._h)
        rg.synchronize()
    run(30)
    t0 = time.perf_counter()
    run(frames)
    dt = time.perf_counter() - t0
    print({"pcie_inclusive_frames_per_s": round(frames / dt, 1), "frames": frames, "h2d_bytes_per_frame": n * m * 5,
           "note": "pageable numpy buffers, staged by hipMemcpyAsync; poses stay on the device"})
    rg.close()


if __name__ == "__main__":
    main()
