"""Decode and mesh of a 256-plane x 512x512 model, a few launches each (for rocprofv3 kernel-trace / PMC passes on the
config-3 kernels).  Usage: python tools/bench_decode_kernel.py [launches]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    planes, N = 256, 512
    ctx = pkg.Context(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    bump = torch.randint(0, 32767, (planes, N, N, 3), dtype=torch.int16, device="cuda", generator=g)
    rgbb = torch.randint(0, 255, (planes, N, N, 3), dtype=torch.uint8, device="cuda", generator=g)
    mask = torch.randint(11, 60, (planes, N, N), dtype=torch.uint8, device="cuda", generator=g)
    param = torch.tensor([0.0, 0.0, 1.0, -2.0, 500.0, 500.0, -1.0, -1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0], device="cuda").repeat(planes, 1)
    vout = torch.zeros(planes, N, N, 3, device="cuda")
    cout = torch.zeros(planes, N, N, 4, device="cuda")
    kout = torch.zeros(planes, N, N, 4, device="cuda")
    nout = torch.zeros(planes, N, N, 3, device="cuda")
    iout = torch.zeros(planes, (N - 1) * (N - 1), 4, dtype=torch.int32, device="cuda")
    tab = lambda t: torch.tensor([t[k].data_ptr() for k in range(planes)], dtype=torch.int64, device="cuda")
    tabs = [tab(x) for x in (vout, cout, kout, bump, rgbb, mask, param)]
    mtabs = tabs[:3] + [tab(nout), tab(iout)] + tabs[3:]
    for _ in range(reps):
        ctx.vertex_from_bump_batch(*tabs, 0, planes, N, N)
    for _ in range(reps):
        ctx.bump_mesh_batch(*mtabs, 0, planes, N, N, 0, 0.1, False, 0)
    ctx.synchronize()
    print("decode: %d B algorithmic per launch; mesh: %d B" % (54 * planes * N * N, 82 * planes * N * N))
    ctx.close()


if __name__ == "__main__":
    main()
