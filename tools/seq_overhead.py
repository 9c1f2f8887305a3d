#!/usr/bin/env python3
"""What a SHORT recorded sequence pays besides its frames (DESIGN 'what a short sequence pays'): host-side spans of one
run_device_sequence call of K frames between the bench's own barriers, K = 20 and 200, and the cost of the pieces around it.
   python tools/seq_overhead.py"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rgb-d-slam_amd")
synth = importlib.import_module("rgb-d-slam_amd.synth")

n, m, POOL = 480, 640, 60
calib = synth.calib_for(n, m)
pool = synth.sequence(POOL, n, m, calib, noise_sigma=0.001, seed=0)      # bench.py's frames and set-up
depth = [torch.from_numpy(p[0]).cuda() for p in pool]
bgr = [torch.from_numpy(p[1]).cuda() for p in pool]
order, k, d = [], 0, 1
for _ in range(3000):
    order.append(k)
    if k + d < 0 or k + d >= POOL:
        d = -d
    k += d
stream = torch.cuda.Stream()
rg = pkg.RG(n, m, calib, device=0, stream=stream.cuda_stream)
rg.set_icp_loop(True)
rg.set_finish(True, 30, 0)
rg.run_device_sequence([depth[i] for i in order[:31]], [bgr[i] for i in order[:31]])
assert rg.finish_stats()["planes"] >= 3


def sync_cost():
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / 200 * 1e6


def run(k, pos):
    seq = rg.prepare_device_sequence([depth[i] for i in order[pos:pos + k]], [bgr[i] for i in order[pos:pos + k]])
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok = rg.run_device_sequence(seq)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    assert ok == k, (ok, k)
    return (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6


def run_frames(frames):
    seq = rg.prepare_device_sequence([depth[i] for i in frames], [bgr[i] for i in frames])
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok = rg.run_device_sequence(seq)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    assert ok == len(frames), (ok, len(frames))
    return (t1 - t0) * 1e6


print("torch.cuda.synchronize() on an idle device: %.2f us" % sync_cost())
# the driver's frames: after the 31 set-up frames and 5 warm-up frames, pool frames 36..55 -- as one 20-frame call, and the SAME
# frames (back and forth through 36..55) as calls of 20 and of 220, so that content does not blur the comparison
run_frames(list(range(31, 36)))
first = run_frames(list(range(36, 56)))
pp, k, d = [], 55, -1
for _ in range(2000):
    if k + d < 36 or k + d > 55:
        d = -d
    k += d
    pp.append(k)
pos = 0
run_frames(pp[pos:pos + 380]); pos += 380            # (19-frame legs: 380 frames end where they started, at frame 55 going down)
print("first 20-frame call (bench's timed frames, 5 warm-up frames before): %.1f us = %.2f us/frame" % (first, first / 20))
for rep in range(3):
    a = run_frames(pp[pos:pos + 19] + pp[pos + 19:pos + 20]); pos += 20
    b = run_frames(pp[pos:pos + 220]); pos += 220
    c = run_frames(pp[pos:pos + 20]); pos += 20
    per = (b - 0.5 * (a + c)) / 200.0
    print("20 frames: %.1f / %.1f us; 220 frames: %.1f us; steady %.2f us/frame -> fixed part of a call %.1f us"
          % (a, c, b, per, 0.5 * (a + c) - 20 * per))
rg.close()
