"""The runnable N-rank forms of BASELINE configs[2] / configs[3] (bench.py --mode planes | rows) on ONE card.

* world size 1 through the real code paths (RCCL not needed for one rank): peer-slot exchange == all-gather path, bit for bit;
* 2 ranks rehearsed on one GPU (gloo for the process group, RGBDSLAM_BENCH_DEVICE=0): the peer exchange maps each other's
  block over HIP IPC and both ranks must hold identical poses (bench.py checks that itself and refuses otherwise); the
  plane-partitioned model after N steps must carry the same checksum as the single-rank run.
Never more than 3 ranks on the card."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(args, world=1, timeout=170, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    if world > 1:
        env.update(RGBDSLAM_BENCH_BACKEND="gloo", RGBDSLAM_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_peer_exchange_equals_allgather_single_rank(pkg, synth, orc):
    multigpu = importlib.import_module("rgb-d-slam_amd.multigpu")
    n, m = 240, 320
    calib = synth.calib_for(n, m)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(2)
    d0, c0 = synth.render(np.eye(3), np.zeros(3), n, m, calib, noise_sigma=0.001, rng=rng)
    R1, t1 = synth.pair_offset_pose()
    d1, c1 = synth.render(R1, t1, n, m, calib, noise_sigma=0.001, rng=rng)
    v0 = orc.vertex_map_kinect(d0, calib); n0 = orc.compute_normal(v0); rgb0 = orc.cpy_char3(c0)
    v1 = orc.vertex_map_kinect(d1, calib); n1 = orc.compute_normal(v1); rgb1 = orc.cpy_char3(c1)
    rgbd0 = orc.add_zbuff_off(d0, rgb0, calib)
    _, gx, gy = orc.intensity_sobel(rgb1)
    I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
    poses = {}
    for mode in ("allgather", "peer", "peer_loop"):
        ctx = pkg.Context(0)
        ctx.set_calibration_matrix(calib)
        ctx.alloc_buffers((n + 7) // 8, (m + 7) // 8)
        al = multigpu.ShardedAligner(ctx, pkg.icp_params(), n, m, 0, 1, exchange=mode)
        if mode == "peer_loop":                         # ONE persistent launch per alignment, the exchange inside it
            state, status = al.pose_block_views()
        else:
            state = dev(np.concatenate([I, z, I, z]).astype(np.float32))
            status = torch.zeros(4, dtype=torch.int32, device="cuda")
        for _ in range(3):                              # repeatedly: tags and slot sets move on, stale lines must not pass
            state.copy_(dev(np.concatenate([I, z, I, z]).astype(np.float32)))
            al.align(dev(v1), dev(n1), dev(rgb1), dev(gx), dev(gy), dev(v0), dev(n0), dev(rgbd0), state, status)
        ctx.synchronize()
        poses[mode] = (state.cpu().numpy().copy(), status.cpu().numpy().copy())
        ctx.close()
    assert int(poses["peer"][1][0]) == 1 and int(poses["peer"][1][1]) >= 6
    assert np.array_equal(poses["peer"][0], poses["allgather"][0]) and np.array_equal(poses["peer"][1], poses["allgather"][1])
    # the persistent form: same pose, same iteration counts (its pose block also carries display()'s "previous := current")
    assert np.array_equal(poses["peer_loop"][0][:12], poses["peer"][0][:12]) and np.array_equal(poses["peer_loop"][1][:3], poses["peer"][1][:3])
    ok, R, t, _ = orc.align_gauss_newton(orc.icp_params(), calib[:4], v1, n1, rgb1, v0, n0, rgbd0, I, z, I, z)
    assert ok and np.array_equal(poses["peer"][0][:9], R) and np.array_equal(poses["peer"][0][9:12], t)   # one band = the single-GPU sums


def test_sharded_persistent_loop_gives_up_cleanly(pkg, synth, orc):
    """rgbdslam_track_frame_sharded with every device-side wait told to give up at once (rgbdslam_debug_set_icp_timeout(0)): the
    launch completes, the frame counts as lost and the fault is reported; with the timeout restored the next alignment tracks."""
    multigpu = importlib.import_module("rgb-d-slam_amd.multigpu")
    n, m = 120, 160
    calib = synth.calib_for(n, m)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d0, c0 = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    R1, t1 = synth.pair_offset_pose()
    d1, c1 = synth.render(R1, t1, n, m, calib)
    v0 = orc.vertex_map_kinect(d0, calib); n0 = orc.compute_normal(v0); rgb0 = orc.cpy_char3(c0)
    v1 = orc.vertex_map_kinect(d1, calib); n1 = orc.compute_normal(v1); rgb1 = orc.cpy_char3(c1)
    rgbd0 = orc.add_zbuff_off(d0, rgb0, calib)
    _, gx, gy = orc.intensity_sobel(rgb1)
    g = [dev(x) for x in (v1, n1, rgb1, gx, gy, v0, n0, rgbd0)]
    I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
    start = dev(np.concatenate([I, z, I, z]).astype(np.float32))
    ctx = pkg.Context(0)
    try:
        ctx.set_calibration_matrix(calib)
        ctx.alloc_buffers((n + 7) // 8, (m + 7) // 8)
        al = multigpu.ShardedAligner(ctx, pkg.icp_params(), n, m, 0, 1, exchange="peer_loop")
        state, status = al.pose_block_views()
        ctx.debug_set_icp_timeout(0)
        state.copy_(start)
        al.align(*g, state, status)
        ctx.synchronize()
        assert int(status[0]) == 0 and ctx.icp_loop_faults() > 0
        ctx.debug_set_icp_timeout(5000000)
        ctx.set_icp_loop(True)                          # (a fault switches the mode off)
        state.copy_(start)
        al.align(*g, state, status)
        ctx.synchronize()
        assert int(status[0]) == 1 and ctx.icp_loop_faults() == 0
        ok, R, t, _ = orc.align_gauss_newton(orc.icp_params(), calib[:4], v1, n1, rgb1, v0, n0, rgbd0, I, z, I, z)
        s = state.cpu().numpy()
        assert ok and np.array_equal(s[:9], R) and np.array_equal(s[9:12], t)
    finally:
        ctx.close()


def test_rows_mode_two_ranks_on_one_card_peer_and_allgather():
    size = ["--mode", "rows", "--rows", "240", "--cols", "320", "--steps", "4", "--warmup", "1"]
    one = _bench(size + ["--exchange", "allgather"])
    outs = {ex: _bench(size + ["--exchange", ex], world=2) for ex in ("allgather", "peer")}
    assert one["n_gpus"] == 1 and all(o["n_gpus"] == 2 and o["scaling"] == "strong" for o in outs.values())
    # the two exchange forms add the same numbers in the same order: identical bits
    assert outs["peer"]["config"]["pose"] == outs["allgather"]["config"]["pose"]
    assert outs["peer"]["config"]["icp_iterations_per_level"] == outs["allgather"]["config"]["icp_iterations_per_level"]
    # against the unsharded run only the float32 summation order differs: BASELINE's 1e-5 rad / 1e-5 m bar
    assert np.allclose(outs["peer"]["config"]["pose"], one["config"]["pose"], atol=1e-5, rtol=0)
    assert "peer" in outs["peer"]["config"]["workload"]


def test_rows_mode_three_ranks_one_of_them_delayed():
    """The peer exchange with a rank that falls behind: rank 2 sleeps 300 us between publishing its line and its
    first poll, in every iteration.  Ranks 0 and 1 meanwhile gather iteration s and publish s + 1.  With ONE slot per rank that
    store overwrote the line of s the slow rank had not read yet (it then waited 50 ms in vain, lost the frame and the ranks'
    poses diverged); the slots are double-buffered by the parity of the sequence, so the slow rank still finds s.
    Three ranks on one card, poses identical on all ranks (bench.py refuses otherwise) and identical to the all-gather form."""
    size = ["--mode", "rows", "--rows", "240", "--cols", "320", "--steps", "3", "--warmup", "1"]
    ag = _bench(size + ["--exchange", "allgather"], world=3)
    delayed = _bench(size + ["--exchange", "peer"], world=3, extra_env={"RGBDSLAM_BENCH_XCHG_DELAY": "2:30000"})
    assert delayed["n_gpus"] == 3 and delayed["config"]["pose"] == ag["config"]["pose"]
    assert delayed["config"]["icp_iterations_per_level"] == ag["config"]["icp_iterations_per_level"]
    iters = sum(delayed["config"]["icp_iterations_per_level"])
    assert iters >= 6 and delayed["ms_per_step"] > iters * 0.3 * 0.9, "the delay was applied (300 us in every executed iteration)"


def test_planes_mode_two_ranks_on_one_card_same_model_as_one_rank():
    args = ["--mode", "planes", "--steps", "6", "--warmup", "2"]
    one = _bench(args)                                           # thin pack, tracking pipelined with the fusion (the defaults)
    two = _bench(args, world=2)
    # rounds 1-2's form -- VMap / RGBD broadcast as they are, one stream, the phases one after the other -- leaves the same
    # model and predicts the same view: what the thin pack drops is rebuilt bit for bit, and the pipeline reorders nothing
    fat = _bench(args + ["--pack", "fat", "--pipeline", "0"])
    assert fat["config"]["broadcast_bytes"] > 3 * one["config"]["broadcast_bytes"] and one["config"]["pipelined"] and not fat["config"]["pipelined"]
    assert (fat["model_checksum"], fat["cells_fused"], fat["prediction_checksum"]) == (one["model_checksum"], one["cells_fused"], one["prediction_checksum"])
    assert one["config"]["planes_per_rank"] == 256 and two["config"]["planes_per_rank"] == 128
    assert one["cells_fused"] > 100000
    assert (one["model_checksum"], one["cells_fused"]) == (two["model_checksum"], two["cells_fused"])
    # the view predicted from the planes: two ranks draw half of the planes each, merge the z-buffers by an unsigned minimum and
    # resolve what their planes won -- together exactly the pixels (bits) one rank predicts from all planes
    assert one["predicted_coverage"] > 0.9 and "prediction" in one["metric"]
    assert (one["prediction_checksum"], one["predicted_coverage"]) == (two["prediction_checksum"], two["predicted_coverage"])
    # the tracking rank without planes of its own (the default from four ranks on): rank 1 owns the whole model
    solo = _bench(args + ["--tracker-works", "0"], world=2)
    assert solo["config"]["planes_per_rank"] == 0 and not solo["config"]["tracker_owns_planes"]
    assert (solo["model_checksum"], solo["cells_fused"], solo["prediction_checksum"]) == (one["model_checksum"], one["cells_fused"], one["prediction_checksum"])
    # labels: the default keeps them in the tracker (u16 images through ProjectLabel + Segment_cu, seeded once); the stand-in that
    # labels every frame afresh by plane membership fuses about the same cells (the growth passes cannot tell the patches of one
    # wall apart at a patch border, the stand-in can) and predicts the same coverage
    assert one["config"]["labels"] == "tracker"
    memb = _bench(args + ["--labels", "membership"])
    assert memb["config"]["labels"] == "membership"
    assert abs(memb["cells_fused"] - one["cells_fused"]) < 0.02 * one["cells_fused"]
    assert abs(memb["predicted_coverage"] - one["predicted_coverage"]) < 0.01
    for o in (one, two):
        assert o["roofline"]["kernel"] == "k_vertex_from_bump_batch" and 0.0 < o["roofline"]["frac"] < 1.0
        assert o["unit"] == "steps/s" and o["value"] > 0


def test_sequences_mode_two_ranks_on_one_card():
    """BASELINE configs[4], the driver's own command (`bench.py --gpus N` through torch.distributed.run): one independent
    sequence per rank, no data-path collective, value = all ranks' frames / max time.  Two ranks rehearsed on one card."""
    one = _bench(["--steps", "40", "--warmup", "5", "--cpu-frames", "0", "--no-large-kernels", "--no-h2d"])
    two = _bench(["--steps", "40", "--warmup", "5", "--cpu-frames", "0", "--no-large-kernels", "--no-h2d"], world=2)
    for o, w in ((one, 1), (two, 2)):
        assert o["n_gpus"] == w and o["scaling"] == "weak" and o["metric"].startswith("ICP+fusion frames/sec") and o["config"]["sequences"] == w
        assert abs(o["value"] - 1e3 * w / o["ms_per_step"]) < 0.01 * o["value"]
    assert two["config"]["icp"].startswith("12 k_icp_iter")      # ranks sharing a card cannot run the persistent kernel
    assert two["value"] > 500 and one["value"] > 1000
