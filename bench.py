#!/usr/bin/env python3
"""bench.py -- ICP+fusion frames/s at 640x480 (BASELINE.json metric) on N MI355X GPUs.

One "step" = one full pass of the hot path over one RGB-D frame, i.e. one display() iteration of
the reference (D1/GP-demo.cpp:138-296): preprocess -> 3-level Gauss-Newton ICP (27-term reduction,
6x6 solve) -> warp + rasterised prediction -> mask/label re-projection -> depth merge -> Bump-image
fusion into 3 planes of 1024x1024 cells.  Workload = BASELINE.json configs[1] (TUM fr1/xyz-shaped
640x480 sequence, synthetic).  Frames are resident in HBM before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): configs[4] -- every rank tracks its
own independent sequence (no data-path collective; "weak" scaling); value = all ranks' frames / max
time over ranks.

The JSON line also carries
  roofline     -- the dominant kernel's algorithmic bytes / its HIP-event duration (events recorded
                  on the library's stream around every launch of that kernel) vs. 8 TB/s HBM3E;
  cpu_baseline -- the CPU oracle (scalar C port of the reference arithmetic, 1 thread) timed on a
                  bounded sample of the same sequence on this box's host cores (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
POOL_FRAMES = 60           # distinct synthetic frames, visited ping-pong so steps stay <= 1 cm / 1 deg

# Algorithmic bytes per unit (SURVEY.md 8d / DESIGN.md "kernels"), per launch of each kernel.
def algo_bytes(kernel, n, m, level_iters):
    P = n * m
    if kernel in ("k_icp_iter", "k_icp_loop"):
        # 80 B per sampled model pixel per executed iteration, all three levels summed
        return sum(80 * (n // f) * (m // f) * it for f, it in zip((1, 2, 4), level_iters))
    return PER_PX[kernel] * P


# per pixel, fused frame kernels: inputs read once + outputs written once
PER_PX = {"k_frame_preprocess": 2 + 3 + 12 + 12 + 12 + 4 + 4,          # depth u16, BGR8 -> VMap, NMap, RGB, Gx, Gy
          "k_predict_draw": 12 + 20,                                   # VMap in -> projected vertices out (z-buffer atomics not counted)
          "k_predict_resolve<true>": 8 + 20 + 16 + 1 + 1 + 12 + 16 + 1 + 1,   # zbuf, projected vertices, RGBD/Mask/Label in -> out
          "k_normal_segment": 12 + 12 + 1 + 1,                         # VMap -> NMap, Label in/out
          "k_segment_pass_pb": 12 + 12 + 1 + 1,
          "k_model_merge": 16 + 1 + 2 + 12 + 16 + 1 + 12 + 12 + 1 + 1 + 8,   # RGBD, Mask, depth, RGB in -> RGBD, Mask, VMap, NMap; + claim pass: Label, cell mask, claim word
          "k_update<false>": 12 + 1 + 1 + 8, "k_update<true>": 12 + 16 + 1 + 1 + 10,
          "k_update_apply_preprocess": (12 + 16 + 1 + 1 + 10) + (2 + 3 + 12 + 12 + 12 + 4 + 4),   # apply pass of frame f + preprocess of f+1
          "k_normal_segment_merge": (1 + 1) + (16 + 1 + 2 + 12 + 16 + 1 + 12 + 12),   # first growth pass (labels in/out; normals only where unlabeled) + Merge
          "k_segment_pass_claim": (12 + 12 + 1 + 1) + (12 + 1 + 1 + 8),                         # last growth pass + claim pass
          # Merge + the three growth passes out of LDS (labels in/out) + the claim pass (cell mask, claim word), one launch
          "k_segment3_merge_claim": (16 + 1 + 2 + 12 + 16 + 1 + 12 + 12) + (1 + 1) + (1 + 8)}
# what rides on a launch when the next frame of a recorded sequence is known (every frame of a sequence but the last):
# the next frame's preprocessing on the merge launch, the previous frame's apply pass on the draw launch
RIDER_PX = {"k_segment3_merge_claim": PER_PX["k_frame_preprocess"], "k_predict_draw": PER_PX["k_update<true>"]}


def hbm_bound_kernels(pkg, lib, synth, torch, device):
    """Fusion / raycast kernels at the sizes where they ARE bandwidth-bound (BASELINE configs[2], [3]):
    the Bump decode of a 256-plane x 512x512 model in one launch, and the fused frame kernels at
    1920x1080.  HIP events around every launch (library hook), algorithmic bytes from DESIGN.md."""
    out = []

    def measure(ctx_h, kernel, run, launches_expected=None):
        lib.rgbdslam_profile_kernel(ctx_h, kernel.encode())
        run()
        tot, cnt = C.c_double(), C.c_int()
        lib.rgbdslam_profile_read(ctx_h, C.byref(tot), C.byref(cnt))
        lib.rgbdslam_profile_kernel(ctx_h, None)
        return tot.value, cnt.value

    # ---- what a plain streaming kernel reaches on THIS card (fractions against it, beside the nominal peak) ----
    ctx = pkg.Context(device)
    probe_bytes = 2 << 30
    pa = torch.zeros(probe_bytes // 4, device="cuda"); pb_ = torch.ones(probe_bytes // 4, device="cuda")
    stream = {}
    for mode, name in ((0, "copy"), (1, "fill"), (2, "read")):
        g_ = C.c_double()
        rc = lib.rgbdslam_profile_hbm_stream(ctx._h, C.c_void_p(pa.data_ptr()), C.c_void_p(pb_.data_ptr()), C.c_size_t(probe_bytes),
                                             mode, 10, C.byref(g_))
        if rc != 0:
            raise RuntimeError("rgbdslam_profile_hbm_stream: %d" % rc)
        stream[name] = round(g_.value, 1)
    del pa, pb_
    torch.cuda.empty_cache()
    # a kernel that reads r bytes and writes w bytes is priced against the time the probe needs for r read + w written
    def measured_frac(gbs, rd, wr):
        t_probe = rd / stream["read"] + wr / stream["fill"]
        return round(gbs * t_probe / (rd + wr), 4)

    # ---- config 3: decode (VertexFromBump, 54 B per cell) of 256 planes x 512 x 512 ----
    planes, N = 256, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    bump = torch.randint(0, 32767, (planes, N, N, 3), dtype=torch.int16, device="cuda", generator=g)
    rgbb = torch.randint(0, 255, (planes, N, N, 3), dtype=torch.uint8, device="cuda", generator=g)
    mask = torch.randint(11, 60, (planes, N, N), dtype=torch.uint8, device="cuda", generator=g)
    param = torch.tensor([0.0, 0.0, 1.0, -2.0, 500.0, 500.0, -1.0, -1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0], device="cuda").repeat(planes, 1)
    vout = torch.empty(planes, N, N, 3, device="cuda"); cout = torch.empty(planes, N, N, 4, device="cuda"); kout = torch.empty(planes, N, N, 4, device="cuda")
    tab = lambda t: torch.tensor([t[k].data_ptr() for k in range(planes)], dtype=torch.int64, device="cuda")
    tabs = [tab(x) for x in (vout, cout, kout, bump, rgbb, mask, param)]
    reps = 6

    def run_decode():
        for _ in range(reps):
            ctx.vertex_from_bump_batch(*tabs, 0, planes, N, N)
        ctx.synchronize()
    run_decode()
    tot, cnt = measure(ctx._h, "k_vertex_from_bump_batch", run_decode)
    nbytes = 54 * planes * N * N
    gbs = nbytes * cnt / (tot * 1e-3) / 1e9
    out.append({"kernel": "k_vertex_from_bump_batch", "workload": "decode of 256 planes x 512x512 Bump cells, one launch",
                "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(tot / cnt * 1e3, 1), "achieved": round(gbs, 1),
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_stream": measured_frac(gbs, 10, 44)})
    # ---- the same model as a mesh (decode + normals + QuadTrim fused, 82 B per cell) ----
    nout = torch.empty(planes, N, N, 3, device="cuda")
    iout = torch.empty(planes, (N - 1) * (N - 1), 4, dtype=torch.int32, device="cuda")
    mtabs = tabs[:3] + [tab(nout), tab(iout)] + tabs[3:]

    def run_mesh():
        for _ in range(reps):
            ctx.bump_mesh_batch(*mtabs, 0, planes, N, N, 0, 0.1, False, 0)
        ctx.synchronize()
    run_mesh()
    tot, cnt = measure(ctx._h, "k_bump_mesh", run_mesh)
    nbytes = 82 * planes * N * N
    gbs = nbytes * cnt / (tot * 1e-3) / 1e9
    out.append({"kernel": "k_bump_mesh", "workload": "mesh (decode + normals + quad trim) of 256 planes x 512x512, one launch",
                "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(tot / cnt * 1e3, 1), "achieved": round(gbs, 1),
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_stream": measured_frac(gbs, 10, 72)})
    del bump, rgbb, mask, vout, cout, kout, nout, iout, tabs, mtabs
    ctx.close()
    torch.cuda.empty_cache()

    # ---- the fused frame kernels at config 4's size (1920x1080) and at the headline size (640x480) ----
    # Durations: HIP event pair around every launch (library hook).
    for n, m in ((1080, 1920), (480, 640)):
        calib = synth.calib_for(n, m)
        pool = synth.sequence(4, n, m, calib, noise_sigma=0.001)
        dd = [torch.from_numpy(p[0]).cuda() for p in pool]
        bb = [torch.from_numpy(p[1]).cuda() for p in pool]
        rg = pkg.RG(n, m, calib, device=device)
        host = pkg.load_host()
        host.slamhost_rg_ctx.restype = C.c_void_p
        ch = C.c_void_p(host.slamhost_rg_ctx(rg._h))
        eq, centers = synth.plane_equations()
        rg.run_device_sequence(dd[:1], bb[:1])
        rg.set_planes(eq, centers)
        order = [1, 2, 3, 2, 1, 0, 1, 2] * (1 if n > 480 else 4)
        run_seq = lambda: rg.run_device_sequence([dd[i] for i in order], [bb[i] for i in order])
        run_seq()
        def run_single():                                # frame by frame: no next-frame hint, so the stand-alone kernels run
            for i in order:
                rg.run_device_sequence([dd[i]], [bb[i]])
        ev0 = C.c_double()
        lib.rgbdslam_profile_event_overhead(ch, 200, C.byref(ev0))
        for kernel, run in (("k_segment3_merge_claim", run_seq), ("k_predict_resolve<true>", run_seq), ("k_predict_draw", run_seq),
                            ("k_frame_preprocess", run_single), ("k_update<true>", run_single)):
            tot, cnt = measure(ch, kernel, run)
            if cnt == 0:
                raise RuntimeError("no launch of %s was seen" % kernel)
            px = PER_PX[kernel]
            if run is run_seq and kernel in RIDER_PX:                 # len(order) - 1 of len(order) launches carry the rider
                px += RIDER_PX[kernel] * (len(order) - 1) / len(order)
            nbytes = int(px * n * m)
            # avg_launch_us = the event pair as it is: an upper bound of the kernel's duration (rocprofv3 of the same command reads
            # 1 - 3 us less on these short launches, the pair minus an empty pair another 1 - 2 us less: profiles/), so `frac` is a
            # lower bound; frac_minus_event_overhead is the optimistic end
            us = tot / cnt * 1e3
            gbs = nbytes / (us * 1e-6) / 1e9
            us_lo = max(1e-3, us - ev0.value * 1e3)
            out.append({"kernel": kernel, "workload": "fused frame pipeline at %dx%d" % (m, n), "algorithmic_bytes_per_launch": nbytes,
                        "avg_launch_us": round(us, 2), "empty_event_pair_us": round(ev0.value * 1e3, 2),
                        "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "frac_of_measured_stream": round(gbs / stream["copy"], 4),
                        "frac_minus_event_overhead": round(nbytes / (us_lo * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
        rg.close()
        del dd, bb, rg
        torch.cuda.empty_cache()
    return out, {"bytes": probe_bytes, "unit": "GB/s", **stream,
                 "note": "k_hbm_stream, one 16-byte word per lane, workgroups in address order, 10 launches; copy counts read + written bytes"}


def check_line(d, base_metric=None):
    """The bench line's contract (keys the driver reads, roofline / CPU-baseline arithmetic).  bench.py runs it on its own
    output before printing, tests/test_bench_contract.py runs it on the committed artefact and on a live GPU run."""
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, "missing key %s" % k
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["n_gpus"] >= 1 and d["steps"] > 0 and d["warmup"] >= 0 and d["value"] > 0
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    units = d["n_gpus"] if d["scaling"] == "weak" else 1          # weak: every rank processes `steps` units of its own
    assert abs(d["value"] - 1e3 * units / d["ms_per_step"]) < 0.01 * d["value"], "value is not whole-job units / max time"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == HBM_PEAK_GBS
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    if base_metric is not None:
        assert d["metric"] in base_metric
    if d["metric"].startswith("ICP+fusion frames/sec"):
        assert d["unit"] == "frames/s" and d["scaling"] == "weak"
        gbs = r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9
        assert abs(gbs - r["achieved"]) < 0.01 * r["achieved"], "achieved != algorithmic bytes per launch / average launch time"
        assert r["event_pair_us"] > r["avg_launch_us"] > 0 and r["empty_event_pair_us"] > 0
        if r["kernel"] in ("k_icp_iter", "k_icp_loop"):
            px = [(d_["rows"] // f) * (d_["cols"] // f) for d_ in (d["config"],) for f in (1, 2, 4)]
            want = 80 * sum(a * b for a, b in zip(r["icp_iterations_per_level"], px))
            assert abs(r["algorithmic_bytes_per_frame"] - want) <= 0.01 * want + 80 * sum(px) / d["steps"]
        c = d["cpu_baseline"]
        if c is not None:
            assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "frames/s" and c["value"] > 0 and c["sample"]
        if d.get("value_with_h2d"):
            assert 0 < d["value_with_h2d"]["value"] <= 1.05 * d["value"]
        if d.get("parity_in_run") is not None:        # lines since round 3: the run carries its own parity evidence
            q = d["parity_in_run"]
            assert q["bit_exact"] is True and q["frames"] > 0 and q["max_abs_pose_diff"] <= q["tolerance"] == 1e-5
    for k in d.get("hbm_bound_kernels") or []:
        assert abs(k["frac"] - k["achieved"] / HBM_PEAK_GBS) < 1e-3 and 0 < k["frac_of_measured_stream"] < 1.5
        gbs = k["algorithmic_bytes_per_launch"] / (k["avg_launch_us"] * 1e-6) / 1e9
        assert abs(gbs - k["achieved"]) < 0.02 * k["achieved"]
    return d


def _max_over_ranks(env, seconds):
    if env["world"] == 1:
        return seconds
    t = env["torch"].tensor([seconds], dtype=env["torch"].float64, device=env["coll_dev"])
    env["dist"].all_reduce(t, op=env["dist"].ReduceOp.MAX)
    return float(t.item())


def _barrier(env):
    env["torch"].cuda.synchronize()
    if env["world"] > 1:
        env["dist"].barrier()
    env["torch"].cuda.synchronize()


def _profile(lib, ctx_h, kernel, run):
    lib.rgbdslam_profile_kernel(ctx_h, kernel.encode())
    run()
    tot, cnt = C.c_double(), C.c_int()
    lib.rgbdslam_profile_read(ctx_h, C.byref(tot), C.byref(cnt))
    lib.rgbdslam_profile_kernel(ctx_h, None)
    return tot.value, cnt.value


def run_planes(args, env):
    """BASELINE configs[2]: a model of 256 planes x 512x512 Bump cells, partitioned by plane over the ranks.  One step =
    rank 0 tracks one 640x480 frame (the configs[1] pipeline without planes of its own), labels the merged model frame
    (16-bit labels, plane membership: detection bypassed as in SURVEY 8d config 2) and BROADCASTS VMap / RGBD / Mask /
    Label / pose (9.2 MB, the step's only collective); every rank then fuses the frame into the planes it owns
    (Graph::Update, claim + apply) and decodes them (VertexFromBump).  Total work is fixed: strong scaling."""
    torch, dist, rank, world, local, pkg, synth = (env[k] for k in ("torch", "dist", "rank", "world", "local", "pkg", "synth"))
    multigpu = importlib.import_module("rgb-d-slam_amd.multigpu")
    planes, N, n, m = 256, 512, 480, 640
    calib = synth.calib_for(n, m)
    stream = torch.cuda.Stream()
    lib = pkg.load()
    with torch.cuda.stream(stream):
        ctx = pkg.Context(local, stream.cuda_stream)
        ctx.set_calibration_matrix(calib)
        # from four ranks on the tracking rank owns no planes (auto): its frame time bounds the step then, and beside the tracker a
        # share of the decode would make it the straggler (scaling table)
        first_worker = {"auto": 1 if world >= 4 else 0, "1": 0, "0": 1}[args.tracker_works]
        if first_worker >= world:
            first_worker = 0
        shard = multigpu.PlaneShard(ctx, synth.tiled_planes(planes, N), N, rank, world, first_worker=first_worker)
        # the z-buffer merge of the prediction is a collective of the PLANE OWNERS: a tracking rank without planes has nothing
        # to draw or resolve, and an all-reduce kernel spinning on its GPU for the workers' buffers would only sit beside its tracker
        owners = None
        if world > 1 and first_worker == 1:
            owners = dist.new_group(ranks=list(range(1, world)))       # (every rank makes this call)
        predicts = args.planes_predict and not (first_worker == 1 and rank == 0)
        thin = args.pack == "thin"
        pipelined = bool(args.pipeline)
        # tracker: the labels are maintained by the reference's own per-frame path, ProjectLabel + Segment_cu (Frame.cpp:385-405)
        # with 16-bit label images, seeded ONCE by plane membership (the stand-in for detection);
        # membership: every frame is labelled afresh by the stand-in (rounds 2-3)
        labels_tracker = args.labels == "tracker"
        labels_seeded = [False]
        # Pipelined step: rank 0 tracks frame f + 1 on a stream of its own (track_stream) while every rank --
        # rank 0 included -- fuses / decodes / predicts frame f on the work stream; two packs alternate, ordered by events.
        # The step time is then max(track, broadcast + work / N) instead of their sum.
        packs = [multigpu.FramePack(n, m, thin=thin) for _ in range(2 if pipelined else 1)]
        pack = packs[0]
        track_stream = torch.cuda.Stream() if (pipelined and rank == 0) else stream
        # pipelined: the broadcast has a stream of its own (the tracker goes on with the next frame while the pack travels), and
        # so has the prediction (it only READS the planes, like the decode: the two run side by side after the fusion)
        comm_stream = torch.cuda.Stream() if pipelined else stream
        pred_stream = torch.cuda.Stream() if (pipelined and args.planes_predict) else stream
        ctx_pred = ctx
        if pred_stream is not stream:
            ctx_pred = pkg.Context(local, pred_stream.cuda_stream)
            ctx_pred.set_calibration_matrix(calib)
        ev_exported = [torch.cuda.Event() for _ in packs]    # tracker -> comm: the pack is filled (rank 0)
        ev_ready = [torch.cuda.Event() for _ in packs]       # comm -> work: the pack has arrived
        ev_free = [torch.cuda.Event() for _ in packs]        # work -> tracker / comm: the fusion has read the pack
        ev_fused, ev_pred = torch.cuda.Event(), torch.cuda.Event()
        free_recorded = [False for _ in packs]
        rg = rg_ctx = None
        if rank == 0:
            pool = synth.sequence(POOL_FRAMES, n, m, calib, noise_sigma=0.001, seed=0)
            depth_dev = [torch.from_numpy(p[0]).cuda() for p in pool]
            bgr_dev = [torch.from_numpy(p[1]).cuda() for p in pool]
            rg = pkg.RG(n, m, calib, device=local, stream=track_stream.cuda_stream)
            rg.set_finish(False)                     # the model lives in the plane shards, not in the tracker's own Graph
            rg_ctx = rg.context()                    # label + export run on the tracker's stream
            if labels_tracker:
                rg.set_label16(True)                 # 256 planes: beyond the reference's unsigned char labels (Frame.h:197)
            # The persistent ICP kernel needs every CU for itself (one workgroup per CU, all resident).  Beside the decode of
            # planes on the SAME GPU (pipelined step, tracker owning planes) it does not get them: its workgroups trickle in as
            # the decode's drain, the resident ones spin, and when the two start together the decode runs at half speed
            # (kernel trace, profiles/r03_o_planes_trace.txt: 1 200 instead of 650 us on every third step).  In the pipelined
            # step the loop therefore runs as one launch per iteration, which interleaves -- also on a rank 0 without planes,
            # where RCCL's broadcast kernel still shares the GPU with it (unrehearsed: no multi-GPU node; --icp-loop 2 forces the
            # persistent kernel there); in the serial step it is the one launch per frame.
            loop_ok = args.icp_loop == 2 or (bool(args.icp_loop) and not pipelined)
            icp_form = "one launch per iteration"
            if world == 1 or "RGBDSLAM_BENCH_DEVICE" not in os.environ:
                try:
                    rg.set_icp_loop(loop_ok)
                    icp_form = "persistent kernel" if loop_ok else icp_form
                except RuntimeError:
                    pass
        order, k, d = [], 0, 1
        for _ in range(2 * (args.warmup + args.steps) + 4 + 7 * 20):    # timed pass, phase pass, one short pass per profiled kernel
            order.append(k)
            if k + d < 0 or k + d >= POOL_FRAMES:
                d = -d
            k += d
        tracked_no = [0]                             # frames handed to the tracker / the broadcast so far
        step_no = [0]                                # frames fused so far (every rank)
        view_w2c = []
        for R_, t_ in synth.trajectory(POOL_FRAMES, 0):
            Pw = np.eye(4)
            Pw[:3, :3] = R_.T
            Pw[:3, 3] = -R_.T @ t_
            view_w2c.append(np.ascontiguousarray(Pw.T.reshape(-1), np.float32))    # column-major 4x4
        torch.cuda.synchronize()

        def track_and_send(mark=None):
            """the next frame: rank 0 tracks, labels and exports it into its pack (tracker's stream); every rank takes part in
            the pack's broadcast (comm stream)"""
            f = tracked_no[0]
            tracked_no[0] += 1
            b = f % len(packs)
            pk = packs[b]
            if rank == 0:
                if pipelined and free_recorded[b]:
                    track_stream.wait_event(ev_free[b])          # the fusion of two frames ago has read this pack
                if mark:
                    mark("k0", track_stream)
                i = order[f]
                rg.load_frame_device(depth_dev[i], bgr_dev[i])
                if labels_tracker and labels_seeded[0] and f + 1 < len(order):   # the next frame's LoadFrame rides on this frame's merge launch
                    rg.hint_next_device_frame(depth_dev[order[f + 1]], bgr_dev[order[f + 1]])
                rg.step_async()
                if labels_tracker and not labels_seeded[0]:
                    prm_ = synth.tiled_planes(planes, N)
                    rg.set_plane_equations(np.concatenate([prm_[:, 0:3], prm_[:, 8:14], prm_[:, 3:4]], axis=1))
                    rg.seed_labels_tiled(shard.t_param, planes, N, N)
                    labels_seeded[0] = True
                if thin and labels_tracker:              # RGBD.w, colour bytes, mask, labels and the history entry in ONE launch
                    rg.export_frame_thin_labels(pk.buf, pk.off_w, pk.off_rgb8, pk.off_mask, pk.off_hist, pk.off_label)
                elif thin:
                    rg.export_frame_thin(pk.buf, pk.off_w, pk.off_rgb8, pk.off_mask, pk.off_hist)
                else:
                    rg.export_frame(pk.buf, pk.off_vmap, pk.off_rgbd, pk.off_mask, pk.off_hist)
                if labels_tracker and not thin:
                    rg.export_labels(pk.buf, pk.off_label)
                elif not labels_tracker:
                    shard.label(pk, rg.buffer_ptr("VMap")[0], rg.buffer_ptr("NMap")[0], ctx=rg_ctx)
                if mark:
                    mark("k1", track_stream)
                if pipelined:
                    ev_exported[b].record(track_stream)
                    comm_stream.wait_event(ev_exported[b])
            elif pipelined and free_recorded[b]:
                comm_stream.wait_event(ev_free[b])               # the pack may be overwritten
            if mark:
                mark("t1", comm_stream)
            with torch.cuda.stream(comm_stream):
                pk.broadcast(0)
            if mark:
                mark("t2", comm_stream)
            if pipelined:
                ev_ready[b].record(comm_stream)

        def work(mark=None):
            """every rank: fusion of the frame into the owned planes, then their decode and -- beside it -- the prediction"""
            f = step_no[0]
            step_no[0] += 1
            b = f % len(packs)
            pk = packs[b]
            if pipelined:
                stream.wait_event(ev_ready[b])
            if mark:
                mark("w0", stream)
            shard.fuse(pk)
            if pipelined:
                ev_free[b].record(stream)            # nothing after the fusion reads the pack
                free_recorded[b] = True
            if mark:
                mark("w1", stream)
            if predicts and pred_stream is not stream:
                ev_fused.record(stream)
                pred_stream.wait_event(ev_fused)
            shard.decode()
            if mark:
                mark("w2", stream)
            if predicts:
                # the "raycast" half of configs[2]: the view of the frame just fused, predicted from the planes.  The view is
                # the synthetic trajectory's pose of that frame (known to every rank without reading the tracked pose
                # back from rank 0's device: nothing here waits for the GPU)
                if mark:
                    mark("p0", pred_stream)
                with torch.cuda.stream(pred_stream):
                    shard.predict(view_w2c[order[f]], n, m, ctx=ctx_pred, group=owners)
                if mark:
                    mark("p1", pred_stream)
                if pred_stream is not stream:
                    ev_pred.record(pred_stream)
                    stream.wait_event(ev_pred)       # the next fusion writes the planes the prediction reads
            if mark:
                mark("w3", stream)

        def step(mark=None):
            track_and_send(mark)                     # pipelined: frame f + 1, enqueued first, beside the work on frame f
            work(mark)

        if pipelined:
            track_and_send()                         # prologue: frame 0 (tracker and broadcast stay one frame ahead from here on)
        for _ in range(1 + args.warmup):
            step()
        _barrier(env)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        for st_ in (stream, track_stream, comm_stream, pred_stream):
            st_.synchronize()
        _barrier(env)
        dt = _max_over_ranks(env, time.perf_counter() - t0)
        tracked = 0
        if rank == 0:
            rg.synchronize()
            hn = min(args.steps, pkg.RG.HISTORY_CAP)             # (a longer run: the frames the history ring still holds)
            h = rg.history(rg.frames_done() - hn, hn)
            tracked = int((h[:, 12] == 1.0).sum())
            if tracked != hn:
                raise SystemExit("tracking lost on %d of %d timed frames: result invalid" % (hn - tracked, hn))
            if loop_ok and rg.icp_loop_faults():
                raise SystemExit("the persistent ICP kernel gave up waiting for its grid: result invalid")
        # ---- second pass (untimed): phases by events on rank 0's streams, decode kernel by the library's event hook ----
        phases = None
        marks = []

        def mark(name, on):
            e = torch.cuda.Event(enable_timing=True)
            e.record(on)
            marks.append((name, e))
        reps = min(args.steps, 50)
        tot_ms, launches = _profile(lib, ctx._h, "k_vertex_from_bump_batch", lambda: [step(mark if rank == 0 else None) for _ in range(reps)])
        stream.synchronize()
        track_stream.synchronize()
        for st_ in (comm_stream, pred_stream):
            st_.synchronize()
        if rank == 0:
            by = {}
            for name, e in marks:
                by.setdefault(name, []).append(e)
            span = lambda a_, b_: round(sum(by[a_][q].elapsed_time(by[b_][q]) for q in range(reps)) / reps * 1e3, 1)
            phases = {"track_label_export": span("k0", "k1"), "broadcast": span("t1", "t2"), "fuse": span("w0", "w1"), "decode": span("w1", "w2"),
                      "predict_draw_merge_resolve": span("p0", "p1") if predicts else 0.0,
                      "work_stream_step": round(by["w0"][0].elapsed_time(by["w3"][reps - 1]) / reps * 1e3, 1)}
            phases["unit"] = ("us per step on rank 0 (event pairs, second pass); pipelined: tracking, broadcast, fusion + decode and the "
                              "prediction each on a stream of their own, spans overlap" if pipelined else
                              "us per step on rank 0 (event pairs on its stream, second pass)")
        # the other kernels of the step, one profiled pass each (the library's hook follows one kernel name at a time)
        planes_kernels = []
        ev0 = C.c_double()
        lib.rgbdslam_profile_event_overhead(ctx._h, 100, C.byref(ev0))
        usable = int((shard.mask > 10).sum()) + int((shard.mask == 1).sum())
        owned_cells = (shard.hi - shard.lo) * N * N
        occupied_tiles = int((shard.mask.view(-1, N // 8, 8, N // 64, 64) > 10).any(dim=4).any(dim=2).sum()) if shard.hi > shard.lo else 0
        extra = [("k_update_thin<true>", (4 + 3 + 1 + 2 + 10) * n * m, "Graph::Update apply pass from the thin pack, u16 labels: 20 B per frame pixel") if thin else
                 ("k_update<true>", (12 + 16 + 1 + 2 + 10) * n * m, "Graph::Update apply pass, u16 labels: 41 B per frame pixel")]
        if args.planes_predict:
            covered = int((shard.pred_v[..., 2] != 0).sum()) if getattr(shard, "pred_v", None) is not None else 0
            extra += [("k_planes_scan", owned_cells, "1 B mask per owned cell: every byte has to be looked at, no occupancy is kept between calls"),
                      ("k_planes_draw", 7 * 585 * occupied_tiles + 8 * covered,
                       "7 B (mask + Bump words) per cell of the 9 x 65 cells of every occupied tile + 8 B z-word per drawn pixel"),
                      ("k_planes_resolve", (8 + 3 * 10 + 12 + 16) * n * m, "z-word + three 10-byte cells read, VMap + RGBD written, per pixel")]
        # (every rank walks the same list: each pass runs steps, and a step holds collectives; the label kernel only runs on rank 0)
        if labels_tracker:
            extra.append(("k_segment3_merge_claim", (72 + 2 + 9 + 49) * n * m,
                          "tracker: Merge + three growth passes on u16 labels (no claim: the planes live in the shards) + the next frame's LoadFrame"))
        else:
            extra.append(("k_label_planes_l16", (12 + 12 + 2) * n * m + 56 * planes, "VMap + NMap read, u16 label written per pixel; 56 B per plane"))
        for kname, nbytes, what in extra:
            kctx = rg_ctx._h if (kname in ("k_label_planes_l16", "k_segment3_merge_claim") and rank == 0) else ctx._h      # these run on the tracker's context
            if kname.startswith("k_planes_"):
                kctx = ctx_pred._h                   # the prediction has a context (stream) of its own
            t_ms, cnt = _profile(lib, kctx, kname, lambda: [step() for _ in range(min(reps, 20))])
            for st_ in (stream, track_stream, comm_stream, pred_stream):
                st_.synchronize()
            if cnt == 0:
                continue
            us = max(1e-3, (t_ms / cnt - ev0.value) * 1e3)
            gbs = nbytes / (us * 1e-6) / 1e9
            planes_kernels.append({"kernel": kname, "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": int(nbytes),
                                   "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes": what})
        # "a checksum of checksums": the model after all steps, independent of how the planes were partitioned
        chk = torch.tensor([int(shard.bump.to(torch.int64).sum()) + 3 * int(shard.rgb.to(torch.int64).sum()) + 7 * int(shard.mask.to(torch.int64).sum()),
                            int((shard.mask > 10).sum())], dtype=torch.int64, device=env["coll_dev"])
        if args.planes_predict:
            # the last predicted view, summed over the ranks' disjoint outputs as integers: independent of the partition
            pchk, pcov = 0, 0                        # (a tracking rank without planes predicts nothing)
            if predicts:
                pchk = int(shard.pred_v.view(torch.int32).to(torch.int64).sum()) + 3 * int(shard.pred_c.view(torch.int32).to(torch.int64).sum())
                pcov = int((shard.pred_v[..., 2] != 0).sum())
            chk = torch.cat([chk, torch.tensor([pchk, pcov], dtype=torch.int64, device=env["coll_dev"])])
        if world > 1:
            dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        cells = (shard.hi - shard.lo) * N * N
        ev_ms = C.c_double()
        lib.rgbdslam_profile_event_overhead(ctx._h, 100, C.byref(ev_ms))
        avg_ms = max(1e-9, tot_ms / max(1, launches) - ev_ms.value)
        if world > 1:                                # the roofline is that of the first rank that owns planes
            t_ = torch.tensor([float(cells), avg_ms], dtype=torch.float64, device=env["coll_dev"])
            dist.broadcast(t_, first_worker)
            cells, avg_ms = int(t_[0].item()), float(t_[1].item())
        achieved = 54 * cells / (avg_ms * 1e-3) / 1e9
        _barrier(env)
        if ctx_pred is not ctx:
            ctx_pred.close()
        if rank != 0:
            ctx.close()
            return None
        out = {"metric": "plane-partitioned Bump fusion + decode%s steps/sec (256 planes x 512x512, one 640x480 frame per step)"
                         % (" + prediction" if args.planes_predict else ""),
               "value": round(args.steps / dt, 2), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[2]: 256 planes x 512x512 Bump cells tiled over the three planes of the synthetic room "
                                      "corner; per step rank 0 tracks one 640x480 frame, labels it (u16; " + ("ProjectLabel + Segment_cu in the tracker, seeded once by plane membership" if labels_tracker else "by plane membership, every frame") + ") and broadcasts " + ("RGBD.w/colour bytes" if thin else "VMap/RGBD") + "/Mask/Label/pose "
                                      "(%d bytes), every rank fuses it into and decodes the planes it owns%s" % (pack.nbytes,
                                      ", then draws them into a u64 z-buffer for the frame's view, all-reduce(MIN) of %d bytes, resolve" % (8 * n * m)
                                      if args.planes_predict else ""),
                          "planes_per_rank": shard.hi - shard.lo, "tracker_owns_planes": first_worker == 0,
                          "broadcast_bytes": pack.nbytes, "pack": args.pack, "labels": args.labels, "tracker_icp": icp_form,
                          "pipelined": pipelined, "backend": env["backend"],
                          "frames_resident_in_hbm": True},
               "roofline": {"bound": "hbm", "kernel": "k_vertex_from_bump_batch", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                            "algorithmic_bytes_per_launch": 54 * cells, "avg_launch_us": round(avg_ms * 1e3, 2),
                            "note": "rank %d's share: 54 B per owned Bump cell per launch" % first_worker},
               "phases": phases, "planes_kernels": planes_kernels, "cpu_baseline": None,
               "model_checksum": int(chk[0].item()), "cells_fused": int(chk[1].item()),
               "prediction_checksum": int(chk[2].item()) if args.planes_predict else None,
               "predicted_coverage": round(int(chk[3].item()) / float(n * m), 4) if args.planes_predict else None}
        rg.close()
        ctx.close()
        return out


def run_rows(args, env):
    """BASELINE configs[3]: row-sharded ICP on a 1920x1080 pair.  One step = one coarse-to-fine Gauss-Newton alignment
    (RG::AlignFrameGaussNewton) of the pair from the same start pose; every rank computes the normal equations of its band of
    8x8-block rows, the 27 partial sums are exchanged every iteration, and every rank runs the identical solve.  Total work
    is fixed: strong scaling."""
    torch, dist, rank, world, local, pkg, synth = (env[k] for k in ("torch", "dist", "rank", "world", "local", "pkg", "synth"))
    multigpu = importlib.import_module("rgb-d-slam_amd.multigpu")
    n, m = (args.rows, args.cols) if (args.rows, args.cols) != (480, 640) else (1080, 1920)
    calib = synth.calib_for(n, m)
    stream = torch.cuda.Stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with torch.cuda.stream(stream):
        ctx = pkg.Context(local, stream.cuda_stream)
        ctx.set_calibration_matrix(calib)
        R1, t1 = synth.pair_offset_pose()
        g = {}
        rng = np.random.default_rng(1)
        for k, (R, t) in enumerate(((np.eye(3), np.zeros(3)), (R1, t1))):
            d, c = synth.render(R, t, n, m, calib, noise_sigma=0.001, rng=rng)   # the same pair on every rank (same seed)
            gd, gc = dev(d), dev(c)
            v = torch.empty(n, m, 3, device="cuda"); nm = torch.empty(n, m, 3, device="cuda"); rgb = torch.empty(n, m, 3, device="cuda")
            ctx.vertex_map_kinect(v, gd, 2 * m, n, m)
            ctx.compute_normal(nm, v, n, m, True)
            ctx.gpu_cpy_char3(gc, 3 * m, rgb, n, m)
            g["v%d" % k], g["n%d" % k], g["rgb%d" % k], g["d%d" % k] = v, nm, rgb, gd
        rgbd0 = torch.empty(n, m, 4, device="cuda")
        ctx.gpu_add_zbuff_off(g["d0"], 2 * m, g["rgb0"], rgbd0, n, m)
        gx = torch.empty(n, m, device="cuda"); gy = torch.empty(n, m, device="cuda")
        ctx.intensity_sobel(g["rgb1"], None, gx, gy, n, m)
        aux = torch.empty(n, m, 4, device="cuda")        # the frame's auxiliary ICP records (the persistent form reads them)
        ctx.frame_aux(g["v1"], g["rgb1"], gx, gy, aux, n, m)
        ctx.alloc_buffers((n + 7) // 8, (m + 7) // 8)
        prm = pkg.icp_params()
        aligner = multigpu.ShardedAligner(ctx, prm, n, m, rank, world, exchange=args.exchange)
        # test hook: RGBDSLAM_BENCH_XCHG_DELAY="rank:ticks" makes that rank's tail fall behind (ticks x 10 ns) between its
        # publication and its first poll in every exchanged iteration (the double-buffered slots must absorb it)
        dl = os.environ.get("RGBDSLAM_BENCH_XCHG_DELAY")
        if dl and int(dl.split(":")[0]) == rank:
            if ctx.lib.rgbdslam_debug_set_exchange_delay(ctx._h, int(dl.split(":")[1])) != 0:
                raise SystemExit("rgbdslam_debug_set_exchange_delay refused %s" % dl)
        I = np.eye(3, dtype=np.float32).reshape(-1); z = np.zeros(3, np.float32)
        start = dev(np.concatenate([I, z, I, z]).astype(np.float32))
        if args.exchange == "peer_loop":             # the persistent form keeps its state in a device pose block
            state, status = aligner.pose_block_views()
        else:
            state = start.clone()
            status = torch.zeros(4, dtype=torch.int32, device="cuda")

        def step():
            state.copy_(start)
            aligner.align(g["v1"], g["n1"], g["rgb1"], gx, gy, g["v0"], g["n0"], rgbd0, state, status, aux=aux)

        for _ in range(1 + args.warmup):
            step()
        _barrier(env)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        stream.synchronize()
        _barrier(env)
        dt = _max_over_ranks(env, time.perf_counter() - t0)
        st = status.cpu().numpy()
        pose = state.cpu()
        if int(st[0]) != 1:
            raise SystemExit("rank %d: alignment lost: result invalid" % rank)
        if world > 1:      # every rank must hold the same bits
            ref = pose.clone().to(env["coll_dev"])
            dist.broadcast(ref, 0)
            if not torch.equal(ref.cpu(), pose):
                raise SystemExit("rank %d: pose differs from rank 0's: result invalid" % rank)
        li = int(st[2])
        level_iters = [li & 255, (li >> 8) & 255, (li >> 16) & 255]
        # roofline of this rank's band kernel: 80 B per sampled model pixel of the band per executed iteration
        lib = pkg.load()
        rows_kernel = "k_icp_loop_sharded" if args.exchange == "peer_loop" else "k_icp_iter"
        tot_ms, launches = _profile(lib, ctx._h, rows_kernel, lambda: ([step() for _ in range(min(args.steps, 20))], stream.synchronize()))
        reps = min(args.steps, 20)
        band_bytes = 0
        for lvl, it in enumerate(level_iters):
            rows = (n // (1 << lvl) + 7) // 8 if (n // (1 << lvl)) % 8 else (n // (1 << lvl)) // 8
            lo, hi = multigpu.band_rows(rank, world, rows)
            band_bytes += 80 * min(8 * (hi - lo), n // (1 << lvl)) * (m // (1 << lvl)) * it
        ev_ms = C.c_double()
        lib.rgbdslam_profile_event_overhead(ctx._h, 100, C.byref(ev_ms))
        # k_icp_iter launches that follow convergence exit at once; the event pairs cover them all
        net_ms = max(1e-9, tot_ms - launches * ev_ms.value)
        achieved = band_bytes * reps / (net_ms * 1e-3) / 1e9
        _barrier(env)
        ctx.close()
        if rank != 0:
            return None
        return {"metric": "row-sharded ICP alignments/sec at %dx%d" % (m, n), "value": round(args.steps / dt, 2), "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "BASELINE configs[3]: synthetic %dx%d pair (1 mm depth noise), 3-level Gauss-Newton ICP row-sharded into %d "
                                       "bands of 8x8-block rows, 27-float exchange per iteration (%s), identical solve on every rank"
                                       % (m, n, world, aligner.exchange),
                           "icp_iterations_per_level": level_iters, "backend": env["backend"], "frames_resident_in_hbm": True,
                           "pose": [float(x) for x in pose[:12].tolist()]},
                "roofline": {"bound": "hbm", "kernel": rows_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                             "algorithmic_bytes_per_frame": band_bytes, "launches_per_frame": round(launches / reps, 1),
                             "note": "rank 0's band, all its %s launches of a frame%s" % (rows_kernel, "" if args.exchange == "peer_loop" else " (incl. the no-op launches after convergence)")},
                "cpu_baseline": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--mode", default="sequences", choices=["sequences", "planes", "rows"],
                    help="sequences (default): the BASELINE metric, configs[1] at N=1 / configs[4] at N>1 (one sequence per GPU); "
                         "planes: configs[2], 256 planes x 512x512, rank 0 tracks + broadcasts the frame, every rank fuses + decodes "
                         "its planes; rows: configs[3], row-sharded ICP at 1920x1080 with the 27-float exchange")
    ap.add_argument("--planes-predict", type=int, default=1,
                    help="planes mode: 1 (default) = every step also predicts the frame's view from the planes (draw + all-reduce(MIN) + "
                         "resolve, the raycast half of configs[2]); 0 = fusion + decode only")
    ap.add_argument("--pack", default="thin", choices=["thin", "fat"],
                    help="planes mode: what rank 0 broadcasts per frame: thin (default) = RGBD.w + colour bytes + Mask + Label (10 B/px; every "
                         "rank rebuilds VMap / RGBD.rgb), fat = VMap + RGBD + Mask + Label (31 B/px, rounds 1-2)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="planes mode: 1 (default) = rank 0 tracks frame f+1 on its own stream beside the fusion / decode / prediction of "
                         "frame f; 0 = one stream, the phases one after the other")
    ap.add_argument("--labels", default="tracker", choices=["tracker", "membership"],
                    help="planes mode: tracker (default) = u16 labels maintained by ProjectLabel + Segment in the tracker, seeded once; "
                         "membership = the plane-membership stand-in labels every frame")
    ap.add_argument("--tracker-works", default="auto", choices=["auto", "0", "1"],
                    help="planes mode: does rank 0 (the tracker) own planes too?  auto (default): yes below four ranks, no from four on")
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "peer", "peer_loop"],
                    help="rows mode: how the 27 partial sums travel: allgather = torch.distributed all-gather between three launches per "
                         "iteration; peer = peer-mapped slots written from inside ONE launch per iteration; peer_loop = the same exchange "
                         "inside ONE persistent launch per alignment (needs the GPU to itself: not for ranks sharing a card)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--kernel", default="k_icp_iter", help="kernel whose roofline is reported")
    ap.add_argument("--cpu-frames", type=int, default=40, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--icp-loop", type=int, default=1,
                    help="1 (default): persistent-loop ICP, one launch per frame (rgbdslam_set_icp_loop: this process owns the GPU "
                         "and uses one stream, which is how bench.py runs); 0: twelve k_icp_iter launches per frame; "
                         "planes mode: the pipelined step runs the tracker's loop as one launch per iteration unless this is 2")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: next frame's preprocessing and the Bump update on auxiliary streams (RG::overlap; measured slower)")
    ap.add_argument("--finish-period", type=int, default=0,
                    help="RG::Finish's periodic dump + plane re-detection cycle in the timed region: every P frames (reference: 100; "
                         "default 0 = off, see DESIGN.md section 6); the first detection after frame 30 always runs, in the set-up")
    ap.add_argument("--no-large-kernels", action="store_true", help="skip the config-3/config-4-size kernel rooflines")
    ap.add_argument("--preheat-ms", type=int, default=120,
                    help="untimed load before the warm-up (sequences mode): a second pipeline object runs the same frames for about "
                         "this long, so that the GPU's clocks have settled when a warm-up of a few frames starts (a 30-frame warm-up "
                         "is 3.5 ms: measured 8 640 frames/s against 8 900 after a 1 000-frame one); 0 = none")
    ap.add_argument("--no-h2d", action="store_true",
                    help="skip the PCIe-inclusive measurement (two streams ordered by stream value operations: a counter-collecting "
                         "profiler that serialises dispatches across queues never lets the producer stream run)")
    ap.add_argument("--profile-in-timed-region", action="store_true",
                    help="record the roofline events inside the timed region instead of in a second pass")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # rehearsal on a 1-GPU box: RGBDSLAM_BENCH_BACKEND=gloo RGBDSLAM_BENCH_DEVICE=0 lets N ranks share one card
    backend = os.environ.get("RGBDSLAM_BENCH_BACKEND", "nccl")
    if "RGBDSLAM_BENCH_DEVICE" in os.environ:
        local = int(os.environ["RGBDSLAM_BENCH_DEVICE"])
        if world > 1:
            args.icp_loop = 0          # ranks share one card: the persistent kernel needs the GPU to itself
    torch.cuda.set_device(local)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("rgb-d-slam_amd")
    synth = importlib.import_module("rgb-d-slam_amd.synth")
    if args.mode != "sequences":
        env = {"torch": torch, "dist": dist, "rank": rank, "world": world, "local": local, "coll_dev": coll_dev, "pkg": pkg, "synth": synth,
               "backend": backend}
        out = run_planes(args, env) if args.mode == "planes" else run_rows(args, env)
        if rank == 0:
            print(json.dumps(check_line(out)), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    n, m = args.rows, args.cols
    calib = synth.calib_for(n, m)
    pool = synth.sequence(POOL_FRAMES, n, m, calib, noise_sigma=0.001, seed=rank)
    SETUP = 31      # seed frame + 30 more: RG::Finish runs the first plane detection once _count > 30 (RG.cpp:154)
    total = SETUP + args.warmup + args.steps
    order = []
    k, d = 0, 1
    for _ in range(total + args.steps + 340):
        order.append(k)
        if k + d < 0 or k + d >= POOL_FRAMES:
            d = -d
        k += d
    depth_dev = [torch.from_numpy(p[0]).cuda() for p in pool]
    bgr_dev = [torch.from_numpy(p[1]).cuda() for p in pool]

    stream = torch.cuda.Stream()
    rg = pkg.RG(n, m, calib, device=local, stream=stream.cuda_stream)
    host = pkg.load_host()
    lib = pkg.load()
    host.slamhost_rg_ctx.restype = C.c_void_p
    ctx = C.c_void_p(host.slamhost_rg_ctx(rg._h))
    eq, centers = synth.plane_equations()
    if args.overlap:
        rg.set_overlap(True)
    if args.icp_loop:
        try:
            rg.set_icp_loop(True)
        except RuntimeError:                 # the device cannot hold the loop kernel's grid: multi-launch path
            args.icp_loop = 0
    if args.kernel == "k_icp_iter" and args.icp_loop:
        args.kernel = "k_icp_loop"

    def run(frames):
        return rg.run_device_sequence([depth_dev[i] for i in frames], [bgr_dev[i] for i in frames])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up (untimed, before the warm-up): frame 0 seeds the model, and RG::Finish (step 12 of display(), RG.cpp:149-175) detects
    # the planes itself after frame 30 -- nothing is injected.  The periodic dump + re-detection cycle of RG::Finish
    # (every 100 frames, RG.cpp:187-270: plane detection and file I/O, rows N1 / N2 of SURVEY 8f, a host-synchronous
    # step outside the five stages the metric names) is switched off for the timed region; DESIGN.md section 6 gives its cost.
    rg.set_finish(True, 30, 0)
    run(order[:SETUP])
    fs = rg.finish_stats()
    if fs["detections"] != 1 or fs["planes"] < 3:
        raise SystemExit("plane detection found %d planes after the set-up frames: result invalid" % fs["planes"])
    pos = SETUP
    if args.finish_period:
        rg.set_finish(True, 30, args.finish_period, reset_without_dump=True)   # no dump directory: the cycle's cost without file I/O
    # Untimed pre-heat (not a step of the measured sequence): clocks and caches need tens of milliseconds of load to settle, the
    # warm-up the caller asks for may be a few frames.  A second pipeline object runs the same frames for ~preheat_ms right in
    # front of the warm-up; the timed region stays exactly `steps` frames of `rg` between two barriers.
    if args.preheat_ms > 0:
        rgp = pkg.RG(n, m, calib, device=local, stream=stream.cuda_stream)
        if args.icp_loop:
            rgp.set_icp_loop(True)
        rgp.set_finish(True, 30, 0)
        seqp = lambda frames: rgp.run_device_sequence([depth_dev[i] for i in frames], [bgr_dev[i] for i in frames])
        seqp(order[:SETUP])
        heat, hk, hd = [], order[SETUP - 1], 1                    # goes on back and forth through the pool from where the set-up stopped
        for _ in range(int(args.preheat_ms * 9)):                 # ~9 frames per millisecond
            if hk + hd < 0 or hk + hd >= POOL_FRAMES:
                hd = -hd
            hk += hd
            heat.append(hk)
        seqp(heat)                                                # (rgp is closed after the timed region: freeing device memory waits for the device)
    run(order[pos:pos + args.warmup])
    pos += args.warmup
    if args.profile_in_timed_region:
        lib.rgbdslam_profile_kernel(ctx, args.kernel.encode())
    timed = rg.prepare_device_sequence([depth_dev[i] for i in order[pos:pos + args.steps]], [bgr_dev[i] for i in order[pos:pos + args.steps]])
    barrier()
    t0 = time.perf_counter()
    tracked = rg.run_device_sequence(timed)        # (the pointer tables of the call were built above: 2 x steps look-ups);
    barrier()                                      # the call returns after its own stream synchronisation; device-wide sync + barrier here
    dt = time.perf_counter() - t0
    pos += args.steps
    if args.preheat_ms > 0:
        rgp.close()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tr = torch.tensor([tracked], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MIN)
        tracked = int(tr.item())
    if args.icp_loop and rg.icp_loop_faults():
        raise SystemExit("the persistent ICP kernel gave up waiting for its grid (is another process using this GPU?): result invalid")
    if tracked != args.steps:
        raise SystemExit("tracking lost on %d of %d timed frames: result invalid" % (args.steps - tracked, args.steps))
    # poses of the warm-up and the timed frames as the device left them (compared with the oracle's in the cpu_baseline leg)
    gpu_first = SETUP
    # a soak longer than the history ring has overwritten the frames the oracle leg can reach: the line then says so instead of
    # carrying a comparison (the default run and every run up to HISTORY_CAP frames carry it)
    ring_wrapped = SETUP + args.warmup + args.steps > pkg.RG.HISTORY_CAP
    gpu_hist = (rg.history(gpu_first, min(args.warmup + args.steps, max(0, 3 * args.cpu_frames)))
                if rank == 0 and world == 1 and args.cpu_frames > 0 and not ring_wrapped else None)

    # ---- roofline of the dominant kernel: HIP events around each of its launches ----
    level_iters = [0, 0, 0]
    if not args.profile_in_timed_region:
        lib.rgbdslam_profile_kernel(ctx, args.kernel.encode())
        run(order[pos:pos + args.steps])
    # iteration counts of EVERY profiled frame (history ring: 16 floats per frame, [14] = executed iterations per level,
    # 8 bits each), not just the last one's
    fd = rg.frames_done()
    hn = min(args.steps, pkg.RG.HISTORY_CAP)                 # a longer run (soak): the frames the ring still holds
    hist = rg.history(fd - hn, hn)
    packed = hist[:, 14].astype(np.int64)
    per_level_total = [int(((packed >> (8 * k)) & 255).sum()) for k in range(3)]
    level_iters = [round(t / float(hn), 2) for t in per_level_total]
    tot_ms, launches = C.c_double(), C.c_int()
    lib.rgbdslam_profile_read(ctx, C.byref(tot_ms), C.byref(launches))
    lib.rgbdslam_profile_kernel(ctx, None)
    # ---- per-stage timers (the reference's TIMING switch, Utilities.h:82 / GP-demo.cpp:153-277): a pass of its own, 64 frames,
    # HIP events at the stage boundaries (RG::SetStageTiming) -- the events cost ~2 us each, so this is not the timed region ----
    stage_times = None
    cont = pos if args.profile_in_timed_region else pos + args.steps
    if len(order) >= cont + 64:
        rg.set_stage_timing(True)
        run(order[cont:cont + 64])
        st_, nf_ = rg.stage_times()
        rg.set_stage_timing(False)
        stage_times = dict(st_, frames=nf_, unit="us per frame, HIP events at the stage boundaries in a pass of their own (event cost included: "
                           "the sum exceeds ms_per_step); stage 1 rides on the previous frame's stage-4 launch inside a recorded sequence, "
                           "so its own span is the two event records only")
    per_frame_launches = launches.value / float(args.steps)
    if args.kernel in ("k_icp_iter", "k_icp_loop"):
        bytes_all_frames = algo_bytes(args.kernel, n, m, [t * (args.steps / float(hn)) for t in per_level_total])   # (hn < steps: the mean of the frames still in the ring)
    else:
        bytes_all_frames = PER_PX[args.kernel] * n * m * launches.value
    bytes_per_frame = bytes_all_frames / float(args.steps)
    # every event pair carries the markers' own processing time on top of what it brackets: measured with empty
    # pairs on the same stream and taken off (one pair per frame: the pair around the loop kernel, or around the
    # twelve back-to-back k_icp_iter launches)
    ev_ms = C.c_double()
    lib.rgbdslam_profile_event_overhead(ctx, 200, C.byref(ev_ms))
    pairs = args.steps if args.kernel in ("k_icp_iter", "k_icp_loop") else launches.value
    net_ms = max(0.0, tot_ms.value - pairs * ev_ms.value)
    avg_ms = net_ms / max(1, launches.value)
    achieved = bytes_all_frames / (net_ms * 1e-3) / 1e9 if net_ms > 0 else 0.0
    # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in runs of
    # their own, profiles/*_pmc_traffic.json): counters cannot be read from inside this process
    traffic = traffic_source = None
    try:
        if (n, m) != (480, 640):
            raise LookupError("the committed counter passes were taken at 640x480")
        import glob
        pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
        pmc = json.load(open(pmc_file))
        t = next(v for k_, v in sorted(pmc["bytes_per_launch"].items()) if k_ == args.kernel or k_.startswith(args.kernel + "<false"))
        traffic = t["fetch"] + t["write"]
        traffic_source = {"file": os.path.relpath(pmc_file, ROOT), "source": pmc.get("source"),
                          "note": "counters cannot be read from inside this process: separate rocprofv3 --pmc passes of this same command, committed"}
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": args.kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": int(bytes_per_frame / max(1.0, per_frame_launches)),
                "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_frame": round(per_frame_launches, 2),
                "event_pair_us": round(tot_ms.value / max(1, pairs) * 1e3, 2), "empty_event_pair_us": round(ev_ms.value * 1e3, 2),
                "algorithmic_bytes_per_frame": int(bytes_per_frame), "icp_iterations_per_level": level_iters,
                "icp_iterations_note": "mean executed iterations per level over all profiled frames (levels 0, 1, 2 = full, 1/2, 1/4 resolution)"}

    # ---- PCIe-inclusive rate (never `value`): host-resident frames through RG::LoadFrame, 1.5 MB H2D per frame, pinned ----
    value_with_h2d = None
    if rank == 0 and world == 1 and not args.no_h2d:
        hd = [torch.from_numpy(p[0]).pin_memory() for p in pool]
        hb = [torch.from_numpy(p[1]).pin_memory() for p in pool]
        nh = min(args.steps, 300)

        def run_h2d(r, frames):
            for i in frames:
                host.slamhost_rg_load_frame(r._h, C.c_void_p(hd[i].data_ptr()), C.c_void_p(hb[i].data_ptr()))
                host.slamhost_rg_step_async(r._h)
            r.synchronize()
        # (a) the SAME pipeline object, one stream: upload, preprocessing and the frame's kernels one after the other
        hpos = pos + (0 if args.profile_in_timed_region else args.steps)    # where the device-resident passes stopped
        run_h2d(rg, order[hpos:hpos + 20])
        h0 = time.perf_counter()
        run_h2d(rg, order[hpos + 20:hpos + 20 + nh])
        one_stream = round(nh / (time.perf_counter() - h0), 2)
        # (b) a pipeline of its own with the loads on a second stream (RG overlap mode 1: upload + preprocessing of frame
        # f+1 beside the kernels of frame f, the two streams ordered by stream value operations): what host-fed callers use
        two_streams = 0.0
        under_profiler = any(k.startswith("ROCPROF") for k in os.environ)      # see --no-h2d: (b) can deadlock under counter collection
        load_stream_error = None
        if not under_profiler:
            try:                                     # a secondary figure: a failure here must not take the headline down
                rg2 = pkg.RG(n, m, calib, device=local, stream=stream.cuda_stream)
                rg2.set_overlap_mode(1)
                if args.icp_loop:
                    rg2.set_icp_loop(True)
                rg2.set_finish(True, 30, 0)
                run_h2d(rg2, order[:SETUP])
                run_h2d(rg2, order[SETUP:SETUP + 20])
                h0 = time.perf_counter()
                run_h2d(rg2, order[SETUP + 20:SETUP + 20 + nh])
                two_streams = round(nh / (time.perf_counter() - h0), 2)
                if rg2.finish_stats()["planes"] < 3:
                    two_streams, load_stream_error = 0.0, "plane detection failed in the host-fed pipeline"
                rg2.close()
            except Exception as e:                   # noqa: BLE001
                two_streams, load_stream_error = 0.0, "%s: %s" % (type(e).__name__, e)
        value_with_h2d = {"value": max(one_stream, two_streams), "unit": "frames/s", "frames": nh,
                          "h2d_bytes_per_frame": n * m * 5, "one_stream": one_stream, "load_stream": two_streams,
                          **({"load_stream_error": load_stream_error} if load_stream_error else {}),
                          "note": "the same pipeline fed from pinned host memory (u16 depth + BGR8 uploaded per frame by RG::LoadFrame, "
                                  "Frame.cpp:56-60); poses stay on the device; load_stream = uploads and preprocessing on a second "
                                  "stream (RG overlap mode 1), one_stream = everything in order on one"}

    # ---- CPU baseline: the oracle's display() loop on a bounded sample of the same sequence ----
    cpu = cpu_all = parity = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import oracle as orc
        # parity in the run: the oracle steps over the very frames the GPU was timed on (set-up, warm-up, timed region); every
        # pose it produces is compared with the GPU's history entry of the same frame, bit for bit.  A mismatch is fatal.
        checked = {"frames": 0, "timed": 0, "max_abs": 0.0, "bad": []}

        def check_pose(f, st, R, t):
            k = f - gpu_first
            if gpu_hist is None or k < 0 or k >= len(gpu_hist):
                return
            h = gpu_hist[k]
            d = max(float(np.abs(h[:9] - R).max()), float(np.abs(h[9:12] - t).max()))
            checked["max_abs"] = max(checked["max_abs"], d)
            checked["frames"] += 1
            checked["timed"] += 1 if k >= args.warmup else 0
            if int(h[12]) != (1 if st else 0) or not (np.array_equal(h[:9], R) and np.array_equal(h[9:12], t)):
                checked["bad"].append(f)

        def oracle_pass(frames, check):
            ref = orc.Slam(n, m, calib)
            ref.set_finish(True, 30, 0)
            for i in order[:SETUP]:                  # the same set-up: seed + detection after frame 30
                ref.step(*pool[i])
            if args.finish_period:
                ref.set_finish(True, 30, args.finish_period)
            c0 = time.perf_counter()
            for f in range(SETUP, SETUP + frames):
                st = ref.step(*pool[order[f]])
                if check:
                    R_, t_ = ref.pose()
                    check_pose(f, st, R_, t_)
            cdt_ = time.perf_counter() - c0
            ref.close()
            return cdt_

        cdt = oracle_pass(args.cpu_frames, False)
        cpu = {"value": round(args.cpu_frames / cdt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames of the same %dx%d sequence, oracle display() loop (scalar C, -O2, 1 thread)"
                         % (args.cpu_frames, m, n)}
        # the same sources with OpenMP over rows / blocks on the host cores this job may use.  This pass is
        # long enough to run through the warm-up into the timed region, so it carries the pose comparison (the pose reads cost
        # microseconds per frame; results of the two builds are bit-identical, tests/test_oracle_omp.py)
        orc.use_openmp(True)
        try:
            nf = 3 * args.cpu_frames
            cdt = oracle_pass(nf, True)
            cpu_all = {"value": round(nf / cdt, 3), "unit": "frames/s", "cores": orc.threads(), "kind": "port",
                       "sample": "%d frames of the same sequence, the same oracle sources built with -fopenmp (bit-identical results)" % nf}
        finally:
            orc.use_openmp(False)
        parity = {"frames": checked["frames"], "timed_frames": checked["timed"], "bit_exact": not checked["bad"],
                  "max_abs_pose_diff": checked["max_abs"], "tolerance": 1e-5,
                  "what": "pose history (poseR, poset, tracked flag) of the warm-up + timed frames on the GPU vs the oracle's display() loop "
                          "over the same frames, in this run"}
        if ring_wrapped:
            parity = None
        elif checked["bad"] or checked["frames"] == 0:
            print(json.dumps({"parity_in_run": parity, "first_bad_frames": checked["bad"][:8]}), file=sys.stderr, flush=True)
            raise SystemExit("parity_in_run failed: the GPU pose history differs from the oracle's on %d of %d frames: result invalid"
                             % (len(checked["bad"]), checked["frames"]))

    large = probe = None
    if rank == 0 and world == 1 and not args.no_large_kernels:
        rg.close()
        del depth_dev, bgr_dev
        torch.cuda.empty_cache()
        large, probe = hbm_bound_kernels(pkg, lib, synth, torch, local)

    if rank == 0:
        out = {"metric": "ICP+fusion frames/sec at 640x480", "value": round(world * args.steps / dt, 2), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "TUM fr1/xyz-shaped %dx%d RGB-D sequence (synthetic room corner, 1 mm depth noise), 3-level ICP "
                                      "(<= 3/4/5 iterations) + prediction + merge + Bump fusion into the %d planes of 1024x1024 that RG::Finish detected"
                                      % (m, n, fs["planes"]),
                          "rows": n, "cols": m, "sequences": world, "frames_resident_in_hbm": True,
                          "icp": "one persistent launch per frame (k_icp_loop)" if args.icp_loop else "12 k_icp_iter launches per frame",
                          "finish_period": args.finish_period,
                          "preheat_ms": args.preheat_ms},        # untimed load in front of the warm-up (see --preheat-ms)
               "parity_in_run": parity,
               "parity_in_run_skipped": ("%d frames overwrite the %d-frame pose history ring before it can be read" % (SETUP + args.warmup + args.steps, pkg.RG.HISTORY_CAP)) if ring_wrapped else None,
               "stage_times": stage_times, "value_with_h2d": value_with_h2d, "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all, "hbm_bound_kernels": large, "hbm_stream_probe": probe}
        print(json.dumps(check_line(out)), flush=True)
    rg.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
