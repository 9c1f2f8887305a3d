"""Multi-process CPU tests (gloo, world_size 2) of the multi-GPU forms of the path:
the partitioning / collective logic of rgb-d-slam_amd/multigpu.py with the CPU oracle standing in
for the HIP kernels.  On the GPU box the same functions run over RCCL."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIN20 = float(np.sin(np.deg2rad(20.0)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _scene(n, m):
    from oracle import oracle as orc
    synth = importlib.import_module("rgb-d-slam_amd.synth")
    calib = synth.calib_for(n, m)
    d0, c0 = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    R1, t1 = synth.pair_offset_pose()
    d1, c1 = synth.render(R1, t1, n, m, calib)
    v0 = orc.vertex_map_kinect(d0, calib); n0 = orc.compute_normal(v0); rgb0 = orc.cpy_char3(c0)
    v1 = orc.vertex_map_kinect(d1, calib); n1 = orc.compute_normal(v1); rgb1 = orc.cpy_char3(c1)
    rgbd0 = orc.add_zbuff_off(d0, rgb0, calib)
    _, gx, gy = orc.intensity_sobel(rgb1)
    return orc, synth, calib, dict(v0=v0, n0=n0, rgbd0=rgbd0, v1=v1, n1=n1, rgb1=rgb1, gx=gx, gy=gy, d0=d0)


# ---- row-sharded ICP: band sums -> all-gather -> rank-ordered total -> identical solve --------------
def _sharded_icp(rank, world):
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    n, m, fact = 120, 160, 1
    orc, synth, calib, s = _scene(n, m)
    I, z = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    gr, gc = orc.icp_grid(n, m, fact)
    buf, _ = orc.icp_match(calib[:4], I, z, s["v1"], s["n1"], s["rgb1"], s["gx"], s["gy"], s["v0"], s["n0"], s["rgbd0"], 0.05, SIN20, fact)
    lo, hi = mg.band_rows(rank, world, gr)
    local = np.ascontiguousarray(buf.reshape(27, gc, gr)[:, :, lo:hi]).reshape(27, -1)   # band-local blk = brow + bcol*rows
    sums = orc.icp_reduce(local)
    gathered = mg.allgather_sums(torch.from_numpy(sums))
    total = mg.ordered_total(gathered).numpy()
    A = np.zeros((6, 6), np.float32); b = np.zeros(6, np.float32)
    k = 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6: b[i] = total[k]
            else: A[i, j] = A[j, i] = total[k]
            k += 1
    rc, x = orc.llt_solve6(A, b)
    full = orc.icp_reduce(buf)
    return dict(band=(lo, hi), total=total.tobytes(), x=x.tobytes(), rc=rc, full=full, total_f=total, gathered=gathered.numpy())


def test_row_sharded_icp_allgather_two_ranks():
    out = _spawn(_sharded_icp)
    assert out[0]["band"][1] == out[1]["band"][0] and out[0]["band"][0] == 0       # bands tile the block rows
    assert out[0]["total"] == out[1]["total"] and out[0]["x"] == out[1]["x"]        # bit-identical on every rank
    assert out[0]["rc"] == 0
    assert np.array_equal(out[0]["gathered"], out[1]["gathered"])
    # the sharded total differs from the single-GPU two-stage sum only by float32 summation order
    assert np.allclose(out[0]["total_f"], out[0]["full"], rtol=2e-5, atol=1e-6)
    assert np.abs(out[0]["full"]).max() > 1.0


# ---- plane-partitioned fusion: disjoint plane ranges, union == single-process result ------------------
def _plane_fusion(rank, world):
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    n, m, n_prim, nb = 60, 80, 128, 3
    orc, synth, calib, s = _scene(n, m)
    eq, centers = synth.plane_equations()
    pose = np.eye(4, dtype=np.float32).T.reshape(-1)
    label = orc.seed_labels(s["v0"], s["n0"], pose, eq)
    mask = orc.init_mask(s["rgbd0"])
    params = []
    for k in range(nb):
        p = np.zeros(14, np.float32)
        p[0:3], p[3], p[4:6] = eq[k, 0:3], eq[k, 9], 500.0
        p[6] = centers[k] @ eq[k, 3:6] - n_prim * 0.004 / 2
        p[7] = centers[k] @ eq[k, 6:9] - n_prim * 0.004 / 2
        p[8:11], p[11:14] = eq[k, 3:6], eq[k, 6:9]
        params.append(p)
    bumps = [np.zeros((n_prim, n_prim, 3), np.uint16) for _ in range(nb)]
    rgbs = [np.zeros((n_prim, n_prim, 3), np.uint8) for _ in range(nb)]
    masks = [np.zeros((n_prim, n_prim), np.uint8) for _ in range(nb)]
    claims = [np.zeros((n_prim, n_prim), np.uint64) for _ in range(nb)]
    lo, hi = mg.plane_range(rank, world, nb)
    orc.update(bumps, rgbs, masks, params, claims, 1, s["v0"], s["rgbd0"], mask, label, pose, lo, hi)
    # owner gathers: every rank contributes only its planes; others stay untouched (zeros)
    mine = torch.from_numpy(np.stack(masks).astype(np.int32))
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)                   # disjoint support -> the sum is the union
    untouched = all(not masks[k].any() for k in range(nb) if not (lo <= k < hi))
    ref_m = [np.zeros((n_prim, n_prim), np.uint8) for _ in range(nb)]
    ref = ([np.zeros((n_prim, n_prim, 3), np.uint16) for _ in range(nb)], [np.zeros((n_prim, n_prim, 3), np.uint8) for _ in range(nb)],
           ref_m, params, [np.zeros((n_prim, n_prim), np.uint64) for _ in range(nb)])
    orc.update(*ref, 1, s["v0"], s["rgbd0"], mask, label, pose)
    return dict(range=(lo, hi), untouched=untouched, union_ok=bool(np.array_equal(mine.numpy(), np.stack(ref_m).astype(np.int32))),
                filled=int((np.stack(ref_m) > 10).sum()))


def test_plane_partitioned_fusion_two_ranks():
    out = _spawn(_plane_fusion)
    assert out[0]["range"] == (0, 2) and out[1]["range"] == (2, 3)
    assert all(o["untouched"] and o["union_ok"] for o in out)
    assert out[0]["filled"] > 200


# ---- bench aggregation (independent sequences, configs[4]) --------------------------------------------
def _aggregate(rank, world):
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    return mg.aggregate_throughput(1.0 + rank, 100 + rank)


def test_aggregate_throughput_max_time_total_units():
    out = _spawn(_aggregate)
    assert out[0] == out[1] == (201, 2.0)


def test_partitions_cover_everything():
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    for nb, world in ((256, 8), (3, 2), (5, 8), (135, 4), (60, 7)):
        spans = [mg.plane_range(r, world, nb) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == nb
        assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


# ---- plane-partitioned step with its one collective: the packed frame broadcast (16-bit labels, tiled model) ----------
def _plane_step_broadcast(rank, world):
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    n, m, N, nb = 60, 80, 64, 12
    orc, synth, calib, s = _scene(n, m)
    params = synth.tiled_planes(nb, N)
    pose = np.eye(4, dtype=np.float32).T.reshape(-1)
    pack = mg.FramePack(n, m, device="cpu", thin=False)
    hist = np.zeros(16, np.float32)
    hist[:9] = np.eye(3, dtype=np.float32).reshape(-1); hist[12] = 1.0
    label = orc.label_planes_l16(s["v0"], s["n0"], pose, list(params), N, N)
    mask = orc.init_mask(s["rgbd0"])
    if rank == 0:                                       # the tracking rank fills the pack; the others receive it
        pack.view("vmap").copy_(torch.from_numpy(s["v0"]))
        pack.view("rgbd").copy_(torch.from_numpy(s["rgbd0"]))
        pack.view("hist").copy_(torch.from_numpy(hist))
        pack.view("mask").copy_(torch.from_numpy(mask))
        pack.view("label").copy_(torch.from_numpy(label.view(np.int16)))
    pack.broadcast(0)
    got = dict(v=pack.view("vmap").numpy(), rgbd=pack.view("rgbd").numpy(), mask=pack.view("mask").numpy(),
               label=pack.view("label").numpy().view(np.uint16), hist=pack.view("hist").numpy())
    same_frame = (np.array_equal(got["v"], s["v0"]) and np.array_equal(got["rgbd"], s["rgbd0"]) and np.array_equal(got["mask"], mask)
                  and np.array_equal(got["label"], label) and np.array_equal(got["hist"], hist))
    lo, hi = mg.plane_range(rank, world, nb)
    mk = lambda shape, dt, v=0: [np.full(shape, v, dt) for _ in range(nb)]
    mine = (mk((N, N, 3), np.uint16), mk((N, N, 3), np.uint8), mk((N, N), np.uint8, 10))
    orc.update_l16(mine[0], mine[1], mine[2], list(params), mk((N, N), np.uint64), 1, got["v"], got["rgbd"], got["mask"], got["label"], pose, lo, hi)
    ref = (mk((N, N, 3), np.uint16), mk((N, N, 3), np.uint8), mk((N, N), np.uint8, 10))
    orc.update_l16(ref[0], ref[1], ref[2], list(params), mk((N, N), np.uint64), 1, s["v0"], s["rgbd0"], mask, label, pose)
    owned_ok = all(np.array_equal(mine[c][k], ref[c][k]) for c in range(3) for k in range(lo, hi))
    others_untouched = all((mine[2][k] == 10).all() for k in range(nb) if not lo <= k < hi)
    return dict(same_frame=same_frame, owned_ok=owned_ok, others_untouched=others_untouched, range=(lo, hi),
                labels=sorted(set(np.unique(label).tolist())), filled=int(sum((ref[2][k] > 10).sum() for k in range(nb))),
                nbytes=pack.nbytes)


def test_plane_partitioned_step_with_frame_broadcast_two_ranks():
    out = _spawn(_plane_step_broadcast)
    assert out[0]["range"] == (0, 6) and out[1]["range"] == (6, 12)
    assert all(o["same_frame"] and o["owned_ok"] and o["others_untouched"] for o in out)
    assert len(out[0]["labels"]) >= 4 and out[0]["filled"] > 300
    assert out[0]["nbytes"] >= 30 * 60 * 80 + 64


# ---- prediction from the planes: z-buffers of disjoint plane ranges merge by an unsigned 64-bit minimum ---------------------
def _zmerge(rank, world):
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    n, m, N, nb = 60, 80, 64, 6
    orc, synth, calib, s = _scene(n, m)
    params = synth.tiled_planes(nb, N, res=0.04)        # 4 cm cells: at 60 x 80 a pixel spans ~3 cm, so neighbouring cells fill
    pose = np.eye(4, dtype=np.float32).T.reshape(-1)
    label = orc.label_planes_l16(s["v0"], s["n0"], pose, list(params), N, N)
    mk = lambda shape, dt, v=0: [np.full(shape, v, dt) for _ in range(nb)]
    b, c, k, cl = mk((N, N, 3), np.uint16), mk((N, N, 3), np.uint8), mk((N, N), np.uint8, 10), mk((N, N), np.uint64)
    orc.update_l16(b, c, k, list(params), cl, 1, s["v0"], s["rgbd0"], np.full((n, m), 12, np.uint8), label, pose)
    full = np.full((n, m), 0xFFFFFFFFFFFFFFFF, np.uint64)
    orc.planes_draw(full, b, c, k, list(params), 0, nb, pose, calib, n, m, min_conf=1)
    lo, hi = mg.plane_range(rank, world, nb)
    mine = np.full((n, m), 0xFFFFFFFFFFFFFFFF, np.uint64)
    orc.planes_draw(mine, b, c, k, list(params), lo, hi, pose, calib, n, m, min_conf=1)
    partial = bool(not np.array_equal(mine, full))
    merged = mg.allreduce_min_u64(torch.from_numpy(mine.view(np.int64))).numpy().view(np.uint64)      # in place
    return dict(ok=bool(np.array_equal(merged, full)), covered=float((full != 0xFFFFFFFFFFFFFFFF).mean()), partial=partial)


def test_plane_partitioned_prediction_zbuffer_merge_two_ranks():
    out = _spawn(_zmerge)
    assert all(o["ok"] for o in out) and out[0]["covered"] > 0.05
    assert any(o["partial"] for o in out)               # at least one rank alone did not have the whole picture


# ---- the same merge among the plane OWNERS only: a tracking rank without planes (rank 0, from four ranks on) stays out ------
def _zmerge_owners(rank, world):
    import torch.distributed as dist
    mg = importlib.import_module("rgb-d-slam_amd.multigpu")
    owners = dist.new_group(ranks=list(range(1, world)))            # every rank makes the call; rank 0 is not a member
    n, m, N, nb = 60, 80, 64, 6
    orc, synth, calib, s = _scene(n, m)
    params = synth.tiled_planes(nb, N, res=0.04)
    pose = np.eye(4, dtype=np.float32).T.reshape(-1)
    label = orc.label_planes_l16(s["v0"], s["n0"], pose, list(params), N, N)
    mk = lambda shape, dt, v=0: [np.full(shape, v, dt) for _ in range(nb)]
    b, c, k, cl = mk((N, N, 3), np.uint16), mk((N, N, 3), np.uint8), mk((N, N), np.uint8, 10), mk((N, N), np.uint64)
    orc.update_l16(b, c, k, list(params), cl, 1, s["v0"], s["rgbd0"], np.full((n, m), 12, np.uint8), label, pose)
    full = np.full((n, m), 0xFFFFFFFFFFFFFFFF, np.uint64)
    orc.planes_draw(full, b, c, k, list(params), 0, nb, pose, calib, n, m, min_conf=1)
    if rank == 0:
        return dict(ok=True, tracker=True)                          # no draw, no collective
    lo, hi = mg.plane_range(rank - 1, world - 1, nb)                # PlaneShard(first_worker=1)
    mine = np.full((n, m), 0xFFFFFFFFFFFFFFFF, np.uint64)
    orc.planes_draw(mine, b, c, k, list(params), lo, hi, pose, calib, n, m, min_conf=1)
    merged = mg.allreduce_min_u64(torch.from_numpy(mine.view(np.int64)), group=owners).numpy().view(np.uint64)
    return dict(ok=bool(np.array_equal(merged, full)), tracker=False, range=(lo, hi))


def test_prediction_merge_among_plane_owners_three_ranks():
    out = _spawn(_zmerge_owners, world=3)
    assert out[0]["tracker"] and all(o["ok"] for o in out)
    assert out[1]["range"] == (0, 3) and out[2]["range"] == (3, 6)
