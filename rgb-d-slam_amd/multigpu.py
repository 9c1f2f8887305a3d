"""Multi-GPU forms of the hot path (the reference itself is single-GPU).

One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests).  Only two stages have a real exchange step:

* plane-partitioned Bump fusion / decode (BASELINE configs[2]): planes are independent units -- every
  rank fuses and decodes only the planes in `plane_range(rank, world, nb)`; the frame maps are the
  only shared input (broadcast by the tracking rank), no collective inside the fusion itself.
* row-sharded ICP (configs[3]): each rank owns a band of 8x8-block rows of the level's block grid
  and produces 27 partial sums per Gauss-Newton iteration; `allgather_sums` + `ordered_total` give
  every rank the same float32 total (summed in rank order, so all ranks hold bit-identical poses),
  followed by the redundant 6x6 solve.  The message is 108 bytes: pure latency, which is why it is an
  all-gather of 27 floats and not a ring all-reduce of anything bigger.

Independent sequences (configs[4], `bench.py --gpus N`) need no exchange at all; `aggregate_throughput`
is the max-over-ranks timing the bench contract asks for.
"""
import ctypes as C

import torch
import torch.distributed as dist


def plane_range(rank, world, nb_planes):
    """Contiguous block partition of plane indices: the first nb % world ranks own one plane more."""
    base, rem = divmod(nb_planes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def band_rows(rank, world, grid_rows):
    """Block-row band [lo, hi) of a level's 8x8-block grid (divUp(n / 2^lvl, 8) rows) owned by `rank`."""
    return plane_range(rank, world, grid_rows)


def allgather_sums(local27):
    """[27] float32 tensor (device of the backend) -> [world, 27], row r = rank r's partial sums."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    out = torch.empty(world, 27, dtype=torch.float32, device=local27.device)
    if world == 1:
        out[0].copy_(local27)
    else:
        dist.all_gather_into_tensor(out.view(-1), local27.contiguous())
    return out


def ordered_total(gathered):
    """float32 sum over ranks IN RANK ORDER (the order the device solve uses): identical on every rank."""
    total = gathered[0].clone()
    for r in range(1, gathered.shape[0]):
        total = total + gathered[r]
    return total


def aggregate_throughput(local_seconds, local_units, device="cpu"):
    """bench contract: units of ALL ranks / MAX time over ranks.  Returns (total_units, max_seconds)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_units, local_seconds
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([local_units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return int(u.item()), float(t.item())


class ShardedAligner:
    """Row-sharded RG::AlignFrameGaussNewton (D1/RG.cpp:797-932) on the HIP library.

    Every rank holds the full current-frame and model maps (the projective gather is unbounded) and
    the same pose state; rank r computes the normal equations of its band of model-map block rows.
    The context must have been created on torch's current stream: the RCCL all-gather is enqueued by
    torch on its own stream with event dependencies on that stream, so the whole loop stays
    asynchronous -- no host sync between iterations.
    """

    def __init__(self, ctx, prm, n, m, rank=None, world=None, exchange="allgather"):
        self.ctx, self.prm, self.n, self.m = ctx, prm, n, m
        self.exchange = exchange
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.sums = torch.zeros(27, dtype=torch.float32, device="cuda")
        self.gathered = torch.zeros(self.world, 27, dtype=torch.float32, device="cuda")
        if exchange in ("peer", "peer_loop"):
            self._connect_peers()
        if exchange == "peer_loop":
            # ONE persistent launch per alignment and rank (rgbdslam_track_frame_sharded): the state lives in a device-resident
            # rgbdslam_pose_block (pose_state 24 f32, status 4 i32, three 4x4 poses); align() takes views of it
            ctx.set_icp_loop(True)                   # raises unless this process has the GPU to itself
            self.pb = torch.zeros(76, dtype=torch.float32, device="cuda")
            self.history = torch.zeros(16, dtype=torch.float32, device="cuda")
            self.bands = {}

    def pose_block_views(self):
        """(pose_state [24] f32, status [4] i32): views of the device pose block of the peer_loop form"""
        return self.pb[:24], self.pb[24:28].view(torch.int32)

    HANDLE_BYTES = 64   # RGBDSLAM_EXCHANGE_HANDLE_BYTES

    def _connect_peers(self):
        """Peer-mapped exchange blocks: every rank exports an IPC handle of its block, the handles are all-gathered once
        (through torch.distributed, whatever the backend), and every rank maps its peers' blocks.  After that the
        per-iteration exchange is stores over xGMI from inside the iteration kernel -- no collective call, no host."""
        lib, h = self.ctx.lib, self.ctx._h
        mine = (C.c_ubyte * self.HANDLE_BYTES)()
        rc = lib.rgbdslam_exchange_create(h, self.world, self.rank, mine)
        if rc:
            raise RuntimeError("exchange_create failed: %d" % rc)
        everyone = (C.c_ubyte * (self.HANDLE_BYTES * self.world))()
        if self.world > 1:
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
            out = torch.empty(self.world * self.HANDLE_BYTES, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(out, t)
            for k, v in enumerate(out.cpu().tolist()):
                everyone[k] = v
        rc = lib.rgbdslam_exchange_connect(h, everyone)
        if rc:
            raise RuntimeError("exchange_connect failed: %d" % rc)
        if self.world > 1:
            dist.barrier()          # nobody starts publishing before every mapping exists

    @staticmethod
    def _div_up(x, y):
        return (x + y - 1) // y if x % y else x // y

    def align(self, vmap, nmap, rgb, gx, gy, vprev, nprev, rgbd_prev, pose_state, status, pitch=None, aux=None):
        """aux: the current frame's auxiliary ICP records (Context.frame_aux), used by the persistent form (peer_loop) only."""
        lib, h, prm, n, m = self.ctx.lib, self.ctx._h, self.prm, self.n, self.m
        pitch = pitch or 4 * m
        p = lambda t: C.c_void_p(t.data_ptr())
        if self.exchange == "peer_loop":
            if pose_state.data_ptr() != self.pb.data_ptr() or status.data_ptr() != self.pb.data_ptr() + 96:
                raise ValueError("peer_loop: pose_state / status must be the views of pose_block_views()")
            lo3, hi3 = (C.c_int * 3)(), (C.c_int * 3)()
            for lvl in range(prm.lvl):
                lo3[lvl], hi3[lvl] = band_rows(self.rank, self.world, self._div_up(n // (1 << lvl), 8))
                if hi3[lvl] <= lo3[lvl]:
                    raise RuntimeError("peer exchange needs a non-empty band on every rank (level %d)" % lvl)
            if aux is not None and lib.rgbdslam_set_frame_aux(h, None, p(aux)) != 0:
                raise RuntimeError("set_frame_aux failed")
            rc = lib.rgbdslam_track_frame_sharded(h, C.byref(prm), p(vmap), p(nmap), p(rgb), p(gx), C.c_size_t(pitch), p(gy),
                                                  C.c_size_t(pitch), p(vprev), p(nprev), p(rgbd_prev), n, m, p(self.pb),
                                                  p(self.history), 0, lo3, hi3)
            if rc:
                raise RuntimeError("track_frame_sharded failed: %d" % rc)
            return
        rc = lib.rgbdslam_align_begin(h, p(pose_state), p(status))
        if rc:
            raise RuntimeError("align_begin failed: %d" % rc)
        for lvl in range(prm.lvl - 1, -1, -1):
            grid_rows = self._div_up(n // (1 << lvl), 8)
            lo, hi = band_rows(self.rank, self.world, grid_rows)
            for it in range(prm.max_iter[lvl] + 2):
                if self.exchange == "peer":
                    if hi <= lo:
                        raise RuntimeError("peer exchange needs a non-empty band on every rank (level %d has %d block rows)" % (lvl, grid_rows))
                    rc = lib.rgbdslam_align_band_exchange(h, C.byref(prm), p(vmap), p(nmap), p(rgb), p(gx), C.c_size_t(pitch), p(gy),
                                                          C.c_size_t(pitch), p(vprev), p(nprev), p(rgbd_prev), n, m, lvl, it, lo, hi,
                                                          p(pose_state), None)
                    if rc:
                        raise RuntimeError("align_band_exchange failed: %d" % rc)
                    continue
                if hi > lo:
                    rc = lib.rgbdslam_align_band_system(h, C.byref(prm), p(vmap), p(nmap), p(rgb), p(gx), C.c_size_t(pitch), p(gy),
                                                        C.c_size_t(pitch), p(vprev), p(nprev), p(rgbd_prev), n, m, lvl, it, lo, hi,
                                                        p(self.sums))
                    if rc:
                        raise RuntimeError("align_band_system failed: %d" % rc)
                else:
                    self.sums.zero_()                       # more ranks than block rows: an empty band contributes zeros
                if self.world == 1:
                    self.gathered[0].copy_(self.sums)
                elif dist.get_backend() == "gloo":              # CPU rehearsal of N ranks on one card: staged through the host
                    host = torch.empty(self.world * 27, dtype=torch.float32)
                    dist.all_gather_into_tensor(host, self.sums.cpu())
                    self.gathered.view(-1).copy_(host)
                else:
                    dist.all_gather_into_tensor(self.gathered.view(-1), self.sums)
                rc = lib.rgbdslam_align_solve_gathered(h, C.byref(prm), p(self.gathered), self.world, p(pose_state), lvl, it, None)
                if rc:
                    raise RuntimeError("align_solve_gathered failed: %d" % rc)
        rc = lib.rgbdslam_align_end(h, p(pose_state), p(status))
        if rc:
            raise RuntimeError("align_end failed: %d" % rc)


# ---- plane-partitioned fusion + decode (BASELINE configs[2]) -----------------------------------------------------------
class FramePack:
    """What the plane owners need of one tracked frame, as ONE contiguous buffer so that the step's only collective is one
    broadcast.
    thin (default): only what cannot be derived -- the merged model depth RGBD.w (f32), the frame's colour bytes (u8 x 3),
    Mask (u8), the 16-bit label image and the frame's 64-byte history entry (pose + gate): 10 B/px = 3.1 MB at 640x480.  After
    Merge the model VMap is VertexMapGL's function of RGBD.w (D1/Kernel.cu:54-72) and RGBD.rgb is the frame's byte / 255
    (D1/Kernel.cu:463-465), so every rank rebuilds both with the reference's own operations: the same bits
    (rgbdslam_update_from_history_thin_l16).
    fat (thin=False, rounds 1-2): VMap (f32x3) and RGBD (f32x4) themselves: 31 B/px = 9.5 MB."""

    def __init__(self, n, m, device="cuda", thin=True):
        P = n * m
        up = lambda x: (x + 255) // 256 * 256
        self.n, self.m, self.thin = n, m, thin
        if thin:
            self.off_w = 0
            self.off_rgb8 = up(4 * P)
            self.off_hist = up(self.off_rgb8 + 3 * P)
        else:
            self.off_vmap = 0
            self.off_rgbd = up(12 * P)
            self.off_hist = up(self.off_rgbd + 16 * P)
        self.off_mask = self.off_hist + 256
        self.off_label = up(self.off_mask + P)
        self.nbytes = up(self.off_label + 2 * P)
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)

    def ptr(self, off):
        return self.buf.data_ptr() + off

    def view(self, name):
        P = self.n * self.m
        if name == "vmap":
            return self.buf[self.off_vmap:self.off_vmap + 12 * P].view(torch.float32).view(self.n, self.m, 3)
        if name == "rgbd":
            return self.buf[self.off_rgbd:self.off_rgbd + 16 * P].view(torch.float32).view(self.n, self.m, 4)
        if name == "w":
            return self.buf[self.off_w:self.off_w + 4 * P].view(torch.float32).view(self.n, self.m)
        if name == "rgb8":
            return self.buf[self.off_rgb8:self.off_rgb8 + 3 * P].view(self.n, self.m, 3)
        if name == "hist":
            return self.buf[self.off_hist:self.off_hist + 64].view(torch.float32)
        if name == "mask":
            return self.buf[self.off_mask:self.off_mask + P].view(self.n, self.m)
        if name == "label":
            return self.buf[self.off_label:self.off_label + 2 * P].view(torch.int16).view(self.n, self.m)
        raise KeyError(name)

    def broadcast(self, src=0):
        """The one collective of the plane-partitioned step (RCCL broadcast over xGMI; gloo in the CPU tests)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            if self.buf.is_cuda and dist.get_backend() == "gloo":    # CPU rehearsal of N ranks on one card: staged through the host
                host = self.buf.cpu()
                dist.broadcast(host, src)
                self.buf.copy_(host)
            else:
                dist.broadcast(self.buf, src)


class PlaneShard:
    """The planes [lo, hi) of a model of `nb` Bump images that this rank owns: their slabs, claim words and decode
    outputs live here and nowhere else (the table entries of the other planes stay null; the kernels never touch a
    plane outside [plane_lo, plane_hi))."""

    def __init__(self, ctx, params, N, rank, world, first_worker=0):
        """first_worker = 1: rank 0 (the tracking rank) owns no planes -- it still takes part in every collective.  From about four
        ranks on the tracker's own frame time, not the fusion, bounds the step, and a tracker that also decodes a share of the
        model is the straggler."""
        self.ctx, self.N = ctx, N
        self.nb = int(params.shape[0])
        if first_worker < 0 or first_worker >= world:
            raise ValueError("first_worker %d with %d ranks" % (first_worker, world))
        self.lo, self.hi = plane_range(rank - first_worker, world - first_worker, self.nb) if rank >= first_worker else (0, 0)
        k = self.hi - self.lo
        self.param = torch.as_tensor(params, dtype=torch.float32).cuda().contiguous()   # every rank holds all equations (labels)
        self.bump = torch.zeros(k, N, N, 3, dtype=torch.int16, device="cuda")
        self.rgb = torch.zeros(k, N, N, 3, dtype=torch.uint8, device="cuda")
        self.mask = torch.full((k, N, N), 10, dtype=torch.uint8, device="cuda")
        self.claim = torch.zeros(k, N, N, dtype=torch.int64, device="cuda")
        self.vout = torch.zeros(k, N, N, 3, device="cuda")
        self.cout = torch.zeros(k, N, N, 4, device="cuda")
        self.kout = torch.zeros(k, N, N, 4, device="cuda")
        tab = lambda t: torch.tensor([0] * self.lo + [t[i].data_ptr() for i in range(k)] + [0] * (self.nb - self.hi),
                                     dtype=torch.int64, device="cuda")
        self.t_bump, self.t_rgb, self.t_mask, self.t_claim = tab(self.bump), tab(self.rgb), tab(self.mask), tab(self.claim)
        self.t_vout, self.t_cout, self.t_kout = tab(self.vout), tab(self.cout), tab(self.kout)
        self.t_param = torch.tensor([self.param[i].data_ptr() for i in range(self.nb)], dtype=torch.int64, device="cuda")
        self.stamp = 0

    def label(self, pack, vmap_ptr, nmap_ptr, eps=0.1, alpha=0.8, ctx=None):
        """Tracking rank only: 16-bit labels of the model frame by plane membership, posed by the pack's history entry.
        ctx: the context (stream) to run on -- the tracker's own in the pipelined step."""
        (ctx or self.ctx).label_planes_l16(pack.ptr(pack.off_label), vmap_ptr, nmap_ptr, None, pack.ptr(pack.off_hist), self.t_param,
                                           self.nb, self.N, self.N, pack.n, pack.m, eps, alpha)

    def fuse_and_decode(self, pack):
        """Graph::Update (claim + apply) of the owned planes from the broadcast frame, then VertexFromBump of the owned planes."""
        self.fuse(pack)
        self.decode()

    def decode(self):
        if self.hi > self.lo:
            self.ctx.vertex_from_bump_batch(self.t_vout, self.t_cout, self.t_kout, self.t_bump, self.t_rgb, self.t_mask, self.t_param,
                                            self.lo, self.hi, self.N, self.N)

    def fuse(self, pack):
        self.stamp += 1
        if self.hi > self.lo:
            if pack.thin:
                self.ctx.update_from_history_thin_l16(pack.ptr(pack.off_hist), self.t_bump, self.t_rgb, self.t_mask, self.t_param,
                                                      self.t_claim, self.stamp, pack.ptr(pack.off_w), pack.ptr(pack.off_rgb8),
                                                      pack.ptr(pack.off_mask), pack.ptr(pack.off_label), self.N, self.N, pack.n, pack.m,
                                                      self.lo, self.hi)
            else:
                self.ctx.update_from_history_l16(pack.ptr(pack.off_hist), self.t_bump, self.t_rgb, self.t_mask, self.t_param,
                                                 self.t_claim, self.stamp, pack.ptr(pack.off_vmap), pack.ptr(pack.off_rgbd),
                                                 pack.ptr(pack.off_mask), pack.ptr(pack.off_label), self.N, self.N, pack.n, pack.m,
                                                 self.lo, self.hi)

    def predict(self, pose_w2c16, n, m, min_conf=1, ctx=None, group=None):
        """Prediction of the view `pose_w2c16` (world-to-camera, column-major 4x4) from the planes, partitioned like the
        fusion: this rank draws the planes it owns into its own u64 z-buffer, ONE all-reduce(MIN) merges the ranks'
        buffers (the key is depth bits << 32 | ~triangle id: the nearest surface wins whoever drew it), and the rank
        resolves the pixels its planes won into pred_v / pred_c (the other pixels stay untouched: the ranks' outputs are
        disjoint)."""
        ctx = ctx or self.ctx                            # (a context on another stream: the prediction beside the decode)
        if getattr(self, "zbuf", None) is None or self.zbuf.shape != (n, m):
            self.zbuf = torch.full((n, m), -1, dtype=torch.int64, device="cuda")       # all ones = empty
            self.pred_v = torch.zeros(n, m, 3, device="cuda")
            self.pred_c = torch.zeros(n, m, 4, device="cuda")
        if self.hi > self.lo:
            ctx.planes_draw(self.zbuf, self.t_bump, self.t_rgb, self.t_mask, self.t_param, self.lo, self.hi, pose_w2c16,
                                 self.N, self.N, n, m, min_conf=min_conf)
        allreduce_min_u64(self.zbuf, group=group)    # group: the plane owners (a tracking rank without planes stays out of it)
        self.pred_v.zero_(); self.pred_c.zero_()          # pixels other ranks' planes won stay zero here: the outputs are disjoint
        if self.hi > self.lo:
            ctx.planes_resolve(self.zbuf, self.pred_v, self.pred_c, self.t_bump, self.t_rgb, self.t_mask, self.t_param,
                                    self.lo, self.hi, pose_w2c16, self.N, self.N, n, m, reset=False)
        # every rank holds the MERGED buffer: the resolve only clears the pixels this rank's planes won, the rest is cleared here
        self.zbuf.fill_(-1)


_SIGN64 = -(1 << 63)


def allreduce_min_u64(zbuf, group=None):
    """Merge of the z-buffers that the ranks drew from their own planes (rgbdslam_planes_draw): elementwise UNSIGNED 64-bit
    minimum over all ranks, in place.  The key is depth bits << 32 | ~triangle id, so the minimum is the nearest surface
    whichever rank drew it.  torch has no u64 reductions: the sign bit is flipped (monotonic u64 -> i64), MIN, flipped back."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return zbuf
    z = zbuf.view(torch.int64)
    z ^= _SIGN64
    if z.is_cuda and dist.get_backend() == "gloo":
        h = z.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
        z.copy_(h)
    else:
        dist.all_reduce(z, op=dist.ReduceOp.MIN, group=group)
    z ^= _SIGN64
    return zbuf
