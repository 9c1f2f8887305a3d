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

    def __init__(self, ctx, prm, n, m, rank=None, world=None):
        self.ctx, self.prm, self.n, self.m = ctx, prm, n, m
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.sums = torch.zeros(27, dtype=torch.float32, device="cuda")
        self.gathered = torch.zeros(self.world, 27, dtype=torch.float32, device="cuda")

    @staticmethod
    def _div_up(x, y):
        return (x + y - 1) // y if x % y else x // y

    def align(self, vmap, nmap, rgb, gx, gy, vprev, nprev, rgbd_prev, pose_state, status, pitch=None):
        lib, h, prm, n, m = self.ctx.lib, self.ctx._h, self.prm, self.n, self.m
        pitch = pitch or 4 * m
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = lib.rgbdslam_align_begin(h, p(pose_state), p(status))
        if rc:
            raise RuntimeError("align_begin failed: %d" % rc)
        for lvl in range(prm.lvl - 1, -1, -1):
            grid_rows = self._div_up(n // (1 << lvl), 8)
            lo, hi = band_rows(self.rank, self.world, grid_rows)
            for it in range(prm.max_iter[lvl] + 2):
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
                else:
                    dist.all_gather_into_tensor(self.gathered.view(-1), self.sums)
                rc = lib.rgbdslam_align_solve_gathered(h, C.byref(prm), p(self.gathered), self.world, p(pose_state), lvl, it, None)
                if rc:
                    raise RuntimeError("align_solve_gathered failed: %d" % rc)
        rc = lib.rgbdslam_align_end(h, p(pose_state), p(status))
        if rc:
            raise RuntimeError("align_end failed: %d" % rc)
