"""SURVEY 8f N4 -- Bump image -> mesh on the GPU: QuadTrim (D1/Primitive.cu:95-152, D2 variant
D2/Primitive.cu:367-424), normals of the decoded vertices (D2/Primitive.cu:181-365) and the fused
whole-model pass, bit-exact against the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_gpu_kernels import _plane_setup, _ptr_table, ctx, dev, host, pose16, same  # noqa: E402,F401


def fused_planes(orc, pair, synth, N_prim, rounds=2, seed=11):
    """Bump images of the three synthetic planes, filled by the oracle's fusion from the frame-0 maps."""
    n, m = pair["v0"].shape[:2]
    eq, params = _plane_setup(synth, N_prim)
    I16 = pose16(np.eye(3), np.zeros(3))
    label = orc.seed_labels(pair["v0"], pair["n0"], I16, eq)
    rng = np.random.default_rng(seed)
    bumps = [np.zeros((N_prim, N_prim, 3), np.uint16) for _ in range(3)]
    rgbs = [np.zeros((N_prim, N_prim, 3), np.uint8) for _ in range(3)]
    masks = [np.zeros((N_prim, N_prim), np.uint8) for _ in range(3)]
    claims = [np.zeros((N_prim, N_prim), np.uint64) for _ in range(3)]
    for stamp in range(1, rounds + 1):
        mk = rng.integers(10, 30, (n, m)).astype(np.uint8)
        orc.update(bumps, rgbs, masks, params, claims, stamp, pair["v0"], pair["rgbd0"], mk, label, I16)
    masks[0][40:60, 40:60] = 1            # the "visible but unobserved" marker the D1 rules exempt
    return bumps, rgbs, masks, params


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("lvl,min_conf,thresh", [(1, 0, 0.1), (1, 5, 0.02), (2, 2, 0.1), (3, 0, 0.05)])
def test_quad_trim(ctx, orc, pair, synth, variant, lvl, min_conf, thresh):
    N = 512
    bumps, rgbs, masks, params = fused_planes(orc, pair, synth, N)
    kept_total = 0
    for k in range(3):
        v, _, _ = orc.vertex_from_bump(bumps[k], rgbs[k], masks[k], params[k], lvl)
        ref = orc.quad_trim(v, masks[k], min_conf, thresh, lvl, variant)
        g_idx = torch.full((ref.shape[0], 4), 0xFFFFFFFF, dtype=torch.int64, device="cuda").to(torch.int32)
        ctx.quad_trim(dev(v), dev(masks[k]), g_idx, N, N, min_conf, thresh, lvl, variant)
        same(host(g_idx).view(np.uint32), ref)
        kept_total += int((ref.sum(1) > 0).sum())
    assert kept_total > 1000 or (lvl > 1)   # a real mesh survives at full resolution


def test_quad_trim_rules_known_answer(ctx, orc):
    """3x4 vertex grid, hand-derived: which of the 6 quads survive under the two corner rules."""
    n, m = 3, 4
    v = np.zeros((n, m, 3), np.float32)
    for i in range(n):
        for j in range(m):
            v[i, j] = (0.01 * i, 0.01 * j, 1.0)
    v[2, 3, 2] = 1.5                                      # a depth step: quad (1,2) has a long edge
    mask = np.array([[20, 20, 1, 12], [20, 11, 20, 20], [20, 20, 20, 20]], np.uint8)
    out = {}
    for variant in (0, 1):
        g_idx = torch.zeros(6, 4, dtype=torch.int32, device="cuda")
        ctx.quad_trim(dev(v), dev(mask), g_idx, n, m, 2, 0.1, 1, variant)
        out[variant] = host(g_idx).view(np.uint32)
        same(out[variant], orc.quad_trim(v, mask, 2, 0.1, 1, variant))
    # D1, min_conf 2: corners need mask == 1 or mask >= 12.  mask 11 (cell 1,1) kills quads (0,0),(0,1),(1,0),(1,1);
    # quad (0,2) keeps (corners 1, 12, 20, 20); quad (1,2) falls to the depth step.
    assert [bool(r.any()) for r in out[0]] == [False, False, True, False, False, False]
    assert list(out[0][2]) == [2, 3, 7, 6]
    # D2: corners need mask >= 11: the marker value 1 now kills (0,1) and (0,2); 11 is fine -> (0,0), (1,0), (1,1) keep
    assert [bool(r.any()) for r in out[1]] == [True, False, False, True, True, False]
    assert list(out[1][0]) == [0, 4, 5, 1]               # reversed winding 0,3,2,1


@pytest.mark.parametrize("variant,inverse,min_conf", [(0, False, 0), (1, True, 3)])
def test_bump_mesh_fused_equals_three_passes(ctx, orc, pair, synth, variant, inverse, min_conf):
    N = 512
    bumps, rgbs, masks, params = fused_planes(orc, pair, synth, N)
    g_b, g_c, g_m, g_p = [dev(x) for x in bumps], [dev(x) for x in rgbs], [dev(x) for x in masks], [dev(p) for p in params]
    outs = [[torch.full((N, N, ch), -7.0, device="cuda") for _ in range(3)] for ch in (3, 4, 4, 3)]
    idx = [torch.full(((N - 1) * (N - 1), 4), -1, dtype=torch.int32, device="cuda") for _ in range(3)]
    tabs = [_ptr_table(x) for x in outs] + [_ptr_table(idx)] + [_ptr_table(x) for x in (g_b, g_c, g_m, g_p)]
    ctx.bump_mesh_batch(*tabs, 1, 3, N, N, min_conf, 0.05, inverse, variant)
    assert float(outs[0][0].max()) == -7.0 and int(idx[0].max()) == -1          # plane 0 is outside [1, 3)
    for k in (1, 2):
        rv, rc, rk = orc.vertex_from_bump(bumps[k], rgbs[k], masks[k], params[k], 1)
        same(host(outs[0][k]), rv)
        same(host(outs[1][k]), rc)
        same(host(outs[2][k]), rk)
        same(host(outs[3][k]), orc.compute_normal(rv, inverse))
        ref_idx = orc.quad_trim(rv, masks[k], min_conf, 0.05, 1, variant)
        same(host(idx[k]).view(np.uint32), ref_idx)
        assert (ref_idx.sum(1) > 0).sum() > 1000
        nm = host(outs[3][k])
        lens = np.linalg.norm(nm[(nm != 0).any(2)], axis=1)
        assert len(lens) > 1000 and np.abs(lens - 1).max() < 1e-5
    # the per-operator path gives the same normals (ComputeNormal on the decoded vertices)
    g_n = torch.empty(N, N, 3, device="cuda")
    ctx.compute_normal(g_n, outs[0][1], N, N, inverse)
    same(host(g_n), host(outs[3][1]))


@pytest.mark.parametrize("n,m,skew", [(40, 200, 0), (48, 256, 0), (48, 256, 1), (33, 132, 1)])
def test_bump_mesh_wide_and_narrow_paths_mixed(ctx, orc, n, m, skew):
    """k_bump_mesh takes its inputs and writes its vertices / normals full width where a tile allows it (m % 4 == 0, a full
    64-cell tile column, 16 full rows for the inputs, aligned planes) and narrow elsewhere: sizes with full AND partial
    tiles in one launch, and planes that start 4 (outputs) / 1 (inputs) bytes off alignment (skew), against the oracle."""
    rng = np.random.default_rng(n * 1000 + m + skew)
    bump = rng.integers(20000, 40000, (n, m, 3)).astype(np.uint16)
    rgbb = rng.integers(0, 256, (n, m, 3)).astype(np.uint8)
    mask = rng.choice(np.array([0, 1, 10, 11, 12, 30, 255], np.uint8), (n, m))
    param = np.array([0.0, 0.0, 1.0, 1.2, 500.0, 500.0, -0.1, -0.2, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0], np.float32)

    def dev_skewed(a):                                   # the same data, 2 * skew bytes (u16) / skew bytes (u8) into a larger buffer
        flat = torch.zeros(a.size + 8, dtype=torch.from_numpy(a).dtype, device="cuda")
        flat[skew:skew + a.size] = torch.from_numpy(a).reshape(-1).cuda()
        return flat[skew:skew + a.size].view(*a.shape)

    def out_skewed(ch):
        flat = torch.empty(n * m * ch + 8, device="cuda")
        return flat[skew:skew + n * m * ch].view(n, m, ch)

    g = [dev_skewed(bump), dev_skewed(rgbb), dev_skewed(mask), dev(param)]
    outs = [[out_skewed(ch)] for ch in (3, 4, 4, 3)]
    if skew:
        assert outs[0][0].data_ptr() % 16 == 4 and g[1].data_ptr() % 4 == 1
    idx = [torch.empty((n - 1) * (m - 1), 4, dtype=torch.int32, device="cuda")]
    # RGBA / MaskOut / Index are written as 16-byte words by every path: those planes stay aligned
    outs[1] = [torch.empty(n, m, 4, device="cuda")]
    outs[2] = [torch.empty(n, m, 4, device="cuda")]
    ctx.bump_mesh_batch(*[_ptr_table(x) for x in outs], _ptr_table(idx), *[_ptr_table([x]) for x in g], 0, 1, n, m, 1, 0.01, True, 0)
    rv, rc, rk = orc.vertex_from_bump(bump, rgbb, mask, param, 1)
    same(host(outs[0][0]), rv)
    same(host(outs[1][0]), rc)
    same(host(outs[2][0]), rk)
    same(host(outs[3][0]), orc.compute_normal(rv, True))
    same(host(idx[0]).view(np.uint32), orc.quad_trim(rv, mask, 1, 0.01, 1, 0))


def test_bump_mesh_odd_size_and_errors(ctx, orc, pkg):
    n, m = 37, 101
    rng = np.random.default_rng(1)
    bump = rng.integers(20000, 40000, (n, m, 3)).astype(np.uint16)
    rgbb = rng.integers(0, 256, (n, m, 3)).astype(np.uint8)
    mask = rng.choice(np.array([0, 1, 10, 11, 12, 30, 255], np.uint8), (n, m))
    param = np.array([0.0, 0.0, 1.0, 1.2, 500.0, 500.0, -0.1, -0.2, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0], np.float32)
    g = [dev(bump), dev(rgbb), dev(mask), dev(param)]
    outs = [[torch.empty(n, m, ch, device="cuda")] for ch in (3, 4, 4, 3)]
    idx = [torch.empty((n - 1) * (m - 1), 4, dtype=torch.int32, device="cuda")]
    ctx.bump_mesh_batch(*[_ptr_table(x) for x in outs], _ptr_table(idx), *[_ptr_table([x]) for x in g], 0, 1, n, m, 1, 0.01, True, 0)
    rv, rc, rk = orc.vertex_from_bump(bump, rgbb, mask, param, 1)
    same(host(outs[0][0]), rv)
    same(host(outs[2][0]), rk)
    same(host(outs[3][0]), orc.compute_normal(rv, True))
    same(host(idx[0]).view(np.uint32), orc.quad_trim(rv, mask, 1, 0.01, 1, 0))
    with pytest.raises(pkg.SlamError):
        ctx.quad_trim(outs[0][0], g[2], idx[0], 1, m, 0, 0.1, 1, 0)          # fewer than 2 rows: the reference's guard wraps
    with pytest.raises(pkg.SlamError):
        ctx.quad_trim(outs[0][0], g[2], idx[0], n, m, 0, 0.1, 1, 2)          # unknown variant
