"""The entry point that allocates nothing: rgbdslam_predict_raster_noalloc (static device z-buffer) against
rgbdslam_predict_raster (caller workspace) -- bit-exact, two calls in a row, the second on a smaller input
(the shared array keeps the first call's contents), then the size limit."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _scene(synth, orc, n, m, seed):
    calib = synth.calib_for(n, m)
    d, c = synth.render(np.eye(3), np.zeros(3), n, m, calib)
    v = orc.vertex_map_kinect(d, calib)
    rgbd = orc.add_zbuff_off(d, orc.cpy_char3(c), calib)
    rng = np.random.default_rng(seed)
    v = v.copy()
    i0, j0 = n // 3, m // 3
    box = (slice(i0, i0 + n // 6), slice(j0, j0 + m // 5))
    v[box][..., 2] += 0.5          # a foreground box: discontinuities, occlusion, trimmed quads
    v[box][..., :2] *= v[box][..., 2:3] / (v[box][..., 2:3] - 0.5)
    v[rng.random((n, m)) < 0.01] = 0   # holes
    ang = 0.01
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    t = np.array([0.004, -0.003, 0.005], np.float32)
    P = np.eye(4, dtype=np.float32)
    P[:3, :3] = R
    P[:3, 3] = t
    v2, rgbd2 = orc.transform_vertex_map(P.T.reshape(-1).copy(), v, rgbd)
    return calib, v2, rgbd2


def _both(ctx, calib, v, rgbd, n, m):
    ctx.set_calibration_matrix(calib)
    g_v = torch.from_numpy(np.ascontiguousarray(v)).cuda()
    g_rgbd = torch.from_numpy(np.ascontiguousarray(rgbd)).cuda()
    zbuf = torch.empty(n, m, dtype=torch.int64, device="cuda")
    tex_ref = torch.empty(n, m, 4, device="cuda")
    tex = torch.full((n, m, 4), -7.0, device="cuda")
    ctx.raster_clear(zbuf, n, m)
    ctx.predict_raster(tex_ref, zbuf, g_v, g_rgbd, None, n, m)
    ctx.predict_raster_noalloc(tex, g_v, g_rgbd, None, n, m)
    ctx.synchronize()
    return tex.cpu().numpy(), tex_ref.cpu().numpy()


def test_predict_raster_noalloc_matches_workspace_form(pkg, synth, orc):
    ctx = pkg.Context(0)
    try:
        for k, (n, m) in enumerate([(480, 640), (37, 53)]):      # the second call is smaller than the first
            calib, v, rgbd = _scene(synth, orc, n, m, seed=k)
            got, ref = _both(ctx, calib, v, rgbd, n, m)
            assert (ref[..., 3] > 0).mean() > 0.6
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "call %d differs" % k
            want, _ = orc.raster(v, rgbd, calib)
            assert np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32))
        # beyond the static array: refused, nothing launched
        n, m = 1081, 1920
        t = torch.empty(8, device="cuda")
        with pytest.raises(pkg.SlamError) as e:
            ctx.predict_raster_noalloc(t, t, t, None, n, m)
        assert "code 3" in str(e.value)                          # RGBDSLAM_E_LIMIT
    finally:
        ctx.close()


def test_predict_raster_noalloc_on_a_stream_of_its_own(pkg, synth, orc):
    """The context's stream is not the NULL stream: the memset of the static z-buffer and both kernels must follow the
    context's stream and device (the entry point makes ctx->device current itself)."""
    stream = torch.cuda.Stream()
    n, m = 120, 160
    calib, v, rgbd = _scene(synth, orc, n, m, seed=5)
    g_v = torch.from_numpy(np.ascontiguousarray(v)).cuda()
    g_rgbd = torch.from_numpy(np.ascontiguousarray(rgbd)).cuda()
    tex = torch.full((n, m, 4), -7.0, device="cuda")
    torch.cuda.synchronize()
    ctx = pkg.Context(0, stream.cuda_stream)
    try:
        ctx.set_calibration_matrix(calib)
        for _ in range(2):
            ctx.predict_raster_noalloc(tex, g_v, g_rgbd, None, n, m)
        ctx.synchronize()
        want, _ = orc.raster(v, rgbd, calib)
        assert np.array_equal(tex.cpu().numpy().view(np.uint32), np.asarray(want, np.float32).view(np.uint32))
    finally:
        ctx.close()
