"""rgb-d-slam_amd -- MI355X-native tracking-and-fusion hot path of diegothomas/RGB-D-SLAM.

The product is librgbdslam_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/rgbdslam.h) plus the C++ host mirror of the reference's L3/L4 classes
(librgbdslam_host.so).  This module is only the ctypes binding used by tests and bench.py;
PyTorch supplies device memory and streams, nothing else.  There is NO CPU fallback: a missing
library raises at load time.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librgbdslam_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "librgbdslam_host.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rgbdslam.h")

_lib = None


class IcpParams(C.Structure):
    _fields_ = [("lvl", C.c_int), ("max_iter", C.c_int * 3), ("distThres", C.c_float),
                ("angleThres", C.c_float), ("thres_cvg_alpha", C.c_float), ("thres_cvg_tr", C.c_float)]


def icp_params(dist=0.05, angle=0.3420201433256687, lvl=3):
    """TUM fr1/xyz settings (D1/GP-demo.cpp:409, D1/RG.h:111-126)."""
    p = IcpParams()
    p.lvl = lvl
    p.max_iter[0], p.max_iter[1], p.max_iter[2] = 3, 2, 1
    p.distThres, p.angleThres = dist, angle
    p.thres_cvg_alpha = p.thres_cvg_tr = 1e-6
    return p


def load():
    """Loads the HIP library or raises -- the product path never falls back to CPU code."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("librgbdslam_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(LIB_PATH)
        _lib.rgbdslam_version.restype = C.c_char_p
    return _lib


class SlamError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise SlamError("%s failed with code %d" % (what, rc))


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(int(t))


def _host16(a):
    arr = (C.c_float * len(a))(*[float(x) for x in a])
    return arr


class Context:
    """Thin owner of a rgbdslam_ctx; arguments are torch CUDA tensors (or raw device addresses)."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        self._h = C.c_void_p()
        _chk(self.lib.rgbdslam_create(C.byref(self._h), int(device), C.c_void_p(stream or 0)), "rgbdslam_create")

    @classmethod
    def borrow(cls, handle):
        """A Context over a rgbdslam_ctx that somebody else owns (e.g. the tracker's: RG.context()); close() does not destroy it."""
        self = cls.__new__(cls)
        self.lib = load()
        self._h = C.c_void_p(int(handle))
        self._borrowed = True
        return self

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            self.lib.rgbdslam_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- context --
    def set_stream(self, stream):
        _chk(self.lib.rgbdslam_set_stream(self._h, C.c_void_p(stream or 0)), "set_stream")

    def synchronize(self):
        _chk(self.lib.rgbdslam_synchronize(self._h), "synchronize")

    def set_calibration_matrix(self, calib):
        _chk(self.lib.rgbdslam_set_calibration_matrix(self._h, _host16(calib)), "set_calibration_matrix")

    def set_pose_matrix(self, pose):
        _chk(self.lib.rgbdslam_set_pose_matrix(self._h, _host16(pose)), "set_pose_matrix")

    # -- stage 1 --
    def vertex_map_kinect(self, vmap, depth, pitch, n, m):
        _chk(self.lib.rgbdslam_vertex_map_kinect(self._h, _ptr(vmap), _ptr(depth), C.c_size_t(pitch), n, m), "vertex_map_kinect")

    def gpu_cpy_char3(self, bgr, pitch, rgb, n, m):
        _chk(self.lib.rgbdslam_gpu_cpy_char3(self._h, _ptr(bgr), C.c_size_t(pitch), _ptr(rgb), n, m), "gpu_cpy_char3")

    def bilateral_depth(self, dst, dst_pitch, src, src_pitch, n, m, d=5, sigma_color=100.0, sigma_space=100.0):
        _chk(self.lib.rgbdslam_bilateral_depth(self._h, _ptr(dst), C.c_size_t(dst_pitch), _ptr(src), C.c_size_t(src_pitch), int(d),
                                               C.c_double(sigma_color), C.c_double(sigma_space), n, m), "bilateral_depth")

    def compute_normal(self, nmap, vmap, n, m, inverse=True):
        _chk(self.lib.rgbdslam_compute_normal(self._h, _ptr(nmap), _ptr(vmap), n, m, int(inverse)), "compute_normal")

    def intensity_sobel(self, rgb, img, gx, gy, n, m, pitch=None):
        pitch = pitch or 4 * m
        _chk(self.lib.rgbdslam_intensity_sobel(self._h, _ptr(rgb), _ptr(img), C.c_size_t(pitch), _ptr(gx), C.c_size_t(pitch),
                                               _ptr(gy), C.c_size_t(pitch), n, m), "intensity_sobel")

    # -- stage 2/3 --
    def alloc_buffers(self, grid_rows, grid_cols):
        _chk(self.lib.rgbdslam_alloc_buffers(self._h, grid_rows, grid_cols), "alloc_buffers")

    def estimate_system(self, R_dev, t_dev, vmap, nmap, rgb, gx, gy, vprev, nprev, rgbd_prev, intr_dev, dist, angle,
                        n, m, fact, corr=None, pitch=None):
        pitch = pitch or 4 * m
        A = (C.c_float * 36)()
        b = (C.c_float * 6)()
        _chk(self.lib.rgbdslam_estimate_system_gauss_newton(
            self._h, _ptr(R_dev), _ptr(t_dev), _ptr(vmap), _ptr(nmap), _ptr(rgb), _ptr(gx), C.c_size_t(pitch), _ptr(gy),
            C.c_size_t(pitch), _ptr(vprev), _ptr(nprev), _ptr(rgbd_prev), _ptr(intr_dev), C.c_float(dist), C.c_float(angle),
            n, m, A, b, fact, _ptr(corr)), "estimate_system_gauss_newton")
        return list(A), list(b)

    def align_gauss_newton(self, prm, vmap, nmap, rgb, gx, gy, vprev, nprev, rgbd_prev, n, m, pose_state, status,
                           trace=None, pitch=None):
        pitch = pitch or 4 * m
        _chk(self.lib.rgbdslam_align_gauss_newton(
            self._h, C.byref(prm), _ptr(vmap), _ptr(nmap), _ptr(rgb), _ptr(gx), C.c_size_t(pitch), _ptr(gy),
            C.c_size_t(pitch), _ptr(vprev), _ptr(nprev), _ptr(rgbd_prev), n, m, _ptr(pose_state), _ptr(status),
            _ptr(trace)), "align_gauss_newton")

    def track_frame(self, prm, vmap, nmap, rgb, gx, gy, vprev, nprev, rgbd_prev, n, m, pose_block, history, frame_index,
                    pitch=None):
        """rgbdslam_track_frame: AlignFrameGaussNewton + the pose algebra of display() on a device-resident pose block
        (43 words: pose_state 24, status 4 ... see include/rgbdslam.h); history = f32[frames][16] or None."""
        pitch = pitch or 4 * m
        _chk(self.lib.rgbdslam_track_frame(
            self._h, C.byref(prm), _ptr(vmap), _ptr(nmap), _ptr(rgb), _ptr(gx), C.c_size_t(pitch), _ptr(gy),
            C.c_size_t(pitch), _ptr(vprev), _ptr(nprev), _ptr(rgbd_prev), n, m, _ptr(pose_block), _ptr(history),
            int(frame_index)), "track_frame")

    def frame_aux(self, vmap, rgb, gx, gy, aux, n, m, pitch=None):
        """rgbdslam_frame_aux: the frame's auxiliary ICP records (f32 [n, m, 4]) from maps that already exist."""
        pitch = pitch or 4 * m
        _chk(self.lib.rgbdslam_frame_aux(self._h, _ptr(vmap), _ptr(rgb), _ptr(gx), C.c_size_t(pitch), _ptr(gy), C.c_size_t(pitch),
                                         _ptr(aux), n, m), "frame_aux")

    def set_frame_aux(self, aux_write=None, aux_read=None):
        """rgbdslam_set_frame_aux: records the NEXT preprocessing launch writes / the NEXT track_frame reads (one call each)."""
        _chk(self.lib.rgbdslam_set_frame_aux(self._h, _ptr(aux_write), _ptr(aux_read)), "set_frame_aux")

    def set_icp_loop(self, on):
        """Persistent-loop mode of track_frame (one launch per frame); raises when the device cannot hold the grid."""
        _chk(self.lib.rgbdslam_set_icp_loop(self._h, int(on)), "set_icp_loop")

    def icp_loop_faults(self):
        f = C.c_int(0)
        _chk(self.lib.rgbdslam_icp_loop_faults(self._h, C.byref(f)), "icp_loop_faults")
        return f.value

    def set_icp_line_check(self, on):
        _chk(self.lib.rgbdslam_set_icp_line_check(self._h, int(on)), "set_icp_line_check")

    def icp_line_rejections(self):
        r = C.c_uint(0)
        _chk(self.lib.rgbdslam_icp_line_rejections(self._h, C.byref(r)), "icp_line_rejections")
        return r.value

    def debug_set_icp_timeout(self, ticks):
        _chk(self.lib.rgbdslam_debug_set_icp_timeout(self._h, int(ticks)), "debug_set_icp_timeout")

    # -- stage 4 --
    def merge_depth_off(self, rgbd, depth, pitch, rgb, mask, n, m):
        _chk(self.lib.rgbdslam_merge_depth_off(self._h, _ptr(rgbd), _ptr(depth), C.c_size_t(pitch), _ptr(rgb), _ptr(mask), n, m),
             "merge_depth_off")

    def vertex_map_gl(self, vmap, rgbd, n, m):
        _chk(self.lib.rgbdslam_vertex_map_gl(self._h, _ptr(vmap), _ptr(rgbd), n, m), "vertex_map_gl")

    def update(self, bump_tab, rgb_tab, mask_tab, param_tab, claim_tab, stamp, vmap, rgbd, mask, label, pose, N_prim,
               M_prim, n, m, plane_lo=0, plane_hi=2 ** 31 - 1):
        _chk(self.lib.rgbdslam_update(self._h, _ptr(bump_tab), _ptr(rgb_tab), _ptr(mask_tab), _ptr(param_tab), _ptr(claim_tab),
                                      C.c_uint32(stamp), _ptr(vmap), _ptr(rgbd), _ptr(mask), _ptr(label), _host16(pose),
                                      N_prim, M_prim, n, m, plane_lo, plane_hi), "update")

    def vertex_from_bump(self, vmap, rgba, mask_out, bump, rgb_bump, mask, param, n, m, lvl=1):
        _chk(self.lib.rgbdslam_vertex_from_bump(self._h, _ptr(vmap), _ptr(rgba), _ptr(mask_out), _ptr(bump), _ptr(rgb_bump),
                                                _ptr(mask), _ptr(param), n, m, lvl), "vertex_from_bump")

    def vertex_from_bump_d2(self, vmap, rgba, bump, rgb_bump, mask, param, n, m, lvl, variant):
        _chk(self.lib.rgbdslam_vertex_from_bump_d2(self._h, _ptr(vmap), _ptr(rgba), _ptr(bump), _ptr(rgb_bump), _ptr(mask), _ptr(param),
                                                   n, m, lvl, variant), "vertex_from_bump_d2")

    def vertex_from_bump_batch(self, vmap_tab, rgba_tab, mask_out_tab, bump_tab, rgb_tab, mask_tab, param_tab, plane_lo, plane_hi, n, m):
        _chk(self.lib.rgbdslam_vertex_from_bump_batch(self._h, _ptr(vmap_tab), _ptr(rgba_tab), _ptr(mask_out_tab), _ptr(bump_tab),
                                                      _ptr(rgb_tab), _ptr(mask_tab), _ptr(param_tab), plane_lo, plane_hi, n, m),
             "vertex_from_bump_batch")

    # -- stage 5 --
    def quad_trim(self, vmap, mask, indices, n, m, min_conf=0, thresh=0.1, lvl=1, variant=0):
        _chk(self.lib.rgbdslam_quad_trim(self._h, _ptr(vmap), _ptr(mask), _ptr(indices), C.c_ushort(min_conf), C.c_uint(n), C.c_uint(m),
                                         C.c_float(thresh), int(lvl), int(variant)), "quad_trim")

    def bump_mesh_batch(self, vmap_tab, rgba_tab, mask_out_tab, nmap_tab, index_tab, bump_tab, rgb_tab, mask_tab, param_tab, plane_lo,
                        plane_hi, n, m, min_conf=0, thresh=0.1, inverse_normals=False, variant=0):
        _chk(self.lib.rgbdslam_bump_mesh_batch(self._h, _ptr(vmap_tab), _ptr(rgba_tab), _ptr(mask_out_tab), _ptr(nmap_tab),
                                               _ptr(index_tab), _ptr(bump_tab), _ptr(rgb_tab), _ptr(mask_tab), _ptr(param_tab),
                                               int(plane_lo), int(plane_hi), n, m, C.c_ushort(min_conf), C.c_float(thresh),
                                               int(inverse_normals), int(variant)), "bump_mesh_batch")

    # ---- plane detection front end ----
    def gpu_cpy_n(self, mat, nmap, n, m, chnl, pitch=None):
        _chk(self.lib.rgbdslam_gpu_cpy_n(self._h, _ptr(mat), C.c_size_t(pitch or m), _ptr(nmap), n, m, int(chnl)), "gpu_cpy_n")

    def canny(self, img, edges, work_labels, work_bytes, n, m, low=120.0, high=30.0, img_pitch=None, edges_pitch=None):
        _chk(self.lib.rgbdslam_canny(self._h, _ptr(img), C.c_size_t(img_pitch or m), _ptr(edges), C.c_size_t(edges_pitch or m),
                                     C.c_double(low), C.c_double(high), _ptr(work_labels), _ptr(work_bytes), n, m), "canny")

    def gpu_sum(self, out, r, g, b, n, m):
        s = C.c_size_t(m)
        _chk(self.lib.rgbdslam_gpu_sum(self._h, _ptr(out), s, _ptr(r), s, _ptr(g), s, _ptr(b), s, n, m), "gpu_sum")

    def gpu_discont(self, mat, vmap, n, m, thresh=0.03):
        _chk(self.lib.rgbdslam_gpu_discont(self._h, _ptr(mat), C.c_size_t(m), _ptr(vmap), n, m, C.c_float(thresh)), "gpu_discont")

    def gpu_thresh(self, out, img, label, n, m, thresh=20):
        _chk(self.lib.rgbdslam_gpu_thresh(self._h, _ptr(out), C.c_size_t(m), _ptr(img), C.c_size_t(m), C.c_uint8(thresh), _ptr(label),
                                          n, m), "gpu_thresh")

    def erode_ellipse5(self, dst, src, n, m):
        _chk(self.lib.rgbdslam_erode_ellipse5(self._h, _ptr(dst), C.c_size_t(m), _ptr(src), C.c_size_t(m), n, m), "erode_ellipse5")

    def find_blobs(self, binary, blob, sizes, n, m, pitch=None):
        _chk(self.lib.rgbdslam_find_blobs(self._h, _ptr(binary), C.c_size_t(pitch or m), _ptr(blob), _ptr(sizes), n, m), "find_blobs")

    def detect_regions(self, nmap, vmap, label, blob, sizes, work_labels, work_bytes, n, m, canny_low=120.0, canny_high=30.0,
                       thresh_discont=0.03, thresh_edge=20):
        _chk(self.lib.rgbdslam_detect_regions(self._h, _ptr(nmap), _ptr(vmap), _ptr(label), C.c_double(canny_low), C.c_double(canny_high),
                                              C.c_float(thresh_discont), C.c_uint8(thresh_edge), _ptr(blob), _ptr(sizes),
                                              _ptr(work_labels), _ptr(work_bytes), n, m), "detect_regions")

    def fit_planes(self, blob, sizes, vmap, nmap, pose, label, n, m, min_size=5000, min_normals=1000, max_labels=255):
        """rgbdslam_fit_planes + _read + _label: the device median fit of RG::DetectPrimitives (D1/RG.cpp:469-646).
        -> (blobs found, equations [k,10], centers [k,3]) in FindBlobs' order, or (blobs, None, None) when there are more blobs
        than the device path holds; `label` (u8 image on the device) receives the plane numbers of the blobs' pixels."""
        import numpy as np
        import torch
        self.lib.rgbdslam_fit_planes_workspace_bytes.restype = C.c_size_t
        ws = torch.empty(self.lib.rgbdslam_fit_planes_workspace_bytes(n, m), dtype=torch.uint8, device="cuda")
        _chk(self.lib.rgbdslam_fit_planes(self._h, _ptr(blob), _ptr(sizes), _ptr(vmap), _ptr(nmap), _host16(pose), int(min_size),
                                          int(min_normals), _ptr(ws), n, m), "fit_planes")
        nb = C.c_int()
        rec = np.zeros((64, 20), np.float32)
        _chk(self.lib.rgbdslam_fit_planes_read(self._h, _ptr(ws), C.byref(nb), C.c_void_p(rec.ctypes.data)), "fit_planes_read")
        if nb.value > 64:
            return nb.value, None, None
        ri = rec.view(np.int32)
        order = sorted(range(nb.value), key=lambda k: int(ri[k, 0]))
        lab = (C.c_uint8 * 64)()
        eq, ce = [], []
        for k in order:
            if len(eq) >= max_labels or not ri[k, 3]:
                continue
            lab[k] = len(eq) + 1
            eq.append(rec[k, 4:14].copy()); ce.append(rec[k, 17:20].copy())
        _chk(self.lib.rgbdslam_fit_planes_label(self._h, _ptr(label), _ptr(blob), _ptr(ws), lab, n, m), "fit_planes_label")
        self.synchronize()
        return nb.value, np.array(eq, np.float32).reshape(-1, 10), np.array(ce, np.float32).reshape(-1, 3)

    def transform_vertex_map(self, vmap, rgbd, n, m):
        _chk(self.lib.rgbdslam_transform_vertex_map(self._h, _ptr(vmap), _ptr(rgbd), n, m), "transform_vertex_map")

    def quad_trim_frame(self, vmap, indices, n, m, thresh=0.1):
        _chk(self.lib.rgbdslam_quad_trim_frame(self._h, _ptr(vmap), _ptr(indices), n, m, C.c_float(thresh)), "quad_trim_frame")

    def raster_clear(self, zbuf, n, m):
        _chk(self.lib.rgbdslam_raster_clear(self._h, _ptr(zbuf), n, m), "raster_clear")

    # -- prediction from the planes --
    def planes_draw(self, zbuf, bump_tab, rgb_tab, mask_tab, param_tab, plane_lo, plane_hi, pose_w2c, N_prim, M_prim, n, m,
                    min_conf=1, thresh=0.1, znear=0.5, zfar=10.0):
        _chk(self.lib.rgbdslam_planes_draw(self._h, _ptr(zbuf), _ptr(bump_tab), _ptr(rgb_tab), _ptr(mask_tab), _ptr(param_tab),
                                           plane_lo, plane_hi, _host16(pose_w2c), N_prim, M_prim, C.c_ushort(min_conf),
                                           C.c_float(thresh), C.c_float(znear), C.c_float(zfar), n, m), "planes_draw")

    def planes_resolve(self, zbuf, vmap, rgbd, bump_tab, rgb_tab, mask_tab, param_tab, plane_lo, plane_hi, pose_w2c, N_prim,
                       M_prim, n, m, reset=True):
        _chk(self.lib.rgbdslam_planes_resolve(self._h, _ptr(zbuf), _ptr(vmap), _ptr(rgbd), _ptr(bump_tab), _ptr(rgb_tab),
                                              _ptr(mask_tab), _ptr(param_tab), plane_lo, plane_hi, _host16(pose_w2c), N_prim,
                                              M_prim, int(reset), n, m), "planes_resolve")

    # -- 16-bit labels (models of more than 255 planes) --
    def update_l16(self, bump_tab, rgb_tab, mask_tab, param_tab, claim_tab, stamp, vmap, rgbd, mask, label16, pose, N_prim,
                   M_prim, n, m, plane_lo=0, plane_hi=0x7FFFFFFF):
        _chk(self.lib.rgbdslam_update_l16(self._h, _ptr(bump_tab), _ptr(rgb_tab), _ptr(mask_tab), _ptr(param_tab), _ptr(claim_tab),
                                          C.c_uint32(stamp), _ptr(vmap), _ptr(rgbd), _ptr(mask), _ptr(label16), _host16(pose),
                                          N_prim, M_prim, n, m, plane_lo, plane_hi), "update_l16")

    def update_from_history_l16(self, hist, bump_tab, rgb_tab, mask_tab, param_tab, claim_tab, stamp, vmap, rgbd, mask, label16,
                                N_prim, M_prim, n, m, plane_lo=0, plane_hi=0x7FFFFFFF):
        _chk(self.lib.rgbdslam_update_from_history_l16(self._h, _ptr(hist), _ptr(bump_tab), _ptr(rgb_tab), _ptr(mask_tab),
                                                       _ptr(param_tab), _ptr(claim_tab), C.c_uint32(stamp), _ptr(vmap), _ptr(rgbd),
                                                       _ptr(mask), _ptr(label16), N_prim, M_prim, n, m, plane_lo, plane_hi),
             "update_from_history_l16")

    def update_from_history_thin_l16(self, hist, bump_tab, rgb_tab, mask_tab, param_tab, claim_tab, stamp, w, rgb8, mask, label16,
                                     N_prim, M_prim, n, m, plane_lo=0, plane_hi=0x7FFFFFFF):
        _chk(self.lib.rgbdslam_update_from_history_thin_l16(self._h, _ptr(hist), _ptr(bump_tab), _ptr(rgb_tab), _ptr(mask_tab),
                                                            _ptr(param_tab), _ptr(claim_tab), C.c_uint32(stamp), _ptr(w), _ptr(rgb8),
                                                            _ptr(mask), _ptr(label16), N_prim, M_prim, n, m, plane_lo, plane_hi),
             "update_from_history_thin_l16")

    def export_thin(self, rgbd, mask, w_out, rgb8_out, mask_out, n, m):
        _chk(self.lib.rgbdslam_export_thin(self._h, _ptr(rgbd), _ptr(mask), _ptr(w_out), _ptr(rgb8_out), _ptr(mask_out), n, m), "export_thin")

    def label_planes_l16(self, label16, vmap, nmap, pose, hist, param_tab, nb, N_prim, M_prim, n, m, eps=0.1, alpha=0.8):
        _chk(self.lib.rgbdslam_label_planes_l16(self._h, _ptr(label16), _ptr(vmap), _ptr(nmap),
                                                _host16(pose) if pose is not None else None, _ptr(hist), _ptr(param_tab), nb,
                                                N_prim, M_prim, C.c_float(eps), C.c_float(alpha), n, m), "label_planes_l16")

    # -- the remaining operator functions of the reference headers (csrc/compat.hip) --
    def merge_depth_kinect(self, rgbd, depth, rgb, mask, n, m):
        _chk(self.lib.rgbdslam_merge_depth_kinect(self._h, _ptr(rgbd), _ptr(depth), _ptr(rgb), _ptr(mask), n, m), "merge_depth_kinect")

    def gpu_add_zbuff_kinect(self, depth, rgb_in, rgbd, n, m):
        _chk(self.lib.rgbdslam_gpu_add_zbuff_kinect(self._h, _ptr(depth), _ptr(rgb_in), _ptr(rgbd), n, m), "gpu_add_zbuff_kinect")

    def gpu_cpy(self, img, pitch, rgb, n, m):
        _chk(self.lib.rgbdslam_gpu_cpy(self._h, _ptr(img), C.c_size_t(pitch), _ptr(rgb), n, m), "gpu_cpy")

    def map_data_2_vbo(self, color, depth, coord, vmap, rgb, n, m):
        _chk(self.lib.rgbdslam_map_data_2_vbo(self._h, _ptr(color), _ptr(depth), _ptr(coord), _ptr(vmap), _ptr(rgb), n, m), "map_data_2_vbo")

    def vote_plan(self, label, vmap, centers, count, pose, nb, n, m):
        _chk(self.lib.rgbdslam_vote_plan(self._h, _ptr(label), _ptr(vmap), _ptr(centers), _ptr(count), _host16(pose), nb, n, m), "vote_plan")

    def affect_plan(self, label, buff, n, m):
        _chk(self.lib.rgbdslam_affect_plan(self._h, _ptr(label), _ptr(buff), n, m), "affect_plan")

    def overlap(self, bump, mask, param, param_in, n, m, N_prim, M_prim):
        ratio = C.c_float()
        cnt = (C.c_uint32 * 2)()
        _chk(self.lib.rgbdslam_overlap(self._h, _ptr(bump), _ptr(mask), _ptr(param), _ptr(param_in), n, m, N_prim, M_prim,
                                       C.byref(ratio), cnt), "overlap")
        return ratio.value, (cnt[0], cnt[1])

    def table_store(self, tab, ptr, indx):
        _chk(self.lib.rgbdslam_table_store(self._h, _ptr(tab), _ptr(ptr), indx), "table_store")

    def update_cu(self, bump_tab, rgb_tab, mask_tab, param_tab, vmap, rgbd, mask, label, pose, N_prim, M_prim, n, m, nb_planes):
        _chk(self.lib.rgbdslam_update_cu(self._h, _ptr(bump_tab), _ptr(rgb_tab), _ptr(mask_tab), _ptr(param_tab), _ptr(vmap),
                                         _ptr(rgbd), _ptr(mask), _ptr(label), _host16(pose), N_prim, M_prim, n, m, nb_planes),
             "update_cu")

    def segment_cu(self, label, vmap, nmap, pose, eq, nb, n, m, eps=0.1, alpha=0.9):
        _chk(self.lib.rgbdslam_segment_cu(self._h, _ptr(label), _ptr(vmap), _ptr(nmap), _host16(pose), _ptr(eq), nb,
                                          C.c_float(eps), C.c_float(alpha), n, m), "segment_cu")

    def predict_raster_noalloc(self, tex, vmap, rgbd, indices, n, m, thresh=0.1, znear=0.5, zfar=10.0):
        _chk(self.lib.rgbdslam_predict_raster_noalloc(self._h, _ptr(tex), _ptr(vmap), _ptr(rgbd), _ptr(indices),
                                                      C.c_float(thresh), C.c_float(znear), C.c_float(zfar), n, m),
             "predict_raster_noalloc")

    def predict_raster(self, tex, zbuf, vmap, rgbd, indices, n, m, thresh=0.1, znear=0.5, zfar=10.0):
        _chk(self.lib.rgbdslam_predict_raster(self._h, _ptr(tex), _ptr(zbuf), _ptr(vmap), _ptr(rgbd), _ptr(indices),
                                              C.c_float(thresh), C.c_float(znear), C.c_float(zfar), n, m), "predict_raster")

    def read_frame(self, vmap, rgbd, tex, n, m):
        _chk(self.lib.rgbdslam_read_frame(self._h, _ptr(vmap), _ptr(rgbd), _ptr(tex), n, m), "read_frame")

    def project_mask(self, mask, mask_swap, vmap, n, m):
        _chk(self.lib.rgbdslam_project_mask(self._h, _ptr(mask), _ptr(mask_swap), _ptr(vmap), n, m), "project_mask")

    def project_label(self, label, label_swap, vmap, n, m):
        """ProjectMask_cu on the label image (Frame.cpp:389), read with the context's label width."""
        _chk(self.lib.rgbdslam_project_label(self._h, _ptr(label), _ptr(label_swap), _ptr(vmap), n, m), "project_label")

    def set_label_width(self, nbytes):
        """1 (default): label images are u8 like the reference's (Frame.h:197); 2: every label image this context's entry
        points read or write is a uint16 image (models of more than 255 planes)."""
        _chk(self.lib.rgbdslam_set_label_width(self._h, int(nbytes)), "set_label_width")

    def label_width(self):
        return int(self.lib.rgbdslam_get_label_width(self._h))

    def segment(self, label, label_swap, vmap, nmap, pose, equations, nb, n, m, eps=0.1, alpha=0.9):
        _chk(self.lib.rgbdslam_segment(self._h, _ptr(label), _ptr(label_swap), _ptr(vmap), _ptr(nmap), _host16(pose),
                                       _ptr(equations), nb, C.c_float(eps), C.c_float(alpha), n, m), "segment")

    def gpu_add_zbuff_off(self, depth, pitch, rgb_in, rgbd, n, m):
        _chk(self.lib.rgbdslam_gpu_add_zbuff_off(self._h, _ptr(depth), C.c_size_t(pitch), _ptr(rgb_in), _ptr(rgbd), n, m),
             "gpu_add_zbuff_off")

    def init_mask(self, mask, rgbd, n, m):
        _chk(self.lib.rgbdslam_init_mask(self._h, _ptr(mask), _ptr(rgbd), n, m), "init_mask")

    def seed_labels(self, label, vmap, nmap, pose, equations, nb, n, m, eps=0.1, alpha=0.8):
        _chk(self.lib.rgbdslam_seed_labels(self._h, _ptr(label), _ptr(vmap), _ptr(nmap), _host16(pose), _ptr(equations), nb,
                                           C.c_float(eps), C.c_float(alpha), n, m), "seed_labels")


# ---- host classes (librgbdslam_host.so: C++ mirror of RG / Frame / Graph / Primitive) ---------------
_host = None


def load_host():
    global _host
    if _host is None:
        load()
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError("librgbdslam_host.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _host = C.CDLL(HOST_LIB_PATH)
        _host.slamhost_last_error.restype = C.c_char_p
    return _host


def _hchk(rc, what):
    if rc != 0:
        raise SlamError("%s failed with code %d: %s" % (what, rc, (load_host().slamhost_last_error() or b"").decode()))


class RG:
    """ctypes handle on the C++ `slam::RG` (D1/RG.h) -- the reference's reconstruction driver plus
    the display() loop (D1/GP-demo.cpp:138-296) as `step()`."""

    BUFFER_DTYPES = {"VMap": ("float32", 3), "NMap": ("float32", 3), "RGBD": ("float32", 4), "Mask": ("uint8", 1),
                     "Label": ("uint8", 1), "frame.VMap": ("float32", 3), "frame.NMap": ("float32", 3),
                     "frame.RGB": ("float32", 3)}

    def __init__(self, n, m, calib, lvl=3, dist=0.05, angle=0.3420201433256687, device=0, stream=None, max_planes=50,
                 N_prim=1024, M_prim=1024):
        self.lib = load_host()
        self.n, self.m, self.N_prim, self.M_prim = n, m, N_prim, M_prim
        self._h = C.c_void_p()
        _hchk(self.lib.slamhost_rg_create(C.byref(self._h), int(device), C.c_void_p(stream or 0), n, m, _host16(calib), lvl,
                                          C.c_float(dist), C.c_float(angle), max_planes, N_prim, M_prim), "rg_create")

    def close(self):
        if self._h:
            self.lib.slamhost_rg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_fused(self, on):
        """True (default): fused, sync-free frame pipeline; False: one C-ABI call per reference operator."""
        _hchk(self.lib.slamhost_rg_set_fused(self._h, int(on)), "set_fused")

    def set_overlap(self, on):
        """Fused pipeline: True runs the next frame's preprocessing and the Bump update on auxiliary streams (default off:
        measured slower than one stream at 640x480)."""
        _hchk(self.lib.slamhost_rg_set_overlap(self._h, int(on)), "set_overlap")

    def set_overlap_mode(self, mode):
        """Bits: 1 = frames are uploaded + preprocessed on a load stream (pays for host-fed frames: the copies run beside the
        previous frame's kernels), 2 = Bump update on its own stream.  Before the first frame."""
        _hchk(self.lib.slamhost_rg_set_overlap_mode(self._h, int(mode)), "set_overlap_mode")

    def set_stream_events(self, on):
        """Overlap mode 1: order the load stream and the frame stream by events instead of stream value operations (needed
        under `rocprofv3 --pmc`, which would hang the value waits).  Before the first frame."""
        _hchk(self.lib.slamhost_rg_set_stream_events(self._h, int(on)), "set_stream_events")

    HISTORY_CAP = 8192        # frames the pose history ring holds (slam::RG::HISTORY_CAP, host/rgbdslam_host.hpp)

    def history(self, first, count):
        """Poses of frames [first, first+count): rows of poseR(9, col-major) poset(3) tracked iterations level_iters 0."""
        import numpy as np
        out = np.empty((count, 16), np.float32)
        _hchk(self.lib.slamhost_rg_history(self._h, int(first), int(count), C.c_void_p(out.ctypes.data)), "history")
        return out

    def set_stage_timing(self, on):
        """The reference's TIMING switch (D1/Utilities.h:82, GP-demo.cpp:153-277) for the asynchronous pipeline: HIP events at
        the stage boundaries (an event record per boundary: opt-in)."""
        _hchk(self.lib.slamhost_rg_set_stage_timing(self._h, int(on)), "set_stage_timing")

    def stage_times(self):
        """Mean us per frame since the last read (at most the last 64 frames) -> (dict, frames): stage 1 preprocessing (0 when
        it rode on the previous frame's merge launch), stages 2+3 ICP, stage 5 prediction, stage 4 fusion."""
        us = (C.c_double * 4)()
        frames = C.c_int()
        _hchk(self.lib.slamhost_rg_stage_times(self._h, us, C.byref(frames)), "stage_times")
        names = ("stage1_preprocess", "stage2_3_icp", "stage5_prediction", "stage4_fusion")
        return {k: round(us[i], 2) for i, k in enumerate(names)}, frames.value

    def set_label16(self, on):
        """u16 label images through the whole tracker (ProjectLabel, Segment, Graph::Update) instead of the reference's
        unsigned char (Frame.h:197): models of more than 255 planes.  Before any plane exists."""
        _hchk(self.lib.slamhost_rg_set_label16(self._h, int(on)), "set_label16")
        self._label16 = bool(on)

    def set_plane_equations(self, equations):
        """Tracker of a model whose Bump images live elsewhere (the tracking rank of the plane-partitioned form): the plane
        equations [nb, 10] = (n, e1, e2, d) for ProjectLabel + Segment_cu (D1/Frame.cpp:385-405); nothing is fused here."""
        import numpy as np
        eq = np.ascontiguousarray(equations, np.float32)
        _hchk(self.lib.slamhost_rg_set_plane_equations(self._h, C.c_void_p(eq.ctypes.data), int(eq.size // 10)), "set_plane_equations")

    def seed_labels_tiled(self, param_tab, nb, N_prim, M_prim):
        """Detection stand-in for a tiled model (needs set_label16): the current model frame's labels by plane membership incl.
        the patch bounds (rgbdslam_label_planes_l16); param_tab = device table of nb 14-float parameter blocks."""
        _hchk(self.lib.slamhost_rg_seed_labels_tiled(self._h, _ptr(param_tab), int(nb), int(N_prim), int(M_prim)), "seed_labels_tiled")

    def export_labels(self, dst, off_label):
        """Copies the label image of the frame just processed into dst + off_label (stream-ordered)."""
        _hchk(self.lib.slamhost_rg_export_labels(self._h, _ptr(dst), C.c_size_t(off_label)), "export_labels")

    def set_icp_loop(self, on):
        """True: all Gauss-Newton iterations of a frame run in ONE persistent kernel (needs the GPU to itself and a single
        stream; same results).  Raises when the device cannot hold the kernel's 512 workgroups at once."""
        _hchk(self.lib.slamhost_rg_set_icp_loop(self._h, int(on)), "set_icp_loop")

    def icp_loop_faults(self):
        """Persistent-loop launches that gave up waiting since the last call (synchronises); a fault switches the mode off."""
        f = C.c_int(0)
        _hchk(self.lib.slamhost_rg_icp_loop_faults(self._h, C.byref(f)), "icp_loop_faults")
        return f.value

    def _ctx_ptr(self):
        self.lib.slamhost_rg_ctx.restype = C.c_void_p
        return C.c_void_p(self.lib.slamhost_rg_ctx(self._h))

    def context(self):
        """The tracker's own rgbdslam_ctx (its stream, its calibration) as a borrowed Context."""
        return Context.borrow(self._ctx_ptr().value)

    def set_icp_line_check(self, on):
        """Verify the integrity word of every line the ICP kernels hand between workgroups (rgbdslam_set_icp_line_check)."""
        _chk(load().rgbdslam_set_icp_line_check(self._ctx_ptr(), int(on)), "set_icp_line_check")

    def icp_line_rejections(self):
        r = C.c_uint(0)
        _chk(load().rgbdslam_icp_line_rejections(self._ctx_ptr(), C.byref(r)), "icp_line_rejections")
        return r.value

    def debug_set_icp_timeout(self, ticks):
        _chk(load().rgbdslam_debug_set_icp_timeout(self._ctx_ptr(), int(ticks)), "debug_set_icp_timeout")

    def set_device_loop(self, on):
        _hchk(self.lib.slamhost_rg_set_device_loop(self._h, int(on)), "set_device_loop")

    def set_index_buffer(self, on):
        _hchk(self.lib.slamhost_rg_set_index_buffer(self._h, int(on)), "set_index_buffer")

    def set_plane_range(self, lo, hi):
        _hchk(self.lib.slamhost_rg_set_plane_range(self._h, int(lo), int(hi)), "set_plane_range")

    def load_frame(self, depth_np, bgr_np):
        """Host images (numpy u16 [n,m], u8 [n,m,3]); uploaded like GpuMat::upload (D1/Frame.cpp:56-60)."""
        self._keep = (depth_np, bgr_np)
        _hchk(self.lib.slamhost_rg_load_frame(self._h, C.c_void_p(depth_np.ctypes.data), C.c_void_p(bgr_np.ctypes.data)),
              "load_frame")

    def load_frame_device(self, depth_t, bgr_t, depth_pitch=None, bgr_pitch=None):
        _hchk(self.lib.slamhost_rg_load_frame_device(self._h, _ptr(depth_t), C.c_size_t(depth_pitch or 2 * self.m), _ptr(bgr_t),
                                                     C.c_size_t(bgr_pitch or 3 * self.m)), "load_frame_device")

    def hint_next_device_frame(self, depth_t, bgr_t, depth_pitch=None, bgr_pitch=None):
        """The device frame the NEXT load_frame_device will name: its maps are then computed in this frame's Bump-update
        launch.  A hint that does not come true costs nothing but that work."""
        _hchk(self.lib.slamhost_rg_hint_next_device_frame(self._h, _ptr(depth_t), C.c_size_t(depth_pitch or 2 * self.m), _ptr(bgr_t),
                                                          C.c_size_t(bgr_pitch or 3 * self.m)), "hint_next_device_frame")

    def step_async(self):
        _hchk(self.lib.slamhost_rg_step_async(self._h), "step_async")

    def set_planes(self, equations, centers):
        eq = [float(x) for x in equations.reshape(-1)]
        ce = [float(x) for x in centers.reshape(-1)]
        _hchk(self.lib.slamhost_rg_set_planes(self._h, _host16(eq), _host16(ce), len(eq) // 10), "set_planes")

    def step(self):
        st = C.c_int()
        _hchk(self.lib.slamhost_rg_step(self._h, C.byref(st)), "step")
        return st.value

    def prepare_device_sequence(self, depth_ts, bgr_ts):
        """The two pointer tables of run_device_sequence, built ahead of the call (a caller that times the call, or replays
        the same frames, keeps the 2 k data_ptr() look-ups out of it).  Holds references to the tensors."""
        k = len(depth_ts)
        dt = (C.c_void_p * k)(*[t.data_ptr() for t in depth_ts])
        bt = (C.c_void_p * k)(*[t.data_ptr() for t in bgr_ts])
        return (dt, bt, k, list(depth_ts), list(bgr_ts))

    def run_device_sequence(self, depth_ts, bgr_ts=None):
        """depth_ts / bgr_ts: lists of device tensors (u16 [n, m], u8 [n, m, 3]), or depth_ts = what prepare_device_sequence returned."""
        dt, bt, k = depth_ts[:3] if bgr_ts is None else self.prepare_device_sequence(depth_ts, bgr_ts)[:3]
        ok = C.c_int()
        _hchk(self.lib.slamhost_rg_run_device_sequence(self._h, dt, bt, k, C.c_size_t(2 * self.m), C.c_size_t(3 * self.m),
                                                       C.byref(ok)), "run_device_sequence")
        return ok.value

    def open_dataset(self, path, size_data=0, real=False, rgbd_rep=True):
        """Dataset part of the reference constructor (D1/RG.h:145-156); returns the sequence length."""
        size = C.c_int()
        _hchk(self.lib.slamhost_rg_open_dataset(self._h, os.fsencode(path), int(size_data), int(real), int(rgbd_rep), C.byref(size)),
              "open_dataset")
        return size.value

    def load_next_frame(self):
        """RG::LoadFrame() (D1/RG.cpp:78-147): reads the next image pair from the dataset; False when exhausted."""
        ok = C.c_int()
        _hchk(self.lib.slamhost_rg_load_next_frame(self._h, C.byref(ok)), "load_next_frame")
        return bool(ok.value)

    def save_trajectory(self, filename):
        _hchk(self.lib.slamhost_rg_save_trajectory(self._h, os.fsencode(filename)), "save_trajectory")

    def detect_planes(self, min_size=-1, min_normals=-1, cap=50):
        """First branch of RG::Finish (D1/RG.cpp:154-175): DetectPrimitives + CreatePrimitives on the current model frame.
        -> (equations [k,10], centers [k,3]); min_size / min_normals override MIN_SIZE_PLAN (5000) / 1000 for small images."""
        import numpy as np
        nb = C.c_int()
        _hchk(self.lib.slamhost_rg_detect_planes(self._h, int(min_size), int(min_normals), C.byref(nb)), "detect_planes")
        eq, ce = np.zeros((cap, 10), np.float32), np.zeros((cap, 3), np.float32)
        _hchk(self.lib.slamhost_rg_get_planes(self._h, C.c_void_p(eq.ctypes.data), C.c_void_p(ce.ctypes.data), cap), "get_planes")
        return eq[:nb.value], ce[:nb.value]

    def reorganize_equa(self):
        """RG::ReorganizeEqua (D1/RG.cpp:934-1006): drops the planes with at most MIN_SIZE_PLAN labelled pixels in the model
        frame, renumbers the rest (labels, equations, centres, Bump images).  -> (equations [k,10], centers [k,3])"""
        import numpy as np
        nb = C.c_int()
        _hchk(self.lib.slamhost_rg_reorganize_equa(self._h, C.byref(nb)), "reorganize_equa")
        eq, ce = np.zeros((max(nb.value, 1), 10), np.float32), np.zeros((max(nb.value, 1), 3), np.float32)
        _hchk(self.lib.slamhost_rg_get_planes(self._h, C.c_void_p(eq.ctypes.data), C.c_void_p(ce.ctypes.data), max(nb.value, 1)), "get_planes")
        return eq[:nb.value], ce[:nb.value]

    def set_bilateral(self, on):
        """Optional stage of OffLineFrame::LoadFrame, default OFF like in the reference (the call is commented out there,
        D1/Frame.cpp:50-54): cv::bilateralFilter(depth, 5, 100.0, 100.0) before the vertex map is computed."""
        _hchk(self.lib.slamhost_rg_set_bilateral(self._h, int(on)), "set_bilateral")

    def set_finish(self, enabled=True, first_after=30, period=100, save_dir=None, min_size=-1, min_normals=-1,
                   reset_without_dump=False):
        """RG::Finish schedule (D1/RG.cpp:149-271): first plane detection when _count > first_after, then every `period`
        frames dump into save_dir + plane reset + re-detection; period 0 = never.  As in RG.cpp:177-187 a due period is
        skipped altogether while save_dir holds a Dummy.txt -- and when there is no save_dir, unless reset_without_dump
        says that the fused planes may be thrown away unsaved."""
        _hchk(self.lib.slamhost_rg_set_finish(self._h, int(enabled), int(first_after), int(period),
                                              os.fsencode(save_dir) if save_dir else None, int(min_size), int(min_normals)), "set_finish")
        _hchk(self.lib.slamhost_rg_set_finish_reset_without_dump(self._h, int(reset_without_dump)), "set_finish_reset_without_dump")

    def finish_skipped(self):
        """Periods of RG::Finish that were due but skipped because nothing could be saved."""
        return self.lib.slamhost_rg_finish_skipped(self._h)

    def finish_stats(self):
        out = (C.c_int * 4)()
        _hchk(self.lib.slamhost_rg_finish_stats(self._h, out), "finish_stats")
        return {"detections": out[0], "dumps": out[1], "planes": out[2], "indx": out[3]}

    def save_model(self, directory, indx):
        """Saving half of RG::Finish + SaveThread (D1/RG.cpp:16-59): cleaned planes + key frame in the reference's dump format."""
        _hchk(self.lib.slamhost_rg_save_model(self._h, os.fsencode(directory), int(indx)), "save_model")

    def synchronize(self):
        _hchk(self.lib.slamhost_rg_synchronize(self._h), "synchronize")

    def frames_done(self):
        return self.lib.slamhost_rg_frames_done(self._h)

    def export_frame(self, dst, off_vmap, off_rgbd, off_mask, off_hist):
        """Packs the merged model VMap / RGBD / Mask and the frame's history entry into `dst` (stream-ordered)."""
        _hchk(self.lib.slamhost_rg_export_frame(self._h, _ptr(dst), C.c_size_t(off_vmap), C.c_size_t(off_rgbd), C.c_size_t(off_mask),
                                                C.c_size_t(off_hist)), "export_frame")

    def export_frame_thin_labels(self, dst, off_w, off_rgb8, off_mask, off_hist, off_label):
        """export_frame_thin + the u16 label image the tracker maintains (set_label16), all in one launch."""
        _hchk(self.lib.slamhost_rg_export_frame_thin_labels(self._h, _ptr(dst), C.c_size_t(off_w), C.c_size_t(off_rgb8), C.c_size_t(off_mask),
                                                            C.c_size_t(off_hist), C.c_size_t(off_label)), "export_frame_thin_labels")

    def export_frame_thin(self, dst, off_w, off_rgb8, off_mask, off_hist):
        """Thin pack: RGBD.w (f32), the colour bytes, Mask and the history entry of the merged model frame (stream-ordered)."""
        _hchk(self.lib.slamhost_rg_export_frame_thin(self._h, _ptr(dst), C.c_size_t(off_w), C.c_size_t(off_rgb8), C.c_size_t(off_mask),
                                                     C.c_size_t(off_hist)), "export_frame_thin")

    def pose(self):
        R = (C.c_float * 9)()
        t = (C.c_float * 3)()
        _hchk(self.lib.slamhost_rg_get_pose(self._h, R, t), "get_pose")
        return list(R), list(t)

    def last_iterations(self):
        return self.lib.slamhost_rg_last_iterations(self._h)

    def buffer_ptr(self, name):
        p = C.c_void_p()
        nb = C.c_size_t()
        _hchk(self.lib.slamhost_rg_buffer(self._h, name.encode(), C.byref(p), C.byref(nb)), "buffer(%s)" % name)
        return p.value, nb.value

    def buffer(self, name):
        """Copies a named device buffer to a numpy array (torch-free)."""
        import numpy as np
        ptr, nb = self.buffer_ptr(name)
        self.synchronize()
        out = np.empty(nb, np.uint8)
        _hchk(self.lib.slamhost_copy_to_host(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(nb)), "copy_to_host")
        if name == "Label" and nb == 2 * self.n * self.m:
            return out.view("uint16").reshape(self.n, self.m)
        if name in self.BUFFER_DTYPES:
            dt, ch = self.BUFFER_DTYPES[name]
            a = out.view(dt)
            return a.reshape(self.n, self.m, ch) if ch > 1 else a.reshape(self.n, self.m)
        if name.startswith("Bump"):
            return out.view("uint16").reshape(self.N_prim, self.M_prim, 3)
        if name.startswith("RGBb"):
            return out.reshape(self.N_prim, self.M_prim, 3)
        return out.reshape(self.N_prim, self.M_prim)


# ---- dataset I/O and trajectory (host code, no GPU needed) --------------------------------
def _pairs(buf):
    return [tuple(l.split("\t")) for l in buf.value.decode().split("\n") if l]


def load_timestamps(path, associated=False, cap=1 << 24):
    """LoadTimeStamps / LoadAssociatedTimeStamps (D1/Utilities.cpp:171-304) -> [(depth_name, color_name), ...]."""
    buf = C.create_string_buffer(cap)
    cnt = C.c_int()
    _hchk(load_host().slamhost_load_timestamps(os.fsencode(path), int(associated), buf, C.c_size_t(cap), C.byref(cnt)), "load_timestamps")
    return _pairs(buf)


def dataset_frame_files(path, count, size_data=0, real=False, rgbd_rep=True):
    """File names RG::LoadFrame builds for its counter value `count` (D1/RG.cpp:105-139) + sequence length."""
    buf = C.create_string_buffer(1 << 16)
    size = C.c_int()
    _hchk(load_host().slamhost_dataset_frame_files(os.fsencode(path), int(size_data), int(real), int(rgbd_rep), int(count), buf,
                                                   C.c_size_t(1 << 16), C.byref(size)), "dataset_frame_files")
    return _pairs(buf)[0], size.value


def read_image(filename):
    """cv::imread(file, CV_LOAD_IMAGE_UNCHANGED) for PNG / TIFF -> numpy [rows, cols(, channels)] u8/u16, BGR order."""
    import numpy as np
    r, c, ch, bits = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    h = load_host()
    _hchk(h.slamhost_read_image(os.fsencode(filename), C.byref(r), C.byref(c), C.byref(ch), C.byref(bits), None, C.c_size_t(0)), "read_image")
    out = np.empty((r.value, c.value, ch.value), np.uint16 if bits.value == 16 else np.uint8)
    _hchk(h.slamhost_read_image(os.fsencode(filename), C.byref(r), C.byref(c), C.byref(ch), C.byref(bits), C.c_void_p(out.ctypes.data),
                                C.c_size_t(out.nbytes)), "read_image")
    return out[:, :, 0] if ch.value == 1 else out


def write_image(filename, arr):
    import numpy as np
    a = np.ascontiguousarray(arr)
    if a.dtype not in (np.uint8, np.uint16) or a.ndim not in (2, 3):
        raise ValueError("write_image: u8/u16 arrays of 2 or 3 dimensions")
    ch = 1 if a.ndim == 2 else a.shape[2]
    _hchk(load_host().slamhost_write_image(os.fsencode(filename), a.shape[0], a.shape[1], ch, 8 * a.itemsize, C.c_void_p(a.ctypes.data)),
          "write_image")


def matrix2quaternion(R):
    """D1/Utilities.cpp:11-68; R = 3x3 (row, col) -> (w, x, y, z) float32."""
    import numpy as np
    Rc = np.ascontiguousarray(np.asarray(R, np.float32).T)     # column-major like Eigen
    q = np.empty(4, np.float32)
    load_host().slamhost_matrix2quaternion(C.c_void_p(Rc.ctypes.data), C.c_void_p(q.ctypes.data))
    return q


def save_trajectory(filename, Rs, ts, names):
    """SaveTrajectory (D1/Utilities.cpp:500-546); Rs [k,3,3] (row, col), ts [k,3], names = listing entries."""
    import numpy as np
    Rc = np.ascontiguousarray(np.asarray(Rs, np.float32).transpose(0, 2, 1))
    tc = np.ascontiguousarray(np.asarray(ts, np.float32))
    _hchk(load_host().slamhost_save_trajectory(os.fsencode(filename), C.c_void_p(Rc.ctypes.data), C.c_void_p(tc.ctypes.data),
                                               "\n".join(names).encode(), len(names)), "save_trajectory")


# ---- model dump format (host code) ----------------------------------------------------------
def _eq19(p):
    import numpy as np
    return np.array(list(p["nmle"]) + [p["dist"]] + list(p["e1"]) + list(p["e2"]) + list(p["shift"]) + list(p["scale"]) +
                    [p["size"][0], p["size"][1]] + list(p["anchor"]), np.float32)


def _plane_dict(e):
    return {"nmle": e[0:3].copy(), "dist": e[3], "e1": e[4:7].copy(), "e2": e[7:10].copy(), "shift": e[10:12].copy(),
            "scale": e[12:14].copy(), "size": (int(e[14]), int(e[15])), "anchor": e[16:19].copy()}


def save_graph(directory, indx, pose16, planes, clean=True):
    """Primitive::Clean + Graph::save (D1/Primitive.cpp:4-134, D1/Graph.cpp:97-248).  planes: dicts with nmle, dist, e1,
    e2, shift, scale, size, anchor and the arrays bump [n,m,3] u16, rgb [n,m,3] u8, mask [n,m] u8 (None = no data).
    Returns the plane dicts as written (cleaned shift/size; size (0, 0) for dropped planes)."""
    import numpy as np
    nb = len(planes)
    eq = np.zeros((max(nb, 1), 19), np.float32)
    keep = []
    B, R, M = (C.c_void_p * max(nb, 1))(), (C.c_void_p * max(nb, 1))(), (C.c_void_p * max(nb, 1))()
    for k, p in enumerate(planes):
        eq[k] = _eq19(p)
        if p.get("bump") is None:
            eq[k, 14:16] = 0
            continue
        arrs = (np.ascontiguousarray(p["bump"], np.uint16), np.ascontiguousarray(p["rgb"], np.uint8), np.ascontiguousarray(p["mask"], np.uint8))
        keep.append(arrs)
        B[k], R[k], M[k] = (a.ctypes.data for a in arrs)
    _hchk(load_host().slamhost_save_graph(os.fsencode(directory), int(indx), _host16([float(v) for v in pose16]), nb,
                                          C.c_void_p(eq.ctypes.data), B, R, M, int(clean)), "save_graph")
    return [_plane_dict(eq[k]) for k in range(nb)]


def save_frame(directory, indx, rgbd):
    import numpy as np
    a = np.ascontiguousarray(rgbd, np.float32)
    _hchk(load_host().slamhost_save_frame(os.fsencode(directory), int(indx), C.c_void_p(a.ctypes.data), a.shape[0], a.shape[1]), "save_frame")


def load_model(directory, indx):
    """Model3D::Update's loader + KeyFrame (D2/Model3D.cpp:1050-1184, D2/KeyFrame.h:49-104).  None while the dump is
    not complete; else dict(pose [16], planes [dicts incl. bump, rgb, mask(u16)], depth, rgb)."""
    import numpy as np
    h = load_host()
    handle, ready = C.c_void_p(), C.c_int()
    _hchk(h.slamhost_model_open(os.fsencode(directory), int(indx), C.byref(handle), C.byref(ready)), "model_open")
    if not ready.value:
        return None
    try:
        pose = np.empty(16, np.float32)
        _hchk(h.slamhost_model_pose(handle, C.c_void_p(pose.ctypes.data)), "model_pose")
        planes = []
        for k in range(h.slamhost_model_count(handle)):
            e = np.empty(19, np.float32)
            _hchk(h.slamhost_model_plane(handle, k, C.c_void_p(e.ctypes.data)), "model_plane")
            d = _plane_dict(e)
            n, m = d["size"]
            d["bump"], d["rgb"], d["mask"] = np.empty((n, m, 3), np.uint16), np.empty((n, m, 3), np.uint8), np.empty((n, m), np.uint16)
            _hchk(h.slamhost_model_plane_data(handle, k, C.c_void_p(d["bump"].ctypes.data), C.c_void_p(d["rgb"].ctypes.data),
                                              C.c_void_p(d["mask"].ctypes.data)), "model_plane_data")
            planes.append(d)
        imgs = []
        for which, ch in ((0, 1), (1, 3)):
            r, c = C.c_int(), C.c_int()
            _hchk(h.slamhost_model_keyframe(handle, which, C.byref(r), C.byref(c), None), "model_keyframe")
            a = np.empty((r.value, c.value, ch), np.uint16)
            _hchk(h.slamhost_model_keyframe(handle, which, C.byref(r), C.byref(c), C.c_void_p(a.ctypes.data)), "model_keyframe")
            imgs.append(a[:, :, 0] if ch == 1 else a)
        return {"pose": pose, "planes": planes, "depth": imgs[0], "rgb": imgs[1]}
    finally:
        h.slamhost_model_close(handle)
