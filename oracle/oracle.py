"""ctypes/numpy front end of the CPU oracle (oracle/rgbdslam_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  PARITY UNPINNED: the reference has no golden vectors and cannot be built here;
see rgbdslam_oracle.h.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBS = {}
_OMP = False


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def use_openmp(on):
    """Switches every later call (and every Slam created later) between the default single-thread library and the
    same sources compiled with -fopenmp (bit-identical results; bench.py's multi-core CPU baseline)."""
    global _OMP, _LIB
    _OMP = bool(on)
    _LIB = None


def threads():
    return int(lib().orc_threads())


def lib():
    global _LIB
    if _LIB is None:
        name = "librgbdslam_oracle_omp.so" if _OMP else "librgbdslam_oracle.so"
        if name not in _LIBS:
            path = os.path.join(_HERE, name)
            if not os.path.exists(path):
                build()
            h = C.CDLL(path)
            h.orc_det6.restype = C.c_float
            h.orc_overlap.restype = C.c_float
            h.orc_slam_create.restype = C.c_void_p
            h.orc_slam_buffer.restype = C.c_void_p
            _LIBS[name] = h
        _LIB = _LIBS[name]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class IcpParams(C.Structure):
    _fields_ = [("lvl", C.c_int), ("max_iter", C.c_int * 3), ("distThres", C.c_float),
                ("angleThres", C.c_float), ("thres_cvg_alpha", C.c_float), ("thres_cvg_tr", C.c_float)]


def icp_params(dist=0.05, angle=float(np.sin(np.deg2rad(20.0))), lvl=3):
    p = IcpParams()
    p.lvl = lvl
    p.max_iter[0], p.max_iter[1], p.max_iter[2] = 3, 2, 1
    p.distThres, p.angleThres = dist, angle
    p.thres_cvg_alpha = p.thres_cvg_tr = 1e-6
    return p


# ---- stage 1 ---------------------------------------------------------------------------------
def vertex_map_kinect(depth, calib):
    n, m = depth.shape
    depth = np.ascontiguousarray(depth, np.uint16)
    out = np.empty((n, m, 3), np.float32)
    lib().orc_vertex_map_kinect(_p(out), _p(depth), C.c_size_t(m), _p(_f32(calib)), n, m)
    return out


def cpy_char3(bgr):
    n, m, _ = bgr.shape
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty((n, m, 3), np.float32)
    lib().orc_cpy_char3(_p(bgr), C.c_size_t(3 * m), _p(out), n, m)
    return out


def compute_normal(vmap, inverse=True):
    n, m, _ = vmap.shape
    out = np.empty((n, m, 3), np.float32)
    lib().orc_compute_normal(_p(out), _p(_f32(vmap)), n, m, int(inverse))
    return out


def intensity_sobel(rgb):
    n, m, _ = rgb.shape
    img = np.empty((n, m), np.float32)
    gx = np.empty((n, m), np.float32)
    gy = np.empty((n, m), np.float32)
    lib().orc_intensity(_p(img), _p(_f32(rgb)), n, m)
    lib().orc_sobel(_p(img), _p(gx), _p(gy), n, m)
    return img, gx, gy


# ---- stage 2/3 ------------------------------------------------------------------------------
def icp_grid(n, m, fact):
    gr, gc = C.c_int(), C.c_int()
    lib().orc_icp_grid(n, m, fact, C.byref(gr), C.byref(gc))
    return gr.value, gc.value


def icp_match(intr, R, t, vmap, nmap, rgb, gx, gy, vprev, nprev, rgbd_prev, dist, angle, fact, want_corr=True):
    n, m, _ = vmap.shape
    gr, gc = icp_grid(n, m, fact)
    buf = np.zeros((27, gr * gc), np.float32)
    corr = np.empty((n, m), np.int32) if want_corr else None
    lib().orc_icp_match(_p(_f32(intr)), _p(_f32(R)), _p(_f32(t)), _p(_f32(vmap)), _p(_f32(nmap)), _p(_f32(rgb)),
                        _p(_f32(gx)), _p(_f32(gy)), _p(_f32(vprev)), _p(_f32(nprev)), _p(_f32(rgbd_prev)),
                        C.c_float(dist), C.c_float(angle), n, m, fact, _p(buf), _p(corr))
    return buf, corr


def icp_reduce(buf):
    out = np.empty(27, np.float32)
    lib().orc_icp_reduce(_p(_f32(buf)), buf.shape[1], _p(out))
    return out


def estimate_system(intr, R, t, vmap, nmap, rgb, gx, gy, vprev, nprev, rgbd_prev, dist, angle, fact):
    n, m, _ = vmap.shape
    A = np.empty((6, 6), np.float32)
    b = np.empty(6, np.float32)
    s = np.empty(27, np.float32)
    lib().orc_estimate_system(_p(_f32(intr)), _p(_f32(R)), _p(_f32(t)), _p(_f32(vmap)), _p(_f32(nmap)), _p(_f32(rgb)),
                              _p(_f32(gx)), _p(_f32(gy)), _p(_f32(vprev)), _p(_f32(nprev)), _p(_f32(rgbd_prev)),
                              C.c_float(dist), C.c_float(angle), n, m, fact, _p(A), _p(b), _p(s))
    return A, b, s


def det6(A):
    return float(lib().orc_det6(_p(_f32(A))))


def llt_solve6(A, b):
    x = np.empty(6, np.float32)
    rc = lib().orc_llt_solve6(_p(_f32(A)), _p(_f32(b)), _p(x))
    return rc, x


def inverse3(R_cm):
    o = np.empty(9, np.float32)
    lib().orc_inverse3(_p(_f32(R_cm)), _p(o))
    return o


def align_gauss_newton(prm, intr, vmap, nmap, rgb, vprev, nprev, rgbd_prev, poseR, poset, poseRPrev, posetPrev):
    """Returns (ok, poseR(9, col-major), poset(3), trace[iters,33])."""
    n, m, _ = vmap.shape
    R = _f32(poseR).copy()
    t = _f32(poset).copy()
    trace = np.zeros((16, 33), np.float32)
    nit = C.c_int()
    ok = lib().orc_align_gauss_newton(C.byref(prm), _p(_f32(intr)), _p(_f32(vmap)), _p(_f32(nmap)), _p(_f32(rgb)),
                                      _p(_f32(vprev)), _p(_f32(nprev)), _p(_f32(rgbd_prev)), n, m, _p(R), _p(t),
                                      _p(_f32(poseRPrev)), _p(_f32(posetPrev)), _p(trace), C.byref(nit))
    return bool(ok), R, t, trace[:nit.value]


# ---- stage 4 ----------------------------------------------------------------------------------
def merge_depth_off(rgbd, depth, rgb, mask, calib):
    n, m = depth.shape
    rgbd = _f32(rgbd).copy()
    mask = np.ascontiguousarray(mask, np.uint8).copy()
    depth = np.ascontiguousarray(depth, np.uint16)
    lib().orc_merge_depth_off(_p(rgbd), _p(depth), C.c_size_t(m), _p(_f32(rgb)), _p(mask), _p(_f32(calib)), n, m)
    return rgbd, mask


def vertex_map_gl(rgbd, calib):
    n, m, _ = rgbd.shape
    out = np.empty((n, m, 3), np.float32)
    lib().orc_vertex_map_gl(_p(out), _p(_f32(rgbd)), _p(_f32(calib)), n, m)
    return out


def _ptr_table(arrs):
    tab = (C.c_void_p * len(arrs))()
    for k, a in enumerate(arrs):
        tab[k] = a.ctypes.data
    return tab


def update(bumps, rgbs, masks, params, claims, stamp, vmap, rgbd, mask, label, pose, plane_lo=0, plane_hi=None):
    """In-place on the lists of per-plane numpy arrays."""
    n, m, _ = vmap.shape
    N_prim, M_prim = masks[0].shape
    if plane_hi is None:
        plane_hi = len(masks)
    lib().orc_update(_ptr_table(bumps), _ptr_table(rgbs), _ptr_table(masks), _ptr_table(params), _ptr_table(claims),
                     C.c_uint32(stamp), _p(_f32(vmap)), _p(_f32(rgbd)), _p(np.ascontiguousarray(mask, np.uint8)),
                     _p(np.ascontiguousarray(label, np.uint8)), _p(_f32(pose)), N_prim, M_prim, n, m,
                     plane_lo, plane_hi)


def vertex_from_bump(bump, rgb_bump, mask, param, lvl=1):
    n, m = mask.shape
    no, mo = n // lvl, m // lvl
    v = np.empty((no, mo, 3), np.float32)
    c = np.empty((no, mo, 4), np.float32)
    mk = np.empty((no, mo, 4), np.float32)
    lib().orc_vertex_from_bump(_p(v), _p(c), _p(mk), _p(np.ascontiguousarray(bump, np.uint16)),
                               _p(np.ascontiguousarray(rgb_bump, np.uint8)), _p(np.ascontiguousarray(mask, np.uint8)),
                               _p(_f32(param)), n, m, lvl)
    return v, c, mk


def vertex_from_bump_d2(bump, rgb_bump, mask, param, lvl=1, variant=0):
    """DemoStep2's decode (D2/Primitive.cu:31-179): variant 0 -> vertices of the sub-sampled grid (u16 mask),
    variant 1 -> vertices + RGBA of the full grid (u8 mask)."""
    n, m = mask.shape
    no, mo = (n, m) if variant else (n // lvl, m // lvl)
    v = np.empty((no, mo, 3), np.float32)
    c = np.empty((no, mo, 4), np.float32)
    mk = np.ascontiguousarray(mask, np.uint8 if variant else np.uint16)
    lib().orc_vertex_from_bump_d2(_p(v), _p(c), _p(np.ascontiguousarray(bump, np.uint16)),
                                  _p(np.ascontiguousarray(rgb_bump, np.uint8)), _p(mk), _p(_f32(param)), n, m, lvl, variant)
    return (v, c) if variant else v


# ---- stage 5 ----------------------------------------------------------------------------------
def transform_vertex_map(pose, vmap, rgbd):
    n, m, _ = vmap.shape
    v = _f32(vmap).copy()
    r = _f32(rgbd).copy()
    lib().orc_transform_vertex_map(_p(_f32(pose)), _p(v), _p(r), n, m)
    return v, r


def quad_trim_frame(vmap, thresh=0.1):
    n, m, _ = vmap.shape
    idx = np.empty(((n - 1) * (m - 1), 4), np.uint32)
    lib().orc_quad_trim_frame(_p(idx), _p(_f32(vmap)), n, m, C.c_float(thresh))
    return idx


def quad_trim(vmap, mask, min_conf=0, thresh=0.1, lvl=1, variant=0, n=None, m=None):
    """QuadTrim of a plane mesh; vmap holds the (n/lvl) x (m/lvl) decoded vertices, mask the plane's n x m bytes."""
    if n is None:
        n, m = mask.shape
    idx = np.empty(((n // lvl - 1) * (m // lvl - 1), 4), np.uint32)
    lib().orc_quad_trim(_p(idx), _p(_f32(vmap)), _p(np.ascontiguousarray(mask, np.uint8)), C.c_ushort(min_conf), C.c_uint(n), C.c_uint(m),
                        C.c_float(thresh), int(lvl), int(variant))
    return idx


def raster(vmap, rgbd, calib, index=None, thresh=0.1, znear=0.5, zfar=10.0):
    n, m, _ = vmap.shape
    tex = np.empty((n, m, 4), np.float32)
    zbuf = np.empty((n, m), np.uint64)
    if index is not None:
        index = np.ascontiguousarray(index, np.uint32)
    lib().orc_raster(_p(tex), _p(zbuf), _p(_f32(vmap)), _p(_f32(rgbd)), _p(index), C.c_float(thresh),
                     _p(_f32(calib)), C.c_float(znear), C.c_float(zfar), n, m)
    return tex, zbuf


def read_frame(tex, calib):
    n, m, _ = tex.shape
    v = np.empty((n, m, 3), np.float32)
    r = np.empty((n, m, 4), np.float32)
    lib().orc_read_frame(_p(v), _p(r), _p(_f32(tex)), _p(_f32(calib)), n, m)
    return v, r


def project_mask(pose, calib, mask, vmap):
    n, m, _ = vmap.shape
    out = np.empty((n, m), np.uint8)
    lib().orc_project_mask(_p(_f32(pose)), _p(_f32(calib)), _p(np.ascontiguousarray(mask, np.uint8)), _p(out),
                           _p(_f32(vmap)), n, m)
    return out


def segment(label, vmap, nmap, pose, equations, eps=0.1, alpha=0.9):
    """3 x SegmentKernel; a uint16 label image goes through the 16-bit form (the build's extension, same statements)."""
    n, m, _ = vmap.shape
    wide = np.asarray(label).dtype == np.uint16
    lab = np.ascontiguousarray(label, np.uint16 if wide else np.uint8).copy()
    swap = np.empty_like(lab)
    eq = _f32(equations)
    (lib().orc_segment_l16 if wide else lib().orc_segment)(_p(lab), _p(swap), _p(_f32(vmap)), _p(_f32(nmap)), _p(_f32(pose)), _p(eq),
                                                           eq.size // 10, C.c_float(eps), C.c_float(alpha), n, m)
    return lab


def project_label_l16(pose, calib, label16, vmap):
    n, m, _ = vmap.shape
    out = np.empty((n, m), np.uint16)
    lib().orc_project_label_l16(_p(_f32(pose)), _p(_f32(calib)), _p(np.ascontiguousarray(label16, np.uint16)), _p(out),
                                _p(_f32(vmap)), n, m)
    return out


def add_zbuff_off(depth, rgb, calib):
    n, m = depth.shape
    out = np.empty((n, m, 4), np.float32)
    lib().orc_add_zbuff_off(_p(np.ascontiguousarray(depth, np.uint16)), C.c_size_t(m), _p(_f32(rgb)), _p(out),
                            _p(_f32(calib)), n, m)
    return out


def init_mask(rgbd):
    n, m, _ = rgbd.shape
    out = np.empty((n, m), np.uint8)
    lib().orc_init_mask(_p(out), _p(_f32(rgbd)), n, m)
    return out


def seed_labels(vmap, nmap, pose, equations, eps=0.1, alpha=0.8, wide=False):
    n, m, _ = vmap.shape
    out = np.empty((n, m), np.uint16 if wide else np.uint8)
    eq = _f32(equations)
    (lib().orc_seed_labels_l16 if wide else lib().orc_seed_labels)(_p(out), _p(_f32(vmap)), _p(_f32(nmap)), _p(_f32(pose)), _p(eq),
                                                                   eq.size // 10, C.c_float(eps), C.c_float(alpha), n, m)
    return out


# ---- whole loop -------------------------------------------------------------------------------
class Slam:
    """CPU restatement of the display() loop (D1/GP-demo.cpp:138-296)."""

    def __init__(self, n, m, calib, prm=None, max_planes=50, N_prim=1024, M_prim=1024, znear=0.5, zfar=10.0):
        self.n, self.m, self.N_prim, self.M_prim = n, m, N_prim, M_prim
        self.prm = prm or icp_params()
        self._lib = lib()                                     # the library (single-thread or OpenMP) current at creation
        self._h = C.c_void_p(self._lib.orc_slam_create(n, m, _p(_f32(calib)), C.byref(self.prm), max_planes, N_prim,
                                                   M_prim, C.c_float(znear), C.c_float(zfar)))

    def close(self):
        if self._h:
            self._lib.orc_slam_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def detect_planes(self, min_size=5000, min_normals=1000, max_planes=50):
        """RG::Finish first branch: DetectPrimitives + CreatePrimitives -> (equations [k,10], centers [k,3])."""
        eq = np.zeros((max_planes, 10), np.float32)
        ce = np.zeros((max_planes, 3), np.float32)
        self._lib.orc_slam_detect_planes.restype = C.c_int
        nb = self._lib.orc_slam_detect_planes(self._h, int(min_size), int(min_normals), _p(eq), _p(ce))
        return eq[:nb], ce[:nb]

    def set_bilateral(self, on):
        self._lib.orc_slam_set_bilateral(self._h, int(on))

    def set_label16(self, on):
        """u16 label images (the build's extension for models of more than 255 planes); before any plane exists."""
        self._lib.orc_slam_set_label16(self._h, int(on))
        self._label16 = bool(on)

    def set_finish(self, enabled=True, first_after=30, period=100, min_size=-1, min_normals=-1):
        """RG::Finish schedule (D1/RG.cpp:149-271)."""
        self._lib.orc_slam_set_finish(self._h, int(enabled), int(first_after), int(period), int(min_size), int(min_normals))

    def finish_detections(self):
        return int(self._lib.orc_slam_finish_detections(self._h))

    def nb_planes(self):
        return int(self._lib.orc_slam_nb_planes(self._h))

    def set_planes(self, equations, centers):
        eq = _f32(equations)
        self._lib.orc_slam_set_planes(self._h, _p(eq), _p(_f32(centers)), eq.size // 10)

    def step(self, depth, bgr):
        return self._lib.orc_slam_step(self._h, _p(np.ascontiguousarray(depth, np.uint16)),
                                   _p(np.ascontiguousarray(bgr, np.uint8)))

    def pose(self):
        R = np.empty(9, np.float32)
        t = np.empty(3, np.float32)
        self._lib.orc_slam_get_pose(self._h, _p(R), _p(t))
        return R, t

    def buffer(self, name):
        nb = C.c_size_t()
        ptr = self._lib.orc_slam_buffer(self._h, name.encode(), C.byref(nb))
        if not ptr:
            raise KeyError(name)
        raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nb.value,)).copy()
        P, Cc = self.n * self.m, self.N_prim * self.M_prim
        if name in ("VMap", "NMap", "frame.VMap", "frame.NMap", "frame.RGB"):
            return raw.view(np.float32).reshape(self.n, self.m, 3)
        if name == "RGBD":
            return raw.view(np.float32).reshape(self.n, self.m, 4)
        if name == "Label" and raw.size == 2 * P:
            return raw.view(np.uint16).reshape(self.n, self.m)
        if name in ("Mask", "Label"):
            return raw.reshape(self.n, self.m)
        if name.startswith("Bump"):
            return raw.view(np.uint16).reshape(self.N_prim, self.M_prim, 3)
        if name.startswith("RGBb"):
            return raw.reshape(self.N_prim, self.M_prim, 3)
        return raw.reshape(self.N_prim, self.M_prim)


# ---- plane detection ---------------------------------------------------------------
def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def cpy_n(nmap, chnl):
    n, m, _ = nmap.shape
    out = np.empty((n, m), np.uint8)
    lib().orc_cpy_n(_p(out), _p(_f32(nmap)), n, m, int(chnl))
    return out


def canny(img, low=120.0, high=30.0):
    n, m = img.shape
    out = np.empty((n, m), np.uint8)
    lib().orc_canny(_p(_u8(img)), _p(out), C.c_double(low), C.c_double(high), n, m)
    return out


def edge_sum(r, g, b):
    n, m = r.shape
    out = np.empty((n, m), np.uint8)
    lib().orc_sum(_p(out), _p(_u8(r)), _p(_u8(g)), _p(_u8(b)), n, m)
    return out


def discont(mat, vmap, thresh=0.03):
    n, m = mat.shape
    out = _u8(mat).copy()
    lib().orc_discont(_p(out), _p(_f32(vmap)), n, m, C.c_float(thresh))
    return out


def thresh(img, label, t=20):
    n, m = img.shape
    out = np.empty((n, m), np.uint8)
    lib().orc_thresh(_p(out), _p(_u8(img)), C.c_uint8(t), _p(_u8(label)), n, m)
    return out


def erode_ellipse5(img):
    n, m = img.shape
    out = np.empty((n, m), np.uint8)
    lib().orc_erode_ellipse5(_p(out), _p(_u8(img)), n, m)
    return out


def find_blobs(binary):
    n, m = binary.shape
    blob = np.empty((n, m), np.int32)
    sizes = np.empty((n, m), np.int32)
    lib().orc_find_blobs(_p(_u8(binary)), _p(blob), _p(sizes), n, m)
    return blob, sizes


def fit_planes(blob, sizes, vmap, nmap, pose, label, min_size=5000, min_normals=1000, max_planes=50):
    n, m = blob.shape
    eq = np.zeros((max_planes, 10), np.float32)
    ce = np.zeros((max_planes, 3), np.float32)
    lab = _u8(label).copy()
    lib().orc_fit_planes.restype = C.c_int
    nb = lib().orc_fit_planes(_p(np.ascontiguousarray(blob, np.int32)), _p(np.ascontiguousarray(sizes, np.int32)), _p(_f32(vmap)),
                              _p(_f32(nmap)), _p(_f32(pose)), _p(lab), _p(eq), _p(ce), n, m, int(min_size), int(min_normals),
                              int(max_planes))
    return nb, eq[:nb], ce[:nb], lab


def detect_regions(nmap, vmap, label=None):
    """RG.cpp:293-338 up to the blob list: -> (blob, sizes, binary)."""
    n, m, _ = nmap.shape
    e = [canny(cpy_n(nmap, c)) for c in range(3)]
    img = discont(edge_sum(*e), vmap, 0.03)
    img2 = thresh(img, np.zeros((n, m), np.uint8) if label is None else label, 20)
    binary = (erode_ellipse5(img2) > 0).astype(np.uint8)
    blob, sizes = find_blobs(binary)
    return blob, sizes, binary


# ---- the remaining operator functions of the reference headers (csrc/compat.hip) ---------------------
def merge_depth_kinect(rgbd, depth, rgb, mask, calib):
    n, m = depth.shape
    rgbd = _f32(rgbd).copy()
    mask = np.ascontiguousarray(mask, np.uint8).copy()
    lib().orc_merge_depth_kinect(_p(rgbd), _p(np.ascontiguousarray(depth, np.uint16)), _p(_f32(rgb)), _p(mask), _p(_f32(calib)), n, m)
    return rgbd, mask


def add_zbuff_kinect(depth, rgb, calib):
    n, m = depth.shape
    out = np.empty((n, m, 4), np.float32)
    lib().orc_add_zbuff_kinect(_p(np.ascontiguousarray(depth, np.uint16)), _p(_f32(rgb)), _p(out), _p(_f32(calib)), n, m)
    return out


def gpu_cpy(rgb):
    n, m, _ = rgb.shape
    out = np.empty((n, m), np.float32)
    lib().orc_gpu_cpy(_p(out), C.c_size_t(m), _p(_f32(rgb)), n, m)
    return out


def map_data_2_vbo(color_bgra, depth, coord, calib):
    n, m = depth.shape
    v = np.zeros((n, m, 3), np.float32)
    rgb = np.zeros((n, m, 3), np.float32)
    lib().orc_map_data_2_vbo(_p(np.ascontiguousarray(color_bgra, np.uint8)), _p(np.ascontiguousarray(depth, np.uint16)),
                             _p(np.ascontiguousarray(coord, np.int32)), _p(v), _p(rgb), _p(_f32(calib)), n, m)
    return v, rgb


def vote_plan(label, vmap, pose, nb, centers=None, count=None):
    n, m = label.shape
    centers = np.zeros(3 * nb, np.float32) if centers is None else _f32(centers).copy()
    count = np.zeros(nb, np.float32) if count is None else _f32(count).copy()
    lib().orc_vote_plan(_p(np.ascontiguousarray(label, np.uint8)), _p(_f32(vmap)), _p(centers), _p(count), _p(_f32(pose)), nb, n, m)
    return centers, count


def affect_plan(label, buff):
    n, m = label.shape
    lab = np.ascontiguousarray(label, np.uint8).copy()
    lib().orc_affect_plan(_p(lab), _p(np.ascontiguousarray(buff, np.int32)), n, m)
    return lab


def overlap(bump, mask, param, param_in, N_prim, M_prim):
    n, m = mask.shape
    cnt = np.zeros(2, np.uint32)
    r = lib().orc_overlap(_p(np.ascontiguousarray(bump, np.uint16)), _p(np.ascontiguousarray(mask, np.uint8)), _p(_f32(param)),
                          _p(_f32(param_in)), n, m, N_prim, M_prim, _p(cnt))
    return float(r), cnt


# ---- 16-bit labels (models of more than 255 planes) ------------------------------------------------------
def update_l16(bumps, rgbs, masks, params, claims, stamp, vmap, rgbd, mask, label16, pose, plane_lo=0, plane_hi=None, size=None):
    """In-place on the lists of per-plane numpy arrays (size = (N_prim, M_prim) when masks[0] is only a stand-in)."""
    n, m, _ = vmap.shape
    N_prim, M_prim = size if size is not None else masks[0].shape
    if plane_hi is None:
        plane_hi = len(masks)
    lib().orc_update_l16(_ptr_table(bumps), _ptr_table(rgbs), _ptr_table(masks), _ptr_table(params), _ptr_table(claims),
                         C.c_uint32(stamp), _p(_f32(vmap)), _p(_f32(rgbd)), _p(np.ascontiguousarray(mask, np.uint8)),
                         _p(np.ascontiguousarray(label16, np.uint16)), _p(_f32(pose)), N_prim, M_prim, n, m, plane_lo, plane_hi)


def label_planes_l16(vmap, nmap, pose, params, N_prim, M_prim, eps=0.1, alpha=0.8):
    n, m, _ = vmap.shape
    out = np.empty((n, m), np.uint16)
    params = [_f32(p) for p in params]
    lib().orc_label_planes_l16(_p(out), _p(_f32(vmap)), _p(_f32(nmap)), _p(_f32(pose)), _ptr_table(params), len(params),
                               N_prim, M_prim, C.c_float(eps), C.c_float(alpha), n, m)
    return out


# ---- prediction from the planes ---------------------------------------------------------------------------
def planes_draw(zbuf, bumps, rgbs, masks, params, plane_lo, plane_hi, pose_w2c, calib, n, m, min_conf=0, thresh=0.1, znear=0.5, zfar=10.0,
                size=None):
    """In place on zbuf (np.uint64 [n, m], start from 0xFFFF...)."""
    N, M = size if size is not None else masks[0].shape
    lib().orc_planes_draw(_p(zbuf), _ptr_table(bumps), _ptr_table(rgbs), _ptr_table(masks), _ptr_table(params), plane_lo, plane_hi,
                          _p(_f32(pose_w2c)), N, M, int(min_conf), C.c_float(thresh), _p(_f32(calib)), C.c_float(znear), C.c_float(zfar), n, m)


def planes_resolve(zbuf, vmap, rgbd, bumps, rgbs, masks, params, plane_lo, plane_hi, pose_w2c, calib, size=None):
    """In place on vmap [n, m, 3] / rgbd [n, m, 4] (float32)."""
    n, m = zbuf.shape
    N, M = size if size is not None else masks[0].shape
    lib().orc_planes_resolve(_p(zbuf), _p(vmap), _p(rgbd), _ptr_table(bumps), _ptr_table(rgbs), _ptr_table(masks), _ptr_table(params),
                             plane_lo, plane_hi, _p(_f32(pose_w2c)), N, M, _p(_f32(calib)), n, m)


def bilateral_depth(depth, d=5, sigma_color=100.0, sigma_space=100.0):
    n, m = depth.shape
    src = np.ascontiguousarray(depth, np.uint16)
    out = np.empty_like(src)
    lib().orc_bilateral_depth(_p(out), _p(src), C.c_size_t(m), int(d), C.c_double(sigma_color), C.c_double(sigma_space), n, m)
    return out
