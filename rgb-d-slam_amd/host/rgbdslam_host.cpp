// rgbdslam_host.cpp -- C++ host mirror of the reference's L3/L4 classes (see rgbdslam_host.hpp).
// No kernels here: all device work goes through the C ABI of include/rgbdslam.h.
#include <rocprofiler-sdk-roctx/roctx.h>
#include "rgbdslam_host.hpp"

#include <algorithm>
#include <chrono>
#include <atomic>
#include <future>
#include <thread>
#include <stdio.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace slam {

// roctx ranges round the five stages of the hot path (the reference times them with clock() behind TIMING,
// GP-demo.cpp:153-277): visible in `rocprofv3 --marker-trace`, ~16 ns per range when no tool is attached.
struct StageRange {
    explicit StageRange(const char *name) { roctxRangePush(name); }
    ~StageRange() { roctxRangePop(); }
};



void check(int rc, const char *what) {
    if (rc != 0) throw Error(what, rc);
}
void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e), -(int)e);
}

// ---- small dense algebra (Eigen stand-ins; same operation order as the oracle's) ----------------
Mat3 Mat3::Identity() {
    Mat3 r;
    for (int k = 0; k < 9; k++) r.a[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    return r;
}
Mat3 Mat3::operator*(const Mat3 &b) const {
    Mat3 r;
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 3; i++)
            r.a[c * 3 + i] = ((*this)(i, 0) * b(0, c) + (*this)(i, 1) * b(1, c)) + (*this)(i, 2) * b(2, c);
    return r;
}
Mat3 Mat3::inverse() const {
    const Mat3 &m = *this;
    float cof[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            cof[i][j] = m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
        }
    float det = (cof[0][0] * m(0, 0) + cof[1][0] * m(1, 0)) + cof[2][0] * m(2, 0);
    float invdet = 1.0f / det;
    Mat3 o;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o(i, j) = cof[j][i] * invdet;
    return o;
}
Mat3 Mat3::neg() const {
    Mat3 r;
    for (int k = 0; k < 9; k++) r.a[k] = -a[k];
    return r;
}
float Vec3::norm() const { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
Vec3 operator*(const Mat3 &m, const Vec3 &x) {
    Vec3 r;
    for (int i = 0; i < 3; i++) r.v[i] = (m(i, 0) * x.v[0] + m(i, 1) * x.v[1]) + m(i, 2) * x.v[2];
    return r;
}
float norm_minus_identity(const Mat3 &m) {
    float s = 0.0f;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) {
            float v = m(r, c) - (r == c ? 1.0f : 0.0f);
            s = s + v * v;
        }
    return sqrtf(s);
}

void quaternion2matrix(double *q, double m[3][3]) {   // D1/Utilities.cpp:1145-1167
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q03 = q[0] * q[3], q13 = q[1] * q[3], q23 = q[2] * q[3];
    double q02 = q[0] * q[2], q12 = q[1] * q[2], q01 = q[0] * q[1];
    m[0][0] = q00 + q11 - q22 - q33;
    m[1][1] = q00 - q11 + q22 - q33;
    m[2][2] = q00 - q11 - q22 + q33;
    m[0][1] = 2.0 * (q12 - q03);
    m[1][0] = 2.0 * (q12 + q03);
    m[0][2] = 2.0 * (q13 + q02);
    m[2][0] = 2.0 * (q13 - q02);
    m[1][2] = 2.0 * (q23 - q01);
    m[2][1] = 2.0 * (q23 + q01);
}

void pack_pose(const Mat3 &R, const Vec3 &t, float *P) {   // D1/GP-demo.cpp:206-211
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) P[j * 4 + i] = R(i, j);
    for (int i = 0; i < 3; i++) P[12 + i] = t(i);
    P[3] = 0.0f; P[7] = 0.0f; P[11] = 0.0f; P[15] = 1.0f;
}

float det6(const float *Ain) {
    float a[36];
    memcpy(a, Ain, sizeof(a));
    float det = 1.0f;
    for (int c = 0; c < 6; c++) {
        int p = c;
        float best = fabsf(a[c * 6 + c]);
        for (int r = c + 1; r < 6; r++) {
            float v = fabsf(a[r * 6 + c]);
            if (v > best) { best = v; p = r; }
        }
        if (p != c) {
            for (int k = 0; k < 6; k++) { float tmp = a[c * 6 + k]; a[c * 6 + k] = a[p * 6 + k]; a[p * 6 + k] = tmp; }
            det = -det;
        }
        float piv = a[c * 6 + c];
        det = det * piv;
        if (piv == 0.0f || piv != piv) return det;
        for (int r = c + 1; r < 6; r++) {
            float f = a[r * 6 + c] / piv;
            for (int k = c + 1; k < 6; k++) a[r * 6 + k] = a[r * 6 + k] - f * a[c * 6 + k];
        }
    }
    return det;
}

bool llt_solve6(const float *A, const float *b, float *x) {
    float L[36];
    memset(L, 0, sizeof(L));
    for (int k = 0; k < 6; k++) {
        float s = A[k * 6 + k];
        for (int p = 0; p < k; p++) s = s - L[k * 6 + p] * L[k * 6 + p];
        if (!(s > 0.0f)) return false;
        float lkk = sqrtf(s);
        L[k * 6 + k] = lkk;
        for (int r = k + 1; r < 6; r++) {
            float v = A[r * 6 + k];
            for (int p = 0; p < k; p++) v = v - L[r * 6 + p] * L[k * 6 + p];
            L[r * 6 + k] = v / lkk;
        }
    }
    float y[6];
    for (int r = 0; r < 6; r++) {
        float v = b[r];
        for (int p = 0; p < r; p++) v = v - L[r * 6 + p] * y[p];
        y[r] = v / L[r * 6 + r];
    }
    for (int r = 5; r >= 0; r--) {
        float v = y[r];
        for (int p = r + 1; p < 6; p++) v = v - L[p * 6 + r] * x[p];
        x[r] = v / L[r * 6 + r];
    }
    return true;
}

// ---- device buffers -----------------------------------------------------------------------------
template <typename T>
void DevBuf<T>::alloc(size_t n, bool zero) {
    release();
    hip_check(hipMalloc((void **)&p, n * sizeof(T)), "hipMalloc");
    count = n;
    if (zero) hip_check(hipMemset(p, 0, n * sizeof(T)), "hipMemset");
}
template <typename T>
void DevBuf<T>::release() {
    if (p) { (void)hipFree(p); p = nullptr; count = 0; }
}
template struct DevBuf<float>;
template struct DevBuf<uint8_t>;
template struct DevBuf<uint16_t>;
template struct DevBuf<uint32_t>;
template struct DevBuf<uint64_t>;
template struct DevBuf<int32_t>;
template struct DevBuf<void *>;

static hipStream_t ctx_stream(rgbdslam_ctx *ctx) {
    void *s = nullptr;
    check(rgbdslam_get_stream(ctx, &s), "rgbdslam_get_stream");
    return (hipStream_t)s;
}

// ---- InputFrame / OffLineFrame (D1/Frame.h:19-134, D1/Frame.cpp:43-77) ----------------------------
InputFrame::InputFrame(rgbdslam_ctx *ctx, int n, int m) : _ctx(ctx), _n(n), _m(m) {
    size_t P = (size_t)n * m;
    _VMap_dev[0].alloc(3 * P);
    _NMap_dev[0].alloc(3 * P);
    _RGB_dev[0].alloc(3 * P);
}

OffLineFrame::OffLineFrame(rgbdslam_ctx *ctx, int n, int m) : InputFrame(ctx, n, m) {
    _depth_dev_test[0].alloc((size_t)n * m);
    _color_dev_test[0].alloc(3 * (size_t)n * m);
}
OffLineFrame::~OffLineFrame() {
    for (int k = 0; k < 2; k++) {
        if (_ready[k]) (void)hipEventDestroy(_ready[k]);
        if (_free[k]) (void)hipEventDestroy(_free[k]);
        if (_sig_ready[k]) (void)hipFree(_sig_ready[k]);
        if (_sig_free[k]) (void)hipFree(_sig_free[k]);
    }
}

void OffLineFrame::EnableGradients(bool on) {
    _gradients = on;
    if (!on) return;
    for (int k = 0; k < 2; k++)
        if (_VMap_dev[k].p && !_Gx[k].p) { _Gx[k].alloc((size_t)_n * _m); _Gy[k].alloc((size_t)_n * _m); }
}

void OffLineFrame::AttachLoadStream(hipStream_t load_stream) {
    if (_loaded_once) throw Error("AttachLoadStream: frames were already loaded", RGBDSLAM_E_STATE);
    _load_stream = load_stream;
    if (!load_stream) return;
    size_t P = (size_t)_n * _m;
    if (!_VMap_dev[1].p) {
        _VMap_dev[1].alloc(3 * P); _NMap_dev[1].alloc(3 * P); _RGB_dev[1].alloc(3 * P);
        _depth_dev_test[1].alloc(P); _color_dev_test[1].alloc(3 * P);
    }
    if (_gradients) EnableGradients(true);
    // stream value operations need signal memory: when the allocation is refused, fall back to events (2.5x the cost per
    // hand-over, the same ordering) instead of failing
    if (!_use_events) {
        for (int k = 0; k < 2 && !_use_events; k++) {
            if (_sig_ready[k]) continue;
            if (hipExtMallocWithFlags((void **)&_sig_ready[k], 8, hipMallocSignalMemory) != hipSuccess ||     // 8-byte allocations
                hipExtMallocWithFlags((void **)&_sig_free[k], 8, hipMallocSignalMemory) != hipSuccess) {
                (void)hipGetLastError();
                _use_events = true;
                break;
            }
            hip_check(hipMemset(_sig_ready[k], 0, 8), "hipMemset");
            hip_check(hipMemset(_sig_free[k], 0, 8), "hipMemset");
        }
        if (_use_events)
            for (int k = 0; k < 2; k++) {
                if (_sig_ready[k]) { (void)hipFree(_sig_ready[k]); _sig_ready[k] = nullptr; }
                if (_sig_free[k]) { (void)hipFree(_sig_free[k]); _sig_free[k] = nullptr; }
            }
    }
    if (_use_events)
        for (int k = 0; k < 2; k++) {
            if (!_ready[k]) hip_check(hipEventCreateWithFlags(&_ready[k], hipEventDisableTiming), "hipEventCreate");
            if (!_free[k]) hip_check(hipEventCreateWithFlags(&_free[k], hipEventDisableTiming), "hipEventCreate");
        }
}

void OffLineFrame::wait_ready(hipStream_t consumer) {
    if (!_load_stream) return;
    if (_use_events) hip_check(hipStreamWaitEvent(consumer, _ready[_cur], 0), "wait for the preprocessed frame");
    else hip_check(hipStreamWaitValue32(consumer, _sig_ready[_cur], _seq_ready[_cur], hipStreamWaitValueGte, 0xFFFFFFFFu),
                   "wait for the preprocessed frame");
}

void OffLineFrame::release(hipStream_t consumer) {
    if (!_load_stream) return;
    if (_use_events) hip_check(hipEventRecord(_free[_cur], consumer), "hipEventRecord");
    else hip_check(hipStreamWriteValue32(consumer, _sig_free[_cur], ++_seq_free[_cur], 0), "hipStreamWriteValue32");
    _free_recorded[_cur] = true;
}

// with a load stream: switch to the other buffer set and make the stream wait until that set's last reader is done
void OffLineFrame::begin_load() {
    if (_load_stream && _loaded_once) {
        _cur ^= 1;
        if (_free_recorded[_cur]) {
            if (_use_events) hip_check(hipStreamWaitEvent(_load_stream, _free[_cur], 0), "hipStreamWaitEvent");
            else hip_check(hipStreamWaitValue32(_load_stream, _sig_free[_cur], _seq_free[_cur], hipStreamWaitValueGte, 0xFFFFFFFFu),
                           "hipStreamWaitValue32");
        }
    }
    _loaded_once = true;
}

void OffLineFrame::LoadFrame(const uint16_t *depth_host, const uint8_t *bgr_host) {
    size_t P = (size_t)_n * _m;
    if (_pre.valid && !_pre.other) _pre.valid = false;       // this load overwrites the set a rider preprocessed in place
    begin_load();
    hipStream_t s = _load_stream ? _load_stream : ctx_stream(_ctx);
    if (!_depth_dev_test[_cur].p) { _depth_dev_test[_cur].alloc(P); _color_dev_test[_cur].alloc(3 * P); }   // (after a set switch)
    // GpuMat::upload x2 (Frame.cpp:56-60)
    hip_check(hipMemcpyAsync(_depth_dev_test[_cur].p, depth_host, 2 * P, hipMemcpyHostToDevice, s), "upload depth");
    hip_check(hipMemcpyAsync(_color_dev_test[_cur].p, bgr_host, 3 * P, hipMemcpyHostToDevice, s), "upload colour");
    _depth_cur[_cur] = _depth_dev_test[_cur].p;
    _depth_pitch[_cur] = 2 * (size_t)_m;
    preprocess(_color_dev_test[_cur].p, 3 * (size_t)_m);
}

void OffLineFrame::LoadFrame(const std::string &filename_depth, const std::string &filename_color) {
    auto ends_with = [](const std::string &a, const char *suf) { size_t k = strlen(suf); return a.size() >= k && !a.compare(a.size() - k, k, suf); };
    auto read = [&](const std::string &f) {
        if (ends_with(f, ".png") || ends_with(f, ".PNG")) return read_png(f);
        if (ends_with(f, ".tif") || ends_with(f, ".tiff")) return read_tiff(f);
        throw Error("unsupported image container: " + f, RGBDSLAM_E_ARG);
    };
    Image depth_test = read(filename_depth);                                        // cv::imread(..., UNCHANGED), Frame.cpp:48
    if (depth_test.rows != _n || depth_test.cols != _m || depth_test.channels != 1 || depth_test.bits != 16)
        throw Error("depth image is not " + std::to_string(_m) + "x" + std::to_string(_n) + " 16UC1: " + filename_depth, RGBDSLAM_E_ARG);
    Image color_test = read(filename_color);                                        // Frame.cpp:58
    if (color_test.rows != _n || color_test.cols != _m || color_test.channels != 3 || color_test.bits != 8)
        throw Error("colour image is not " + std::to_string(_m) + "x" + std::to_string(_n) + " 8UC3: " + filename_color, RGBDSLAM_E_ARG);
    LoadFrame((const uint16_t *)depth_test.data.data(), color_test.data.data());
}

void OffLineFrame::LoadFrameDevice(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch) {
    const bool done = _pre.valid && !_load_stream && _pre.depth == depth_dev && _pre.depth_pitch == depth_pitch && _pre.bgr == bgr_dev &&
                      _pre.bgr_pitch == bgr_pitch;
    const bool swap_sets = done && _pre.other;
    _pre.valid = false;
    begin_load();
    if (swap_sets) _cur ^= 1;                         // the frame's maps are in the other set already
    _depth_cur[_cur] = depth_dev;
    _depth_pitch[_cur] = depth_pitch;
    if (!done) preprocess(bgr_dev, bgr_pitch);
}

void OffLineFrame::MarkPreprocessed(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch, bool other_set) {
    _pre.depth = depth_dev; _pre.depth_pitch = depth_pitch; _pre.bgr = bgr_dev; _pre.bgr_pitch = bgr_pitch; _pre.valid = true;
    _pre.other = other_set;
}
void OffLineFrame::PrepareOtherSet() {
    if (_load_stream) throw Error("PrepareOtherSet: the load stream owns the second buffer set", RGBDSLAM_E_STATE);
    const int o = _cur ^ 1;
    const size_t P = (size_t)_n * _m;
    if (!_VMap_dev[o].p) { _VMap_dev[o].alloc(3 * P); _NMap_dev[o].alloc(3 * P); _RGB_dev[o].alloc(3 * P); }
    if (!_Gx[o].p) { _Gx[o].alloc(P); _Gy[o].alloc(P); }
}

void OffLineFrame::announce_aux(bool other) {
    const int set = _cur ^ (other ? 1 : 0);
    if (!_aux_enabled) { _aux_valid[set] = false; return; }
    if (!_Aux[set].p) _Aux[set].alloc(4 * (size_t)_n * _m);
    check(rgbdslam_set_frame_aux(_ctx, _Aux[set].p, nullptr), "set_frame_aux");
    _aux_valid[set] = true;
}

void OffLineFrame::preprocess(const uint8_t *bgr, size_t bgr_pitch) {
    // kernels are enqueued on the context's stream: point it at the load stream for the duration of the load
    hipStream_t main = ctx_stream(_ctx);
    if (_load_stream) check(rgbdslam_set_stream(_ctx, _load_stream), "set_stream");
    int rc;
    if (_bilateral) {                                                               // Frame.cpp:50-54 (optional stage)
        if (_depth_filtered[_cur].count != (size_t)_n * _m) _depth_filtered[_cur].alloc((size_t)_n * _m, false);
        rc = rgbdslam_bilateral_depth(_ctx, _depth_filtered[_cur].p, 2 * (size_t)_m, _depth_cur[_cur], _depth_pitch[_cur], 5, 100.0, 100.0, _n, _m);
        if (rc) { if (_load_stream) (void)rgbdslam_set_stream(_ctx, main); check(rc, "bilateral_depth"); }
        _depth_cur[_cur] = _depth_filtered[_cur].p;
        _depth_pitch[_cur] = 2 * (size_t)_m;
    }
    if (_gradients) {
        announce_aux(false);
        rc = rgbdslam_frame_preprocess(_ctx, _depth_cur[_cur], _depth_pitch[_cur], bgr, bgr_pitch, getVMap(), getNMap(), getRGB(), getGx(),
                                       4 * (size_t)_m, getGy(), 4 * (size_t)_m, _n, _m);
    } else {
        _aux_valid[_cur] = false;
        rc = rgbdslam_gpu_cpy_char3(_ctx, bgr, bgr_pitch, getRGB(), _n, _m);                                  // Frame.cpp:65
        if (!rc) rc = rgbdslam_vertex_map_kinect(_ctx, getVMap(), _depth_cur[_cur], _depth_pitch[_cur], _n, _m);   // :68
        if (!rc) rc = rgbdslam_compute_normal(_ctx, getNMap(), getVMap(), _n, _m, 1);                         // :71
    }
    if (_load_stream) {
        check(rgbdslam_set_stream(_ctx, main), "set_stream");
        if (!rc) {
            if (_use_events) hip_check(hipEventRecord(_ready[_cur], _load_stream), "hipEventRecord");
            else hip_check(hipStreamWriteValue32(_load_stream, _sig_ready[_cur], ++_seq_ready[_cur], 0), "hipStreamWriteValue32");
        }
    }
    check(rc, "frame preprocess");
}

// ---- PredictedFrame (D1/Frame.h:188-282) -----------------------------------------------------------
PredictedFrame::PredictedFrame(rgbdslam_ctx *ctx, int n, int m) : _n(n), _m(m), _ctx(ctx) {
    size_t P = (size_t)n * m;
    _VMap_dev.alloc(3 * P);
    _NMap_dev.alloc(3 * P);
    _RGB_dev.alloc(4 * P);
    _tex.alloc(4 * P);
    _VMap_alt.alloc(3 * P);
    _RGB_alt.alloc(4 * P);
    _mask_a.alloc(P); _mask_b.alloc(P); _label_a.alloc(P); _label_b.alloc(P);
    _Mask_dev = _mask_a.p; _Mask_dev_swap = _mask_b.p;
    _Label_dev = _label_a.p; _Label_dev_swap = _label_b.p;
    _Index_dev.alloc(4 * (size_t)(n - 1) * (m - 1));
    _zbuf.alloc(P, false);
    check(rgbdslam_raster_clear(ctx, _zbuf.p, n, m), "raster_clear");
}

// 16-bit label images (models of more than 255 planes; the reference's are unsigned char, Frame.h:197): the context reads
// and writes every label image with that width from here on; the two images start again at zero
void PredictedFrame::SetLabelWidth(int bytes) {
    check(rgbdslam_set_label_width(_ctx, bytes), "set_label_width");
    _label_width = bytes;
    const size_t P = (size_t)_n * _m;
    _label_a.alloc(P * (size_t)bytes); _label_b.alloc(P * (size_t)bytes);
    _Label_dev = _label_a.p; _Label_dev_swap = _label_b.p;
}

void PredictedFrame::Transform(float *pose, bool quad) {
    check(rgbdslam_set_pose_matrix(_ctx, pose), "SetPoseMatrix");                                           // Frame.cpp:283
    check(rgbdslam_transform_vertex_map(_ctx, _VMap_dev.p, _RGB_dev.p, _n, _m), "TransformVertexMap");      // :290
    if (quad && use_index_buffer)
        check(rgbdslam_quad_trim_frame(_ctx, _VMap_dev.p, _Index_dev.p, _n, _m, 0.1f), "QuadTrimFrame");    // :293
}

void PredictedFrame::ReadFrame(float znear, float zfar) {
    // DisplayPredictedFBO (GP-demo.cpp:98-113): the GL_QUADS draw becomes the compute rasteriser;
    // without an index buffer the quads are trimmed inline (same 0.1 threshold, Frame.cpp:293)
    check(rgbdslam_predict_raster(_ctx, _tex.p, _zbuf.p, _VMap_dev.p, _RGB_dev.p, use_index_buffer ? _Index_dev.p : nullptr,
                                  0.1f, znear, zfar, _n, _m), "predict_raster");
    check(rgbdslam_read_frame(_ctx, _VMap_dev.p, _RGB_dev.p, _tex.p, _n, _m), "ReadFrame_cu");              // Frame.cpp:239
    // the cudaMemset of NMap (Frame.cpp:241) is dropped: compute_normal writes every pixel
    check(rgbdslam_compute_normal(_ctx, _NMap_dev.p, _VMap_dev.p, _n, _m, 1), "ComputeNormal");             // :242
}

void PredictedFrame::Merge(OffLineFrame *frame) {
    check(rgbdslam_merge_depth_off(_ctx, _RGB_dev.p, frame->depth(), frame->depth_pitch(), frame->getRGB(), _Mask_dev, _n, _m),
          "MergeDepthOff");                                                                                 // Frame.cpp:313
    // the two cudaMemsets (Frame.cpp:317-318) are dropped: both kernels below write every pixel
    check(rgbdslam_vertex_map_gl(_ctx, _VMap_dev.p, _RGB_dev.p, _n, _m), "VertexMapGL");                    // :319
    check(rgbdslam_compute_normal(_ctx, _NMap_dev.p, _VMap_dev.p, _n, _m, 1), "ComputeNormal");             // :320
}

void PredictedFrame::Cpy(OffLineFrame *frame) {
    size_t P = (size_t)_n * _m;
    hipStream_t s = ctx_stream(_ctx);
    hip_check(hipMemcpyAsync(_VMap_dev.p, frame->getVMap(), 12 * P, hipMemcpyDeviceToDevice, s), "Cpy VMap");   // Frame.cpp:346
    hip_check(hipMemcpyAsync(_NMap_dev.p, frame->getNMap(), 12 * P, hipMemcpyDeviceToDevice, s), "Cpy NMap");   // :347
    check(rgbdslam_gpu_add_zbuff_off(_ctx, frame->depth(), frame->depth_pitch(), frame->getRGB(), _RGB_dev.p, _n, _m),
          "gpu_add_zbuffOff");                                                                              // :351
}

void PredictedFrame::InitMask() {
    check(rgbdslam_init_mask(_ctx, _Mask_dev, _RGB_dev.p, _n, _m), "InitMask_cu");                          // Frame.cpp:367
}

void PredictedFrame::ProjectMask(float *pose) {
    check(rgbdslam_set_pose_matrix(_ctx, pose), "SetPoseMatrix");
    check(rgbdslam_project_mask(_ctx, _Mask_dev, _Mask_dev_swap, _VMap_dev.p, _n, _m), "ProjectMask_cu");   // Frame.cpp:376
    uint8_t *tmp = _Mask_dev; _Mask_dev = _Mask_dev_swap; _Mask_dev_swap = tmp;
}

void PredictedFrame::ProjectLabel(float *pose_loc, float *pose_glob, float *Equations_dev, int nbPlans) {
    check(rgbdslam_set_pose_matrix(_ctx, pose_loc), "SetPoseMatrix");
    check(rgbdslam_project_label(_ctx, _Label_dev, _Label_dev_swap, _VMap_dev.p, _n, _m), "ProjectMask_cu"); // Frame.cpp:389
    uint8_t *tmp = _Label_dev; _Label_dev = _Label_dev_swap; _Label_dev_swap = tmp;
    // Segment_cu with nbPlans == 0 launches kernels that can never relabel anything (no labels
    // exist before the first detection); skipped.
    if (nbPlans > 0)
        check(rgbdslam_segment(_ctx, _Label_dev, _Label_dev_swap, _VMap_dev.p, _NMap_dev.p, pose_glob, Equations_dev, nbPlans,
                               0.1f, 0.9f, _n, _m), "Segment_cu");                                          // Frame.cpp:395
}

// ---- fused forms ------------------------------------------------------------------------------------
void PredictedFrame::ReseedFused(const rgbdslam_pose_block *pb, int force, OffLineFrame *frame) {
    check(rgbdslam_model_reseed(_ctx, pb, force, frame->getVMap(), frame->getNMap(), frame->getRGB(), frame->depth(),
                                frame->depth_pitch(), _VMap_dev.p, _NMap_dev.p, _RGB_dev.p, _Mask_dev, _n, _m), "model_reseed");
}
void PredictedFrame::PredictFused(const rgbdslam_pose_block *pb, float znear, float zfar, OffLineFrame *frame,
                                  const UpdateTables *prev_apply, const float *prev_history_entry) {
    if (!_vertex_scratch.p && !getenv("RGBDSLAM_NO_VERTEX_SCRATCH"))      // projections handed from the draw to the resolve
        _vertex_scratch.alloc((RGBDSLAM_PREDICT_SCRATCH_BYTES(_n, _m) + 3) / 4, false);
    // frame != null: a lost frame re-seeds the model inside the same launch (Cpy + InitMask, GP-demo.cpp:190-191)
    if (prev_apply) {
        check(rgbdslam_model_predict_apply(_ctx, pb, _VMap_dev.p, _RGB_dev.p, _Mask_dev, _Label_dev, _zbuf.p, _VMap_alt.p, _RGB_alt.p,
                                           _Mask_dev_swap, _Label_dev_swap, 0.1f, znear, zfar, _n, _m, frame ? frame->getVMap() : nullptr,
                                           frame ? frame->getNMap() : nullptr, frame ? frame->getRGB() : nullptr,
                                           frame ? frame->depth() : nullptr, frame ? frame->depth_pitch() : 0, _NMap_dev.p,
                                           _vertex_scratch.p, prev_history_entry, prev_apply->Bump, prev_apply->RGB, prev_apply->Mask, prev_apply->Param,
                                           prev_apply->Claim, prev_apply->stamp, prev_apply->N_prim, prev_apply->M_prim,
                                           prev_apply->plane_lo, prev_apply->plane_hi), "model_predict + previous Update_cu (apply)");
        return;
    }
    check(rgbdslam_model_predict(_ctx, pb, _VMap_dev.p, _RGB_dev.p, _Mask_dev, _Label_dev, _zbuf.p, _VMap_alt.p, _RGB_alt.p,
                                 _Mask_dev_swap, _Label_dev_swap, 0.1f, znear, zfar, _n, _m, frame ? frame->getVMap() : nullptr,
                                 frame ? frame->getNMap() : nullptr, frame ? frame->getRGB() : nullptr,
                                 frame ? frame->depth() : nullptr, frame ? frame->depth_pitch() : 0, _NMap_dev.p, _vertex_scratch.p),
          "model_predict");
}
void PredictedFrame::NormalSegmentFused(const rgbdslam_pose_block *pb, float *Equations_dev, int nbPlans) {
    // projected labels are in _Label_dev_swap; an odd number of passes brings them back to _Label_dev
    check(rgbdslam_model_normal_segment(_ctx, pb, _NMap_dev.p, _VMap_alt.p, _Label_dev_swap, _Label_dev, Equations_dev, nbPlans,
                                        0.1f, 0.9f, _n, _m), "model_normal_segment");
}
void PredictedFrame::SegmentMergeClaimFused(const rgbdslam_pose_block *pb, float *Equations_dev, int nbPlans, OffLineFrame *frame,
                                            const UpdateTables &claim, const uint16_t *next_depth, size_t next_depth_pitch,
                                            const uint8_t *next_bgr, size_t next_bgr_pitch) {
    if (!_NMap_seg.p) _NMap_seg.alloc(3 * (size_t)_n * _m);   // normals of the predicted frame for the growth passes (Merge owns _NMap_dev)
    if (next_depth) {
        frame->PrepareOtherSet();
        const size_t gp = 4 * (size_t)_m;
        frame->announce_aux(true);
        check(rgbdslam_model_segment_merge_claim_preprocess(
                  _ctx, pb, _NMap_seg.p, _VMap_alt.p, _Label_dev_swap, _Label_dev, Equations_dev, nbPlans, 0.1f, 0.9f, _RGB_alt.p,
                  _Mask_dev_swap, frame->depth(), frame->depth_pitch(), frame->getRGB(), _RGB_dev.p, _Mask_dev, _VMap_dev.p, _NMap_dev.p,
                  claim.Mask, claim.Param, claim.Claim, claim.stamp, claim.N_prim, claim.M_prim, _n, _m, claim.plane_lo, claim.plane_hi,
                  next_depth, next_depth_pitch, next_bgr, next_bgr_pitch, frame->otherVMap(), frame->otherNMap(), frame->otherRGB(),
                  frame->otherGx(), gp, frame->otherGy(), gp),
              "model_segment_merge_claim + next LoadFrame");
        frame->MarkPreprocessed(next_depth, next_depth_pitch, next_bgr, next_bgr_pitch, true);
        return;
    }
    check(rgbdslam_model_segment_merge_claim(_ctx, pb, _NMap_seg.p, _VMap_alt.p, _Label_dev_swap, _Label_dev, Equations_dev, nbPlans, 0.1f,
                                             0.9f, _RGB_alt.p, _Mask_dev_swap, frame->depth(), frame->depth_pitch(), frame->getRGB(),
                                             _RGB_dev.p, _Mask_dev, _VMap_dev.p, _NMap_dev.p, claim.Mask, claim.Param, claim.Claim,
                                             claim.stamp, claim.N_prim, claim.M_prim, _n, _m, claim.plane_lo, claim.plane_hi),
          "model_segment_merge_claim");
}
void PredictedFrame::MergeFused(const rgbdslam_pose_block *pb, OffLineFrame *frame, const UpdateTables *claim) {
    if (claim) {
        check(rgbdslam_model_merge_claim(_ctx, pb, _RGB_alt.p, _Mask_dev_swap, frame->depth(), frame->depth_pitch(), frame->getRGB(),
                                         _RGB_dev.p, _Mask_dev, _VMap_dev.p, _NMap_dev.p, _Label_dev, claim->Mask, claim->Param,
                                         claim->Claim, claim->stamp, claim->N_prim, claim->M_prim, _n, _m, claim->plane_lo,
                                         claim->plane_hi), "model_merge_claim");
        return;
    }
    check(rgbdslam_model_merge(_ctx, pb, _RGB_alt.p, _Mask_dev_swap, frame->depth(), frame->depth_pitch(), frame->getRGB(),
                               _RGB_dev.p, _Mask_dev, _VMap_dev.p, _NMap_dev.p, _n, _m), "model_merge");
}

// ---- Primitive (D1/Primitive.h:79-112,165-190) --------------------------------------------------------
Primitive::Primitive(float *nmle, float *e1, float *e2, float dist, float *scale, int *Size, float *Shift, float *anchor) {
    for (int k = 0; k < 3; k++) { _nmle[k] = nmle[k]; _e1[k] = e1[k]; _e2[k] = e2[k]; _anchor[k] = anchor[k]; }
    _dist = dist;
    for (int k = 0; k < 2; k++) { _scale[k] = scale[k]; _Size[k] = Size[k]; _Shift[k] = Shift[k]; }
}
void Primitive::Allocate() { _param_dev.alloc(14); }
void Primitive::SetParam() {
    float tmp[14];
    tmp[0] = _nmle[0]; tmp[1] = _nmle[1]; tmp[2] = _nmle[2]; tmp[3] = _dist;
    tmp[4] = _scale[0]; tmp[5] = _scale[1];
    tmp[6] = _Shift[0]; tmp[7] = _Shift[1];
    tmp[8] = _e1[0]; tmp[9] = _e1[1]; tmp[10] = _e1[2];
    tmp[11] = _e2[0]; tmp[12] = _e2[1]; tmp[13] = _e2[2];
    hip_check(hipMemcpy(_param_dev.p, tmp, sizeof(tmp), hipMemcpyHostToDevice), "SetParam");
}

// ---- Graph (D1/Graph.h:54-116, D1/Graph.cpp:50-95) ----------------------------------------------------
Graph::Graph(rgbdslam_ctx *ctx, int frag_n, int frag_m, int max_planes) : _ctx(ctx), _max_planes(max_planes) {
    _SizeVBUFF[0] = frag_n;
    _SizeVBUFF[1] = frag_m;
}
Graph::~Graph() {
    Clear();
    for (auto *b : _data_Bump_dev) delete b;
    for (auto *b : _data_RGB_dev) delete b;
    for (auto *b : _data_Mask_dev) delete b;
    for (auto *b : _data_Claim_dev) delete b;
}
void Graph::Clear() {
    for (Primitive *p : _nodes) delete p;
    _nodes.clear();
    _tables_built = false;
}
void Graph::Compact(const std::vector<int32_t> &Assoc) {
    // a primitive's slabs are pool entries at its own index: the pool is reordered with the nodes (kept ones first, in order;
    // the slabs of dropped primitives go back behind them and are wiped when AddPrimitive hands them out again)
    std::vector<Primitive *> kept;
    std::vector<DevBuf<uint16_t> *> b;
    std::vector<DevBuf<uint8_t> *> r, k;
    std::vector<DevBuf<uint64_t> *> c;
    std::vector<size_t> rest;
    for (size_t i = 0; i < _data_Bump_dev.size(); i++) {
        const bool keep = i < _nodes.size() && i < Assoc.size() && Assoc[i] != 0;
        if (keep) {
            kept.push_back(_nodes[i]);
            b.push_back(_data_Bump_dev[i]); r.push_back(_data_RGB_dev[i]); k.push_back(_data_Mask_dev[i]); c.push_back(_data_Claim_dev[i]);
        } else {
            if (i < _nodes.size()) delete _nodes[i];
            rest.push_back(i);
        }
    }
    for (size_t i : rest) { b.push_back(_data_Bump_dev[i]); r.push_back(_data_RGB_dev[i]); k.push_back(_data_Mask_dev[i]); c.push_back(_data_Claim_dev[i]); }
    _nodes.swap(kept);
    _data_Bump_dev.swap(b); _data_RGB_dev.swap(r); _data_Mask_dev.swap(k); _data_Claim_dev.swap(c);
    _tables_built = false;
}
bool Graph::AddPrimitive(Primitive *Prim) {
    size_t idx = _nodes.size();
    if ((int)idx > _max_planes - 1) return false;                                  // Graph.h:103 (50 in the reference)
    size_t C = (size_t)_SizeVBUFF[0] * _SizeVBUFF[1];
    // The reference preallocates all 50 slabs in the constructor (Graph.h:64-72); here slabs are
    // created on first use so that small test configurations stay small.  Bump/RGB start at zero
    // (uninitialised cudaMalloc memory in the reference).
    if (_data_Bump_dev.size() <= idx) {
        auto *b = new DevBuf<uint16_t>(); b->alloc(3 * C); _data_Bump_dev.push_back(b);
        auto *r = new DevBuf<uint8_t>(); r->alloc(3 * C); _data_RGB_dev.push_back(r);
        auto *k = new DevBuf<uint8_t>(); k->alloc(C); _data_Mask_dev.push_back(k);
        auto *c = new DevBuf<uint64_t>(); c->alloc(C); _data_Claim_dev.push_back(c);
    } else {
        hip_check(hipMemset(_data_Bump_dev[idx]->p, 0, 6 * C), "memset bump");
        hip_check(hipMemset(_data_RGB_dev[idx]->p, 0, 3 * C), "memset rgb");
    }
    Prim->_Bump_dev = _data_Bump_dev[idx]->p;
    Prim->_RGB_dev = _data_RGB_dev[idx]->p;
    Prim->_Mask_dev = _data_Mask_dev[idx]->p;
    Prim->_Claim_dev = _data_Claim_dev[idx]->p;
    hip_check(hipMemset(Prim->_Mask_dev, 0, C), "memset mask");                    // Graph.h:112
    _nodes.push_back(Prim);
    _tables_built = false;
    return true;
}
void Graph::build_tables() {                                                       // Graph.cpp:56-70
    size_t S = _nodes.size();
    std::vector<void *> bt(S), rt(S), mt(S), pt(S), ct(S);
    for (size_t i = 0; i < S; i++) {
        Primitive *Prim = _nodes[i];
        Prim->SetParam();
        bt[i] = Prim->_Bump_dev; rt[i] = Prim->_RGB_dev; mt[i] = Prim->_Mask_dev; pt[i] = Prim->_param_dev.p;
        ct[i] = Prim->_Claim_dev;
    }
    // one H2D copy per table replaces the 1-thread Affect* kernels (D1/RG.cu:557-567)
    DevBuf<void *> *tabs[5] = {&_BumpTab, &_RGBTab, &_MaskTab, &_ParamTab, &_ClaimTab};
    std::vector<void *> *src[5] = {&bt, &rt, &mt, &pt, &ct};
    for (int k = 0; k < 5; k++) {
        tabs[k]->alloc(S, false);
        hip_check(hipMemcpy(tabs[k]->p, src[k]->data(), S * sizeof(void *), hipMemcpyHostToDevice), "pointer table");
    }
    _tables_built = true;
}
void Graph::Update(PredictedFrame *frame, float *pose) {
    if (_nodes.size() == 0) return;                                                // Graph.cpp:53-54
    if (!_tables_built) build_tables();
    _stamp++;
    check(rgbdslam_update(_ctx, (uint16_t **)_BumpTab.p, (uint8_t **)_RGBTab.p, (uint8_t **)_MaskTab.p, (float **)_ParamTab.p,
                          (uint64_t **)_ClaimTab.p, _stamp, frame->getVMap(), frame->getRGB(), frame->_Mask_dev,
                          frame->_Label_dev, pose, _SizeVBUFF[0], _SizeVBUFF[1], frame->_n, frame->_m, plane_lo, plane_hi),
          "Update_cu");                                                            // Graph.cpp:74
}
bool Graph::begin_update(UpdateTables *t) {
    if (_nodes.size() == 0) return false;                                          // Graph.cpp:53-54
    if (!_tables_built) build_tables();
    _stamp++;
    *t = UpdateTables{(uint16_t **)_BumpTab.p, (uint8_t **)_RGBTab.p, (uint8_t **)_MaskTab.p, (float **)_ParamTab.p,
                      (uint64_t **)_ClaimTab.p, _stamp, _SizeVBUFF[0], _SizeVBUFF[1], plane_lo, plane_hi};
    return true;
}
void Graph::ApplyTracked(PredictedFrame *frame, const rgbdslam_pose_block *pb, const UpdateTables &t) {
    check(rgbdslam_update_apply_tracked(_ctx, pb, t.Bump, t.RGB, t.Mask, t.Param, t.Claim, t.stamp, frame->getVMap(), frame->getRGB(),
                                        frame->_Mask_dev, frame->_Label_dev, t.N_prim, t.M_prim, frame->_n, frame->_m, t.plane_lo,
                                        t.plane_hi), "Update_cu (apply)");
}
void Graph::ApplyTrackedAndPreprocess(PredictedFrame *frame, const rgbdslam_pose_block *pb, const UpdateTables &t, OffLineFrame *next,
                                      const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch) {
    const size_t gp = 4 * (size_t)frame->_m;
    next->announce_aux(false);
    check(rgbdslam_update_apply_tracked_preprocess(_ctx, pb, t.Bump, t.RGB, t.Mask, t.Param, t.Claim, t.stamp, frame->getVMap(),
                                                   frame->getRGB(), frame->_Mask_dev, frame->_Label_dev, t.N_prim, t.M_prim, frame->_n,
                                                   frame->_m, t.plane_lo, t.plane_hi, depth_dev, depth_pitch, bgr_dev, bgr_pitch,
                                                   next->getVMap(), next->getNMap(), next->getRGB(), next->getGx(), gp, next->getGy(), gp),
          "Update_cu (apply) + next LoadFrame");
    next->MarkPreprocessed(depth_dev, depth_pitch, bgr_dev, bgr_pitch);
}
void Graph::UpdateFromHistory(PredictedFrame *frame, const float *history_entry_dev) {
    if (_nodes.size() == 0) return;
    if (!_tables_built) build_tables();
    _stamp++;
    check(rgbdslam_update_from_history(_ctx, history_entry_dev, (uint16_t **)_BumpTab.p, (uint8_t **)_RGBTab.p, (uint8_t **)_MaskTab.p,
                                       (float **)_ParamTab.p, (uint64_t **)_ClaimTab.p, _stamp, frame->getVMap(), frame->getRGB(),
                                       frame->_Mask_dev, frame->_Label_dev, _SizeVBUFF[0], _SizeVBUFF[1], frame->_n, frame->_m, plane_lo,
                                       plane_hi), "Update_cu (history)");
}
void Graph::UpdateTracked(PredictedFrame *frame, const rgbdslam_pose_block *pb) {
    if (_nodes.size() == 0) return;
    if (!_tables_built) build_tables();
    _stamp++;
    check(rgbdslam_update_tracked(_ctx, pb, (uint16_t **)_BumpTab.p, (uint8_t **)_RGBTab.p, (uint8_t **)_MaskTab.p,
                                  (float **)_ParamTab.p, (uint64_t **)_ClaimTab.p, _stamp, frame->getVMap(), frame->getRGB(),
                                  frame->_Mask_dev, frame->_Label_dev, _SizeVBUFF[0], _SizeVBUFF[1], frame->_n, frame->_m, plane_lo,
                                  plane_hi), "Update_cu (tracked)");
}

// ---- RG (D1/RG.h:111-250, D1/RG.cpp) -------------------------------------------------------------------
RG::RG(int device, void *stream, int n, int m, const float *Calib, int lvl, float distThres, float angleThres, int max_planes,
       int N_prim, int M_prim)
    : _count(1), _nbPlan(0), _lvl(lvl), _distThres(distThres), _angleThres(angleThres), _thres_cvg_alpha(1e-6f),
      _thres_cvg_tr(1e-6f) {
    check(rgbdslam_create(&_ctx, device, stream), "rgbdslam_create");
    memcpy(_calib, Calib, sizeof(_calib));
    check(rgbdslam_set_calibration_matrix(_ctx, _calib), "SetCalibrationMatrix");   // RG.h:186
    int gr = ((n % 8) ? (n + 7) / 8 : n / 8), gc = ((m % 8) ? (m + 7) / 8 : m / 8);
    check(rgbdslam_alloc_buffers(_ctx, gr, gc), "AllocBuffers");                     // RG.h:189
    _max_iter[0] = 3; _max_iter[1] = 2; _max_iter[2] = 1;                           // RG.h:124-126
    _poseR = _poseRPrev = Mat3::Identity();
    _poset = _posetPrev = Vec3::Zero();
    _frame = new OffLineFrame(_ctx, n, m);
    _predicted_frame = new PredictedFrame(_ctx, n, m);
    _graph = new Graph(_ctx, N_prim, M_prim, max_planes);
    size_t P = (size_t)n * m;
    _intr_dev.alloc(4); _Rcurr_dev.alloc(9); _tcurr_dev.alloc(3);
    hip_check(hipMemcpy(_intr_dev.p, _calib, 16, hipMemcpyHostToDevice), "intr");   // RG.h:200-204
    _img1.alloc(P); _Gx.alloc(P); _Gy.alloc(P);
    _pose_state_dev.alloc(24); _status_dev.alloc(4); _trace_dev.alloc(33 * 16);
    hip_check(hipHostMalloc((void **)&_pinned, 32 * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
    // fused pipeline state: device pose block (identity poses), pinned mirror, pose history ring
    hip_check(hipMalloc((void **)&_pb_dev, sizeof(rgbdslam_pose_block)), "hipMalloc pose block");
    hip_check(hipHostMalloc((void **)&_pb_host, sizeof(rgbdslam_pose_block), hipHostMallocDefault), "hipHostMalloc");
    memset(_pb_host, 0, sizeof(rgbdslam_pose_block));
    memcpy(_pb_host->pose_state, _poseR.a, 36);
    memcpy(_pb_host->pose_state + 12, _poseRPrev.a, 36);
    hip_check(hipMemcpy(_pb_dev, _pb_host, sizeof(rgbdslam_pose_block), hipMemcpyHostToDevice), "pose block init");
    _history_dev.alloc(16 * (size_t)HISTORY_CAP);
    hip_check(hipHostMalloc((void **)&_history_host, 16 * (size_t)HISTORY_CAP * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
    _frame->EnableGradients(true);
    hip_check(hipStreamCreateWithFlags(&_load_stream, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipStreamCreateWithFlags(&_update_stream, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipEventCreateWithFlags(&_ev_merged, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&_ev_updated, hipEventDisableTiming), "hipEventCreate");
}

RG::~RG() {
    if (_ctx) (void)rgbdslam_synchronize(_ctx);
    if (_load_stream) { (void)hipStreamSynchronize(_load_stream); }
    if (_update_stream) { (void)hipStreamSynchronize(_update_stream); }
    delete _dataset;
    delete _graph;
    delete _predicted_frame;
    delete _frame;
    if (_load_stream) (void)hipStreamDestroy(_load_stream);
    if (_update_stream) (void)hipStreamDestroy(_update_stream);
    if (_ev_merged) (void)hipEventDestroy(_ev_merged);
    if (_ev_updated) (void)hipEventDestroy(_ev_updated);
    for (auto &e : _stage_ev) if (e) (void)hipEventDestroy(e);
    if (_pinned) (void)hipHostFree(_pinned);
    if (_pb_dev) (void)hipFree(_pb_dev);
    if (_pb_host) (void)hipHostFree(_pb_host);
    if (_history_host) (void)hipHostFree(_history_host);
    _history_dev.release();
    _intr_dev.release(); _Rcurr_dev.release(); _tcurr_dev.release(); _img1.release(); _Gx.release(); _Gy.release();
    _pose_state_dev.release(); _status_dev.release(); _trace_dev.release(); _Equations_dev.release();
    if (_ctx) (void)rgbdslam_destroy(_ctx);
}

// before the first frame: the fused pipeline loads through the auxiliary stream into alternating buffer sets
void RG::configure_streams() {
    const char *e = getenv("RGBDSLAM_OVERLAP_MODE");
    if (e && !_overlap_mode_set) _overlap_mode = atoi(e);
    if (_count == 1 && fused && overlap && (_overlap_mode & 1) && !_frame->load_stream()) _frame->AttachLoadStream(_load_stream);
}

bool RG::LoadFrame(const uint16_t *depth_host, const uint8_t *bgr_host) {
    StageRange r("stage 1: preprocessing (upload + OffLineFrame::LoadFrame)");
    configure_streams();
    _frame->LoadFrame(depth_host, bgr_host);                                        // RG.cpp:141
    _count++;                                                                       // RG.cpp:144
    return true;
}
bool RG::LoadFrameDevice(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch) {
    StageRange r("stage 1: preprocessing (OffLineFrame::LoadFrame)");
    configure_streams();
    stage_mark(0);
    _frame->LoadFrameDevice(depth_dev, depth_pitch, bgr_dev, bgr_pitch);
    stage_mark(1);
    _count++;
    return true;
}

// ---- per-stage timers (the reference's TIMING, Utilities.h:82 / GP-demo.cpp:153-277) ----------------------------------------
// marks of one frame: 0 / 1 round stage 1 (LoadFrame), 2 before the tracking, 3 after it, 4 after the prediction, 5 after the fusion
void RG::SetStageTiming(bool on) {
    if (on && _stage_ev.empty()) {
        _stage_ev.resize((size_t)STAGE_RING * STAGE_MARKS, nullptr);
        _stage_set.assign((size_t)STAGE_RING, 0);
        for (auto &e : _stage_ev) hip_check(hipEventCreate(&e), "hipEventCreate");
    }
    if (on && !_stage_timing) { _stage_frames = 0; std::fill(_stage_set.begin(), _stage_set.end(), 0); }
    _stage_timing = on;
}
void RG::stage_mark(int mark) {
    if (!_stage_timing) return;
    // marks 0 / 1 belong to the frame that is being loaded (the one the next StepAsync processes), the others to that step
    const int slot = _stage_frames % STAGE_RING;
    hipStream_t s = (mark <= 1 && _frame->load_stream()) ? (hipStream_t)_frame->load_stream() : ctx_stream(_ctx);
    hip_check(hipEventRecord(_stage_ev[(size_t)slot * STAGE_MARKS + mark], s), "hipEventRecord");
    _stage_set[(size_t)slot] |= (unsigned char)(1u << mark);
    if (mark == 5) { _stage_frames++; _stage_set[(size_t)(_stage_frames % STAGE_RING)] = 0; }   // the next frame starts clean
}
int RG::StageTimes(double us[4]) {
    for (int k = 0; k < 4; k++) us[k] = 0.0;
    if (_stage_ev.empty() || _stage_frames == 0) return 0;
    SyncAll();
    const int frames = std::min(_stage_frames, (int)STAGE_RING);
    const int pairs[4][2] = {{0, 1}, {2, 3}, {3, 4}, {4, 5}};
    for (int f = 0; f < frames; f++) {
        const int slot = (_stage_frames - 1 - f) % STAGE_RING;
        for (int k = 0; k < 4; k++) {
            const unsigned need = (1u << pairs[k][0]) | (1u << pairs[k][1]);
            if ((_stage_set[(size_t)slot] & need) != need) continue;            // (a frame loaded before the timers were on)
            float ms = 0.0f;
            hip_check(hipEventElapsedTime(&ms, _stage_ev[(size_t)slot * STAGE_MARKS + pairs[k][0]],
                                          _stage_ev[(size_t)slot * STAGE_MARKS + pairs[k][1]]), "hipEventElapsedTime");
            us[k] += 1e3 * (double)ms;
        }
    }
    for (int k = 0; k < 4; k++) us[k] /= (double)frames;
    _stage_frames = 0;
    std::fill(_stage_set.begin(), _stage_set.end(), 0);
    return frames;
}

void RG::HintNextDeviceFrame(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch) {
    _next_hint.depth = depth_dev; _next_hint.depth_pitch = depth_pitch; _next_hint.bgr = bgr_dev; _next_hint.bgr_pitch = bgr_pitch;
    _next_hint.valid = depth_dev && bgr_dev;
}

void RG::OpenDataset(const char *path, int Sizedata, bool real, bool RGBDRep) {
    if (_count != 1) throw Error("OpenDataset: frames were already loaded", RGBDSLAM_E_STATE);
    delete _dataset;
    _dataset = new Dataset(path, Sizedata, real, RGBDRep);
    _first_out = true;
}

bool RG::LoadFrame() {
    if (!_dataset) throw Error("LoadFrame(): no dataset opened", RGBDSLAM_E_STATE);
    if (_count > _dataset->size() - 1) {                                            // RG.cpp:81-90 (Finish() belongs to N1/N2)
        if (_first_out) {
            memset(_poseRPrev.a, 0, sizeof(_poseRPrev.a));
            _posetPrev = Vec3::Zero();
        }
        _first_out = false;
        return false;
    }
    std::string filename_depth, filename_color;
    _dataset->frame_files(_count, &filename_depth, &filename_color);                // RG.cpp:105-139
    configure_streams();
    _frame->LoadFrame(filename_depth, filename_color);                              // RG.cpp:141
    _traj_count.push_back(_count);
    _count++;                                                                       // RG.cpp:144
    return true;
}

void RG::SaveTrajectory(const char *filename) {
    if (!_dataset || !_dataset->_RGBDRep || _dataset->_real)
        throw Error("SaveTrajectory: needs a TUM listing (the stamp is cut out of its file names)", RGBDSLAM_E_STATE);
    std::vector<Mat3> Rs;
    std::vector<Vec3> ts;
    std::vector<std::string> stamps;
    if (fused) {
        Sync();
        if (frames_done > HISTORY_CAP) throw Error("SaveTrajectory: sequence longer than the pose history", RGBDSLAM_E_LIMIT);
        for (int f = 0; f < frames_done; f++) {
            const float *h = history_host(f);
            Mat3 R; Vec3 t;
            memcpy(R.a, h, 36); memcpy(t.v, h + 9, 12);
            if (h[12] == 0.0f) t(0) = 100.0f;                                       // lost frame: the marker SaveTrajectory skips
            Rs.push_back(R); ts.push_back(t);
        }
    } else {
        Rs = _traj_R; ts = _traj_t;
    }
    if (_traj_count.size() < Rs.size()) throw Error("SaveTrajectory: frames were not loaded through LoadFrame()", RGBDSLAM_E_STATE);
    for (size_t f = 0; f < Rs.size(); f++) stamps.push_back(_dataset->timestamp_depth(_traj_count[f]));
    slam::SaveTrajectory(filename, Rs, ts, stamps);
}

void RG::SaveModel(const char *dir, int indx) {
    if (fused) Sync();
    SyncAll();
    const int N = _graph->_SizeVBUFF[0], M = _graph->_SizeVBUFF[1];
    const size_t C = (size_t)N * M;
    std::vector<PlanarPatch> nodes;
    std::vector<uint8_t> mask8(C);
    for (Primitive *Prim : _graph->_nodes) {                                        // RG.cpp:216-223 + Primitive.h:150-158
        PlanarPatch P;
        memcpy(P.nmle, Prim->_nmle, 12); memcpy(P.e1, Prim->_e1, 12); memcpy(P.e2, Prim->_e2, 12); memcpy(P.anchor, Prim->_anchor, 12);
        P.dist = Prim->_dist;
        memcpy(P.scale, Prim->_scale, 8); memcpy(P.Shift, Prim->_Shift, 8);
        P.Size[0] = Prim->_Size[0]; P.Size[1] = Prim->_Size[1];
        if (P.Size[0] != N || P.Size[1] != M) throw Error("SaveModel: plane size differs from the slab size", RGBDSLAM_E_STATE);
        P.Bump.resize(3 * C); P.RGB.resize(3 * C); P.Mask.resize(C);
        hip_check(hipMemcpy(P.Bump.data(), Prim->_Bump_dev, 6 * C, hipMemcpyDeviceToHost), "LoadOnCPU bump");
        hip_check(hipMemcpy(P.RGB.data(), Prim->_RGB_dev, 3 * C, hipMemcpyDeviceToHost), "LoadOnCPU rgb");
        hip_check(hipMemcpy(mask8.data(), Prim->_Mask_dev, C, hipMemcpyDeviceToHost), "LoadOnCPU mask");
        for (size_t c = 0; c < C; c++) P.Mask[c] = mask8[c];
        P.has_data = true;
        nodes.push_back(std::move(P));
    }
    const int n = _frame->getN(), m = _frame->getM();
    std::vector<float> rgbd((size_t)4 * n * m);                                     // RG.cpp:227
    hip_check(hipMemcpy(rgbd.data(), _predicted_frame->getRGB(), rgbd.size() * sizeof(float), hipMemcpyDeviceToHost), "key frame download");
    float pose[16];
    pack_pose(_poseR, _poset, pose);                                                // RG.cpp:209-214
    for (PlanarPatch &P : nodes) P.Clean();                                         // RG.cpp:27 (Graph::Clean)
    SaveFrame(dir, indx + 1, rgbd.data(), n, m);                                    // RG.cpp:43
    SaveGraph(dir, indx, pose, nodes);                                              // RG.cpp:44
}

bool RG::AlignFrameGaussNewton() {
    int n = _frame->getN(), m = _frame->getM();
    size_t pitch = 4 * (size_t)m;
    hipStream_t s = ctx_stream(_ctx);
    // gpu_cpy + 2 x cv::gpu::Sobel, fused (RG.cpp:807-810); already done by frame_preprocess when attached
    float *Gx = _frame->has_gradients() ? _frame->getGx() : _Gx.p, *Gy = _frame->has_gradients() ? _frame->getGy() : _Gy.p;
    if (!_frame->has_gradients()) check(rgbdslam_intensity_sobel(_ctx, _frame->getRGB(), nullptr, 0, Gx, pitch, Gy, pitch, n, m), "intensity_sobel");
    float *VMap_prev = _predicted_frame->getVMap(), *NMap_prev = _predicted_frame->getNMap();
    float *VMap = _frame->getVMap(), *NMap = _frame->getNMap();

    if (device_loop) {
        // the whole loop of RG.cpp:832-926 on the device; one host sync for the result
        float *ps = _pinned;
        memcpy(ps, _poseR.a, 36); memcpy(ps + 9, _poset.v, 12); memcpy(ps + 12, _poseRPrev.a, 36); memcpy(ps + 21, _posetPrev.v, 12);
        hip_check(hipMemcpyAsync(_pose_state_dev.p, ps, 96, hipMemcpyHostToDevice, s), "pose upload");
        rgbdslam_icp_params prm = icp_params();
        check(rgbdslam_align_gauss_newton(_ctx, &prm, VMap, NMap, _frame->getRGB(), Gx, pitch, Gy, pitch, VMap_prev,
                                          NMap_prev, _predicted_frame->getRGB(), n, m, _pose_state_dev.p, _status_dev.p, nullptr),
              "align_gauss_newton");
        hip_check(hipMemcpyAsync(ps, _pose_state_dev.p, 48, hipMemcpyDeviceToHost, s), "pose download");
        hip_check(hipMemcpyAsync(ps + 24, _status_dev.p, 16, hipMemcpyDeviceToHost, s), "status download");
        hip_check(hipStreamSynchronize(s), "sync");
        const int32_t *st = (const int32_t *)(ps + 24);
        last_iterations = st[1];
        last_level_iterations = st[2];
        if (!st[0]) return false;
        memcpy(_poseR.a, ps, 36); memcpy(_poset.v, ps + 9, 12);
        return true;
    }

    // per-iteration host loop, statement for statement (RG.cpp:823-926)
    Mat3 Rcurr = _poseR;
    Vec3 tcurr = _poset;
    int fact = 1 << (_lvl - 1);                                                     // RG.cpp:832
    Mat3 Rcurr_inv = Rcurr.inverse();
    Vec3 tcurr_inv = Rcurr_inv.neg() * tcurr;
    last_iterations = 0;
    for (int lvl = _lvl - 1; lvl > -1; lvl--) {
        float A[36], b[6], x[6];
        bool cvg = false;
        int iter = 0;
        while (!cvg) {
            Rcurr = Rcurr_inv * _poseRPrev;                                         // RG.cpp:851
            tcurr = Rcurr_inv * _posetPrev + tcurr_inv;
            hip_check(hipMemcpyAsync(_Rcurr_dev.p, Rcurr.a, 36, hipMemcpyHostToDevice, s), "Rcurr");
            hip_check(hipMemcpyAsync(_tcurr_dev.p, tcurr.v, 12, hipMemcpyHostToDevice, s), "tcurr");
            check(rgbdslam_estimate_system_gauss_newton(_ctx, _Rcurr_dev.p, _tcurr_dev.p, VMap, NMap, _frame->getRGB(), Gx,
                                                        pitch, Gy, pitch, VMap_prev, NMap_prev, _predicted_frame->getRGB(),
                                                        _intr_dev.p, _distThres, _angleThres, n, m, A, b, fact, nullptr),
                  "EstimateSystemGaussNewton");                                     // RG.cpp:858
            last_iterations++;
            float det = det6(A);                                                    // RG.cpp:862
            if (fabsf(det) < 1e-15 || det != det) return false;
            if (!llt_solve6(A, b, x)) return false;                                 // RG.cpp:873
            double q[4];
            double norm = (double)((x[3] * x[3] + x[4] * x[4]) + x[5] * x[5]);
            q[1] = x[3]; q[2] = x[4]; q[3] = x[5]; q[0] = sqrt(1 - norm);
            double tmp[3][3];
            quaternion2matrix(q, tmp);
            Mat3 Rinc;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) Rinc(r, c) = (float)tmp[r][c];
            Vec3 tinc{{x[0], x[1], x[2]}};
            tcurr_inv = Rinc * tcurr_inv + tinc;                                    // RG.cpp:886
            Rcurr_inv = Rinc * Rcurr_inv;
            if (iter > _max_iter[lvl] || (norm_minus_identity(Rinc) < _thres_cvg_alpha && tinc.norm() < _thres_cvg_tr)) cvg = true;
            iter++;
        }
        fact /= 2;
    }
    if (norm_minus_identity(Rcurr) > 0.3 || tcurr.norm() > 0.3) return false;       // RG.cpp:910
    Rcurr = Rcurr_inv.inverse();                                                    // RG.cpp:923
    tcurr = Rcurr.neg() * tcurr_inv;
    _poseR = Rcurr;
    _poset = tcurr;
    return true;
}

// Label images as u16 instead of the reference's unsigned char (Frame.h:197), so that a model of more than 255 planes is
// tracked end to end (ProjectLabel, Segment, Graph::Update).  Before the first frame, or at least before any plane exists.
void RG::SetLabel16(bool on) {
    if (_nbPlan != 0 || !_graph->_nodes.empty()) throw Error("SetLabel16: the label width cannot change while planes exist", RGBDSLAM_E_STATE);
    if (fused && frames_done) Sync();
    SyncAll();
    _predicted_frame->SetLabelWidth(on ? 2 : 1);
}

// A tracker that maintains the LABELS of a model whose Bump images live elsewhere (the tracking rank of the plane-partitioned
// form): the plane equations (n, e1, e2, d) x nbPlan for ProjectLabel + Segment_cu (Frame.cpp:385-405), no
// Primitives, no Graph::Update here.  The labels of the current model frame are seeded by the caller (SeedLabelsTiled).
void RG::SetPlaneEquations(const float *equations, int nbPlan) {
    if (nbPlan <= 0 || nbPlan > (_predicted_frame->label_width() == 2 ? 65535 : 255))
        throw Error("SetPlaneEquations: plane count beyond the label width", RGBDSLAM_E_LIMIT);
    if (fused && frames_done) Sync();
    SyncAll();
    _graph->Clear();
    _Equations.assign(equations, equations + 10 * (size_t)nbPlan);
    _Centers.assign(3 * (size_t)nbPlan, 0.0f);
    _nbPlan = nbPlan;
    _Equations_dev.alloc(10 * (size_t)nbPlan, false);
    hip_check(hipMemcpy(_Equations_dev.p, _Equations.data(), 40 * (size_t)nbPlan, hipMemcpyHostToDevice), "Equations");
    _equations_only = true;
}
// Stand-in for plane detection on a TILED model (16-bit labels): every pixel of the current model frame takes the first plane
// whose equation it satisfies and whose N_prim x M_prim Bump image contains its cell (rgbdslam_label_planes_l16), posed by
// the current global pose.  From then on the labels are maintained by ProjectLabel + Segment like any others.
void RG::SeedLabelsTiled(float **ParamTab_dev, int nbPlan, int N_prim, int M_prim) {
    if (_predicted_frame->label_width() != 2) throw Error("SeedLabelsTiled needs 16-bit labels (SetLabel16)", RGBDSLAM_E_STATE);
    if (fused && frames_done) Sync();
    SyncAll();
    float pose[16];
    pack_pose(_poseR, _poset, pose);
    check(rgbdslam_label_planes_l16(_ctx, (uint16_t *)_predicted_frame->_Label_dev, _predicted_frame->getVMap(), _predicted_frame->getNMap(),
                                    pose, nullptr, ParamTab_dev, nbPlan, N_prim, M_prim, 0.1f, 0.8f, _predicted_frame->_n,
                                    _predicted_frame->_m), "label_planes_l16");
}
// the label image of the frame just processed, into a frame pack (stream-ordered copy)
void RG::ExportLabels(void *dst_dev, size_t off_label) {
    if (!dst_dev || frames_done == 0) throw Error("ExportLabels: no frame processed yet", RGBDSLAM_E_STATE);
    const size_t P = (size_t)_predicted_frame->_n * _predicted_frame->_m;
    hip_check(hipMemcpyAsync((char *)dst_dev + off_label, _predicted_frame->_Label_dev, P * (size_t)_predicted_frame->label_width(),
                             hipMemcpyDeviceToDevice, ctx_stream(_ctx)), "export labels");
}

void RG::SetPlanes(const float *equations, const float *centers, int nbPlan) {
    _equations_only = false;
    if (fused && frames_done) Sync();
    SyncAll();
    _graph->Clear();
    _Equations.assign(equations, equations + 10 * (size_t)nbPlan);
    _Centers.assign(centers, centers + 3 * (size_t)nbPlan);
    _nbPlan = nbPlan;
    float pose[16];
    pack_pose(_poseR, _poset, pose);
    CreatePrimitives(pose);
}

// RGBDSLAM_FINISH_TIMING=1: stage times of a detection / of RG::Finish's periodic cycle on stderr (wall clock, ms)
struct StageTimer {
    bool on; const char *what; std::chrono::steady_clock::time_point t0;
    explicit StageTimer(const char *w) : what(w) {
        static const bool enabled = getenv("RGBDSLAM_FINISH_TIMING") != nullptr;
        on = enabled;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~StageTimer() {
        if (on) fprintf(stderr, "[finish timing] %-28s %8.3f ms\n", what,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// RG.cpp:277-694.  Image stages on the GPU, median fit on the host (as in the reference, RG.cpp:393-679).
void RG::DetectPrimitives(float *pose) {
    StageTimer st_all("DetectPrimitives");
    const int n = _predicted_frame->_n, m = _predicted_frame->_m;
    const size_t P = (size_t)n * m;
    hipStream_t s = ctx_stream(_ctx);
    if (_blob_dev.count != P) {
        _blob_dev.alloc(P, false); _blob_sizes_dev.alloc(P, false); _detect_labels.alloc(3 * P, false); _detect_bytes.alloc(8 * P, false);
    }
    const size_t LW = (size_t)_predicted_frame->label_width();
    hip_check(hipMemsetAsync(_predicted_frame->_Label_dev, 0, P * LW, s), "memset label");                 // RG.cpp:313
    check(rgbdslam_detect_regions(_ctx, _predicted_frame->getNMap(), _predicted_frame->getVMap(), nullptr, 120.0, 30.0, 0.03f, 20,
                                  _blob_dev.p, _blob_sizes_dev.p, _detect_labels.p, _detect_bytes.p, n, m), "detect_regions");
    _nbPlan = 0;                                                                      // RG.cpp:401
    _Equations.clear(); _Centers.clear();
    const int max_labels = LW == 2 ? 65535 : 255;
    // The median fit on the device (rgbdslam_fit_planes: radix select over the image instead of four downloads and three
    // sorts per blob): only 80 bytes per blob come back.  Blob order, plane numbers and the 255 cap are applied here, in
    // FindBlobs' order (ascending first pixel).  More blobs than the device path holds, or RGBDSLAM_HOST_FIT=1: the host fit below.
    static const bool force_host_fit = getenv("RGBDSLAM_HOST_FIT") != nullptr;
    if (!force_host_fit) {
        StageTimer st_dev("  device median fit + labels");
        const size_t need = rgbdslam_fit_planes_workspace_bytes(n, m);
        if (_fit_ws.count < need) _fit_ws.alloc(need, false);
        check(rgbdslam_fit_planes(_ctx, _blob_dev.p, _blob_sizes_dev.p, _predicted_frame->getVMap(), _predicted_frame->getNMap(), pose,
                                  min_size_plan, min_normals_plan, _fit_ws.p, n, m), "fit_planes");
        int nb = 0;
        std::vector<float> rec((size_t)RGBDSLAM_FIT_MAX_BLOBS * RGBDSLAM_FIT_RECORD);
        check(rgbdslam_fit_planes_read(_ctx, _fit_ws.p, &nb, rec.data()), "fit_planes_read");
        if (nb <= RGBDSLAM_FIT_MAX_BLOBS) {
            _Detect = nb > 0;                                                         // RG.cpp:385-390
            if (!_Detect) return;
            auto ival = [&](int slot, int k) { int v; memcpy(&v, &rec[(size_t)slot * RGBDSLAM_FIT_RECORD + k], 4); return v; };
            std::vector<int> order((size_t)nb);
            for (int k = 0; k < nb; k++) order[(size_t)k] = k;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return ival(a, 0) < ival(b, 0); });
            uint8_t label_of_slot[RGBDSLAM_FIT_MAX_BLOBS] = {0};
            for (int slot : order) {
                if (_nbPlan >= max_labels) break;
                if (!ival(slot, 3)) continue;                                         // RG.cpp:500-516: too few normals
                const float *r = &rec[(size_t)slot * RGBDSLAM_FIT_RECORD];
                label_of_slot[slot] = (uint8_t)(_nbPlan + 1);
                for (int k = 0; k < 3; k++) _Centers.push_back(r[17 + k]);
                _Equations.insert(_Equations.end(), r + 4, r + 14);
                _nbPlan++;
            }
            check(rgbdslam_fit_planes_label(_ctx, _predicted_frame->_Label_dev, _blob_dev.p, _fit_ws.p, label_of_slot, n, m), "fit_planes_label");   // RG.cpp:590
            hip_check(hipStreamSynchronize(s), "sync");                               // label_of_slot lives on this stack frame
            return;
        }
    }
    StageTimer *st_dl = new StageTimer("  regions + 4 downloads");
    std::vector<int32_t> blob(P), sizes(P);
    std::vector<float> VMap_cpu(3 * P), NMap_cpu(3 * P);                              // RG.cpp:394-396
    std::vector<uint16_t> Label_cpu(P, 0);
    hip_check(hipMemcpyAsync(blob.data(), _blob_dev.p, 4 * P, hipMemcpyDeviceToHost, s), "blob download");
    hip_check(hipMemcpyAsync(sizes.data(), _blob_sizes_dev.p, 4 * P, hipMemcpyDeviceToHost, s), "sizes download");
    hip_check(hipMemcpyAsync(VMap_cpu.data(), _predicted_frame->getVMap(), 12 * P, hipMemcpyDeviceToHost, s), "VMap download");
    hip_check(hipMemcpyAsync(NMap_cpu.data(), _predicted_frame->getNMap(), 12 * P, hipMemcpyDeviceToHost, s), "NMap download");
    hip_check(hipStreamSynchronize(s), "sync");
    delete st_dl;
    StageTimer st_fit("  host median fit + labels");

    // blobs in FindBlobs' order = ascending first-pixel index; a blob's points = its pixels in raster order
    std::vector<size_t> roots;
    for (size_t p = 0; p < P; p++)
        if (sizes[p] > min_size_plan) roots.push_back(p);                             // RG.cpp:342-347
    _Detect = !roots.empty();                                                         // RG.cpp:385-390
    if (!_Detect) return;
    // The fit of one blob is independent of the others' (the reference runs them one after the other on one core: ~8 ms for
    // three 100 000-pixel planes, the bulk of RG::Finish's periodic cycle): one task per blob, its three normal medians side
    // by side; what depends on the order of the blobs -- plane numbers, the 255 cap -- is applied afterwards, in order.
    // Inside a blob every sum runs in the reference's pixel order, so the results are the same bits.
    struct BlobFit { bool accepted = false; float eq[10]; float center[3]; std::vector<uint32_t> px; };
    std::vector<BlobFit> fits(roots.size());
    auto fit_blob = [&](size_t root, BlobFit &out) {
        const int curr_size = sizes[root];
        std::vector<uint32_t> &px = out.px;
        px.reserve((size_t)curr_size);
        for (size_t p = root; p < P && (int)px.size() < curr_size; p++)
            if (blob[p] == (int32_t)root) px.push_back((uint32_t)p);
        std::vector<float> Nx((size_t)curr_size, 0.0f), Ny((size_t)curr_size, 0.0f), Nz((size_t)curr_size, 0.0f);
        int count = 0;
        for (int j = 0; j < curr_size; j++) {                                         // RG.cpp:479-497
            const float *N = &NMap_cpu[3 * (size_t)px[j]];
            if (N[0] == 0.0f && N[1] == 0.0f && N[2] == 0.0f) continue;               // such entries stay 0 in the lists
            count++;
            Nx[j] = (N[0] * pose[0] + N[1] * pose[4]) + N[2] * pose[8];
            Ny[j] = (N[0] * pose[1] + N[1] * pose[5]) + N[2] * pose[9];
            Nz[j] = (N[0] * pose[2] + N[1] * pose[6]) + N[2] * pose[10];
        }
        if (count < min_normals_plan) return;                                         // RG.cpp:500-516
        // RG.cpp:521-533 sorts the three lists and takes the middle elements: the same elements by selection
        auto mid = [](std::vector<float> &v) { std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return v[v.size() / 2]; };
        float nmle[3];
        nmle[0] = mid(Nx); nmle[1] = mid(Ny); nmle[2] = mid(Nz);
        float norm_nmle = sqrtf((nmle[0] * nmle[0] + nmle[1] * nmle[1]) + nmle[2] * nmle[2]);
        nmle[0] = nmle[0] / norm_nmle; nmle[1] = nmle[1] / norm_nmle; nmle[2] = nmle[2] / norm_nmle;
        std::vector<float> Dists((size_t)curr_size);
        float center_curr[3] = {0.0f, 0.0f, 0.0f};
        for (int j = 0; j < curr_size; j++) {                                         // RG.cpp:557-578
            const float *V = &VMap_cpu[3 * (size_t)px[j]];
            float pt[3];
            pt[0] = ((V[0] * pose[0] + V[1] * pose[4]) + V[2] * pose[8]) + pose[12];
            pt[1] = ((V[0] * pose[1] + V[1] * pose[5]) + V[2] * pose[9]) + pose[13];
            pt[2] = ((V[0] * pose[2] + V[1] * pose[6]) + V[2] * pose[10]) + pose[14];
            center_curr[0] = center_curr[0] + pt[0]; center_curr[1] = center_curr[1] + pt[1]; center_curr[2] = center_curr[2] + pt[2];
            Dists[j] = (pt[0] * nmle[0] + pt[1] * nmle[1]) + pt[2] * nmle[2];
        }
        for (int k = 0; k < 3; k++) out.center[k] = center_curr[k] / (float)curr_size;   // RG.cpp:581-583
        const float dist = mid(Dists);                                                // RG.cpp:592-598 (sort + middle element)
        float e1[3], e2[3];                                                           // RG.cpp:607-637
        if (nmle[0] > nmle[1] && nmle[0] > nmle[2]) { e1[0] = -nmle[1]; e1[1] = nmle[0]; e1[2] = 0.0f; }
        else if (nmle[1] > nmle[0] && nmle[1] > nmle[2]) { e1[0] = 0.0f; e1[1] = -nmle[2]; e1[2] = nmle[1]; }
        else { e1[0] = nmle[2]; e1[1] = 0.0f; e1[2] = -nmle[0]; }
        float norm_e = sqrtf((e1[0] * e1[0] + e1[1] * e1[1]) + e1[2] * e1[2]);
        e1[0] /= norm_e; e1[1] /= norm_e; e1[2] /= norm_e;
        e2[0] = nmle[1] * e1[2] - nmle[2] * e1[1];
        e2[1] = nmle[2] * e1[0] - nmle[0] * e1[2];
        e2[2] = nmle[0] * e1[1] - nmle[1] * e1[0];
        const float eq[10] = {nmle[0], nmle[1], nmle[2], e1[0], e1[1], e1[2], e2[0], e2[1], e2[2], dist};
        memcpy(out.eq, eq, sizeof(eq));
        out.accepted = true;
    };
    {
        // a bounded pool: at most 8 workers pull blob indices from a counter -- a 1080p frame with a small
        // min_size can hold hundreds of blobs, and a thread per blob (plus two per blob for the medians) could fail to start
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t workers = std::min<size_t>(roots.size(), std::min<size_t>(8, hw ? hw : 1));
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t r; (r = next.fetch_add(1)) < roots.size();) fit_blob(roots[r], fits[r]); };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < workers; w++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    for (size_t r = 0; r < roots.size(); r++) {
        if (_nbPlan >= max_labels) break;
        const BlobFit &f = fits[r];
        if (!f.accepted) continue;
        const uint16_t label = (uint16_t)(_nbPlan + 1);
        for (uint32_t p : f.px) Label_cpu[p] = label;
        for (int k = 0; k < 3; k++) _Centers.push_back(f.center[k]);
        _Equations.insert(_Equations.end(), f.eq, f.eq + 10);
        _nbPlan++;
    }
    if (LW == 2) {
        hip_check(hipMemcpy(_predicted_frame->_Label_dev, Label_cpu.data(), 2 * P, hipMemcpyHostToDevice), "label upload");
        return;
    }
    std::vector<uint8_t> narrow(Label_cpu.begin(), Label_cpu.end());
    hip_check(hipMemcpy(_predicted_frame->_Label_dev, narrow.data(), P, hipMemcpyHostToDevice), "label upload");       // RG.cpp:590
}

// RG.cpp:934-1006: planes whose label is left on at most MIN_SIZE_PLAN pixels of the model frame are dropped, the others
// renumbered in order; labels remapped (AffectPlan), equations compacted, centres = mean of the labelled points in the
// global frame (VotePlan).  Two deliberate differences: the reference adds its votes onto UNINITIALISED device buffers
// (RG.cpp:939-942: cudaMalloc without a memset) -- zeroed here -- and its unordered float atomics are a fixed summation
// order in rgbdslam_vote_plan.  The reference calls it from RG::Finish right before a detection that overwrites everything
// it produces (RG.cpp:266-267, labels cleared :313, _nbPlan = 0 :401), so Finish only runs it on request
// (finish_reorganize); as a method of its own it compacts the plane set of a live tracker.
int RG::ReorganizeEqua(float *pose) {
    if (fused) Sync();
    SyncAll();
    if (_nbPlan <= 0) return 0;
    const int n = _predicted_frame->_n, m = _predicted_frame->_m;
    DevBuf<int32_t> Buff_dev; Buff_dev.alloc((size_t)_nbPlan);
    DevBuf<float> Centers_dev; Centers_dev.alloc(3 * (size_t)_nbPlan);
    DevBuf<float> count_dev; count_dev.alloc((size_t)_nbPlan);
    check(rgbdslam_vote_plan(_ctx, _predicted_frame->_Label_dev, _predicted_frame->getVMap(), Centers_dev.p, count_dev.p, pose, _nbPlan,
                             n, m), "VotePlan_cu");                                   // RG.cpp:946
    std::vector<float> count_c((size_t)_nbPlan), centers_c(3 * (size_t)_nbPlan);
    hip_check(hipMemcpy(count_c.data(), count_dev.p, 4 * (size_t)_nbPlan, hipMemcpyDeviceToHost), "count download");     // RG.cpp:951
    hip_check(hipMemcpy(centers_c.data(), Centers_dev.p, 12 * (size_t)_nbPlan, hipMemcpyDeviceToHost), "centres download");
    std::vector<int32_t> Assoc((size_t)_nbPlan);
    int count = 0;
    for (int i = 0; i < _nbPlan; i++) Assoc[i] = ((int)count_c[i] > min_size_plan) ? ++count : 0;                        // RG.cpp:955-962
    hip_check(hipMemcpy(Buff_dev.p, Assoc.data(), 4 * (size_t)_nbPlan, hipMemcpyHostToDevice), "Assoc upload");
    check(rgbdslam_affect_plan(_ctx, _predicted_frame->_Label_dev, Buff_dev.p, n, m), "AffectPlan_cu");                  // RG.cpp:965
    std::vector<float> Eq(10 * (size_t)count), Ce(3 * (size_t)count);
    for (int i = 0; i < _nbPlan; i++) {                                              // RG.cpp:972-990
        if (Assoc[i] == 0) continue;
        memcpy(&Eq[10 * (size_t)(Assoc[i] - 1)], &_Equations[10 * (size_t)i], 40);
        for (int k = 0; k < 3; k++) Ce[3 * (size_t)(Assoc[i] - 1) + k] = centers_c[3 * (size_t)i + k] / count_c[i];
    }
    _Equations.swap(Eq); _Centers.swap(Ce);
    _nbPlan = count;
    // the reference leaves _Equations_dev NULL here (RG.cpp:997-998: the detection that follows allocates it again); a live
    // tracker needs the compacted equations on the device, and its Graph must follow the renumbering
    SyncAll();
    _graph->Compact(Assoc);
    if (count > 0) {
        _Equations_dev.alloc(10 * (size_t)count, false);
        hip_check(hipMemcpy(_Equations_dev.p, _Equations.data(), 40 * (size_t)count, hipMemcpyHostToDevice), "Equations");
    }
    return count;
}

int RG::DetectPlanes() {                                                              // RG.cpp:154-175
    if (fused) Sync();
    SyncAll();
    float pose[16];
    pack_pose(_poseR, _poset, pose);
    { StageTimer st("Graph::Clear"); _graph->Clear(); }
    DetectPrimitives(pose);
    StageTimer st("CreatePrimitives");
    CreatePrimitives(pose, true);
    return _nbPlan;
}

// RG.cpp:149-271.  ReorganizeEqua (RG.cpp:266, 934-1006) is not run: everything it produces -- remapped labels, compacted
// equations / centres, _nbPlan -- is overwritten by the DetectPrimitives call that follows it (labels cleared RG.cpp:313,
// _nbPlan = 0 RG.cpp:401); its two kernels exist as rgbdslam_vote_plan / rgbdslam_affect_plan for callers that want them.
void RG::Finish() {
    if (!finish_enabled) return;
    const bool first_due = _first_in && _count > finish_first_after;                  // RG.cpp:154
    const bool period_due = !first_due && finish_period > 0 && (_count % finish_period) == 0;   // RG.cpp:187
    if (!first_due && !period_due) return;
    // RG.cpp:177-187: `(_count % 100) == 0 && _finished && ok2Save` gates the WHOLE periodic block -- while a file Dummy.txt sits
    // in the data directory the reference neither dumps, nor resets the plane set, nor detects again.  The reference always
    // has somewhere to save to (a hard-coded directory, RG.cpp:229); here no save_dir means that nothing CAN be saved, and
    // throwing the fused planes away without a dump must be asked for (finish_reset_without_dump: bench.py's
    // --finish-period, the oracle's schedule) -- otherwise such a period is skipped like one with a Dummy.txt.
    // Host-decidable, so a skipped period does not even drain the queue.
    bool can_dump = false;
    if (period_due) {
        can_dump = !save_dir.empty();
        if (can_dump) {
            FILE *f = fopen((save_dir + "/Dummy.txt").c_str(), "r");
            if (f) { can_dump = false; fclose(f); }
        }
        const bool ok2Save = can_dump || (save_dir.empty() && finish_reset_without_dump);
        if (!ok2Save) { finish_skipped++; return; }
    }
    StageTimer st_cycle("RG::Finish (whole)");
    if (fused) {
        StageTimer st("Sync (drain the queue)");
        if (Sync() == 0) return;                                                      // lost frame: display() jumps over Finish
    }
    if (first_due) {
        DetectPlanes();                                                               // RG.cpp:163-169
        _first_in = false;
        finish_detections++;
        return;
    }
    _poseRPrev = _poseR;                                                              // RG.cpp:189-190
    _posetPrev = _poset;
    if (can_dump) { StageTimer st("SaveModel"); SaveModel(save_dir.c_str(), _indx); finish_dumps++; }   // RG.cpp:209-232 (+ SaveThread, RG.cpp:16-59)
    _indx++;                                                                          // RG.cpp:234
    if (finish_reorganize) { float pose[16]; pack_pose(_poseR, _poset, pose); ReorganizeEqua(pose); }   // RG.cpp:266 (its output is dead, see above)
    DetectPlanes();                                                                   // RG.cpp:253-268: tables freed, planes re-detected
    finish_detections++;
}

void RG::CreatePrimitives(float *pose, bool labels_from_detection) {
    std::vector<int> Assoc(_nbPlan, -1);
    int count = 0;
    for (int i = 0; i < _nbPlan; i++) {
        float scale[2] = {(float)(2.0 / RES_PLANE), (float)(2.0 / RES_PLANE)};       // RG.cpp:706
        int Size_frag[2] = {_graph->_SizeVBUFF[0], _graph->_SizeVBUFF[1]};
        float center[2], pt[2];
        const float *E = &_Equations[10 * i], *Cn = &_Centers[3 * i];
        pt[0] = (Cn[0] * E[3] + Cn[1] * E[4]) + Cn[2] * E[5];                        // RG.cpp:717-718
        pt[1] = (Cn[0] * E[6] + Cn[1] * E[7]) + Cn[2] * E[8];
        center[0] = (float)((double)pt[0] - ((double)(float)Size_frag[0] * RES_PLANE) / 2.0);   // RG.cpp:720-721
        center[1] = (float)((double)pt[1] - ((double)(float)Size_frag[1] * RES_PLANE) / 2.0);
        Primitive *ThePrim = new Primitive(&_Equations[10 * i], &_Equations[10 * i + 3], &_Equations[10 * i + 6],
                                           _Equations[10 * i + 9], scale, Size_frag, center, &_Centers[3 * i]);
        if (!_graph->AddPrimitive(ThePrim)) { delete ThePrim; continue; }            // RG.cpp:733-737
        ThePrim->Allocate();
        Assoc[i] = count++;
    }
    std::vector<float> Eq(10 * (size_t)count), Ce(3 * (size_t)count);
    for (int i = 0; i < _nbPlan; i++)
        if (Assoc[i] != -1) {
            memcpy(&Eq[10 * Assoc[i]], &_Equations[10 * i], 40);
            memcpy(&Ce[3 * Assoc[i]], &_Centers[3 * i], 12);
        }
    _Equations.swap(Eq);
    _Centers.swap(Ce);
    _nbPlan = count;
    if (count == 0) return;
    _Equations_dev.alloc(10 * (size_t)count, false);                                 // RG.cpp:779-780
    hip_check(hipMemcpy(_Equations_dev.p, _Equations.data(), 40 * (size_t)count, hipMemcpyHostToDevice), "Equations");
    if (labels_from_detection) {                                                     // RG.cpp:786: grow the blob labels
        check(rgbdslam_segment(_ctx, _predicted_frame->_Label_dev, _predicted_frame->_Label_dev_swap, _predicted_frame->getVMap(),
                               _predicted_frame->getNMap(), pose, _Equations_dev.p, _nbPlan, 0.1f, 0.8f, _predicted_frame->_n,
                               _predicted_frame->_m), "Segment_cu");
        return;
    }
    // injected planes (SetPlanes) come without blob labels: they are seeded by plane membership with the
    // same thresholds (ACCEPT_THRESH 0.1, 0.8)
    check(rgbdslam_seed_labels(_ctx, _predicted_frame->_Label_dev, _predicted_frame->getVMap(), _predicted_frame->getNMap(), pose,
                               _Equations_dev.p, _nbPlan, 0.1f, 0.8f, _predicted_frame->_n, _predicted_frame->_m), "seed_labels");
}

rgbdslam_icp_params RG::icp_params() const {
    rgbdslam_icp_params prm;
    prm.lvl = _lvl;
    for (int k = 0; k < 3; k++) prm.max_iter[k] = _max_iter[k];
    prm.distThres = _distThres; prm.angleThres = _angleThres;
    prm.thres_cvg_alpha = _thres_cvg_alpha; prm.thres_cvg_tr = _thres_cvg_tr;
    return prm;
}

// display() (GP-demo.cpp:138-296) as a fixed launch sequence: no host decision depends on device
// results, so nothing here waits for the GPU.
int RG::StepAsync() {
    int n = _frame->getN(), m = _frame->getM();
    size_t pitch = 4 * (size_t)m;
    bool first = (_count == 2);
    hipStream_t main = ctx_stream(_ctx);
    const bool ovl = _frame->load_stream() != nullptr;
    const bool ov = fused && overlap && (_overlap_mode & 2);
    if (ovl) _frame->wait_ready(main);
    if (first) _predicted_frame->ReseedFused(nullptr, 1, _frame);                   // GP-demo.cpp:164-167
    StageRange r_frame("rgbdslam:display() pass");
    rgbdslam_icp_params prm = icp_params();
    const int slot = frames_done % HISTORY_CAP;
    stage_mark(2);
    { StageRange r("stage 2+3: ICP + 27-term reduction (RG::AlignFrameGaussNewton)");
    if (_frame->getAux()) check(rgbdslam_set_frame_aux(_ctx, nullptr, _frame->getAux()), "set_frame_aux");   // this call's records
    check(rgbdslam_track_frame(_ctx, &prm, _frame->getVMap(), _frame->getNMap(), _frame->getRGB(), _frame->getGx(), pitch, _frame->getGy(), pitch,
                               _predicted_frame->getVMap(), _predicted_frame->getNMap(), _predicted_frame->getRGB(), n, m, _pb_dev,
                               _history_dev.p, slot), "track_frame");               // GP-demo.cpp:189 + 201-245
    }
    stage_mark(3);
    // the previous frame's Bump update (other stream) reads the model this frame is about to rewrite
    if (ov && _update_pending) hip_check(hipStreamWaitEvent(main, _ev_updated, 0), "wait for the previous update");
    // GP-demo.cpp:215-237 + ProjectLabel gather; if lost: Cpy + InitMask instead (GP-demo.cpp:190-191)
    roctxRangePush("stage 5: prediction (Transform + raster + ReadFrame + ProjectMask/Label)");
    if (_deferred.valid) {                                                          // the previous frame's apply pass rides along
        _deferred.valid = false;
        _predicted_frame->PredictFused(_pb_dev, Znear, Zfar, _frame, &_deferred.t, _deferred.hist);
    } else {
        _predicted_frame->PredictFused(_pb_dev, Znear, Zfar, _frame);
    }
    roctxRangePop();
    stage_mark(4);
    roctxRangePush("stage 4: fusion (Segment + Merge + Graph::Update; next frame's stage 1 may ride)");
    UpdateTables upd;
    const bool claim_in_merge = !ov && _graph->begin_update(&upd);
    // equations without Bump images (SetPlaneEquations): the same one-launch form, its claim pass switched off
    const bool eq_only = !ov && !claim_in_merge && _equations_only && _nbPlan > 0;
    if (eq_only) upd = UpdateTables{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0, 0, 0, 0};
    // when the caller has named the next device frame, its LoadFrame (Frame.cpp:43-77) rides on a launch of this frame
    const bool next_rides = (claim_in_merge || eq_only) && _next_hint.valid && !ovl && _frame->has_gradients() && !_frame->bilateral();
    bool next_done = false;
    if (eq_only) {
        if (next_rides) {
            _predicted_frame->SegmentMergeClaimFused(_pb_dev, _Equations_dev.p, _nbPlan, _frame, upd, _next_hint.depth,
                                                     _next_hint.depth_pitch, _next_hint.bgr, _next_hint.bgr_pitch);
            next_done = true;
        } else {
            _predicted_frame->SegmentMergeClaimFused(_pb_dev, _Equations_dev.p, _nbPlan, _frame, upd);
        }
    } else if (claim_in_merge && _nbPlan > 0) {
        // Frame.cpp:242 + 395, GP-demo.cpp:249 and the claim pass of :260 in one launch
        if (next_rides && _next_on_merge) {
            _predicted_frame->SegmentMergeClaimFused(_pb_dev, _Equations_dev.p, _nbPlan, _frame, upd, _next_hint.depth,
                                                     _next_hint.depth_pitch, _next_hint.bgr, _next_hint.bgr_pitch);
            next_done = true;
        } else {
            _predicted_frame->SegmentMergeClaimFused(_pb_dev, _Equations_dev.p, _nbPlan, _frame, upd);
        }
    } else {
        _predicted_frame->NormalSegmentFused(_pb_dev, _Equations_dev.p, _nbPlan);   // Frame.cpp:242 + 395
        _predicted_frame->MergeFused(_pb_dev, _frame, claim_in_merge ? &upd : nullptr);  // GP-demo.cpp:249 (+ claim pass of :260)
    }
    if (ovl) _frame->release(main);                                                 // last reader of this frame's buffer set
    if (!ov) {
        // GP-demo.cpp:260; the next frame's LoadFrame rides here if it has not yet: every reader of this frame's maps
        // has been enqueued, so the same buffers take the next frame's
        if (next_rides && !next_done)
            _graph->ApplyTrackedAndPreprocess(_predicted_frame, _pb_dev, upd, _frame, _next_hint.depth, _next_hint.depth_pitch,
                                              _next_hint.bgr, _next_hint.bgr_pitch);
        else if (claim_in_merge && _defer_apply && _defer_apply_enabled && _next_hint.valid) {
            _deferred.t = upd;                                                      // -> the next frame's draw launch
            _deferred.hist = _history_dev.p + 16 * (size_t)slot;
            _deferred.valid = true;
        }
        else if (claim_in_merge) _graph->ApplyTracked(_predicted_frame, _pb_dev, upd);
    } else {
        if (_graph->_nodes.size() != 0) {
            // Graph::Update on its own stream, posed by this frame's history entry (the pose block is rewritten
            // by the next frame's tracking, which now runs concurrently)
            hip_check(hipEventRecord(_ev_merged, main), "hipEventRecord");
            hip_check(hipStreamWaitEvent(_update_stream, _ev_merged, 0), "hipStreamWaitEvent");
            check(rgbdslam_set_stream(_ctx, _update_stream), "set_stream");
            try { _graph->UpdateFromHistory(_predicted_frame, _history_dev.p + 16 * (size_t)slot); }
            catch (...) { (void)rgbdslam_set_stream(_ctx, main); throw; }
            check(rgbdslam_set_stream(_ctx, main), "set_stream");
            hip_check(hipEventRecord(_ev_updated, _update_stream), "hipEventRecord");
            _update_pending = true;
        }
    }
    roctxRangePop();
    stage_mark(5);
    _next_hint.valid = false;
    frames_done++;
    { StageRange r("RG::Finish"); Finish(); }                                       // GP-demo.cpp:274
    return first ? 2 : 1;
}

void RG::FlushDeferredApply() {
    if (!_deferred.valid) return;
    _deferred.valid = false;
    const UpdateTables &t = _deferred.t;
    check(rgbdslam_update_apply_from_history(_ctx, _deferred.hist, t.Bump, t.RGB, t.Mask, t.Param, t.Claim, t.stamp,
                                             _predicted_frame->getVMap(), _predicted_frame->getRGB(), _predicted_frame->_Mask_dev,
                                             _predicted_frame->_Label_dev, t.N_prim, t.M_prim, _predicted_frame->_n, _predicted_frame->_m,
                                             t.plane_lo, t.plane_hi), "Update_cu (deferred apply)");
}

void RG::SyncAll() {
    FlushDeferredApply();
    if (_load_stream) hip_check(hipStreamSynchronize(_load_stream), "sync load stream");
    check(rgbdslam_synchronize(_ctx), "synchronize");
    if (_update_stream) hip_check(hipStreamSynchronize(_update_stream), "sync update stream");
    _update_pending = false;
}

// Sync() only if the host's view is behind the device: frames enqueued since the last Sync, or an apply pass pending
void RG::SyncIfStale() {
    static const bool always = getenv("RGBDSLAM_SEQ_LEADING_SYNC") != nullptr;      // A/B switch: the unconditional Sync of before
    if (always || _history_synced != frames_done || _deferred.valid || _update_pending) Sync();
}

int RG::Sync() {
    FlushDeferredApply();
    // the pose block and the new history entries (at most two ring segments) go to their pinned mirrors with ONE launch that
    // stores them there (rgbdslam_copy_segments) instead of two or three copy commands, each a blit launch of its own;
    // RGBDSLAM_SYNC_COPY_CMDS=1 restores the copy commands
    static const bool copy_cmds = getenv("RGBDSLAM_SYNC_COPY_CMDS") != nullptr;
    hipStream_t s = ctx_stream(_ctx);
    const void *src[3]; void *dst[3]; size_t bytes[3]; int nseg = 0;
    src[nseg] = _pb_dev; dst[nseg] = _pb_host; bytes[nseg] = sizeof(rgbdslam_pose_block); nseg++;
    int lo = _history_synced, hi = frames_done;
    if (hi - lo > HISTORY_CAP) lo = hi - HISTORY_CAP;
    for (int f = lo; f < hi;) {                                                     // ring segments
        int slot = f % HISTORY_CAP, cnt = hi - f;
        if (slot + cnt > HISTORY_CAP) cnt = HISTORY_CAP - slot;
        src[nseg] = _history_dev.p + 16 * (size_t)slot; dst[nseg] = _history_host + 16 * (size_t)slot; bytes[nseg] = 64 * (size_t)cnt; nseg++;
        f += cnt;
    }
    if (copy_cmds) {
        for (int k = 0; k < nseg; k++) hip_check(hipMemcpyAsync(dst[k], src[k], bytes[k], hipMemcpyDeviceToHost, s), "pose / history download");
    } else {
        check(rgbdslam_copy_segments(_ctx, src, dst, bytes, nseg), "copy_segments");
    }
    check(rgbdslam_synchronize(_ctx), "sync");                                     // polls first, then blocks (rgbdslam_set_sync_spin)
    if (_update_pending) { hip_check(hipStreamSynchronize(_update_stream), "sync update stream"); _update_pending = false; }
    _history_synced = hi;
    memcpy(_poseR.a, _pb_host->pose_state, 36); memcpy(_poset.v, _pb_host->pose_state + 9, 12);
    memcpy(_poseRPrev.a, _pb_host->pose_state + 12, 36); memcpy(_posetPrev.v, _pb_host->pose_state + 21, 12);
    last_iterations = _pb_host->status[1];
    last_level_iterations = _pb_host->status[2];
    tracked_total = _pb_host->status[3];
    return _pb_host->status[0];
}

void RG::ExportFrame(void *dst_dev, size_t off_vmap, size_t off_rgbd, size_t off_mask, size_t off_hist) {
    if (!dst_dev || frames_done == 0) throw Error("ExportFrame: no frame processed yet", RGBDSLAM_E_STATE);
    hipStream_t s = ctx_stream(_ctx);
    const size_t P = (size_t)_predicted_frame->_n * _predicted_frame->_m;
    char *d = (char *)dst_dev;
    hip_check(hipMemcpyAsync(d + off_vmap, _predicted_frame->getVMap(), 12 * P, hipMemcpyDeviceToDevice, s), "export VMap");
    hip_check(hipMemcpyAsync(d + off_rgbd, _predicted_frame->getRGB(), 16 * P, hipMemcpyDeviceToDevice, s), "export RGBD");
    hip_check(hipMemcpyAsync(d + off_mask, _predicted_frame->_Mask_dev, P, hipMemcpyDeviceToDevice, s), "export Mask");
    hip_check(hipMemcpyAsync(d + off_hist, _history_dev.p + 16 * (size_t)((frames_done - 1) % HISTORY_CAP), 64, hipMemcpyDeviceToDevice, s),
              "export history entry");
}

// The thin form of ExportFrame (rgbdslam_export_thin): RGBD.w, the colour bytes and the mask -- what the plane owners cannot
// derive -- plus the history entry; one kernel and one 64-byte copy.
// off_label != SIZE_MAX (16-bit labels only): the label image travels too.  One launch, no stream-ordered copies.
void RG::ExportFrameThin(void *dst_dev, size_t off_w, size_t off_rgb8, size_t off_mask, size_t off_hist, size_t off_label) {
    if (!dst_dev || frames_done == 0) throw Error("ExportFrameThin: no frame processed yet", RGBDSLAM_E_STATE);
    const bool labels = off_label != (size_t)-1;
    if (labels && _predicted_frame->label_width() != 2) throw Error("ExportFrameThin: the pack carries 16-bit labels (SetLabel16)", RGBDSLAM_E_STATE);
    char *d = (char *)dst_dev;
    check(rgbdslam_export_thin_l16(_ctx, _predicted_frame->getRGB(), _predicted_frame->_Mask_dev,
                                   labels ? (const uint16_t *)_predicted_frame->_Label_dev : nullptr,
                                   _history_dev.p + 16 * (size_t)((frames_done - 1) % HISTORY_CAP), (float *)(d + off_w), (uint8_t *)(d + off_rgb8),
                                   (uint8_t *)(d + off_mask), labels ? (uint16_t *)(d + off_label) : nullptr, (float *)(d + off_hist),
                                   _predicted_frame->_n, _predicted_frame->_m), "export_thin");
}

const float *RG::history_host(int frame_index) { return _history_host + 16 * (size_t)(frame_index % HISTORY_CAP); }

int RG::Step() {
    if (fused) {
        int r = StepAsync();
        int ok = Sync();
        return ok ? r : 0;
    }
    float pose[16], pose_inv[16];
    bool first = false;
    if (_count == 2) {                                                              // GP-demo.cpp:164-167
        _predicted_frame->Cpy(_frame);
        _predicted_frame->InitMask();
        first = true;
    }
    if (!AlignFrameGaussNewton()) {                                                 // GP-demo.cpp:189-196
        _predicted_frame->Cpy(_frame);
        _predicted_frame->InitMask();
        _poseRPrev = _poseR;
        _posetPrev = _poset;
        if (_dataset) { Vec3 lost = _poset; lost(0) = 100.0f; _traj_R.push_back(_poseR); _traj_t.push_back(lost); }
        return 0;
    }
    if (_dataset) { _traj_R.push_back(_poseR); _traj_t.push_back(_poset); }
    Mat3 poseRinv = _poseR.inverse();                                               // GP-demo.cpp:201-211
    Vec3 posetinv = poseRinv.neg() * _poset;
    Mat3 poseR = poseRinv * _poseRPrev;
    Vec3 poset = poseRinv * _posetPrev + posetinv;
    pack_pose(poseR, poset, pose_inv);
    _predicted_frame->Transform(pose_inv, true);                                    // GP-demo.cpp:215
    _predicted_frame->ReadFrame(Znear, Zfar);                                       // GP-demo.cpp:222-223
    poseRinv = poseR.inverse();                                                     // GP-demo.cpp:225-236
    posetinv = poseRinv.neg() * poset;
    pack_pose(poseRinv, posetinv, pose_inv);
    pack_pose(_poseR, _poset, pose);
    _predicted_frame->ProjectMask(pose_inv);                                        // GP-demo.cpp:237
    _predicted_frame->ProjectLabel(pose_inv, pose, _Equations_dev.p, _nbPlan);      // GP-demo.cpp:238
    _poseRPrev = _poseR;                                                            // GP-demo.cpp:244-245
    _posetPrev = _poset;
    _predicted_frame->Merge(_frame);                                                // GP-demo.cpp:249
    _graph->Update(_predicted_frame, pose);                                         // GP-demo.cpp:260
    Finish();                                                                       // GP-demo.cpp:274
    return first ? 2 : 1;
}

}  // namespace slam

// =====================================================================================================
// C facade over the classes (for ctypes drivers: tests, bench.py).  Exceptions become error codes.
// =====================================================================================================
using slam::RG;

#define FACADE_TRY(stmt)                                   \
    try { stmt; }                                          \
    catch (const slam::Error &e) { g_last_error = e.what(); return e.code ? e.code : -1; } \
    catch (const std::exception &e) { g_last_error = e.what(); return -1; }

static thread_local std::string g_last_error;

extern "C" {

const char *slamhost_last_error(void) { return g_last_error.c_str(); }

int slamhost_rg_create(void **out, int device, void *stream, int n, int m, const float *calib11, int lvl, float distThres,
                       float angleThres, int max_planes, int N_prim, int M_prim) {
    if (!out || !calib11 || n < 2 || m < 2) return RGBDSLAM_E_ARG;
    FACADE_TRY(*out = new RG(device, stream, n, m, calib11, lvl, distThres, angleThres, max_planes, N_prim, M_prim));
    return 0;
}
int slamhost_rg_destroy(void *rg) {
    delete (RG *)rg;
    return 0;
}
int slamhost_rg_set_fused(void *rg, int on);
int slamhost_rg_set_device_loop(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    ((RG *)rg)->device_loop = on != 0;
    if (!on) return slamhost_rg_set_fused(rg, 0);              // the per-iteration host solve is the unfused path
    return 0;
}
int slamhost_rg_set_fused(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    RG *r = (RG *)rg;
    if (r->_count != 1) return RGBDSLAM_E_STATE;               // choose before the first frame
    r->fused = on != 0;
    r->_frame->EnableGradients(r->fused);
    return 0;
}
// poses of frames [first, first+count) as 16 floats each: poseR(9) poset(3) tracked iterations level_iters 0
int slamhost_rg_history(void *rg, int first, int count, float *out) {
    if (!rg || !out || first < 0 || count < 0) return RGBDSLAM_E_ARG;
    RG *r = (RG *)rg;
    if (first + count > r->frames_done || r->frames_done - first > RG::HISTORY_CAP) return RGBDSLAM_E_LIMIT;
    FACADE_TRY(r->Sync());
    for (int k = 0; k < count; k++) memcpy(out + 16 * (size_t)k, r->history_host(first + k), 64);
    return 0;
}
// persistent-loop ICP (rgbdslam_set_icp_loop): one launch per frame for the whole coarse-to-fine loop
int slamhost_rg_set_icp_loop(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    return rgbdslam_set_icp_loop(((RG *)rg)->ctx(), on);
}
int slamhost_rg_icp_loop_faults(void *rg, int *faults) {
    if (!rg || !faults) return RGBDSLAM_E_ARG;
    return rgbdslam_icp_loop_faults(((RG *)rg)->ctx(), faults);
}
int slamhost_rg_set_index_buffer(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    ((RG *)rg)->_predicted_frame->use_index_buffer = on != 0;
    if (on) return slamhost_rg_set_fused(rg, 0);               // an explicit GL_QUADS index buffer only exists unfused
    return 0;
}
int slamhost_rg_set_plane_range(void *rg, int lo, int hi) {
    if (!rg) return RGBDSLAM_E_ARG;
    ((RG *)rg)->_graph->plane_lo = lo;
    ((RG *)rg)->_graph->plane_hi = hi;
    return 0;
}
int slamhost_rg_load_frame(void *rg, const uint16_t *depth_host, const uint8_t *bgr_host) {
    if (!rg || !depth_host || !bgr_host) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->LoadFrame(depth_host, bgr_host));
    return 0;
}
int slamhost_rg_load_frame_device(void *rg, const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch) {
    if (!rg || !depth_dev || !bgr_dev) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->LoadFrameDevice(depth_dev, depth_pitch, bgr_dev, bgr_pitch));
    return 0;
}
int slamhost_rg_set_planes(void *rg, const float *equations, const float *centers, int nb) {
    if (!rg || !equations || !centers || nb <= 0) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SetPlanes(equations, centers, nb));
    return 0;
}
// *status: 2 = model seeded, 1 = tracked, 0 = lost (model re-seeded)
int slamhost_rg_step(void *rg, int *status) {
    if (!rg) return RGBDSLAM_E_ARG;
    int st = 0;
    FACADE_TRY(st = ((RG *)rg)->Step());
    if (status) *status = st;
    return 0;
}
// fused pipeline: enqueue one display() pass without waiting for it (poses via slamhost_rg_history / _synchronize)
// names the device frame the NEXT load_frame_device will pass (RG::HintNextDeviceFrame)
int slamhost_rg_hint_next_device_frame(void *rg, const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch) {
    if (!rg) return RGBDSLAM_E_ARG;
    ((RG *)rg)->HintNextDeviceFrame(depth_dev, depth_pitch, bgr_dev, bgr_pitch);
    return 0;
}
int slamhost_rg_step_async(void *rg) {
    if (!rg) return RGBDSLAM_E_ARG;
    if (!((RG *)rg)->fused) return RGBDSLAM_E_STATE;
    FACADE_TRY(((RG *)rg)->StepAsync());
    return 0;
}
// Runs `count` display() passes over device-resident frames (depth_dev[k], bgr_dev[k]); returns
// the number of tracked frames in *tracked.  Fused pipeline: the whole sequence is enqueued without a host wait and
// synchronised once at the end (a frame on which RG::Finish runs a detection also waits, by design); unfused pipeline:
// one host sync per frame (the pose read).
int slamhost_rg_run_device_sequence(void *rg, const uint16_t *const *depth_dev, const uint8_t *const *bgr_dev, int count,
                                    size_t depth_pitch, size_t bgr_pitch, int *tracked) {
    if (!rg || !depth_dev || !bgr_dev || count < 0) return RGBDSLAM_E_ARG;
    int ok = 0;
    RG *r = (RG *)rg;
    FACADE_TRY(if (r->fused) {
        static const bool timing = getenv("RGBDSLAM_SEQ_TIMING") != nullptr;    // host-side phases of the call, to stderr
        const auto c0 = std::chrono::steady_clock::now();
        r->SyncIfStale();                                      // tracked_total is current unless frames were enqueued since the last Sync
        const auto c1 = std::chrono::steady_clock::now();
        long before = r->tracked_total;
        for (int k = 0; k < count; k++) {
            r->LoadFrameDevice(depth_dev[k], depth_pitch, bgr_dev[k], bgr_pitch);
            if (k + 1 < count) r->HintNextDeviceFrame(depth_dev[k + 1], depth_pitch, bgr_dev[k + 1], bgr_pitch);
            r->_defer_apply = k + 1 < count;                   // the next StepAsync is certain to come
            try { r->StepAsync(); } catch (...) { r->_defer_apply = false; throw; }
        }
        r->_defer_apply = false;
        const auto c2 = std::chrono::steady_clock::now();
        r->Sync();                                             // the only host<->device sync of the sequence
        ok = (int)(r->tracked_total - before);
        if (timing) {
            const auto c3 = std::chrono::steady_clock::now();
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
                return std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
            fprintf(stderr, "[seq timing] frames %d: leading sync %.1f us, enqueue %.1f us, final sync %.1f us, total %.1f us\n", count,
                    us(c0, c1), us(c1, c2), us(c2, c3), us(c0, c3));
        }
    } else for (int k = 0; k < count; k++) {
        r->LoadFrameDevice(depth_dev[k], depth_pitch, bgr_dev[k], bgr_pitch);
        if (r->Step() != 0) ok++;
    });
    if (tracked) *tracked = ok;
    return 0;
}
int slamhost_rg_synchronize(void *rg) {
    if (!rg) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SyncAll());
    return 0;
}
// fused pipeline: 1 = preprocessing of the next frame and the Bump update run on auxiliary streams (default 0)
int slamhost_rg_set_overlap(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    if (((RG *)rg)->_count != 1) return RGBDSLAM_E_STATE;
    ((RG *)rg)->overlap = on != 0;
    return 0;
}
// mode bits: 1 = frames are loaded (uploaded + preprocessed) on the load stream, into alternating buffer sets; 2 = the Bump
// update runs on the update stream.  0 = everything on one stream.  Before the first frame only.
int slamhost_rg_set_overlap_mode(void *rg, int mode) {
    if (!rg || mode < 0 || mode > 3) return RGBDSLAM_E_ARG;
    if (((RG *)rg)->_count != 1) return RGBDSLAM_E_STATE;
    ((RG *)rg)->overlap = mode != 0;
    ((RG *)rg)->set_overlap_mode(mode);
    return 0;
}
// 1: the load stream and the frame's stream are ordered by events instead of stream value operations (2.5x the cost per
// hand-over; needed under counter-collecting profilers, which never let the producer queue run while the consumer waits on a
// value).  Before the first frame only.  Same switch as RGBDSLAM_STREAM_EVENTS=1.
int slamhost_rg_set_stream_events(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    if (((RG *)rg)->_count != 1) return RGBDSLAM_E_STATE;
    FACADE_TRY(((RG *)rg)->_frame->set_stream_events(on != 0));
    return 0;
}
int slamhost_rg_export_frame_thin_labels(void *rg, void *dst_dev, size_t off_w, size_t off_rgb8, size_t off_mask, size_t off_hist, size_t off_label) {
    if (!rg) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->ExportFrameThin(dst_dev, off_w, off_rgb8, off_mask, off_hist, off_label));
    return 0;
}
int slamhost_rg_export_frame_thin(void *rg, void *dst_dev, size_t off_w, size_t off_rgb8, size_t off_mask, size_t off_hist) {
    if (!rg) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->ExportFrameThin(dst_dev, off_w, off_rgb8, off_mask, off_hist));
    return 0;
}
int slamhost_rg_get_pose(void *rg, float *R9_cm, float *t3) {
    if (!rg || !R9_cm || !t3) return RGBDSLAM_E_ARG;
    memcpy(R9_cm, ((RG *)rg)->_poseR.a, 36);
    memcpy(t3, ((RG *)rg)->_poset.v, 12);
    return 0;
}
// ---- dataset I/O ---------------------------------------------------------------------
int slamhost_rg_open_dataset(void *rg, const char *path, int Sizedata, int real, int RGBDRep, int *size) {
    if (!rg || !path) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->OpenDataset(path, Sizedata, real != 0, RGBDRep != 0));
    if (size) *size = ((RG *)rg)->_dataset->size();
    return 0;
}
int slamhost_rg_load_next_frame(void *rg, int *loaded) {
    if (!rg) return RGBDSLAM_E_ARG;
    bool ok = false;
    FACADE_TRY(ok = ((RG *)rg)->LoadFrame());
    if (loaded) *loaded = ok ? 1 : 0;
    return 0;
}
int slamhost_rg_save_trajectory(void *rg, const char *filename) {
    if (!rg || !filename) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SaveTrajectory(filename));
    return 0;
}
static int join_out(const std::vector<std::string> &a, const std::vector<std::string> &b, char *out, size_t cap, int *count) {
    std::string s;
    for (size_t k = 0; k < a.size(); k++) s += a[k] + "\t" + b[k] + "\n";
    if (count) *count = (int)a.size();
    if (s.size() + 1 > cap) return RGBDSLAM_E_LIMIT;
    memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}
// lines "depth_name\tcolor_name\n" of the associated pairs, in order
int slamhost_load_timestamps(const char *path, int associated, char *out, size_t cap, int *count) {
    if (!path || !out) return RGBDSLAM_E_ARG;
    std::vector<std::string> d, c;
    FACADE_TRY(if (associated) slam::LoadAssociatedTimeStamps(&d, &c, path); else slam::LoadTimeStamps(&d, &c, path));
    return join_out(d, c, out, cap, count);
}
int slamhost_dataset_frame_files(const char *path, int Sizedata, int real, int RGBDRep, int count, char *out, size_t cap, int *size) {
    if (!path || !out) return RGBDSLAM_E_ARG;
    std::vector<std::string> d(1), c(1);
    FACADE_TRY(slam::Dataset ds(path, Sizedata, real != 0, RGBDRep != 0); if (size) *size = ds.size(); ds.frame_files(count, &d[0], &c[0]));
    return join_out(d, c, out, cap, nullptr);
}
int slamhost_read_image(const char *filename, int *rows, int *cols, int *channels, int *bits, void *buf, size_t cap) {
    if (!filename || !rows || !cols || !channels || !bits) return RGBDSLAM_E_ARG;
    slam::Image img;
    std::string f(filename);
    FACADE_TRY(img = (f.size() > 4 && f.substr(f.size() - 4) == ".png") ? slam::read_png(f) : slam::read_tiff(f));
    *rows = img.rows; *cols = img.cols; *channels = img.channels; *bits = img.bits;
    if (!buf) return 0;                                        // size query
    if (img.data.size() > cap) return RGBDSLAM_E_LIMIT;
    memcpy(buf, img.data.data(), img.data.size());
    return 0;
}
int slamhost_write_image(const char *filename, int rows, int cols, int channels, int bits, const void *buf) {
    if (!filename || !buf || rows <= 0 || cols <= 0) return RGBDSLAM_E_ARG;
    slam::Image img;
    img.rows = rows; img.cols = cols; img.channels = channels; img.bits = bits;
    img.data.assign((const uint8_t *)buf, (const uint8_t *)buf + (size_t)rows * cols * channels * (bits / 8));
    std::string f(filename);
    FACADE_TRY(if (f.size() > 4 && f.substr(f.size() - 4) == ".png") slam::write_png(f, img); else slam::write_tiff(f, img));
    return 0;
}
void slamhost_matrix2quaternion(const float *R_cm, float *q_wxyz) {
    slam::Mat3 R; memcpy(R.a, R_cm, 36);
    slam::matrix2quaternion(R, q_wxyz);
}
// names: `count` entries separated by '\n'
int slamhost_save_trajectory(const char *filename, const float *R9_cm, const float *t3, const char *names, int count) {
    if (!filename || !R9_cm || !t3 || !names || count < 0) return RGBDSLAM_E_ARG;
    std::vector<slam::Mat3> Rs((size_t)count);
    std::vector<slam::Vec3> ts((size_t)count);
    std::vector<std::string> st;
    for (int k = 0; k < count; k++) { memcpy(Rs[k].a, R9_cm + 9 * k, 36); memcpy(ts[k].v, t3 + 3 * k, 12); }
    std::string all(names);
    for (size_t pos = 0; st.size() < (size_t)count;) {
        size_t nl = all.find('\n', pos);
        st.push_back(all.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos));
        if (nl == std::string::npos) break;
        pos = nl + 1;
    }
    if (st.size() != (size_t)count) return RGBDSLAM_E_ARG;
    FACADE_TRY(slam::SaveTrajectory(filename, Rs, ts, st));
    return 0;
}
// ---- model dump format ---------------------------------------------------------------
int slamhost_rg_save_model(void *rg, const char *dir, int indx) {
    if (!rg || !dir) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SaveModel(dir, indx));
    return 0;
}
// eq19 per plane: nmle(3) dist e1(3) e2(3) shift(2) scale(2) size(2) anchor(3); size 0 x 0 = plane without data
static void patch_from(const float *e, slam::PlanarPatch *P) {
    memcpy(P->nmle, e, 12); P->dist = e[3]; memcpy(P->e1, e + 4, 12); memcpy(P->e2, e + 7, 12); memcpy(P->Shift, e + 10, 8);
    memcpy(P->scale, e + 12, 8); P->Size[0] = (int)e[14]; P->Size[1] = (int)e[15]; memcpy(P->anchor, e + 16, 12);
}
static void patch_to(const slam::PlanarPatch &P, float *e) {
    memcpy(e, P.nmle, 12); e[3] = P.dist; memcpy(e + 4, P.e1, 12); memcpy(e + 7, P.e2, 12); memcpy(e + 10, P.Shift, 8);
    memcpy(e + 12, P.scale, 8); e[14] = (float)P.Size[0]; e[15] = (float)P.Size[1]; memcpy(e + 16, P.anchor, 12);
}
// Clean (if `clean`) + Graph::save of host arrays; eq19 is updated in place with the cleaned shift/size
int slamhost_save_graph(const char *dir, int indx, const float *pose16, int nb, float *eq19, const uint16_t *const *bumps,
                        const uint8_t *const *rgbs, const uint8_t *const *masks, int clean) {
    if (!dir || !pose16 || nb < 0 || (nb && (!eq19 || !bumps || !rgbs || !masks))) return RGBDSLAM_E_ARG;
    std::vector<slam::PlanarPatch> nodes((size_t)nb);
    for (int k = 0; k < nb; k++) {
        slam::PlanarPatch &P = nodes[k];
        patch_from(eq19 + 19 * k, &P);
        size_t C = (size_t)P.Size[0] * P.Size[1];
        if (!C || !bumps[k]) continue;
        P.Bump.assign(bumps[k], bumps[k] + 3 * C); P.RGB.assign(rgbs[k], rgbs[k] + 3 * C); P.Mask.assign(masks[k], masks[k] + C);
        P.has_data = true;
        if (clean) P.Clean();
    }
    FACADE_TRY(slam::SaveGraph(dir, indx, pose16, nodes));
    for (int k = 0; k < nb; k++) {
        if (!nodes[k].has_data) { nodes[k].Size[0] = nodes[k].Size[1] = 0; }
        patch_to(nodes[k], eq19 + 19 * k);
    }
    return 0;
}
int slamhost_save_frame(const char *dir, int indx, const float *rgbd, int n, int m) {
    if (!dir || !rgbd || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    FACADE_TRY(slam::SaveFrame(dir, indx, rgbd, n, m));
    return 0;
}
// reader: *ready = 0 while the dump's flag file is missing
int slamhost_model_open(const char *dir, int indx, void **out, int *ready) {
    if (!dir || !out || !ready) return RGBDSLAM_E_ARG;
    slam::LoadedModel *M = new slam::LoadedModel();
    bool ok = false;
    try { ok = slam::LoadModel(dir, indx, M); }
    catch (const std::exception &e) { delete M; g_last_error = e.what(); return RGBDSLAM_E_ARG; }
    *ready = ok ? 1 : 0;
    if (!ok) { delete M; M = nullptr; }
    *out = M;
    return 0;
}
int slamhost_model_close(void *h) { delete (slam::LoadedModel *)h; return 0; }
int slamhost_model_count(void *h) { return h ? (int)((slam::LoadedModel *)h)->patches.size() : -1; }
int slamhost_model_pose(void *h, float *pose16) {
    if (!h || !pose16) return RGBDSLAM_E_ARG;
    memcpy(pose16, ((slam::LoadedModel *)h)->pose, 64);
    return 0;
}
int slamhost_model_plane(void *h, int k, float *eq19) {
    slam::LoadedModel *M = (slam::LoadedModel *)h;
    if (!M || !eq19 || k < 0 || k >= (int)M->patches.size()) return RGBDSLAM_E_ARG;
    patch_to(M->patches[k], eq19);
    return 0;
}
int slamhost_model_plane_data(void *h, int k, uint16_t *bump, uint8_t *rgb, uint16_t *mask) {
    slam::LoadedModel *M = (slam::LoadedModel *)h;
    if (!M || !bump || !rgb || !mask || k < 0 || k >= (int)M->patches.size()) return RGBDSLAM_E_ARG;
    const slam::PlanarPatch &P = M->patches[k];
    memcpy(bump, P.Bump.data(), P.Bump.size() * 2); memcpy(rgb, P.RGB.data(), P.RGB.size()); memcpy(mask, P.Mask.data(), P.Mask.size() * 2);
    return 0;
}
// key-frame images: which = 0 depth (u16 [n,m]), 1 colour (u16 [n,m,3], B,G,R); buf = NULL queries the shape
int slamhost_model_keyframe(void *h, int which, int *rows, int *cols, uint16_t *buf) {
    slam::LoadedModel *M = (slam::LoadedModel *)h;
    if (!M || !rows || !cols) return RGBDSLAM_E_ARG;
    const slam::Image &I = which ? M->rgb : M->depth;
    *rows = I.rows; *cols = I.cols;
    if (buf) memcpy(buf, I.data.data(), I.data.size());
    return 0;
}
// ---- plane detection --------------------------------------------------------------------
int slamhost_rg_detect_planes(void *rg, int min_size, int min_normals, int *nb) {
    if (!rg) return RGBDSLAM_E_ARG;
    RG *r = (RG *)rg;
    if (min_size >= 0) r->min_size_plan = min_size;
    if (min_normals >= 0) r->min_normals_plan = min_normals;
    int k = 0;
    FACADE_TRY(k = r->DetectPlanes());
    if (nb) *nb = k;
    return 0;
}
// RG::Finish schedule: enabled, first detection when _count > first_after, dump + re-detection when _count % period == 0
// (period 0 = never); save_dir NULL / "" = nowhere to dump: a due period is then skipped like one that finds a Dummy.txt
// (RG.cpp:177-187) unless slamhost_rg_set_finish_reset_without_dump allows the reset; min_size / min_normals >= 0 override
// MIN_SIZE_PLAN / 1000 (small images)
// optional bilateral stage of OffLineFrame::LoadFrame (Frame.cpp:50-54, commented out in the reference): default off
// 16-bit label images through the tracker (models of more than 255 planes); before the first frame
int slamhost_rg_set_label16(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SetLabel16(on != 0));
    return 0;
}
int slamhost_rg_set_plane_equations(void *rg, const float *equations, int nb) {
    if (!rg || !equations) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SetPlaneEquations(equations, nb));
    return 0;
}
int slamhost_rg_seed_labels_tiled(void *rg, float **ParamTab_dev, int nb, int N_prim, int M_prim) {
    if (!rg || !ParamTab_dev) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SeedLabelsTiled(ParamTab_dev, nb, N_prim, M_prim));
    return 0;
}
int slamhost_rg_export_labels(void *rg, void *dst_dev, size_t off_label) {
    if (!rg) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->ExportLabels(dst_dev, off_label));
    return 0;
}
// per-stage timers (the reference's TIMING switch): RG::SetStageTiming / RG::StageTimes
int slamhost_rg_set_stage_timing(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->SetStageTiming(on != 0));
    return 0;
}
int slamhost_rg_stage_times(void *rg, double *us4, int *frames) {
    if (!rg || !us4 || !frames) return RGBDSLAM_E_ARG;
    FACADE_TRY(*frames = ((RG *)rg)->StageTimes(us4));
    return 0;
}
int slamhost_rg_set_bilateral(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    ((RG *)rg)->_frame->EnableBilateral(on != 0);
    return 0;
}
int slamhost_rg_set_finish(void *rg, int enabled, int first_after, int period, const char *save_dir, int min_size, int min_normals) {
    if (!rg || first_after < 0 || period < 0) return RGBDSLAM_E_ARG;
    RG *r = (RG *)rg;
    r->finish_enabled = enabled != 0; r->finish_first_after = first_after; r->finish_period = period;
    r->save_dir = save_dir ? save_dir : "";
    if (min_size >= 0) r->min_size_plan = min_size;
    if (min_normals >= 0) r->min_normals_plan = min_normals;
    return 0;
}
// 1: a period that is due with no save_dir still resets the plane set and detects again (nothing is dumped: the fused planes
// are lost -- what a benchmark of the cycle or a comparison with the oracle's schedule wants); default 0
int slamhost_rg_set_finish_reset_without_dump(void *rg, int on) {
    if (!rg) return RGBDSLAM_E_ARG;
    ((RG *)rg)->finish_reset_without_dump = on != 0;
    return 0;
}
// RG::ReorganizeEqua (RG.cpp:934-1006) on the current model frame; *nb = planes kept
int slamhost_rg_reorganize_equa(void *rg, int *nb) {
    if (!rg) return RGBDSLAM_E_ARG;
    RG *r = (RG *)rg;
    int k = 0;
    FACADE_TRY(float pose[16]; slam::pack_pose(r->_poseR, r->_poset, pose); k = r->ReorganizeEqua(pose));
    if (nb) *nb = k;
    return 0;
}
int slamhost_rg_finish_skipped(void *rg) { return rg ? ((RG *)rg)->finish_skipped : -1; }
// [0] detections run by Finish, [1] dumps written, [2] current plane count, [3] _indx
int slamhost_rg_finish_stats(void *rg, int *out4) {
    if (!rg || !out4) return RGBDSLAM_E_ARG;
    RG *r = (RG *)rg;
    out4[0] = r->finish_detections; out4[1] = r->finish_dumps; out4[2] = r->_nbPlan; out4[3] = r->_indx;
    return 0;
}
int slamhost_rg_get_planes(void *rg, float *equations, float *centers, int cap) {
    if (!rg || !equations || !centers) return RGBDSLAM_E_ARG;
    RG *r = (RG *)rg;
    if (r->_nbPlan > cap) return RGBDSLAM_E_LIMIT;
    memcpy(equations, r->_Equations.data(), 40 * (size_t)r->_nbPlan);
    memcpy(centers, r->_Centers.data(), 12 * (size_t)r->_nbPlan);
    return 0;
}
// ---- host-logic test hooks (CPU only: no device work) ---------------------------------------------
void slamhost_test_inverse3(const float *a_cm, float *out_cm) {
    slam::Mat3 m; memcpy(m.a, a_cm, 36);
    slam::Mat3 o = m.inverse(); memcpy(out_cm, o.a, 36);
}
void slamhost_test_mul3(const float *a_cm, const float *b_cm, float *out_cm) {
    slam::Mat3 a, b; memcpy(a.a, a_cm, 36); memcpy(b.a, b_cm, 36);
    slam::Mat3 o = a * b; memcpy(out_cm, o.a, 36);
}
float slamhost_test_det6(const float *A) { return slam::det6(A); }
int slamhost_test_llt_solve6(const float *A, const float *b, float *x) { return slam::llt_solve6(A, b, x) ? 0 : 1; }
void slamhost_test_quaternion2matrix(const double *q, double *m9_rowmajor) {
    double qq[4] = {q[0], q[1], q[2], q[3]}, m[3][3];
    slam::quaternion2matrix(qq, m);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m9_rowmajor[3 * r + c] = m[r][c];
}
void slamhost_test_pack_pose(const float *R_cm, const float *t, float *pose16) {
    slam::Mat3 R; memcpy(R.a, R_cm, 36);
    slam::Vec3 v{{t[0], t[1], t[2]}};
    slam::pack_pose(R, v, pose16);
}

int slamhost_copy_to_host(void *dst_host, const void *src_dev, size_t bytes) {
    if (!dst_host || !src_dev) return RGBDSLAM_E_ARG;
    hipError_t e = hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : -(int)e;
}
int slamhost_rg_frames_done(void *rg) { return rg ? ((RG *)rg)->frames_done : -1; }
int slamhost_rg_export_frame(void *rg, void *dst_dev, size_t off_vmap, size_t off_rgbd, size_t off_mask, size_t off_hist) {
    if (!rg || !dst_dev) return RGBDSLAM_E_ARG;
    FACADE_TRY(((RG *)rg)->ExportFrame(dst_dev, off_vmap, off_rgbd, off_mask, off_hist));
    return 0;
}
int slamhost_rg_last_iterations(void *rg) { return rg ? ((RG *)rg)->last_iterations : -1; }
int slamhost_rg_last_level_iterations(void *rg) { return rg ? ((RG *)rg)->last_level_iterations : -1; }
void *slamhost_rg_ctx(void *rg) { return rg ? (void *)((RG *)rg)->ctx() : nullptr; }
int slamhost_rg_count(void *rg) { return rg ? ((RG *)rg)->_count : -1; }
// Device address + size of a named buffer: VMap NMap RGBD Mask Label frame.VMap frame.NMap
// frame.RGB Bump<k> RGBb<k> Maskb<k>
int slamhost_rg_buffer(void *rgp, const char *name, void **ptr, size_t *bytes) {
    if (!rgp || !name || !ptr || !bytes) return RGBDSLAM_E_ARG;
    RG *rg = (RG *)rgp;
    slam::PredictedFrame *pf = rg->_predicted_frame;
    size_t P = (size_t)pf->_n * pf->_m, C = (size_t)rg->_graph->_SizeVBUFF[0] * rg->_graph->_SizeVBUFF[1];
    *ptr = nullptr; *bytes = 0;
    std::string s(name);
    if (s == "VMap") { *ptr = pf->getVMap(); *bytes = 12 * P; }
    else if (s == "NMap") { *ptr = pf->getNMap(); *bytes = 12 * P; }
    else if (s == "RGBD") { *ptr = pf->getRGB(); *bytes = 16 * P; }
    else if (s == "Mask") { *ptr = pf->_Mask_dev; *bytes = P; }
    else if (s == "Label") { *ptr = pf->_Label_dev; *bytes = P * (size_t)pf->label_width(); }
    else if (s == "history") { *ptr = rg->history_dev(); *bytes = 64 * (size_t)RG::HISTORY_CAP; }
    else if (s == "frame.VMap") { *ptr = rg->_frame->getVMap(); *bytes = 12 * P; }
    else if (s == "frame.NMap") { *ptr = rg->_frame->getNMap(); *bytes = 12 * P; }
    else if (s == "frame.RGB") { *ptr = rg->_frame->getRGB(); *bytes = 12 * P; }
    else {
        auto plane = [&](size_t off) -> slam::Primitive * {
            int k = atoi(name + off);
            return (k >= 0 && (size_t)k < rg->_graph->_nodes.size()) ? rg->_graph->_nodes[k] : nullptr;
        };
        if (s.rfind("Bump", 0) == 0) { if (auto *p = plane(4)) { *ptr = p->_Bump_dev; *bytes = 6 * C; } }
        else if (s.rfind("RGBb", 0) == 0) { if (auto *p = plane(4)) { *ptr = p->_RGB_dev; *bytes = 3 * C; } }
        else if (s.rfind("Maskb", 0) == 0) { if (auto *p = plane(5)) { *ptr = p->_Mask_dev; *bytes = C; } }
    }
    return *ptr ? 0 : RGBDSLAM_E_ARG;
}

}  // extern "C"
