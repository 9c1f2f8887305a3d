// context.hip -- per-GPU context replacing the reference's file-scope device globals
// (calib_dev/pose_dev D1/Kernel.cu:6-7, buf_dev/outbuf_dev/buf D1/ICP.cu:5-7).
#include "common.hpp"
#include <new>

extern "C" const char *rgbdslam_version(void) { return "rgbdslam-hip 0.1 (gfx950)"; }

extern "C" int rgbdslam_create(rgbdslam_ctx **out, int device, void *stream) {
    if (!out) return RGBDSLAM_E_ARG;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(device));
    rgbdslam_ctx *c = new (std::nothrow) rgbdslam_ctx();
    if (!c) return -(int)hipErrorOutOfMemory;
    c->device = device;
    c->stream = (hipStream_t)stream;
    c->have_calib = false;
    for (int k = 0; k < 16; k++) c->pose.v[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    c->buf_dev = nullptr; c->buf_blocks = 0; c->outbuf_dev = nullptr; c->out_host = nullptr; c->icp_state = nullptr;
    *out = c;
    return 0;
}

extern "C" int rgbdslam_destroy(rgbdslam_ctx *ctx) {
    if (!ctx) return RGBDSLAM_E_ARG;
    rgbdslam_free_buffers(ctx);
    delete ctx;
    return 0;
}

extern "C" int rgbdslam_set_stream(rgbdslam_ctx *ctx, void *stream) {
    if (!ctx) return RGBDSLAM_E_ARG;
    ctx->stream = (hipStream_t)stream;
    return 0;
}

extern "C" int rgbdslam_synchronize(rgbdslam_ctx *ctx) {
    if (!ctx) return RGBDSLAM_E_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int rgbdslam_set_calibration_matrix(rgbdslam_ctx *ctx, const float *calib11) {
    if (!ctx || !calib11) return RGBDSLAM_E_ARG;
    for (int k = 0; k < 11; k++) ctx->calib.v[k] = calib11[k];
    ctx->have_calib = true;
    return 0;
}

extern "C" int rgbdslam_set_pose_matrix(rgbdslam_ctx *ctx, const float *pose16) {
    if (!ctx || !pose16) return RGBDSLAM_E_ARG;
    for (int k = 0; k < 16; k++) ctx->pose.v[k] = pose16[k];
    return 0;
}
