// context.hip -- per-GPU context replacing the reference's file-scope device globals
// (calib_dev/pose_dev D1/Kernel.cu:6-7, buf_dev/outbuf_dev/buf D1/ICP.cu:5-7).
#include "common.hpp"
#include <new>
#include <chrono>
#include <string.h>
#include <math.h>
#include <stdlib.h>

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
    c->buf_dev = nullptr; c->icp_timing_dev = nullptr; c->outbuf_dev = nullptr; c->out_host = nullptr; c->icp_state = nullptr; c->icp_epoch_dev = nullptr; c->icp_block = nullptr; c->icp_epoch_base = 0; c->icp_loop = 0; c->icp_line_check = 0; c->icp_timeout_ticks = 5000000;
    { const char *e = getenv("RGBDSLAM_SEG3"); c->seg3_fused = !(e && e[0] == '0'); }
    c->aux_write = nullptr; c->track_aux = nullptr;
    { const char *e = getenv("RGBDSLAM_SYNC_SPIN_US"); c->sync_spin_us = e ? atoi(e) : 5000; if (c->sync_spin_us < 0) c->sync_spin_us = 0; }
    for (int k = 0; k < RGBDSLAM_COMPAT_MAX_PLANES; k++) c->claim_slab[k] = nullptr;
    c->claim_tab_dev = nullptr; c->claim_cells = 0; c->claim_stamp = 0; c->label_swap = nullptr; c->label_swap_px = 0; c->compat_counter = nullptr;
    for (int k = 0; k < 8; k++) c->xchg_peer[k] = nullptr;
    c->xchg_world = 0; c->xchg_rank = 0; c->xchg_seq = 0; c->xchg_delay_ticks = 0; c->xchg_tab_dev = nullptr;
    c->bil_lut = nullptr; c->bil_sigma_color = -1.0;
    c->label_width = 1;
    c->prof_name = nullptr; c->prof_ev = nullptr; c->prof_cap = 0; c->prof_n = 0; c->prof_launches = 0; c->prof_in_group = false;
    // this device's copy of the static z-buffer of rgbdslam_predict_raster_noalloc (predict.hip), resolved here so that
    // the entry point itself never has to
    c->noalloc_zbuf = nullptr; c->pd_scratch = nullptr; c->pd_cap = 0; c->pd_stamp = 0;
    int rc = raster_noalloc_setup(device, &c->noalloc_zbuf);
    if (rc) { delete c; return rc; }
    *out = c;
    return 0;
}

extern "C" int rgbdslam_destroy(rgbdslam_ctx *ctx) {
    if (!ctx) return RGBDSLAM_E_ARG;
    rgbdslam_free_buffers(ctx);
    compat_release(ctx);
    if (ctx->bil_lut) { (void)hipFree(ctx->bil_lut); ctx->bil_lut = nullptr; }
    if (ctx->pd_scratch) { (void)hipFree(ctx->pd_scratch); ctx->pd_scratch = nullptr; }
    rgbdslam_exchange_destroy(ctx);
    rgbdslam_profile_kernel(ctx, nullptr);
    for (int k = 0; k < 2 * ctx->prof_cap; k++) (void)hipEventDestroy(ctx->prof_ev[k]);
    free(ctx->prof_ev);
    delete ctx;
    return 0;
}

extern "C" int rgbdslam_set_stream(rgbdslam_ctx *ctx, void *stream) {
    if (!ctx) return RGBDSLAM_E_ARG;
    ctx->stream = (hipStream_t)stream;
    return 0;
}

// Element width of every label image this context's entry points read or write: 1 = u8 (the reference's, Frame.h:197),
// 2 = u16 (models of more than 255 planes; the pointers keep their uint8_t type in the signatures).
extern "C" int rgbdslam_set_label_width(rgbdslam_ctx *ctx, int bytes) {
    if (!ctx || (bytes != 1 && bytes != 2)) return RGBDSLAM_E_ARG;
    ctx->label_width = bytes;
    return 0;
}
extern "C" int rgbdslam_get_label_width(const rgbdslam_ctx *ctx) { return ctx ? ctx->label_width : RGBDSLAM_E_ARG; }

// ---- one POD configuration (the reference's #defines + RG constructor arguments) ---------------------------------
extern "C" void rgbdslam_config_defaults(rgbdslam_config *cfg) {
    if (!cfg) return;
    const float tum[11] = {525.0f, 525.0f, 319.5f, 239.5f, 0.2624f, -0.9531f, -0.0054f, 0.0026f, 1.1633f, 1.035f, 5000.0f};   // GP-demo.cpp:404
    memcpy(cfg->calib, tum, sizeof(tum));
    cfg->icp.lvl = 3;                                                        // GP-demo.cpp:409
    cfg->icp.max_iter[0] = 3; cfg->icp.max_iter[1] = 2; cfg->icp.max_iter[2] = 1;   // RG.h:124-126
    cfg->icp.distThres = 0.05f; cfg->icp.angleThres = 0.3420201433256687f;   // sin 20 deg, GP-demo.cpp:409
    cfg->icp.thres_cvg_alpha = 1e-6f; cfg->icp.thres_cvg_tr = 1e-6f;         // RG.h:111
    cfg->icp_loop = 0; cfg->icp_line_check = 0; cfg->label_width = 1;
}
extern "C" int rgbdslam_configure(rgbdslam_ctx *ctx, const rgbdslam_config *cfg) {
    if (!ctx || !cfg) return RGBDSLAM_E_ARG;
    if (cfg->label_width != 1 && cfg->label_width != 2) return RGBDSLAM_E_ARG;
    int rc = rgbdslam_set_calibration_matrix(ctx, cfg->calib);
    if (rc) return rc;
    if ((rc = rgbdslam_set_label_width(ctx, cfg->label_width))) return rc;
    if ((rc = rgbdslam_set_icp_line_check(ctx, cfg->icp_line_check))) return rc;
    if (cfg->icp_loop != ctx->icp_loop && (rc = rgbdslam_set_icp_loop(ctx, cfg->icp_loop))) return rc;   // needs the ICP buffers (rgbdslam_alloc_buffers)
    return 0;
}
extern "C" int rgbdslam_get_config(const rgbdslam_ctx *ctx, rgbdslam_config *cfg) {
    if (!ctx || !cfg) return RGBDSLAM_E_ARG;
    for (int k = 0; k < 11; k++) cfg->calib[k] = ctx->calib.v[k];
    cfg->icp_loop = ctx->icp_loop; cfg->icp_line_check = ctx->icp_line_check; cfg->label_width = ctx->label_width;
    return 0;
}

extern "C" int rgbdslam_get_stream(rgbdslam_ctx *ctx, void **stream) {
    if (!ctx || !stream) return RGBDSLAM_E_ARG;
    *stream = (void *)ctx->stream;
    return 0;
}

// A blocked hipStreamSynchronize is woken by an interrupt, tens of microseconds after the stream has drained -- a quarter of a
// 112-us frame for a caller that waits once per frame, and ~1 % of a 20-frame sequence.  The wait therefore polls the stream
// first (hipStreamQuery, ~1 us a call) for at most sync_spin_us (default 5 ms, RGBDSLAM_SYNC_SPIN_US / rgbdslam_set_sync_spin;
// 0 = block at once) and only then blocks: short waits end within a microsecond or two, long ones do not burn a core.
extern "C" int rgbdslam_synchronize(rgbdslam_ctx *ctx) {
    if (!ctx) return RGBDSLAM_E_ARG;
    if (ctx->sync_spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) return 0;
            (void)hipGetLastError();                                   // "not ready" is not an error to leave behind
            if (e != hipErrorNotReady) return -(int)e;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= ctx->sync_spin_us) break;
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// The frame's auxiliary ICP records (16 bytes per pixel: the scaled gradients, the intensity and the depth weight the ICP rows
// take from the current frame, ICP.cu:167-173): aux_write -> the NEXT launch that preprocesses a frame (rgbdslam_frame_preprocess
// and the launches it rides on) also leaves the records of that frame there; aux_read -> the NEXT rgbdslam_track_frame reads
// them instead of forming the four values per sample and iteration.  Either may be NULL; each is consumed by that one call.
extern "C" int rgbdslam_set_frame_aux(rgbdslam_ctx *ctx, float *aux_write, const float *aux_read) {
    if (!ctx) return RGBDSLAM_E_ARG;
    if ((((uintptr_t)aux_write | (uintptr_t)aux_read) & 15) != 0) return RGBDSLAM_E_ARG;
    ctx->aux_write = aux_write; ctx->track_aux = aux_read;
    return 0;
}

extern "C" int rgbdslam_set_sync_spin(rgbdslam_ctx *ctx, int microseconds) {
    if (!ctx || microseconds < 0) return RGBDSLAM_E_ARG;
    ctx->sync_spin_us = microseconds;
    return 0;
}

// ---- small results to the host without copy commands (no reference counterpart: the reference reads poses back with
// blocking cudaMemcpy, RG.cpp:898-903) ----------------------------------------------------------------------------------
// A hipMemcpyAsync device-to-host on a compute stream is a blit launch of its own with its own completion handling
// (~7 us each in the frame loop's trace); the pose block and the pose history entries a caller reads after a sequence are a
// few hundred bytes, so ONE launch stores all of them into pinned host memory directly.
struct CopySegs { const uint32_t *src[RGBDSLAM_COPY_MAX_SEGMENTS]; uint32_t *dst[RGBDSLAM_COPY_MAX_SEGMENTS]; uint32_t words[RGBDSLAM_COPY_MAX_SEGMENTS]; int n; };
__global__ void k_copy_segments(CopySegs a) {
    const uint32_t stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < a.n; k++) {
        const uint32_t *s = a.src[k];
        uint32_t *d = a.dst[k];
        const uint32_t w = a.words[k];
        if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
            const uint32_t q = w >> 2;
            for (uint32_t i = first; i < q; i += stride) ((uint4 *)d)[i] = ((const uint4 *)s)[i];
            for (uint32_t i = 4 * q + first; i < w; i += stride) d[i] = s[i];
        } else {
            for (uint32_t i = first; i < w; i += stride) d[i] = s[i];
        }
    }
}

extern "C" int rgbdslam_copy_segments(rgbdslam_ctx *ctx, const void *const *src_dev, void *const *dst, const size_t *bytes, int count) {
    if (!ctx || count < 0 || count > RGBDSLAM_COPY_MAX_SEGMENTS || (count && (!src_dev || !dst || !bytes))) return RGBDSLAM_E_ARG;
    if (count == 0) return 0;
    CopySegs a = {};
    size_t most = 0;
    for (int k = 0; k < count; k++) {
        if (!src_dev[k] || !dst[k] || (bytes[k] & 3) || (((uintptr_t)src_dev[k] | (uintptr_t)dst[k]) & 3) || bytes[k] > 0xffffffffull)
            return RGBDSLAM_E_ARG;
        a.src[k] = (const uint32_t *)src_dev[k]; a.dst[k] = (uint32_t *)dst[k]; a.words[k] = (uint32_t)(bytes[k] / 4);
        if (bytes[k] > most) most = bytes[k];
    }
    a.n = count;
    HIP_TRY(hipSetDevice(ctx->device));
    const int blocks = (int)((most / 16 + 255) / 256);
    KLAUNCH(ctx, k_copy_segments, dim3(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks)), dim3(256), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_set_calibration_matrix(rgbdslam_ctx *ctx, const float *calib11) {
    if (!ctx || !calib11) return RGBDSLAM_E_ARG;
    CalibV &c = ctx->calib;
    for (int k = 0; k < 11; k++) c.v[k] = calib11[k];
    // host IEEE divisions: the same correctly rounded values the kernels' divisions give
    c.inv_fx = 1.0f / c.v[0]; c.inv_fy = 1.0f / c.v[1]; c.s = c.v[9] / c.v[10];
    auto focal_ok = [](float f) {                  // finite, 2^-20 <= |f| <= 2^40, significand not all ones (Markstein's condition)
        uint32_t b; memcpy(&b, &f, 4);
        return fabsf(f) >= 0x1p-20f && fabsf(f) <= 0x1p40f && (b & 0x7FFFFFu) != 0x7FFFFFu;
    };
    auto centre_ok = [](float x) { return x == 0.0f || (fabsf(x) >= 0x1p-60f && fabsf(x) <= 0x1p20f); };
    c.fast = focal_ok(c.v[0]) && focal_ok(c.v[1]) && centre_ok(c.v[2]) && centre_ok(c.v[3]) &&
             (c.s == 0.0f || (fabsf(c.s) >= 0x1p-40f && fabsf(c.s) <= 0x1p40f));
    // Markstein's theorem wants q = RN(a y) within an ulp of a / b, which holds for most but not all (a, b).  The numerators of
    // these two quotients are (k - centre) for a pixel index k < 65536 -- a finite set once the calibration is known -- so the
    // claim is simply checked for every one of them here, with the FMA sequence the kernels use against the division; a
    // calibration with a single differing quotient keeps the divisions.  (~0.3 ms, once per calibration.)
    for (int axis = 0; axis < 2 && c.fast; axis++) {
        const float f = c.v[axis], ctr = c.v[2 + axis], y = axis ? c.inv_fy : c.inv_fx;
        for (int k = 0; k < 65536; k++) {
            const float a = (float)k - ctr, q = a * y;
            if (fmaf(fmaf(-f, q, a), y, q) != a / f) { c.fast = 0; break; }
        }
    }
    ctx->have_calib = true;
    return 0;
}

extern "C" int rgbdslam_set_pose_matrix(rgbdslam_ctx *ctx, const float *pose16) {
    if (!ctx || !pose16) return RGBDSLAM_E_ARG;
    for (int k = 0; k < 16; k++) ctx->pose.v[k] = pose16[k];
    return 0;
}

// ---- instrumentation -----------------------------------------------------------------------------
static bool prof_matches(const rgbdslam_ctx *ctx, const char *kname) {
    // exact match, or -- for a requested name without template arguments -- every instantiation of that kernel
    if (strcmp(ctx->prof_name, kname) == 0) return true;
    size_t len = strlen(ctx->prof_name);
    return !strchr(ctx->prof_name, '<') && strncmp(ctx->prof_name, kname, len) == 0 && kname[len] == '<';
}
static bool prof_record_first(rgbdslam_ctx *ctx) {
    if (ctx->prof_n == ctx->prof_cap) {
        int cap = ctx->prof_cap ? 2 * ctx->prof_cap : 1024;
        hipEvent_t *ev = (hipEvent_t *)realloc(ctx->prof_ev, sizeof(hipEvent_t) * 2 * cap);
        if (!ev) return false;
        ctx->prof_ev = ev;
        for (int k = 2 * ctx->prof_cap; k < 2 * cap; k++)
            if (hipEventCreate(&ev[k]) != hipSuccess) return false;
        ctx->prof_cap = cap;
    }
    return hipEventRecord(ctx->prof_ev[2 * ctx->prof_n], ctx->stream) == hipSuccess;
}
bool prof_begin(rgbdslam_ctx *ctx, const char *kname) {
    if (!prof_matches(ctx, kname)) return false;
    if (ctx->prof_in_group) { ctx->prof_launches++; return false; }
    if (!prof_record_first(ctx)) return false;
    ctx->prof_launches++;
    return true;
}
void prof_end(rgbdslam_ctx *ctx) {
    (void)hipEventRecord(ctx->prof_ev[2 * ctx->prof_n + 1], ctx->stream);
    ctx->prof_n++;
}
bool prof_group_begin(rgbdslam_ctx *ctx, const char *kname) {
    if (!ctx->prof_name || ctx->prof_in_group || !prof_matches(ctx, kname) || !prof_record_first(ctx)) return false;
    ctx->prof_in_group = true;
    return true;
}
void prof_group_end(rgbdslam_ctx *ctx) {
    ctx->prof_in_group = false;
    prof_end(ctx);
}

extern "C" int rgbdslam_profile_kernel(rgbdslam_ctx *ctx, const char *kernel_name) {
    if (!ctx) return RGBDSLAM_E_ARG;
    free((void *)ctx->prof_name);
    ctx->prof_name = kernel_name ? strdup(kernel_name) : nullptr;
    ctx->prof_n = 0;
    ctx->prof_launches = 0;
    ctx->prof_in_group = false;
    return 0;
}

extern "C" int rgbdslam_profile_read(rgbdslam_ctx *ctx, double *total_ms, int *launches) {
    if (!ctx || !total_ms || !launches) return RGBDSLAM_E_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (int k = 0; k < ctx->prof_n; k++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_ev[2 * k], ctx->prof_ev[2 * k + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = ctx->prof_launches;
    ctx->prof_n = 0;
    ctx->prof_launches = 0;
    return 0;
}

// What an event pair measures with NOTHING between its two records, on the context's stream: the marker's own
// processing time, which every bracketed launch carries on top of the kernel's duration.
extern "C" int rgbdslam_profile_event_overhead(rgbdslam_ctx *ctx, int pairs, double *mean_ms) {
    if (!ctx || !mean_ms || pairs <= 0 || pairs > 4096) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double tot = 0.0;
    for (int k = 0; k < pairs; k++) {
        HIP_TRY(hipEventRecord(e0, ctx->stream));
        HIP_TRY(hipEventRecord(e1, ctx->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mean_ms = tot / pairs;
    return 0;
}

// ---- what this card's HBM delivers to a plain streaming kernel (the measured figure the per-kernel
// fractions are ALSO quoted against, beside the nominal 8 TB/s).  One 16-byte word per lane, workgroups in address
// order: the densest possible moving window of open DRAM pages.  (The same bytes through a grid-stride loop of 4096
// workgroups reach only 3.9 TB/s when filling and 4.8 TB/s when copying, against 6.9 and 6.1 TB/s this way.)
typedef float probe4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 copy (1 read + 1 write per byte), 1 fill (write only), 2 read only
__global__ __launch_bounds__(256) void k_hbm_stream(probe4 *__restrict__ dst, const probe4 *__restrict__ src, size_t n16) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n16) return;
    probe4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (MODE == 0) dst[k] = src[k];
    if (MODE == 1) dst[k] = acc;
    if (MODE == 2) {
        acc = src[k];
        // keeps the load alive; the probe buffers hold finite values, so nothing is written
        if (acc.x + acc.y + acc.z + acc.w == __builtin_inff()) dst[0] = acc;
    }
}

extern "C" int rgbdslam_profile_hbm_stream(rgbdslam_ctx *ctx, void *dst, const void *src, size_t bytes, int mode, int reps,
                                           double *gbytes_per_s) {
    if (!ctx || !dst || !src || !gbytes_per_s || bytes < 16 || mode < 0 || mode > 2 || reps <= 0 || reps > 1000) return RGBDSLAM_E_ARG;
    if (((uintptr_t)dst | (uintptr_t)src) & 15) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n16 = bytes / 16;
    if (n16 > 0xFFFFFF00ull) return RGBDSLAM_E_LIMIT;          // one thread per word: a grid holds fewer than 2^32 threads (64 GiB)
    const unsigned grid = (unsigned)((n16 + 255) / 256);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int r = -1; r < reps; r++) {                           // one untimed launch first
        if (r == 0) HIP_TRY(hipEventRecord(e0, ctx->stream));
        if (mode == 0) hipLaunchKernelGGL(k_hbm_stream<0>, dim3(grid), dim3(256), 0, ctx->stream, (probe4 *)dst, (const probe4 *)src, n16);
        if (mode == 1) hipLaunchKernelGGL(k_hbm_stream<1>, dim3(grid), dim3(256), 0, ctx->stream, (probe4 *)dst, (const probe4 *)src, n16);
        if (mode == 2) hipLaunchKernelGGL(k_hbm_stream<2>, dim3(grid), dim3(256), 0, ctx->stream, (probe4 *)dst, (const probe4 *)src, n16);
    }
    HIP_TRY(hipEventRecord(e1, ctx->stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double moved = (double)(n16 * 16) * (mode == 0 ? 2.0 : 1.0) * reps;
    *gbytes_per_s = moved / (ms * 1e-3) / 1e9;
    return 0;
}

// debug: 7 x int64 device words receiving phase time stamps of the LAST k_icp_iter launch
extern "C" int rgbdslam_debug_icp_timing(rgbdslam_ctx *ctx, long long *timing7_dev) {
    if (!ctx) return RGBDSLAM_E_ARG;
    ctx->icp_timing_dev = timing7_dev;
    return 0;
}
