// predict.hip -- stage 5: model prediction.  Rigid warp of the model depth image, quad trimming,
// the compute rasteriser that replaces the reference's OpenGL GL_QUADS pass, read-back to vertex
// maps, mask/label re-projection and the plane-label growing passes.
#include "device_fns.hpp"
#include <mutex>

// ---- D1/Kernel.cu:345-379 TransformVertexMapKernel --------------------------------------------
__global__ void k_transform_vertex_map(PoseV P, float *__restrict__ VMap, float *__restrict__ RGBD, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    F3 v = ld3(VMap, idx);
    if (is_zero3(v)) { st3(VMap, idx, 0.0f, 0.0f, 0.0f); RGBD[4 * idx + 3] = 0.0f; return; }
    F3 r = xf_point(P.v, v);
    st3(VMap, idx, r.x, r.y, r.z);
    RGBD[4 * idx + 3] = depth_attr(r.z);
}

// ---- D1/Kernel.cu:381-438 QuadTrimFrameKernel ---------------------------------------------------
__global__ void k_quad_trim_frame(uint32_t *__restrict__ index, const float *__restrict__ VMap, int n, int m, float thresh) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i > n - 2 || j > m - 2) return;
    size_t q = (size_t)i * (m - 1) + j;
    uint32_t v0 = i * m + j, v1 = v0 + 1, v2 = v0 + m + 1, v3 = v0 + m;
    bool cut = quad_trimmed(ld3(VMap, v0), ld3(VMap, v1), ld3(VMap, v2), ld3(VMap, v3), thresh);
    uint4 o;
    if (cut) { o.x = o.y = o.z = o.w = 0u; } else { o.x = v0; o.y = v1; o.z = v2; o.w = v3; }
    *(uint4 *)(index + 4 * q) = o;
}

// ---- compute rasteriser (device_fns.hpp) ------------------------------------------
// pass 1: one thread per quad
__global__ void k_raster_draw(unsigned long long *__restrict__ zbuf, const float *__restrict__ VMap,
                              const uint32_t *__restrict__ index, float thresh, CalibV c, float znear, float zfar,
                              int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i > n - 2 || j > m - 2) return;
    uint32_t q = (uint32_t)(i * (m - 1) + j), id[4];
    quad_vertex_ids(q, index, m, id);
    F3 v0 = ld3(VMap, id[0]), v1 = ld3(VMap, id[1]), v2 = ld3(VMap, id[2]), v3 = ld3(VMap, id[3]);
    if (!index && quad_trimmed(v0, v1, v2, v3, thresh)) return;
    // triangles (0,1,2) and (0,2,3): the four vertices are projected once
    const RasVert p0 = ras_project(v0, c), p1 = ras_project(v1, c), p2 = ras_project(v2, c), p3 = ras_project(v3, c);
    RasTri T;
    if (ras_setup_projected(p0, p1, p2, T)) ras_draw_tri(zbuf, T, 2 * q, znear, zfar, n, m);
    if (ras_setup_projected(p0, p2, p3, T)) ras_draw_tri(zbuf, T, 2 * q + 1, znear, zfar, n, m);
}

// pass 2: one thread per pixel; re-derives the winning triangle's weights, writes the texel and
// resets the z-buffer word so the next frame needs no clear.
__global__ void k_raster_resolve(float *__restrict__ tex, unsigned long long *__restrict__ zbuf,
                                 const float *__restrict__ VMap, const float *__restrict__ RGBD,
                                 const uint32_t *__restrict__ index, CalibV c, int n, int m) {
    int px = blockIdx.x * TILE_W + threadIdx.x, py = blockIdx.y * TILE_H + threadIdx.y;
    if (py >= n || px >= m) return;
    size_t p = (size_t)py * m + px;
    unsigned long long key = zbuf[p];
    float4 o; o.x = o.y = o.z = o.w = 0.0f;
    if (key != ZBUF_EMPTY) {
        zbuf[p] = ZBUF_EMPTY;
        uint32_t tri_id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull), id[4], vid[3];
        quad_vertex_ids(tri_id >> 1, index, m, id);
        if (tri_id & 1) { vid[0] = id[0]; vid[1] = id[2]; vid[2] = id[3]; } else { vid[0] = id[0]; vid[1] = id[1]; vid[2] = id[2]; }
        F3 vtx[3] = {ld3(VMap, vid[0]), ld3(VMap, vid[1]), ld3(VMap, vid[2])};
        float4 att[3] = {*(const float4 *)(RGBD + 4 * (size_t)vid[0]), *(const float4 *)(RGBD + 4 * (size_t)vid[1]),
                         *(const float4 *)(RGBD + 4 * (size_t)vid[2])};
        o = ras_shade(vtx, att, c, px, py);
    }
    *(float4 *)(tex + 4 * p) = o;
}

// ---- D1/Kernel.cu:309-343 ReadKernel (+ the cudaMemset of RGBD, Kernel.cu:900) -------------------
__global__ void k_read_frame(float *__restrict__ VMap, float *__restrict__ RGBD, const float *__restrict__ tex,
                             CalibV c, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    float4 t = *(const float4 *)(tex + 4 * idx);
    if (t.w == 0.0f) {
        st3(VMap, idx, 0.0f, 0.0f, 0.0f);
        float4 z; z.x = z.y = z.z = z.w = 0.0f;
        *(float4 *)(RGBD + 4 * idx) = z;
        return;
    }
    F3 v = vertex_from_texel(t.w, i, j, c);
    st3(VMap, idx, v.x, v.y, v.z);
    *(float4 *)(RGBD + 4 * idx) = t;
}

// ---- D1/Kernel.cu:634-667 ProjectMaskKernel -------------------------------------------------------
// T = uint8_t (Mask, Label as in the reference) or uint16_t (16-bit label images)
template <typename T>
__global__ void k_project_mask(PoseV P, CalibV c, const T *__restrict__ mask, T *__restrict__ mask_swap,
                               const float *__restrict__ VMap, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    long src = project_source(P.v, c, ld3(VMap, idx), n, m);
    mask_swap[idx] = (src < 0) ? (T)0 : mask[src];
}

// ---- D1/RG.cu:252-320 SegmentKernel, one double-buffered pass --------------------------------------
template <typename LT>
__global__ void k_segment_pass(const LT *__restrict__ in, LT *__restrict__ out, const float *__restrict__ VMap,
                               const float *__restrict__ NMap, PoseV P, const float *__restrict__ Eq, float epsilon,
                               float alpha, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    LT mine = in[idx];
    if (mine > 0) { out[idx] = mine; return; }
    out[idx] = segment_pixel(in, VMap, ld3(NMap, idx), P.v, Eq, epsilon, alpha, i, j, n, m);
}

// plane-membership labelling (test-harness stand-in for plane detection)
template <typename LT>
__global__ void k_seed_labels(LT *__restrict__ Label, const float *__restrict__ VMap, const float *__restrict__ NMap,
                              PoseV P, const float *__restrict__ Eq, int nb, float epsilon, float alpha, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    LT res = 0;
    F3 nl = ld3(NMap, idx);
    if (!is_zero3(nl)) {
        F3 p = xf_point(P.v, ld3(VMap, idx));
        F3 q = xf_dir(P.v, nl);
        for (int k = 0; k < nb && k < (sizeof(LT) == 2 ? 65535 : 255); k++) {
            const float *E = Eq + 10 * k;
            float ed = ((p.x * E[0] + p.y * E[1]) + p.z * E[2]) - E[9];
            float ea = (q.x * E[0] + q.y * E[1]) + q.z * E[2];
            if (__builtin_fabsf(ed) > epsilon || ea < alpha) continue;
            res = (LT)(k + 1);
            break;
        }
    }
    Label[idx] = res;
}

// =================================================================================================
// C ABI
// =================================================================================================
static PoseV pose_from_host(const float *p) { PoseV r; for (int k = 0; k < 16; k++) r.v[k] = p[k]; return r; }

extern "C" int rgbdslam_transform_vertex_map(rgbdslam_ctx *ctx, float *VMap, float *RGBD, int n, int m) {
    if (!ctx || !VMap || !RGBD || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_transform_vertex_map, grid2d(n, m), block2d(), ctx->pose, VMap, RGBD, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_quad_trim_frame(rgbdslam_ctx *ctx, const float *VMap, uint32_t *indices, int n, int m, float thresh) {
    if (!ctx || !VMap || !indices || n < 2 || m < 2) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_quad_trim_frame, grid2d(n - 1, m - 1), block2d(), indices, VMap, n, m, thresh);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_raster_clear(rgbdslam_ctx *ctx, uint64_t *zbuf, int n, int m) {
    if (!ctx || !zbuf || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    HIP_TRY(hipMemsetAsync(zbuf, 0xFF, sizeof(uint64_t) * (size_t)n * m, ctx->stream));
    return 0;
}
extern "C" int rgbdslam_predict_raster(rgbdslam_ctx *ctx, float *tex, uint64_t *zbuf, const float *VMap, const float *RGBD,
                                       const uint32_t *indices, float thresh, float znear, float zfar, int n, int m) {
    if (!ctx || !tex || !zbuf || !VMap || !RGBD || n < 2 || m < 2) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x3FFFFFFFu) return RGBDSLAM_E_LIMIT;   // triangle ids are 32 bit
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_raster_draw, grid2d(n - 1, m - 1), block2d(), (unsigned long long *)zbuf, VMap,
                       indices, thresh, ctx->calib, znear, zfar, n, m);
    KLAUNCH(ctx, k_raster_resolve, grid2d(n, m), block2d(), tex, (unsigned long long *)zbuf, VMap,
                       RGBD, indices, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
// ---- the one entry point that allocates nothing (no counterpart in the reference) ---------------------------------
// rgbdslam_predict_raster with the z-buffer taken from a file-scope device array instead of a caller workspace, for
// callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc).  Every GPU the
// library runs on holds its own copy of the array (16.6 MB) at its own address until the process ends; `scratch` is
// only the host-side stand-in of that symbol, so host code never passes it anywhere: the device address is resolved
// once per device with hipGetSymbolAddress (below) and only that pointer is used.
#define RASTER_NOALLOC_MAX_PIXELS (1920 * 1080)
#define RASTER_NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned long long scratch[RASTER_NOALLOC_MAX_PIXELS];
static std::mutex g_scratch_mutex;
static unsigned long long *g_scratch_ptr[RASTER_NOALLOC_MAX_DEVICES];   // per device: that GPU's copy of `scratch`

// Called from rgbdslam_create (the library's init call) with `device` current; idempotent.  *out = that device's copy of the
// array; the context keeps it (ctx->noalloc_zbuf), so the entry point never reads the table.
int raster_noalloc_setup(int device, unsigned long long **out) {
    if (device < 0 || device >= RASTER_NOALLOC_MAX_DEVICES || !out) return RGBDSLAM_E_LIMIT;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        HIP_TRY(hipSetDevice(device));
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return -(int)hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (unsigned long long *)ptr;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

extern "C" int rgbdslam_predict_raster_noalloc(rgbdslam_ctx *ctx, float *tex, const float *VMap, const float *RGBD,
                                               const uint32_t *indices, float thresh, float znear, float zfar, int n, int m) {
    if (!ctx || !tex || !VMap || !RGBD || n < 2 || m < 2) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > (size_t)RASTER_NOALLOC_MAX_PIXELS) return RGBDSLAM_E_LIMIT;   // beyond the static z-buffer
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    unsigned long long *zbuf = ctx->noalloc_zbuf;            // this device's copy of `scratch`, resolved by rgbdslam_create
    if (!zbuf) return RGBDSLAM_E_STATE;
    // the array lives on ctx->device: make that device current whatever the caller's is (the NULL stream and the kernel
    // launches follow the current device).  hipSetDevice allocates nothing and is legal under stream capture.
    HIP_TRY(hipSetDevice(ctx->device));
    // the array keeps the last call's contents (an aborted call could leave claimed words): the words this call
    // uses start at "empty" whatever ran before
    HIP_TRY(hipMemsetAsync(zbuf, 0xFF, sizeof(unsigned long long) * (size_t)n * m, ctx->stream));
    KLAUNCH(ctx, k_raster_draw, grid2d(n - 1, m - 1), block2d(), zbuf, VMap, indices, thresh, ctx->calib, znear, zfar, n, m);
    KLAUNCH(ctx, k_raster_resolve, grid2d(n, m), block2d(), tex, zbuf, VMap, RGBD, indices, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_read_frame(rgbdslam_ctx *ctx, float *VMap, float *RGBD, const float *tex, int n, int m) {
    if (!ctx || !VMap || !RGBD || !tex || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_read_frame, grid2d(n, m), block2d(), VMap, RGBD, tex, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_project_mask(rgbdslam_ctx *ctx, const uint8_t *mask, uint8_t *mask_swap, const float *VMap, int n, int m) {
    if (!ctx || !mask || !mask_swap || mask == mask_swap || !VMap || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH_AS(ctx, "k_project_mask", (k_project_mask<uint8_t>), grid2d(n, m), block2d(), ctx->pose, ctx->calib, mask, mask_swap, VMap, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
// ProjectMask_cu on the LABEL image (Frame.cpp:389): the same gather with the context's label width
extern "C" int rgbdslam_project_label(rgbdslam_ctx *ctx, const uint8_t *Label, uint8_t *Label_swap, const float *VMap, int n, int m) {
    if (!ctx || !Label || !Label_swap || Label == Label_swap || !VMap || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    if (!labels16(ctx)) return rgbdslam_project_mask(ctx, Label, Label_swap, VMap, n, m);
    KLAUNCH_AS(ctx, "k_project_mask", (k_project_mask<uint16_t>), grid2d(n, m), block2d(), ctx->pose, ctx->calib, (const uint16_t *)Label,
               (uint16_t *)Label_swap, VMap, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <typename LT>
static int segment_impl(rgbdslam_ctx *ctx, LT *Label, LT *Label_swap, const float *VMap, const float *NMap, const float *pose,
                        const float *Eq, float epsilon, float alpha, int n, int m) {
    PoseV P = pose_from_host(pose);
    LT *a = Label, *b = Label_swap;
    for (int pass = 0; pass < 3; pass++) {                      // D1/RG.cu:910-911
        KLAUNCH_AS(ctx, "k_segment_pass", (k_segment_pass<LT>), grid2d(n, m), block2d(), (const LT *)a, b, VMap, NMap, P, Eq, epsilon, alpha, n, m);
        LT *t = a; a = b; b = t;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(Label, Label_swap, sizeof(LT) * (size_t)n * m, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
extern "C" int rgbdslam_segment(rgbdslam_ctx *ctx, uint8_t *Label, uint8_t *Label_swap, const float *VMap, const float *NMap,
                                const float *pose, const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    if (!ctx || !Label || !Label_swap || Label == Label_swap || !VMap || !NMap || !pose || !Eq || n <= 0 || m <= 0)
        return RGBDSLAM_E_ARG;
    if (nbPlans <= 0 || nbPlans > max_label(ctx)) return RGBDSLAM_E_LIMIT;
    if (labels16(ctx)) return segment_impl(ctx, (uint16_t *)Label, (uint16_t *)Label_swap, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
    return segment_impl(ctx, Label, Label_swap, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
}
extern "C" int rgbdslam_seed_labels(rgbdslam_ctx *ctx, uint8_t *Label, const float *VMap, const float *NMap, const float *pose,
                                    const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    if (!ctx || !Label || !VMap || !NMap || !pose || !Eq || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (nbPlans <= 0 || nbPlans > max_label(ctx)) return RGBDSLAM_E_LIMIT;
    if (labels16(ctx))
        KLAUNCH_AS(ctx, "k_seed_labels", (k_seed_labels<uint16_t>), grid2d(n, m), block2d(), (uint16_t *)Label, VMap, NMap,
                   pose_from_host(pose), Eq, nbPlans, epsilon, alpha, n, m);
    else
        KLAUNCH_AS(ctx, "k_seed_labels", (k_seed_labels<uint8_t>), grid2d(n, m), block2d(), Label, VMap, NMap, pose_from_host(pose), Eq,
                   nbPlans, epsilon, alpha, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
