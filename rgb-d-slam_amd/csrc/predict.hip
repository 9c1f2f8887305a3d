// predict.hip -- stage 5: model prediction.  Rigid warp of the model depth image, quad trimming,
// the compute rasteriser that replaces the reference's OpenGL GL_QUADS pass, read-back to vertex
// maps, mask/label re-projection and the plane-label growing passes.
#include "common.hpp"

// ---- D1/Kernel.cu:345-379 TransformVertexMapKernel --------------------------------------------
__global__ void k_transform_vertex_map(PoseV P, float *__restrict__ VMap, float *__restrict__ RGBD, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    F3 v = ld3(VMap, idx);
    if (is_zero3(v)) { st3(VMap, idx, 0.0f, 0.0f, 0.0f); RGBD[4 * idx + 3] = 0.0f; return; }
    F3 r = xf_point(P.v, v);
    st3(VMap, idx, r.x, r.y, r.z);
    RGBD[4 * idx + 3] = ((double)(-r.z) > SLAM_MAX_DEPTH || (double)(-r.z) < 0.0) ? 0.0f
                                                                                   : (float)((double)(-r.z) / SLAM_MAX_DEPTH);
}

// ---- D1/Kernel.cu:381-438 QuadTrimFrameKernel ---------------------------------------------------
__device__ __forceinline__ float dist3(F3 a, F3 b) {
    float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
    return __builtin_sqrtf((x * x + y * y) + z * z);
}
__device__ __forceinline__ bool quad_trimmed(F3 a, F3 b, F3 c, F3 d, float thresh) {
    float mx = fmaxf(dist3(a, b), dist3(a, c));
    mx = fmaxf(dist3(a, d), mx);
    mx = fmaxf(dist3(b, c), mx);
    mx = fmaxf(dist3(b, d), mx);
    mx = fmaxf(dist3(c, d), mx);
    return mx > thresh;
}
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

// ---- compute rasteriser -------------------------------------------------------------
// Vertices are snapped to 1/256 pixel; coverage uses exact 64-bit integer edge functions with a
// top-left rule (watertight, each pixel of a shared edge drawn once); depth and attributes are
// interpolated perspective-correctly in double; the z-buffer is a u64 atomicMin on
// (f32 depth bits << 32 | ~triangle id): nearest wins, the later primitive wins ties (GL_LEQUAL).
#define RASTER_MIN_DEPTH 1e-3f
#define RASTER_GUARD 16384.0f
#define ZBUF_EMPTY 0xFFFFFFFFFFFFFFFFull

struct RasTri { int X[3], Y[3]; double inv_d[3]; long long area; int order[3]; };

__device__ __forceinline__ bool ras_setup(const float *__restrict__ VMap, const uint32_t *vid, const CalibV &c, RasTri &T) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        F3 v = ld3(VMap, vid[k]);
        float dk = -v.z;
        if (!(dk >= RASTER_MIN_DEPTH)) return false;
        float u = (v.x / dk) * c.v[0] + c.v[2];
        float w = (v.y / dk) * c.v[1] + c.v[3];
        if (!(__builtin_fabsf(u) <= RASTER_GUARD) || !(__builtin_fabsf(w) <= RASTER_GUARD)) return false;
        T.X[k] = f2i_rn(u * 256.0f);
        T.Y[k] = f2i_rn(w * 256.0f);
        T.inv_d[k] = 1.0 / (double)dk;
        T.order[k] = k;
    }
    long long area = (long long)(T.X[1] - T.X[0]) * (T.Y[2] - T.Y[0]) - (long long)(T.Y[1] - T.Y[0]) * (T.X[2] - T.X[0]);
    if (area == 0) return false;
    if (area < 0) {
        int t;
        t = T.X[1]; T.X[1] = T.X[2]; T.X[2] = t;
        t = T.Y[1]; T.Y[1] = T.Y[2]; T.Y[2] = t;
        double dd = T.inv_d[1]; T.inv_d[1] = T.inv_d[2]; T.inv_d[2] = dd;
        T.order[1] = 2; T.order[2] = 1;
        area = -area;
    }
    T.area = area;
    return true;
}
__device__ __forceinline__ long long ras_edge(int ax, int ay, int bx, int by, int px, int py) {
    return (long long)(bx - ax) * (py - ay) - (long long)(by - ay) * (px - ax);
}
__device__ __forceinline__ bool ras_topleft(int ax, int ay, int bx, int by) {
    int dx = bx - ax, dy = by - ay;
    return (dy < 0) || (dy == 0 && dx < 0);
}
__device__ __forceinline__ bool ras_cover(const RasTri &T, int px, int py, long long *w) {
    int PX = px * 256, PY = py * 256;
    w[0] = ras_edge(T.X[1], T.Y[1], T.X[2], T.Y[2], PX, PY);
    w[1] = ras_edge(T.X[2], T.Y[2], T.X[0], T.Y[0], PX, PY);
    w[2] = ras_edge(T.X[0], T.Y[0], T.X[1], T.Y[1], PX, PY);
    if (w[0] < 0 || w[1] < 0 || w[2] < 0) return false;
    if (w[0] == 0 && !ras_topleft(T.X[1], T.Y[1], T.X[2], T.Y[2])) return false;
    if (w[1] == 0 && !ras_topleft(T.X[2], T.Y[2], T.X[0], T.Y[0])) return false;
    if (w[2] == 0 && !ras_topleft(T.X[0], T.Y[0], T.X[1], T.Y[1])) return false;
    return true;
}

__device__ __forceinline__ void ras_draw(unsigned long long *__restrict__ zbuf, const float *__restrict__ VMap,
                                         const uint32_t *vid, uint32_t tri_id, const CalibV &c, float znear, float zfar,
                                         int n, int m) {
    RasTri T;
    if (!ras_setup(VMap, vid, c, T)) return;
    int xmn = min(T.X[0], min(T.X[1], T.X[2])), xmx = max(T.X[0], max(T.X[1], T.X[2]));
    int ymn = min(T.Y[0], min(T.Y[1], T.Y[2])), ymx = max(T.Y[0], max(T.Y[1], T.Y[2]));
    int x0 = max(0, (xmn + 255) >> 8), x1 = min(m - 1, xmx >> 8);
    int y0 = max(0, (ymn + 255) >> 8), y1 = min(n - 1, ymx >> 8);
    for (int py = y0; py <= y1; py++)
        for (int px = x0; px <= x1; px++) {
            long long w[3];
            if (!ras_cover(T, px, py, w)) continue;
            double W = ((double)w[0] * T.inv_d[0] + (double)w[1] * T.inv_d[1]) + (double)w[2] * T.inv_d[2];
            float dpx = (float)((double)T.area / W);
            if (!(dpx >= znear) || !(dpx <= zfar)) continue;
            unsigned long long key = ((unsigned long long)__float_as_uint(dpx) << 32) | (unsigned long long)(0xFFFFFFFFu - tri_id);
            atomicMin(&zbuf[(size_t)py * m + px], key);
        }
}

__device__ __forceinline__ void quad_vertex_ids(uint32_t q, const uint32_t *__restrict__ index, int m, uint32_t *id) {
    if (index) {
        uint4 v = *(const uint4 *)(index + 4 * (size_t)q);
        id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
    } else {
        uint32_t i = q / (uint32_t)(m - 1), j = q % (uint32_t)(m - 1);
        id[0] = i * m + j; id[1] = id[0] + 1; id[2] = id[0] + m + 1; id[3] = id[0] + m;
    }
}

// pass 1: one thread per quad
__global__ void k_raster_draw(unsigned long long *__restrict__ zbuf, const float *__restrict__ VMap,
                              const uint32_t *__restrict__ index, float thresh, CalibV c, float znear, float zfar,
                              int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i > n - 2 || j > m - 2) return;
    uint32_t q = (uint32_t)(i * (m - 1) + j), id[4];
    quad_vertex_ids(q, index, m, id);
    if (!index && quad_trimmed(ld3(VMap, id[0]), ld3(VMap, id[1]), ld3(VMap, id[2]), ld3(VMap, id[3]), thresh)) return;
    uint32_t t0[3] = {id[0], id[1], id[2]}, t1[3] = {id[0], id[2], id[3]};
    ras_draw(zbuf, VMap, t0, 2 * q, c, znear, zfar, n, m);
    ras_draw(zbuf, VMap, t1, 2 * q + 1, c, znear, zfar, n, m);
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
        RasTri T; long long w[3];
        ras_setup(VMap, vid, c, T);
        ras_cover(T, px, py, w);
        double b0 = (double)w[0] * T.inv_d[0], b1 = (double)w[1] * T.inv_d[1], b2 = (double)w[2] * T.inv_d[2];
        double W = (b0 + b1) + b2;
        // T.order is a permutation of {0,1,2}: select without dynamic indexing
        uint32_t va = vid[0], vb = (T.order[1] == 1) ? vid[1] : vid[2], vc = (T.order[1] == 1) ? vid[2] : vid[1];
        float4 A0 = *(const float4 *)(RGBD + 4 * (size_t)va), A1 = *(const float4 *)(RGBD + 4 * (size_t)vb),
               A2 = *(const float4 *)(RGBD + 4 * (size_t)vc);
        o.x = (float)(((b0 * (double)A0.x + b1 * (double)A1.x) + b2 * (double)A2.x) / W);
        o.y = (float)(((b0 * (double)A0.y + b1 * (double)A1.y) + b2 * (double)A2.y) / W);
        o.z = (float)(((b0 * (double)A0.z + b1 * (double)A1.z) + b2 * (double)A2.z) / W);
        o.w = (float)(((b0 * (double)A0.w + b1 * (double)A1.w) + b2 * (double)A2.w) / W);
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
    float z = (float)((double)t.w * SLAM_MAX_DEPTH);
    st3(VMap, idx, z * (((float)j - c.v[2]) / c.v[0]), z * (((float)i - c.v[3]) / c.v[1]), -z);
    *(float4 *)(RGBD + 4 * idx) = t;
}

// ---- D1/Kernel.cu:634-667 ProjectMaskKernel -------------------------------------------------------
__global__ void k_project_mask(PoseV P, CalibV c, const uint8_t *__restrict__ mask, uint8_t *__restrict__ mask_swap,
                               const float *__restrict__ VMap, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    F3 v = ld3(VMap, idx);
    if (is_zero3(v)) { mask_swap[idx] = 0; return; }
    F3 r = xf_point(P.v, v);
    float az = __builtin_fabsf(r.z);
    int pu = f2i_rn((r.x / az) * c.v[0] + c.v[2]);
    int pv = f2i_rn((r.y / az) * c.v[1] + c.v[3]);
    // the reference only tests the upper bounds; negative projections are defined as out of view
    if (pv > n - 1 || pu > m - 1 || pv < 0 || pu < 0) { mask_swap[idx] = 0; return; }
    mask_swap[idx] = mask[(size_t)pv * m + pu];
}

// ---- D1/RG.cu:252-320 SegmentKernel, one double-buffered pass --------------------------------------
__global__ void k_segment_pass(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const float *__restrict__ VMap,
                               const float *__restrict__ NMap, PoseV P, const float *__restrict__ Eq, float epsilon,
                               float alpha, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    uint8_t mine = in[idx];
    if (mine > 0) { out[idx] = mine; return; }
    uint8_t res = 0;
    F3 nl = ld3(NMap, idx);
    F3 q = xf_dir(P.v, nl);
    if (!is_zero3(q)) {
        F3 pl = ld3(VMap, idx);
        F3 p = xf_point(P.v, pl);
        float depth = pl.z;
        const float thresh_depth = 0.003f;
        int lb = max(0, i - 1), ub = min(n, i + 2), lr = max(0, j - 1), ur = min(m, j + 2);
        for (int ki = lb; ki < ub && res == 0; ki++)
            for (int kj = lr; kj < ur; kj++) {
                size_t nb = (size_t)ki * m + kj;
                uint8_t l = in[nb];
                if (l > 0 && __builtin_fabsf(VMap[3 * nb + 2] - depth) < thresh_depth) {
                    const float *E = Eq + 10 * ((int)l - 1);
                    float ed = ((p.x * E[0] + p.y * E[1]) + p.z * E[2]) - E[9];
                    float ea = (q.x * E[0] + q.y * E[1]) + q.z * E[2];
                    if (__builtin_fabsf(ed) > epsilon || ea < alpha) continue;
                    res = l;
                    break;
                }
            }
    }
    out[idx] = res;
}

// plane-membership labelling (test-harness stand-in for plane detection)
__global__ void k_seed_labels(uint8_t *__restrict__ Label, const float *__restrict__ VMap, const float *__restrict__ NMap,
                              PoseV P, const float *__restrict__ Eq, int nb, float epsilon, float alpha, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    uint8_t res = 0;
    F3 nl = ld3(NMap, idx);
    if (!is_zero3(nl)) {
        F3 p = xf_point(P.v, ld3(VMap, idx));
        F3 q = xf_dir(P.v, nl);
        for (int k = 0; k < nb && k < 255; k++) {
            const float *E = Eq + 10 * k;
            float ed = ((p.x * E[0] + p.y * E[1]) + p.z * E[2]) - E[9];
            float ea = (q.x * E[0] + q.y * E[1]) + q.z * E[2];
            if (__builtin_fabsf(ed) > epsilon || ea < alpha) continue;
            res = (uint8_t)(k + 1);
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
    KLAUNCH(ctx, k_project_mask, grid2d(n, m), block2d(), ctx->pose, ctx->calib, mask, mask_swap, VMap, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_segment(rgbdslam_ctx *ctx, uint8_t *Label, uint8_t *Label_swap, const float *VMap, const float *NMap,
                                const float *pose, const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    if (!ctx || !Label || !Label_swap || Label == Label_swap || !VMap || !NMap || !pose || !Eq || n <= 0 || m <= 0)
        return RGBDSLAM_E_ARG;
    if (nbPlans <= 0 || nbPlans > 255) return RGBDSLAM_E_LIMIT;
    PoseV P = pose_from_host(pose);
    uint8_t *a = Label, *b = Label_swap;
    for (int pass = 0; pass < 3; pass++) {                      // D1/RG.cu:910-911
        KLAUNCH(ctx, k_segment_pass, grid2d(n, m), block2d(), a, b, VMap, NMap, P, Eq, epsilon, alpha, n, m);
        uint8_t *t = a; a = b; b = t;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(Label, Label_swap, (size_t)n * m, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
extern "C" int rgbdslam_seed_labels(rgbdslam_ctx *ctx, uint8_t *Label, const float *VMap, const float *NMap, const float *pose,
                                    const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    if (!ctx || !Label || !VMap || !NMap || !pose || !Eq || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (nbPlans <= 0 || nbPlans > 255) return RGBDSLAM_E_LIMIT;
    KLAUNCH(ctx, k_seed_labels, grid2d(n, m), block2d(), Label, VMap, NMap, pose_from_host(pose), Eq,
                       nbPlans, epsilon, alpha, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
