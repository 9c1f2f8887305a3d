// compat.hip -- the remaining free functions of the reference's operator headers that its unchanged host code still
// calls (Kernel.cuh, Primitive.cuh, RG.cuh), so that integration/rgbdslam_shim.cpp can define EVERY symbol Frame.cpp,
// Graph.cpp and RG.cpp reference: the live-Kinect variants of the merge / model-seed kernels, the stand-alone grey
// copy, the label vote / remap pair of RG::ReorganizeEqua, the plane-overlap count, the device pointer-table store, and
// reference-shaped forms of Update_cu / Segment_cu whose extra work buffers (claim words, label swap) the library owns.
#include "device_fns.hpp"
#include <stdlib.h>

// ---- D1/Kernel.cu:528-614 MergeDepthKinectKernel: depth[(n-1-i)*m + (m-1-j)], no pitch -------------------------------
__global__ void k_merge_depth_kinect(float *__restrict__ RGBD, const uint16_t *__restrict__ depth, const float *__restrict__ RGB,
                                     uint8_t *__restrict__ mask, CalibV c, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    float4 px = *(float4 *)(RGBD + 4 * idx);
    F3 col = ld3(RGB, idx);
    int newmask;
    px.w = merge_pixel(px.w, depth[(size_t)(n - 1 - i) * m + (m - 1 - j)], mask[idx], c, newmask);
    px.x = col.x; px.y = col.y; px.z = col.z;
    *(float4 *)(RGBD + 4 * idx) = px;
    if (newmask >= 0) mask[idx] = (uint8_t)newmask;
}

// ---- D1/Kernel.cu:804-823 AddZBUFFKinectKernel: depth[idx] ------------------------------------------------------------
__global__ void k_add_zbuff_kinect(const uint16_t *__restrict__ depth, const float *__restrict__ rgb_in, float *__restrict__ RGBD,
                                   CalibV c, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    F3 col = ld3(rgb_in, idx);
    float4 o;
    o.x = col.x; o.y = col.y; o.z = col.z;
    o.w = zbuff_w(depth[idx], c);
    *(float4 *)(RGBD + 4 * idx) = o;
}

// ---- D1/Kernel.cu:670-682 CpyKernel -------------------------------------------------------------------------------------
__global__ void k_gpu_cpy(float *__restrict__ img, size_t pitch, const float *__restrict__ rgb, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    F3 c = ld3(rgb, (size_t)i * m + j);
    *(float *)((char *)img + (size_t)i * pitch + 4 * (size_t)j) = (float)((double)((c.z + c.y) + c.x) / 3.0);
}

// ---- D1/Kernel.cu:74-121 MapData2VBOKernel (live Kinect frame; LONG = 32 bit) -----------------------------------------
// lanes run along the OUTPUT row (idx_new), so the stores are coalesced and the reads run backwards
__global__ void k_map_data_2_vbo(const uint8_t *__restrict__ color, const uint16_t *__restrict__ depth,
                                 const int32_t *__restrict__ coord, float *__restrict__ VMap, float *__restrict__ RGB, CalibV c,
                                 int n, int m) {
    int jo = blockIdx.x * TILE_W + threadIdx.x, io = blockIdx.y * TILE_H + threadIdx.y;
    if (io >= n || jo >= m) return;
    const int i = n - 1 - io, j = m - 1 - jo;
    const size_t idx = (size_t)i * m + j, idx_new = (size_t)io * m + jo;
    float z = ((float)(-c.v[9]) * (float)depth[idx]) / c.v[10];
    float x = (((float)j - c.v[2]) * z) / c.v[0];
    float y = (((float)i - c.v[3]) * z) / c.v[1];
    int u = coord[2 * idx], v = coord[2 * idx + 1];
    if (u >= 0 && u < m && v >= 0 && v < n && (double)z > -SLAM_MAX_DEPTH) {
        const uint8_t *px = color + (size_t)v * ((size_t)m * 4) + (size_t)u * 4;
        st3(VMap, idx_new, x, y, z);
        st3(RGB, idx_new, (float)px[2] / 255.0f, (float)px[1] / 255.0f, (float)px[0] / 255.0f);
    }
}

// ---- D1/RG.cu:508-537 VotePlanKernel ------------------------------------------------------------------------------------
// The reference sums with unordered float atomics (scheduler dependent).  Defined order (oracle: orc_vote_plan): per label,
// a row's contributions left to right from 0, then the row sums top to bottom from 0.  One workgroup per label, one
// thread per row (the call runs once per 100 frames), row sums through LDS.
#define VOTE_MAX_ROWS 4096
template <typename LT>
__global__ __launch_bounds__(256)
void k_vote_plan(const LT *__restrict__ Label, const float *__restrict__ VMap, float *__restrict__ centers,
                 float *__restrict__ count, PoseV P, int n, int m) {
    __shared__ float4 rows[VOTE_MAX_ROWS];
    __shared__ unsigned char any_row[VOTE_MAX_ROWS];
    const int L = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) {
        float4 r; r.x = r.y = r.z = r.w = 0.0f;
        bool any = false;
        const LT *lab = Label + (size_t)i * m;
        for (int j = 0; j < m; j++) {
            if (lab[j] != L + 1) continue;
            F3 p = xf_point(P.v, ld3(VMap, (size_t)i * m + j));
            r.x = r.x + p.x; r.y = r.y + p.y; r.z = r.z + p.z; r.w = r.w + 1.0f;
            any = true;
        }
        rows[i] = r;
        any_row[i] = any ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 s; s.x = s.y = s.z = s.w = 0.0f;
        for (int i = 0; i < n; i++)
            if (any_row[i]) { s.x = s.x + rows[i].x; s.y = s.y + rows[i].y; s.z = s.z + rows[i].z; s.w = s.w + rows[i].w; }
        // like the reference's atomics, only labels that occur touch memory (the buffers hold _nbPlan entries, RG.cpp:939-942)
        if (s.w > 0.0f) { centers[3 * L] += s.x; centers[3 * L + 1] += s.y; centers[3 * L + 2] += s.z; count[L] += s.w; }
    }
}

// ---- D1/RG.cu:539-555 AffectPlanKernel -------------------------------------------------------------------------------------
template <typename LT>
__global__ void k_affect_plan(LT *__restrict__ Label, const int32_t *__restrict__ Buff, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    LT l = Label[idx];
    if (l != 0) Label[idx] = (LT)Buff[l - 1];
}

// ---- D1/Primitive.cu:502-556 OverlapKernel: counter[1] = occupied cells, counter[0] = those that land on plane param_in ------
__global__ void k_overlap(const uint16_t *__restrict__ Bump, const uint8_t *__restrict__ Mask, const float *__restrict__ param,
                          const float *__restrict__ param_in, unsigned int *__restrict__ counter, int n, int m, int N_prim, int M_prim) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    bool occupied = false, hit = false;
    if (i < n && j < m) {
        size_t idx = (size_t)i * m + j;
        const uint8_t mk = Mask[idx];
        if (mk >= 11) {
            BumpCell o = decode_cell(Bump[3 * idx], Bump[3 * idx + 1], Bump[3 * idx + 2], 0, 0, 0, mk, load_param(param), (float)i, (float)j, unorm8);
            F3 p; p.x = o.v.x; p.y = o.v.y; p.z = o.v.z;
            if (!is_zero3(p)) {
                occupied = true;
                BumpParam Q = load_param(param_in);
                UpdCell u;
                const bool inside = update_cell(p, Q.p, N_prim, M_prim, u);
                hit = !((double)__builtin_fabsf(u.err) > 0.1) && inside;     // EPSILON, cudaTypes.cuh:48
            }
        }
    }
    // one atomic per wave and counter instead of one per cell
    const unsigned long long mo = __ballot(occupied), mh = __ballot(hit);
    if ((threadIdx.x & 63) == 0) {
        if (mo) atomicAdd(&counter[1], (unsigned int)__builtin_popcountll(mo));
        if (mh) atomicAdd(&counter[0], (unsigned int)__builtin_popcountll(mh));
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int rgbdslam_merge_depth_kinect(rgbdslam_ctx *ctx, float *RGBD, const uint16_t *depth, const float *RGB, uint8_t *mask,
                                           int n, int m) {
    if (!ctx || !RGBD || !depth || !RGB || !mask || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_merge_depth_kinect, grid2d(n, m), block2d(), RGBD, depth, RGB, mask, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_gpu_add_zbuff_kinect(rgbdslam_ctx *ctx, const uint16_t *depth, const float *rgb_in, float *RGBD, int n, int m) {
    if (!ctx || !depth || !rgb_in || !RGBD || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_add_zbuff_kinect, grid2d(n, m), block2d(), depth, rgb_in, RGBD, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_gpu_cpy(rgbdslam_ctx *ctx, float *img, size_t img_pitch, const float *rgb, int n, int m) {
    if (!ctx || !img || !rgb || n <= 0 || m <= 0 || img_pitch < 4 * (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_gpu_cpy, grid2d(n, m), block2d(), img, img_pitch, rgb, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_map_data_2_vbo(rgbdslam_ctx *ctx, const uint8_t *colorFrame, const uint16_t *depthFrame,
                                       const int32_t *colorCoord, float *VMap, float *RGB, int n, int m) {
    if (!ctx || !colorFrame || !depthFrame || !colorCoord || !VMap || !RGB || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_map_data_2_vbo, grid2d(n, m), block2d(), colorFrame, depthFrame, colorCoord, VMap, RGB, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_vote_plan(rgbdslam_ctx *ctx, const uint8_t *Label, const float *VMap, float *centers_dev, float *count_dev,
                                  const float *pose16_host, int nbPlans, int n, int m) {
    if (!ctx || !Label || !VMap || !centers_dev || !count_dev || !pose16_host || n <= 0 || m <= 0 || nbPlans < 0) return RGBDSLAM_E_ARG;
    if (nbPlans > max_label(ctx) || n > VOTE_MAX_ROWS) return RGBDSLAM_E_LIMIT;
    if (nbPlans == 0) return 0;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = pose16_host[k];
    if (labels16(ctx))
        KLAUNCH_AS(ctx, "k_vote_plan", (k_vote_plan<uint16_t>), dim3(nbPlans), dim3(256), (const uint16_t *)Label, VMap, centers_dev, count_dev, P, n, m);
    else
        KLAUNCH_AS(ctx, "k_vote_plan", (k_vote_plan<uint8_t>), dim3(nbPlans), dim3(256), Label, VMap, centers_dev, count_dev, P, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_affect_plan(rgbdslam_ctx *ctx, uint8_t *Label, const int32_t *Buff_dev, int n, int m) {
    if (!ctx || !Label || !Buff_dev || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (labels16(ctx)) KLAUNCH_AS(ctx, "k_affect_plan", (k_affect_plan<uint16_t>), grid2d(n, m), block2d(), (uint16_t *)Label, Buff_dev, n, m);
    else KLAUNCH_AS(ctx, "k_affect_plan", (k_affect_plan<uint8_t>), grid2d(n, m), block2d(), Label, Buff_dev, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
// blocking, like the reference's cudaMemcpy of the two counters (Primitive.cu:745-770)
extern "C" int rgbdslam_overlap(rgbdslam_ctx *ctx, const uint16_t *Bump, const uint8_t *Mask, const float *param_dev,
                                const float *param_in_dev, int n, int m, int N_prim, int M_prim, float *ratio, uint32_t *counters2) {
    if (!ctx || !Bump || !Mask || !param_dev || !param_in_dev || !ratio || n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->compat_counter) HIP_TRY(hipMalloc((void **)&ctx->compat_counter, 2 * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(ctx->compat_counter, 0, 2 * sizeof(unsigned int), ctx->stream));
    KLAUNCH(ctx, k_overlap, grid2d(n, m), block2d(), Bump, Mask, param_dev, param_in_dev, ctx->compat_counter, n, m, N_prim, M_prim);
    HIP_TRY(hipGetLastError());
    unsigned int c[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(c, ctx->compat_counter, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (counters2) { counters2[0] = c[0]; counters2[1] = c[1]; }
    *ratio = c[1] == 0 ? 1.0f : (float)c[0] / (float)c[1];
    return 0;
}
// RG.cuh:27-29 Affect / AffectChar / AffectShort: Tab[indx] = ptr, as a stream-ordered 8-byte copy instead of a 1-thread kernel
extern "C" int rgbdslam_table_store(rgbdslam_ctx *ctx, void **tab_dev, const void *ptr, int indx) {
    if (!ctx || !tab_dev || indx < 0) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    // the source is a by-value local: the copy must have left the host before this function returns
    HIP_TRY(hipMemcpyAsync(tab_dev + indx, &ptr, sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Update_cu with the reference's exact argument list: the claim slabs of the winner rule and the call stamp
// are the library's.  nb_planes = number of valid entries of the four tables (labels beyond it are ignored).
extern "C" int rgbdslam_update_cu(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                                  const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label,
                                  const float *pose16_host, int N_prim, int M_prim, int n, int m, int nb_planes) {
    if (!ctx || nb_planes < 0 || N_prim <= 0 || M_prim <= 0) return RGBDSLAM_E_ARG;
    if (nb_planes > RGBDSLAM_COMPAT_MAX_PLANES) return RGBDSLAM_E_LIMIT;
    if (nb_planes == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t cells = (size_t)N_prim * M_prim;
    if (ctx->claim_cells != cells) {                            // plane size changed: start over
        for (int k = 0; k < RGBDSLAM_COMPAT_MAX_PLANES; k++)
            if (ctx->claim_slab[k]) { (void)hipFree(ctx->claim_slab[k]); ctx->claim_slab[k] = nullptr; }
        ctx->claim_cells = cells;
    }
    if (!ctx->claim_tab_dev) HIP_TRY(hipMalloc((void **)&ctx->claim_tab_dev, sizeof(uint64_t *) * RGBDSLAM_COMPAT_MAX_PLANES));
    bool grown = false;
    for (int k = 0; k < nb_planes; k++)
        if (!ctx->claim_slab[k]) {
            HIP_TRY(hipMalloc((void **)&ctx->claim_slab[k], sizeof(uint64_t) * cells));
            HIP_TRY(hipMemsetAsync(ctx->claim_slab[k], 0, sizeof(uint64_t) * cells, ctx->stream));
            grown = true;
        }
    if (grown) {
        HIP_TRY(hipMemcpyAsync(ctx->claim_tab_dev, ctx->claim_slab, sizeof(uint64_t *) * RGBDSLAM_COMPAT_MAX_PLANES, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return rgbdslam_update(ctx, BumpTab, RGBTab, MaskTab, ParamTab, ctx->claim_tab_dev, ++ctx->claim_stamp, VMap, RGBD, Mask, Label,
                           pose16_host, N_prim, M_prim, n, m, 0, nb_planes);
}

// Segment_cu with the reference's exact argument list: the swap image of the double-buffered passes is the library's
extern "C" int rgbdslam_segment_cu(rgbdslam_ctx *ctx, uint8_t *Label, const float *VMap, const float *NMap, const float *pose16_host,
                                   const float *Equations_dev, int nbPlans, float epsilon, float alpha, int n, int m) {
    if (!ctx || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t px = (size_t)n * m * (size_t)ctx->label_width;   // bytes of one label image
    if (ctx->label_swap_px < px) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->label_swap) { (void)hipFree(ctx->label_swap); ctx->label_swap = nullptr; ctx->label_swap_px = 0; }
        HIP_TRY(hipMalloc((void **)&ctx->label_swap, px));
        ctx->label_swap_px = px;
    }
    return rgbdslam_segment(ctx, Label, ctx->label_swap, VMap, NMap, pose16_host, Equations_dev, nbPlans, epsilon, alpha, n, m);
}

void compat_release(rgbdslam_ctx *ctx) {
    for (int k = 0; k < RGBDSLAM_COMPAT_MAX_PLANES; k++)
        if (ctx->claim_slab[k]) { (void)hipFree(ctx->claim_slab[k]); ctx->claim_slab[k] = nullptr; }
    if (ctx->claim_tab_dev) { (void)hipFree(ctx->claim_tab_dev); ctx->claim_tab_dev = nullptr; }
    if (ctx->label_swap) { (void)hipFree(ctx->label_swap); ctx->label_swap = nullptr; ctx->label_swap_px = 0; }
    if (ctx->compat_counter) { (void)hipFree(ctx->compat_counter); ctx->compat_counter = nullptr; }
}
