// preprocess.hip -- stage 1 (depth/colour -> vertex, normal, intensity-gradient maps) and the
// per-pixel model maintenance kernels of stage 4/5 (merge, vertex-from-depth, model seeding).
// Lanes run along image columns so every row segment is one coalesced burst.
#include "device_fns.hpp"
#include <math.h>

// ---- D1/Kernel.cu:33-52 VertexMapKinectKernel ------------------------------------------------
__global__ void k_vertex_map_kinect(float *__restrict__ VMap, const uint16_t *__restrict__ depth,
                                    size_t pitch, CalibV c, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    const uint16_t *row = (const uint16_t *)((const char *)depth + (size_t)(n - i - 1) * pitch);
    F3 v = vertex_from_depth(row[j], i, j, c);
    st3(VMap, idx, v.x, v.y, v.z);
}

// ---- D1/Kernel.cu:739-753 CpyKernelChar3 ------------------------------------------------------
__global__ void k_cpy_char3(const uint8_t *__restrict__ bgr, size_t pitch, float *__restrict__ rgb, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    const uint8_t *px = bgr + (size_t)(n - i - 1) * pitch + 3 * (size_t)j;
    st3(rgb, idx, (float)((double)(float)px[2] / 255.0), (float)((double)(float)px[1] / 255.0),
        (float)((double)(float)px[0] / 255.0));
}

// ---- D1/Kernel.cu:123-307 NormalKernel (device_fns.hpp) --------------------------------------------
__global__ void k_compute_normal(float *__restrict__ NMap, const float *__restrict__ VMap, int n, int m, int inverse) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    F3 r = normal_global(VMap, i, j, n, m, inverse);
    st3(NMap, (size_t)i * m + j, r.x, r.y, r.z);
}

// ---- D1/Kernel.cu:670-682 CpyKernel + cv::gpu::Sobel x2 (D1/RG.cpp:807-810), fused -------------
__global__ void k_intensity_sobel(const float *__restrict__ rgb, float *__restrict__ img, size_t img_pitch,
                                  float *__restrict__ gx, size_t gx_pitch, float *__restrict__ gy,
                                  size_t gy_pitch, int n, int m) {
    // LDS tile of intensities with a one-pixel reflect-101 halo
    __shared__ float tile[TILE_H + 2][TILE_W + 2];
    int j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * TILE_H;
    int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int e = tid; e < (TILE_H + 2) * (TILE_W + 2); e += TILE_W * TILE_H) {
        int r = e / (TILE_W + 2), c = e % (TILE_W + 2);
        int ii = i0 + r - 1, jj = j0 + c - 1;
        float v = 0.0f;
        if (ii <= n && jj <= m) {   // one past the edge is still needed (reflected)
            size_t s = (size_t)reflect101(ii, n) * m + reflect101(jj, m);
            float sum = (rgb[3 * s + 2] + rgb[3 * s + 1]) + rgb[3 * s];   // Kernel.cu:678
            v = (float)((double)sum / 3.0);
        }
        tile[r][c] = v;
    }
    __syncthreads();
    int j = j0 + threadIdx.x, i = i0 + threadIdx.y;
    if (i >= n || j >= m) return;
    int r = threadIdx.y + 1, c = threadIdx.x + 1;
    float rowd[3], rows[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = tile[r - 1 + k][c - 1], b = tile[r - 1 + k][c], cc = tile[r - 1 + k][c + 1];
        float s = 0.0f;
        s = s + a * -1.0f; s = s + b * 0.0f; s = s + cc * 1.0f;
        rowd[k] = s;
        s = 0.0f;
        s = s + a * 1.0f; s = s + b * 2.0f; s = s + cc * 1.0f;
        rows[k] = s;
    }
    float s = 0.0f;
    s = s + rowd[0] * 1.0f; s = s + rowd[1] * 2.0f; s = s + rowd[2] * 1.0f;
    *(float *)((char *)gx + (size_t)i * gx_pitch + 4 * (size_t)j) = s;
    s = 0.0f;
    s = s + rows[0] * -1.0f; s = s + rows[1] * 0.0f; s = s + rows[2] * 1.0f;
    *(float *)((char *)gy + (size_t)i * gy_pitch + 4 * (size_t)j) = s;
    if (img) *(float *)((char *)img + (size_t)i * img_pitch + 4 * (size_t)j) = tile[r][c];
}

// ---- D1/Kernel.cu:440-526 MergeDepthOffKernel ---------------------------------------------------
__global__ void k_merge_depth_off(float *__restrict__ RGBD, const uint16_t *__restrict__ depth, size_t pitch,
                                  const float *__restrict__ RGB, uint8_t *__restrict__ mask, CalibV c, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    const uint16_t *row = (const uint16_t *)((const char *)depth + (size_t)(n - i - 1) * pitch);
    float4 px = *(float4 *)(RGBD + 4 * idx);
    F3 col = ld3(RGB, idx);
    int newmask;
    px.w = merge_pixel(px.w, row[j], mask[idx], c, newmask);
    px.x = col.x; px.y = col.y; px.z = col.z;
    *(float4 *)(RGBD + 4 * idx) = px;
    if (newmask >= 0) mask[idx] = (uint8_t)newmask;
}

// ---- D1/Kernel.cu:54-72 VertexMapGLKernel -------------------------------------------------------
__global__ void k_vertex_map_gl(float *__restrict__ VMap, const float *__restrict__ RGBD, CalibV c, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    F3 v = vertex_from_w(RGBD[4 * idx + 3], i, j, c);
    st3(VMap, idx, v.x, v.y, v.z);
}

// ---- D1/Kernel.cu:783-802 AddZBUFFOffKernel, 616-632 InitMaskKernel -----------------------------
__global__ void k_add_zbuff_off(const uint16_t *__restrict__ depth, size_t pitch, const float *__restrict__ rgb_in,
                                float *__restrict__ RGBD, CalibV c, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    const uint16_t *row = (const uint16_t *)((const char *)depth + (size_t)(n - i - 1) * pitch);
    F3 col = ld3(rgb_in, idx);
    float4 o;
    o.x = col.x; o.y = col.y; o.z = col.z;
    o.w = zbuff_w(row[j], c);
    *(float4 *)(RGBD + 4 * idx) = o;
}
__global__ void k_init_mask(uint8_t *__restrict__ mask, const float *__restrict__ RGBD, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    mask[idx] = (RGBD[4 * idx + 3] == 0.0f) ? 10 : 11;
}

// ---- optional, default-OFF stage of OffLineFrame::LoadFrame: the bilateral filter the reference has commented out ---------
// D1/Frame.cpp:50-54:  depth.convertTo(tmp, CV_32FC1); bilateralFilter(tmp, out, 5, 100.0, 100.0); out.convertTo(depth, CV_16UC1)
// cv::bilateralFilter is third-party (OpenCV 2.4, version unpinned): restated from its published float algorithm -- circular
// window of radius d/2, reflect-101 border, spatial weights exp(-r^2 / 2 sigma_s^2), colour weights from a 4096-bin
// exponential table with linear interpolation -- with ONE stated difference: the table spans the full u16 range instead of
// the image's own min..max (no reduction pass).  Both tables are computed on the host in double, so the kernel and the
// oracle only multiply and add: bit-exact.  LDS tile of the depth samples with a `radius` halo.
#define BIL_MAX_RADIUS 4
#define BIL_LUT_BINS 4096
struct BilTaps { int count; signed char dy[81], dx[81]; float w[81]; float scale_index; int radius; };
__global__ __launch_bounds__(TILE_W * TILE_H)
void k_bilateral_depth(uint16_t *__restrict__ dst, size_t dpitch, const uint16_t *__restrict__ src, size_t spitch,
                       const float *__restrict__ lut, BilTaps T, int n, int m) {
    __shared__ float tile[TILE_H + 2 * BIL_MAX_RADIUS][TILE_W + 2 * BIL_MAX_RADIUS];
    const int R = T.radius, j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * TILE_H;
    const int tw = TILE_W + 2 * R, th = TILE_H + 2 * R;
    for (int e = threadIdx.y * TILE_W + threadIdx.x; e < tw * th; e += TILE_W * TILE_H) {
        const int r = e / tw, c = e % tw;
        const int ii = reflect101(min(i0 + r - R, n - 1 + R), n), jj = reflect101(min(j0 + c - R, m - 1 + R), m);
        tile[r][c] = (float)((const uint16_t *)((const char *)src + (size_t)ii * spitch))[jj];
    }
    __syncthreads();
    const int j = j0 + threadIdx.x, i = i0 + threadIdx.y;
    if (i >= n || j >= m) return;
    const float val0 = tile[threadIdx.y + R][threadIdx.x + R];
    float sum = 0.0f, wsum = 0.0f;
    for (int k = 0; k < T.count; k++) {
        const float val = tile[threadIdx.y + R + T.dy[k]][threadIdx.x + R + T.dx[k]];
        float alpha = __builtin_fabsf(val - val0) * T.scale_index;
        const int idx = (int)__builtin_floorf(alpha);
        alpha = alpha - (float)idx;
        const float l0 = lut[idx], l1 = lut[idx + 1];
        const float w = T.w[k] * (l0 + alpha * (l1 - l0));
        sum = sum + val * w;
        wsum = wsum + w;
    }
    float o = __builtin_rintf(sum / wsum);                     // convertTo(CV_16UC1): cvRound + saturate
    o = o < 0.0f ? 0.0f : (o > 65535.0f ? 65535.0f : o);
    ((uint16_t *)((char *)dst + (size_t)i * dpitch))[j] = (uint16_t)o;
}

// =================================================================================================
// C ABI
// =================================================================================================
#define CHECK_CTX(ctx) do { if (!(ctx)) return RGBDSLAM_E_ARG; } while (0)
#define CHECK_DIM(n, m) do { if ((n) <= 0 || (m) <= 0) return RGBDSLAM_E_ARG; } while (0)

extern "C" int rgbdslam_vertex_map_kinect(rgbdslam_ctx *ctx, float *VMap, const uint16_t *depth, size_t pitch, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!VMap || !depth || pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_vertex_map_kinect, grid2d(n, m), block2d(), VMap, depth, pitch, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_gpu_cpy_char3(rgbdslam_ctx *ctx, const uint8_t *bgr, size_t pitch, float *rgb, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!bgr || !rgb || pitch < 3 * (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_cpy_char3, grid2d(n, m), block2d(), bgr, pitch, rgb, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_compute_normal(rgbdslam_ctx *ctx, float *NMap, const float *VMap, int n, int m, int inverse) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!NMap || !VMap || NMap == VMap) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_compute_normal, grid2d(n, m), block2d(), NMap, VMap, n, m, inverse);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_intensity_sobel(rgbdslam_ctx *ctx, const float *rgb, float *img, size_t img_pitch, float *gx,
                                        size_t gx_pitch, float *gy, size_t gy_pitch, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!rgb || !gx || !gy || gx_pitch < 4 * (size_t)m || gy_pitch < 4 * (size_t)m) return RGBDSLAM_E_ARG;
    if (img && img_pitch < 4 * (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_intensity_sobel, grid2d(n, m), block2d(), rgb, img, img_pitch, gx, gx_pitch, gy, gy_pitch, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_merge_depth_off(rgbdslam_ctx *ctx, float *RGBD, const uint16_t *depth, size_t pitch,
                                        const float *RGB, uint8_t *mask, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!RGBD || !depth || !RGB || !mask || pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_merge_depth_off, grid2d(n, m), block2d(), RGBD, depth, pitch, RGB, mask, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_vertex_map_gl(rgbdslam_ctx *ctx, float *VMap, const float *RGBD, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!VMap || !RGBD) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_vertex_map_gl, grid2d(n, m), block2d(), VMap, RGBD, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_gpu_add_zbuff_off(rgbdslam_ctx *ctx, const uint16_t *depth, size_t pitch, const float *rgb_in,
                                          float *RGBD, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!depth || !rgb_in || !RGBD || pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_add_zbuff_off, grid2d(n, m), block2d(), depth, pitch, rgb_in, RGBD, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_init_mask(rgbdslam_ctx *ctx, uint8_t *mask, const float *RGBD, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!mask || !RGBD) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_init_mask, grid2d(n, m), block2d(), mask, RGBD, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

// cv::bilateralFilter(depth as f32, d, sigmaColor, sigmaSpace) + conversion back to u16 (D1/Frame.cpp:50-54, commented out in the
// reference: an OPTIONAL stage, off unless the caller asks for it).  dst != src.
extern "C" int rgbdslam_bilateral_depth(rgbdslam_ctx *ctx, uint16_t *dst, size_t dst_pitch, const uint16_t *src, size_t src_pitch, int d,
                                        double sigma_color, double sigma_space, int n, int m) {
    CHECK_CTX(ctx); CHECK_DIM(n, m);
    if (!dst || !src || dst == src || dst_pitch < 2 * (size_t)m || src_pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if (sigma_color <= 0.0) sigma_color = 1.0;                 // OpenCV's own defaults for non-positive sigmas
    if (sigma_space <= 0.0) sigma_space = 1.0;
    int radius = d <= 0 ? (int)__builtin_rint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    if (radius > BIL_MAX_RADIUS) return RGBDSLAM_E_LIMIT;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->bil_lut) HIP_TRY(hipMalloc((void **)&ctx->bil_lut, sizeof(float) * (BIL_LUT_BINS + 2)));
    const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color), gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
    const float len = 65535.0f, scale_index = (float)BIL_LUT_BINS / len;
    if (ctx->bil_sigma_color != sigma_color) {
        float host_lut[BIL_LUT_BINS + 2];
        for (int i = 0; i < BIL_LUT_BINS + 2; i++) {
            const double val = (double)i / (double)scale_index;
            host_lut[i] = (float)exp(val * val * gauss_color_coeff);
        }
        HIP_TRY(hipMemcpyAsync(ctx->bil_lut, host_lut, sizeof(host_lut), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));            // host_lut is a local
        ctx->bil_sigma_color = sigma_color;
    }
    BilTaps T;
    T.count = 0; T.radius = radius; T.scale_index = scale_index;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            T.dy[T.count] = (signed char)i; T.dx[T.count] = (signed char)j;
            T.w[T.count++] = (float)exp(r * r * gauss_space_coeff);
        }
    KLAUNCH(ctx, k_bilateral_depth, grid2d(n, m), block2d(), dst, dst_pitch, src, src_pitch, ctx->bil_lut, T, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the frame's auxiliary ICP records from maps that already exist (no reference counterpart; rgbdslam_set_frame_aux) ----
// For callers that preprocess with the per-operator entry points: the four values MatchKernelGaussNewton takes from a pixel of
// the current frame (ICP.cu:167-173), by its own expressions on the stored maps.
__global__ void k_frame_aux(const float *__restrict__ VMap, const float *__restrict__ RGB, const float *__restrict__ Gx, size_t gxp,
                            const float *__restrict__ Gy, size_t gyp, float *__restrict__ aux, int n, int m) {
    const int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    const size_t idx = (size_t)i * m + j;
    const float gx = *(const float *)((const char *)Gx + (size_t)i * gxp + 4 * (size_t)j);
    const float gy = *(const float *)((const char *)Gy + (size_t)i * gyp + 4 * (size_t)j);
    float4 x;
    x.x = (float)((double)gx / 8.0);
    x.y = (float)((double)gy / 8.0);
    x.z = ((RGB[3 * idx] + RGB[3 * idx + 1]) + RGB[3 * idx + 2]) / 3.0f;
    x.w = icp_depth_weight(VMap[3 * idx + 2]);
    *(float4 *)(aux + 4 * idx) = x;
}

extern "C" int rgbdslam_frame_aux(rgbdslam_ctx *ctx, const float *VMap, const float *RGB, const float *Gx, size_t gx_pitch,
                                  const float *Gy, size_t gy_pitch, float *aux, int n, int m) {
    if (!ctx || !VMap || !RGB || !Gx || !Gy || !aux || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (((uintptr_t)aux & 15) != 0 || gx_pitch < 4 * (size_t)m || gy_pitch < 4 * (size_t)m || ((gx_pitch | gy_pitch) & 3)) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    KLAUNCH(ctx, k_frame_aux, grid2d(n, m), block2d(), VMap, RGB, Gx, gx_pitch, Gy, gy_pitch, aux, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
