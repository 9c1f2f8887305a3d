// detect.hip -- plane detection front end: everything of RG::DetectPrimitives
// (D1/RG.cpp:277-338) up to the list of candidate regions, on the GPU.
//   CpyKernelN / SumKernel / DiscontKernel / ThreshKernel      D1/Kernel.cu:684-781 (one kernel each, same arithmetic)
//   cv::gpu::Canny(img, edges, 120, 30, 3)                     RG.cpp:296-302   -> k_canny_map + union-find hysteresis
//   cv::erode with the 5x5 ellipse (CPU in the reference)      RG.cpp:320-326   -> k_erode5
//   FindBlobs = cv::floodFill, 4-connected (CPU)               Utilities.cpp:1052-1092 -> union-find components
// The three OpenCV stages are restated from their published algorithms (see oracle/detect_oracle.c: parity
// unpinned).  Connected components (4- and 8-connected) use one lock-free union-find over the pixel
// grid: every component ends up labelled with the raster index of its FIRST pixel, which is exactly the
// order in which FindBlobs numbers its blobs -- no host round trip, no flood fill.
// Two launch shapes: one entry point per reference operator, and rgbdslam_detect_regions, which keeps the
// three channel images, the edge images and the summed image out of memory altogether.
#include "common.hpp"
#include "device_fns.hpp"

// ---- per-pixel functions shared by both shapes ------------------------------------------------------------
__device__ __forceinline__ uint8_t normal_to_byte(float c) { return (uint8_t)(int)(125.0 * (1.0 + (double)c)); }   // Kernel.cu:691
__device__ __forceinline__ uint8_t edge_mean(uint8_t r, uint8_t g, uint8_t b) {                                      // Kernel.cu:762
    return (uint8_t)(int)((double)(((float)r + (float)g) + (float)b) / 3.0);
}
// Kernel.cu:704-717: the loops stop BEFORE i+1 / j+1, so only the pixel, its upper and its left neighbours count
__device__ __forceinline__ bool depth_step(const float *__restrict__ VMap, int i, int j, int n, int m, float thresh) {
    float depth = VMap[3 * ((size_t)i * m + j) + 2];
    for (int k = max(i - 1, 0); k < min(i + 1, n); k++)
        for (int h = max(j - 1, 0); h < min(j + 1, m); h++) {
            float z = VMap[3 * ((size_t)k * m + h) + 2];
            if (z != 0.0f && __builtin_fabsf(depth - z) > thresh) return true;
        }
    return false;
}

// ---- Canny stage 1: Sobel + L1 magnitude + non-maximum suppression -> 0 none / 1 weak / 2 strong ------------
// GRAY(i, j) must return the byte image with clamped (replicated) coordinates already applied.
#define CANNY_GH (TILE_H + 4)
#define CANNY_GW (TILE_W + 4)
#define CANNY_MH (TILE_H + 2)
#define CANNY_MW (TILE_W + 2)
template <typename GRAY>
__device__ __forceinline__ uint8_t canny_map_tile(GRAY gray, int low, int high, int n, int m) {
    __shared__ uint8_t sg[CANNY_GH][CANNY_GW];
    __shared__ int smag[CANNY_MH][CANNY_MW];
    const int j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * TILE_H;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int e = tid; e < CANNY_GH * CANNY_GW; e += TILE_W * TILE_H) {
        int r = e / CANNY_GW, c = e % CANNY_GW;
        int ii = min(max(i0 + r - 2, 0), n - 1), jj = min(max(j0 + c - 2, 0), m - 1);
        sg[r][c] = gray(ii, jj);
    }
    __syncthreads();
    auto sobel = [&](int r, int c, int &dx, int &dy) {         // gradient at magnitude position (r, c); grey centre at (r + 1, c + 1)
        int gu = r, gd = r + 2, gl = c, gr = c + 2, gc = r + 1, gcc = c + 1;
        dx = ((int)sg[gu][gr] + 2 * (int)sg[gc][gr] + (int)sg[gd][gr]) - ((int)sg[gu][gl] + 2 * (int)sg[gc][gl] + (int)sg[gd][gl]);
        dy = ((int)sg[gd][gl] + 2 * (int)sg[gd][gcc] + (int)sg[gd][gr]) - ((int)sg[gu][gl] + 2 * (int)sg[gu][gcc] + (int)sg[gu][gr]);
    };
    for (int e = tid; e < CANNY_MH * CANNY_MW; e += TILE_W * TILE_H) {
        int r = e / CANNY_MW, c = e % CANNY_MW;             // magnitude position (i0 + r - 1, j0 + c - 1)
        int ii = i0 + r - 1, jj = j0 + c - 1;
        int mg = 0;                                         // the magnitude is 0 outside the image
        if (ii >= 0 && ii < n && jj >= 0 && jj < m) {       // inside: the clamped tile load IS the replicated border
            int dx, dy;
            sobel(r, c, dx, dy);
            mg = abs(dx) + abs(dy);
        }
        smag[r][c] = mg;
    }
    __syncthreads();
    const int r = threadIdx.y + 1, c = threadIdx.x + 1;
    int my_dx, my_dy;
    sobel(r, c, my_dx, my_dy);
    const int mm = smag[r][c];
    uint8_t out = 0;
    if (mm > low) {
        int x = abs(my_dx), y = abs(my_dy) << 15;                // |dx|, |dy| <= 1020: everything fits 32 bits
        int tg22x = x * 13573, tg67x = tg22x + (x << 16);
        bool keep;
        if (y < tg22x) keep = mm > smag[r][c - 1] && mm >= smag[r][c + 1];
        else if (y > tg67x) keep = mm > smag[r - 1][c] && mm >= smag[r + 1][c];
        else { int s = ((my_dx ^ my_dy) < 0) ? -1 : 1; keep = mm > smag[r - 1][c - s] && mm > smag[r + 1][c + s]; }
        if (keep) out = mm > high ? 2 : 1;
    }
    return out;
}

__global__ __launch_bounds__(TILE_W *TILE_H)
void k_canny_map(const uint8_t *__restrict__ img, size_t pitch, uint8_t *__restrict__ map, int low, int high, int n, int m) {
    uint8_t v = canny_map_tile([&](int i, int j) { return img[(size_t)i * pitch + j]; }, low, high, n, m);
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i < n && j < m) map[(size_t)i * m + j] = v;
}
// all three normal channels at once (blockIdx.z = channel), grey value computed from the normal map on the fly
__global__ __launch_bounds__(TILE_W *TILE_H)
void k_canny_map3(const float *__restrict__ NMap, uint8_t *__restrict__ map, int low, int high, int n, int m) {
    const int ch = blockIdx.z;
    uint8_t v = canny_map_tile([&](int i, int j) { return normal_to_byte(NMap[3 * ((size_t)i * m + j) + ch]); }, low, high, n, m);
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i < n && j < m) map[(size_t)ch * n * m + (size_t)i * m + j] = v;
}

// ---- lock-free union-find over the pixel grid ---------------------------------------------------------------
// label[p] = p for a foreground pixel, -1 for background; roots only ever move to SMALLER indices, so the
// final root of a component is its first pixel in raster order.  blockIdx.z selects an independent image.
__device__ __forceinline__ int cc_find(const int *label, int x) {
    int p;
    while ((p = __atomic_load_n(&label[x], __ATOMIC_RELAXED)) != x) x = p;
    return x;
}
__device__ __forceinline__ void cc_union(int *label, int a, int b) {
    while (true) {
        a = cc_find(label, a);
        b = cc_find(label, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&label[a], b);                      // hang the larger root under the smaller one
        if (old == a) return;
        a = old;                                                // a was re-parented meanwhile: merge its new parent with b
    }
}
// A wavefront is one 64-pixel row segment (block = 64 x 4).  Horizontal runs inside a segment are resolved with
// ballots instead of unions: a pixel starts out labelled with the first pixel of its run.
__global__ __launch_bounds__(TILE_W *TILE_H)
void k_cc_init(const uint8_t *__restrict__ mask, size_t pitch, int *__restrict__ label, int n, int m) {
    const int lane = threadIdx.x;
    int j = blockIdx.x * TILE_W + lane, i = blockIdx.y * TILE_H + threadIdx.y;
    const bool in = i < n && j < m;
    const bool fg = in && mask[(size_t)blockIdx.z * n * pitch + (size_t)i * pitch + j] != 0;
    const unsigned long long bits = __ballot(fg);
    if (!in) return;
    int p = i * m + j;
    int v = -1;
    if (fg) {
        unsigned long long zeros = ~bits & ((1ull << lane) - 1ull);
        int start = zeros ? 64 - __builtin_clzll(zeros) : 0;           // lane after the highest background lane below this one
        v = p - (lane - start);
    }
    label[(size_t)blockIdx.z * n * m + p] = v;
}
// Unions that are still needed: segment starts with their left neighbour, and ONE union per interval in which a
// run touches a run of the row above (the other pixels of the interval are connected through the runs).
template <int CONN>
__global__ __launch_bounds__(TILE_W *TILE_H)
void k_cc_merge(int *__restrict__ label, int n, int m) {
    const int lane = threadIdx.x;
    int j = blockIdx.x * TILE_W + lane, i = blockIdx.y * TILE_H + threadIdx.y;
    int *L = label + (size_t)blockIdx.z * n * m;
    const bool in = i < n && j < m;
    const int p = i * m + j;
    const bool fg = in && L[p] >= 0;
    const bool upfg = in && i > 0 && L[p - m] >= 0;
    const unsigned long long bits = __ballot(fg), upbits = __ballot(upfg);
    if (!fg) return;
    const bool left = lane > 0 ? ((bits >> (lane - 1)) & 1ull) != 0 : false;       // left neighbour in this segment is foreground
    if (lane == 0 && j > 0 && L[p - 1] >= 0) cc_union(L, p, p - 1);
    if (i == 0) return;
    const bool ul = lane > 0 ? ((upbits >> (lane - 1)) & 1ull) != 0 : (j > 0 && L[p - m - 1] >= 0);
    if (upfg) {
        if (lane == 0 || !left || !ul) cc_union(L, p, p - m);
    } else if (CONN == 8) {
        const bool right = lane < 63 ? ((bits >> (lane + 1)) & 1ull) != 0 : false;
        const bool ur = lane < 63 ? ((upbits >> (lane + 1)) & 1ull) != 0 : (j < m - 1 && L[p - m + 1] >= 0);
        if (ul && !left) cc_union(L, p, p - m - 1);
        if (ur && !(right && lane < 63)) cc_union(L, p, p - m + 1);
    }
}
// sizes (may be null): sizes[root] = number of pixels of the component, 0 elsewhere (cleared by the caller);
// lanes of a wave that share a root add their count with one atomic
__global__ __launch_bounds__(TILE_W *TILE_H)
void k_cc_flatten(int *__restrict__ label, int *__restrict__ sizes, int n, int m) {
    const int lane = threadIdx.x;
    int j = blockIdx.x * TILE_W + lane, i = blockIdx.y * TILE_H + threadIdx.y;
    int *L = label + (size_t)blockIdx.z * n * m;
    const bool in = i < n && j < m;
    const int p = i * m + j;
    const bool fg = in && L[p] >= 0;
    int root = -1;
    if (fg) {
        root = cc_find(L, p);
        __atomic_store_n(&L[p], root, __ATOMIC_RELAXED);
    }
    if (!sizes) return;
    unsigned long long remaining = __ballot(fg);
    while (remaining) {                                               // wave-uniform loop: one round per distinct root
        const int leader = __builtin_ctzll(remaining);
        const int r0 = __shfl(root, leader);
        const unsigned long long same = __ballot(fg && root == r0);
        if (lane == leader) atomicAdd(&sizes[(size_t)blockIdx.z * n * m + r0], (int)__builtin_popcountll(same));
        remaining &= ~same;
    }
}

// ---- Canny stage 2: hysteresis = components of (weak or strong) pixels that contain a strong one ------------
__global__ void k_hyst_flag(const uint8_t *__restrict__ map, const int *__restrict__ label, uint8_t *__restrict__ flag, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t base = (size_t)blockIdx.z * n * m, p = (size_t)i * m + j;
    if (map[base + p] == 2) flag[base + label[base + p]] = 1;
}
__global__ void k_hyst_out(const uint8_t *__restrict__ map, const int *__restrict__ label, const uint8_t *__restrict__ flag,
                           uint8_t *__restrict__ edges, size_t pitch, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t p = (size_t)i * m + j;
    edges[(size_t)i * pitch + j] = (map[p] && flag[label[p]]) ? 255 : 0;
}

// ---- the reference's own small kernels ------------------------------------------------------------------------
__global__ void k_cpy_n(uint8_t *__restrict__ mat, size_t pitch, const float *__restrict__ NMap, int n, int m, int chnl) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    mat[(size_t)i * pitch + j] = normal_to_byte(NMap[3 * ((size_t)i * m + j) + chnl]);
}
__global__ void k_sum(uint8_t *__restrict__ out, size_t op, const uint8_t *__restrict__ r, size_t rp, const uint8_t *__restrict__ g,
                      size_t gp, const uint8_t *__restrict__ b, size_t bp, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    out[(size_t)i * op + j] = edge_mean(r[(size_t)i * rp + j], g[(size_t)i * gp + j], b[(size_t)i * bp + j]);
}
__global__ void k_discont(uint8_t *__restrict__ mat, size_t pitch, const float *__restrict__ VMap, int n, int m, float thresh) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    if (depth_step(VMap, i, j, n, m, thresh)) mat[(size_t)i * pitch + j] = 255;
}
__global__ void k_thresh(uint8_t *__restrict__ out, size_t op, const uint8_t *__restrict__ in, size_t ip, uint8_t thresh,
                         const uint8_t *__restrict__ Label, int l16, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    const size_t p = (size_t)i * m + j;
    const bool labelled = l16 ? ((const uint16_t *)Label)[p] > 0 : Label[p] > 0;   // rgbdslam_set_label_width
    out[(size_t)i * op + j] = (in[(size_t)i * ip + j] > thresh || labelled) ? 0 : 255;
}
// cv::erode with getStructuringElement(MORPH_ELLIPSE, 5x5): rows 00100 / 11111 / 11111 / 11111 / 00100; pixels
// outside the image do not constrain the minimum
__global__ void k_erode5(uint8_t *__restrict__ dst, size_t dp, const uint8_t *__restrict__ src, size_t sp, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    int v = 255;
#pragma unroll
    for (int a = -2; a <= 2; a++) {
        const int half = (a == -2 || a == 2) ? 0 : 2;
        int ii = i + a;
        if (ii < 0 || ii >= n) continue;
        for (int b = -half; b <= half; b++) {
            int jj = j + b;
            if (jj < 0 || jj >= m) continue;
            v = min(v, (int)src[(size_t)ii * sp + jj]);
        }
    }
    dst[(size_t)i * dp + j] = (uint8_t)v;
}
// fused: hysteresis output of the three channels -> mean -> depth step -> threshold (RG.cpp:304-315); img2 only
__global__ void k_edge_binary(const uint8_t *__restrict__ map, const int *__restrict__ label, const uint8_t *__restrict__ flag,
                              const float *__restrict__ VMap, const uint8_t *__restrict__ Label, uint8_t *__restrict__ img2,
                              float thresh_discont, uint8_t thresh_edge, int l16, int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    const size_t P = (size_t)n * m, p = (size_t)i * m + j;
    uint8_t e[3];
#pragma unroll
    for (int c = 0; c < 3; c++) e[c] = (map[c * P + p] && flag[c * P + label[c * P + p]]) ? 255 : 0;
    uint8_t v = edge_mean(e[0], e[1], e[2]);
    if (depth_step(VMap, i, j, n, m, thresh_discont)) v = 255;
    const bool labelled = Label && (l16 ? ((const uint16_t *)Label)[p] > 0 : Label[p] > 0);
    img2[p] = (v > thresh_edge || labelled) ? 0 : 255;
}

// =================================================================================================
// C ABI
// =================================================================================================
static inline dim3 grid3(int n, int m, int z) { dim3 g = grid2d(n, m); g.z = z; return g; }
static void canny_thresholds(double low, double high, int *lo, int *hi) {
    if (low > high) { double t = low; low = high; high = t; }                      // OpenCV swaps them
    *lo = (int)floor(low); *hi = (int)floor(high);
}

extern "C" int rgbdslam_gpu_cpy_n(rgbdslam_ctx *ctx, uint8_t *mat, size_t pitch, const float *NMap, int n, int m, int chnl) {
    if (!ctx || !mat || !NMap || n <= 0 || m <= 0 || chnl < 0 || chnl > 2 || pitch < (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_cpy_n, grid2d(n, m), block2d(), mat, pitch, NMap, n, m, chnl);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int run_cc(rgbdslam_ctx *ctx, const uint8_t *mask, size_t pitch, int *label, int *sizes, int n, int m, int images, int conn) {
    KLAUNCH(ctx, k_cc_init, grid3(n, m, images), block2d(), mask, pitch, label, n, m);
    if (conn == 8) { KLAUNCH(ctx, k_cc_merge<8>, grid3(n, m, images), block2d(), label, n, m); }
    else { KLAUNCH(ctx, k_cc_merge<4>, grid3(n, m, images), block2d(), label, n, m); }
    KLAUNCH(ctx, k_cc_flatten, grid3(n, m, images), block2d(), label, sizes, n, m);
    return 0;
}

extern "C" int rgbdslam_canny(rgbdslam_ctx *ctx, const uint8_t *img, size_t img_pitch, uint8_t *edges, size_t edges_pitch,
                              double low_thresh, double high_thresh, int32_t *work_labels, uint8_t *work_bytes, int n, int m) {
    if (!ctx || !img || !edges || !work_labels || !work_bytes || n <= 0 || m <= 0 || img_pitch < (size_t)m || edges_pitch < (size_t)m)
        return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFFFu) return RGBDSLAM_E_LIMIT;
    int lo, hi;
    canny_thresholds(low_thresh, high_thresh, &lo, &hi);
    const size_t P = (size_t)n * m;
    uint8_t *map = work_bytes, *flag = work_bytes + P;
    HIP_TRY(hipMemsetAsync(flag, 0, P, ctx->stream));
    KLAUNCH(ctx, k_canny_map, grid2d(n, m), block2d(), img, img_pitch, map, lo, hi, n, m);
    run_cc(ctx, map, (size_t)m, work_labels, nullptr, n, m, 1, 8);
    KLAUNCH(ctx, k_hyst_flag, grid2d(n, m), block2d(), (const uint8_t *)map, (const int *)work_labels, flag, n, m);
    KLAUNCH(ctx, k_hyst_out, grid2d(n, m), block2d(), (const uint8_t *)map, (const int *)work_labels, (const uint8_t *)flag, edges,
            edges_pitch, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_gpu_sum(rgbdslam_ctx *ctx, uint8_t *out, size_t out_pitch, const uint8_t *inR, size_t r_pitch,
                                const uint8_t *inG, size_t g_pitch, const uint8_t *inB, size_t b_pitch, int n, int m) {
    if (!ctx || !out || !inR || !inG || !inB || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (out_pitch < (size_t)m || r_pitch < (size_t)m || g_pitch < (size_t)m || b_pitch < (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_sum, grid2d(n, m), block2d(), out, out_pitch, inR, r_pitch, inG, g_pitch, inB, b_pitch, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_gpu_discont(rgbdslam_ctx *ctx, uint8_t *mat, size_t pitch, const float *VMap, int n, int m, float thresh) {
    if (!ctx || !mat || !VMap || n <= 0 || m <= 0 || pitch < (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_discont, grid2d(n, m), block2d(), mat, pitch, VMap, n, m, thresh);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_gpu_thresh(rgbdslam_ctx *ctx, uint8_t *out, size_t out_pitch, const uint8_t *in, size_t in_pitch,
                                   uint8_t thresh, const uint8_t *Label, int n, int m) {
    if (!ctx || !out || !in || !Label || n <= 0 || m <= 0 || out_pitch < (size_t)m || in_pitch < (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_thresh, grid2d(n, m), block2d(), out, out_pitch, in, in_pitch, thresh, Label, labels16(ctx) ? 1 : 0, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_erode_ellipse5(rgbdslam_ctx *ctx, uint8_t *dst, size_t dst_pitch, const uint8_t *src, size_t src_pitch,
                                       int n, int m) {
    if (!ctx || !dst || !src || dst == src || n <= 0 || m <= 0 || dst_pitch < (size_t)m || src_pitch < (size_t)m) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_erode5, grid2d(n, m), block2d(), dst, dst_pitch, src, src_pitch, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_find_blobs(rgbdslam_ctx *ctx, const uint8_t *binary, size_t pitch, int32_t *blob, int32_t *sizes, int n,
                                   int m) {
    if (!ctx || !binary || !blob || !sizes || n <= 0 || m <= 0 || pitch < (size_t)m) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFFFu) return RGBDSLAM_E_LIMIT;
    HIP_TRY(hipMemsetAsync(sizes, 0, sizeof(int32_t) * (size_t)n * m, ctx->stream));
    run_cc(ctx, binary, pitch, blob, sizes, n, m, 1, 4);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_detect_regions(rgbdslam_ctx *ctx, const float *NMap, const float *VMap, const uint8_t *Label,
                                       double canny_low, double canny_high, float thresh_discont, uint8_t thresh_edge,
                                       int32_t *blob, int32_t *sizes, int32_t *work_labels, uint8_t *work_bytes, int n, int m) {
    if (!ctx || !NMap || !VMap || !blob || !sizes || !work_labels || !work_bytes || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFFFu / 3) return RGBDSLAM_E_LIMIT;
    int lo, hi;
    canny_thresholds(canny_low, canny_high, &lo, &hi);
    const size_t P = (size_t)n * m;
    uint8_t *map = work_bytes, *flag = work_bytes + 3 * P, *img2 = work_bytes + 6 * P, *binary = work_bytes + 7 * P;
    HIP_TRY(hipMemsetAsync(flag, 0, 3 * P, ctx->stream));
    HIP_TRY(hipMemsetAsync(sizes, 0, sizeof(int32_t) * P, ctx->stream));
    KLAUNCH(ctx, k_canny_map3, grid3(n, m, 3), block2d(), NMap, map, lo, hi, n, m);
    run_cc(ctx, map, (size_t)m, work_labels, nullptr, n, m, 3, 8);
    KLAUNCH(ctx, k_hyst_flag, grid3(n, m, 3), block2d(), (const uint8_t *)map, (const int *)work_labels, flag, n, m);
    KLAUNCH(ctx, k_edge_binary, grid2d(n, m), block2d(), (const uint8_t *)map, (const int *)work_labels, (const uint8_t *)flag, VMap,
            Label, img2, thresh_discont, thresh_edge, labels16(ctx) ? 1 : 0, n, m);
    KLAUNCH(ctx, k_erode5, grid2d(n, m), block2d(), binary, (size_t)m, (const uint8_t *)img2, (size_t)m, n, m);
    run_cc(ctx, binary, (size_t)m, blob, sizes, n, m, 1, 4);
    HIP_TRY(hipGetLastError());
    return 0;
}
