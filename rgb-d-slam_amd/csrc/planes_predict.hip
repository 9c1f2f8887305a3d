// planes_predict.hip -- prediction ("raycast") of the model DIRECTLY FROM THE BUMP IMAGES (SURVEY 8e row 1 (iii), 8f N4;
// BASELINE configs[2] "fusion + raycast partitioned by plane").  The reference's live loop predicts from the previous
// depth image; its Bump meshes (VertexFromBump, D1/Primitive.cu:31-93, + QuadTrim, D1/Primitive.cu:95-152) are only drawn by
// DemoStep2's viewer.  Here they are drawn into the SAME u64 z-buffer, by the SAME rasteriser rules (device_fns.hpp), as the
// frame mesh of k_raster_draw: key = f32 depth bits << 32 | ~triangle id with
//     triangle id = ((plane * (N-1)*(M-1) + quad) * 2 + t),
// so the nearest surface wins whatever plane -- or GPU -- drew it: z-buffers of disjoint plane ranges merge by an
// elementwise u64 minimum (one RCCL all-reduce in the plane-partitioned multi-GPU form), and every rank resolves the
// pixels whose winning plane it owns.
#include "device_fns.hpp"

#define PD_ROWS 8                      // quad rows per tile (tile: PD_ROWS+1 x 65 cells in LDS)
#define PD_DRAW_WGS 2048               // workgroups of the draw launch (each walks the queue with that stride)
struct PlaneTabs {
    const uint16_t *const *Bump; const uint8_t *const *RGB; const uint8_t *const *Mask; const float *const *Param;
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool pd_corner_rejected(uint8_t mk, int min_conf) { return mk != 1 && (int)mk - 10 < min_conf; }
// which of the four mask bytes of w are usable corners?  (!pd_corner_rejected: mk == 1 or mk >= min_conf + 10 = thr); bit k = byte k
__device__ __forceinline__ uint32_t pd_accepted4(uint32_t w, int thr) {
    uint32_t a = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int b = (int)((w >> (8 * k)) & 255u);
        a |= (uint32_t)((b == 1) | (b >= thr)) << k;
    }
    return a;
}

// Two launches.  Round 2 drew with one workgroup per 64 x 8-quad tile: 131 072 workgroups for 256 planes x 512^2, each reading
// its 9 x 65 mask bytes one byte per lane through a division by 65 and meeting at two barriers only to find -- 99.5 % of
// the time -- that its tile is empty: 226 us.  Every mask byte has to be looked at (67 MB: no occupancy is kept between
// calls), so the first launch does just that, shaped for bandwidth, and leaves a queue of the occupied tiles; the second
// draws the queue, one tile per workgroup, however the occupied tiles are spread over the planes (in the tiled scene they
// sit in a dozen planes at the END of the grid: drawn where they were found, they ran after everything else, four to a
// workgroup one after the other).
//
// k_planes_scan: one thread = one 16-column piece of one band of PD_ROWS quad rows (PD_ROWS + 1 cell rows): nine 16-byte
// loads in flight per lane (9 KB per wave; one load per lane kept only 3.6 MB in flight on the whole chip: 1.65 TB/s).  A
// piece that holds a usable corner marks its tile, and -- through its first column, the 65th of the tile to its left --
// that one; the first to mark a tile (atomicMax of the call's stamp on the tile's word: no clearing between calls)
// appends it to the queue.  Rows that are not 16-byte aligned (M_prim % 16 != 0, or a slab at an odd address) are read
// bytewise.
__global__ __launch_bounds__(256)
void k_planes_scan(PlaneTabs T, int plane_lo, int N, int M, int min_conf, int pieces, int bands, int tiles_x,
                   uint32_t *__restrict__ flags, uint32_t *__restrict__ queue, uint32_t *__restrict__ count, uint32_t stamp) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= pieces * bands) return;
    const int band = g / pieces, pc = g % pieces;
    const int plane = plane_lo + blockIdx.y;
    const GLOBAL_AS uint8_t *__restrict__ Maskb = (const GLOBAL_AS uint8_t *)T.Mask[plane];
    const int thr = min_conf + 10;
    const int i0 = band * PD_ROWS, j0 = 16 * pc;
    uint32_t acc = 0;                                // bit c: column j0 + c holds a usable corner in one of the band's rows
    if (((((uintptr_t)Maskb) | (uintptr_t)(unsigned)M) & (uintptr_t)15) == 0) {
        u32x4 w[PD_ROWS + 1];
#pragma unroll
        for (int r = 0; r <= PD_ROWS; r++) {         // all loads first; rows below the image repeat the last one
            const int i = min(i0 + r, N - 1);
            w[r] = *(const GLOBAL_AS u32x4 *)(Maskb + (size_t)i * M + j0);
        }
#pragma unroll
        for (int r = 0; r <= PD_ROWS; r++)
            acc |= pd_accepted4(w[r].x, thr) | pd_accepted4(w[r].y, thr) << 4 | pd_accepted4(w[r].z, thr) << 8 | pd_accepted4(w[r].w, thr) << 12;
    } else {
        for (int r = 0; r <= PD_ROWS; r++) {
            const int i = i0 + r;
            if (i >= N) break;
            for (int cc = 0; cc < 16 && j0 + cc < M; cc++)
                acc |= (uint32_t)!pd_corner_rejected(Maskb[(size_t)i * M + j0 + cc], min_conf) << cc;
        }
    }
    if (!acc) return;
    const uint32_t tiles_per_plane = (uint32_t)tiles_x * (uint32_t)bands;
    const uint32_t tbase = (uint32_t)blockIdx.y * tiles_per_plane + (uint32_t)band * (uint32_t)tiles_x;
    const int tx = pc >> 2;
    if (tx < tiles_x) {                              // (the last column of an image of 64 k + 1 columns belongs to no tile of its own)
        if (atomicMax(&flags[tbase + tx], stamp) < stamp) queue[atomicAdd(count, 1u)] = tbase + tx;
    }
    if ((acc & 1u) && (pc & 3) == 0 && tx > 0) {     // column 64 of the tile to the left
        if (atomicMax(&flags[tbase + tx - 1], stamp) < stamp) queue[atomicAdd(count, 1u)] = tbase + tx - 1;
    }
}

// k_planes_draw: one queued tile at a time per workgroup.  The (PD_ROWS+1) x 65 cells of the tile are decoded ONCE into LDS
// (world-space vertex, camera-space projection, mask), then one lane per quad trims and draws.  Block 0 clears the OTHER
// call parity's counter for the next call's scan.
__global__ __launch_bounds__(256)
void k_planes_draw(unsigned long long *__restrict__ zbuf, PlaneTabs T, int plane_lo, PoseV P, CalibV c, int N, int M, int min_conf,
                   float thresh, float znear, float zfar, int n, int m, int bands, int tiles_x, const uint32_t *__restrict__ queue,
                   const uint32_t *__restrict__ count, uint32_t *__restrict__ next_count) {
    __shared__ float sw[PD_ROWS + 1][65][3];       // world-space vertices (trim test, Primitive.cu:112-133)
    __shared__ RasVert sp[PD_ROWS + 1][65];        // projected camera-space vertices
    __shared__ uint8_t sm[PD_ROWS + 1][65];
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0u;
    const uint32_t total = *count;
    const uint32_t tiles_per_plane = (uint32_t)tiles_x * (uint32_t)bands;
    const uint32_t Qn = (uint32_t)(N - 1) * (uint32_t)(M - 1);
    for (uint32_t qi = blockIdx.x; qi < total; qi += gridDim.x) {
        const uint32_t tile = queue[qi];
        const int plane = plane_lo + (int)(tile / tiles_per_plane);
        const uint32_t rem = tile % tiles_per_plane;
        const int i0 = (int)(rem / (uint32_t)tiles_x) * PD_ROWS, j0 = (int)(rem % (uint32_t)tiles_x) * 64;
        const GLOBAL_AS uint16_t *__restrict__ Bump = (const GLOBAL_AS uint16_t *)T.Bump[plane];
        const GLOBAL_AS uint8_t *__restrict__ Maskb = (const GLOBAL_AS uint8_t *)T.Mask[plane];
        const BumpParam Q = load_param(T.Param[plane]);
        for (int e = threadIdx.x; e < (PD_ROWS + 1) * 65; e += 256) {
            const int r = e / 65, cc = e % 65, i = i0 + r, j = j0 + cc;
            uint8_t mk = 0;
            float3 v; v.x = v.y = v.z = 0.0f;
            RasVert pv; pv.X = 0; pv.Y = 0; pv.inv_d = 0.0; pv.ok = false;
            if (i < N && j < M) {
                const size_t cell = (size_t)i * M + j;
                mk = Maskb[cell];
                if (!pd_corner_rejected(mk, min_conf)) {
                    BumpCell o = decode_cell(Bump[3 * cell], Bump[3 * cell + 1], Bump[3 * cell + 2], 0, 0, 0, mk, Q, (float)i, (float)j, unorm8);
                    v = o.v;
                    F3 wv; wv.x = v.x; wv.y = v.y; wv.z = v.z;
                    pv = ras_project(xf_point(P.v, wv), c);
                }
            }
            sw[r][cc][0] = v.x; sw[r][cc][1] = v.y; sw[r][cc][2] = v.z;
            sp[r][cc] = pv;
            sm[r][cc] = mk;
        }
        __syncthreads();
        const int cc = threadIdx.x & 63;
        for (int r = threadIdx.x >> 6; r < PD_ROWS; r += 4) {
            const int i = i0 + r, j = j0 + cc;
            if (i > N - 2 || j > M - 2) continue;
            // corners in QuadTrim's order: (i,j) (i,j+1) (i+1,j+1) (i+1,j)
            if (pd_corner_rejected(sm[r][cc], min_conf) || pd_corner_rejected(sm[r][cc + 1], min_conf) ||
                pd_corner_rejected(sm[r + 1][cc + 1], min_conf) || pd_corner_rejected(sm[r + 1][cc], min_conf))
                continue;
            auto lw = [&](int rr, int c2) { F3 q; q.x = sw[rr][c2][0]; q.y = sw[rr][c2][1]; q.z = sw[rr][c2][2]; return q; };
            if (quad_trimmed(lw(r, cc), lw(r, cc + 1), lw(r + 1, cc + 1), lw(r + 1, cc), thresh)) continue;
            const uint32_t q = (uint32_t)i * (uint32_t)(M - 1) + (uint32_t)j;
            const uint32_t base = ((uint32_t)plane * Qn + q) * 2u;
            const RasVert p0 = sp[r][cc], p1 = sp[r][cc + 1], p2 = sp[r + 1][cc + 1], p3 = sp[r + 1][cc];
            RasTri Tr;
            if (ras_setup_projected(p0, p1, p2, Tr)) ras_draw_tri(zbuf, Tr, base, znear, zfar, n, m);
            if (ras_setup_projected(p0, p2, p3, Tr)) ras_draw_tri(zbuf, Tr, base + 1, znear, zfar, n, m);
        }
        __syncthreads();                             // the LDS tile is reused by the workgroup's next tile
    }
}

// One lane per pixel: winner's three cells decoded again (vertex + colour), shaded perspective-correctly, then the
// ReadKernel arithmetic (Kernel.cu:309-343).  Pixels won by a plane outside [plane_lo, plane_hi) are left alone.
__global__ void k_planes_resolve(unsigned long long *__restrict__ zbuf, float *__restrict__ VMap, float *__restrict__ RGBD, PlaneTabs T,
                                 int plane_lo, int plane_hi, PoseV P, CalibV c, int N, int M, int reset, int n, int m) {
    int px = blockIdx.x * TILE_W + threadIdx.x, py = blockIdx.y * TILE_H + threadIdx.y;
    if (py >= n || px >= m) return;
    const size_t p = (size_t)py * m + px;
    const unsigned long long key = zbuf[p];
    float4 o; o.x = o.y = o.z = o.w = 0.0f;
    if (key != ZBUF_EMPTY) {
        const uint32_t Qn = (uint32_t)(N - 1) * (uint32_t)(M - 1);
        const uint32_t tri_id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
        const uint32_t plane = (tri_id >> 1) / Qn, q = (tri_id >> 1) % Qn;
        if ((int)plane < plane_lo || (int)plane >= plane_hi) return;
        if (reset) zbuf[p] = ZBUF_EMPTY;
        const int i = (int)(q / (uint32_t)(M - 1)), j = (int)(q % (uint32_t)(M - 1));
        const uint16_t *__restrict__ Bump = T.Bump[plane];
        const uint8_t *__restrict__ RGBb = T.RGB[plane];
        const uint8_t *__restrict__ Maskb = T.Mask[plane];
        const BumpParam Q = load_param(T.Param[plane]);
        F3 vtx[3];
        float4 att[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            // t = 0: corners 0,1,2; t = 1: corners 0,2,3 of (i,j) (i,j+1) (i+1,j+1) (i+1,j)
            const int cn = (e == 0) ? 0 : ((tri_id & 1) ? e + 1 : e);
            const int ci = i + (cn >= 2 ? 1 : 0), cj = j + ((cn == 1 || cn == 2) ? 1 : 0);
            const size_t cell = (size_t)ci * M + cj;
            BumpCell d = decode_cell(Bump[3 * cell], Bump[3 * cell + 1], Bump[3 * cell + 2], RGBb[3 * cell], RGBb[3 * cell + 1],
                                     RGBb[3 * cell + 2], Maskb[cell], Q, (float)ci, (float)cj, unorm8);
            F3 w; w.x = d.v.x; w.y = d.v.y; w.z = d.v.z;
            vtx[e] = xf_point(P.v, w);
            att[e].x = d.c.x; att[e].y = d.c.y; att[e].z = d.c.z; att[e].w = depth_attr(vtx[e].z);
        }
        o = ras_shade(vtx, att, c, px, py);
    } else if (plane_lo > 0) {
        return;                                       // empty pixels are written by the owner of plane 0 only
    }
    if (o.w == 0.0f) {
        st3(VMap, p, 0.0f, 0.0f, 0.0f);
        float4 z; z.x = z.y = z.z = z.w = 0.0f;
        *(float4 *)(RGBD + 4 * p) = z;
        return;
    }
    F3 v = vertex_from_texel(o.w, py, px, c);
    st3(VMap, p, v.x, v.y, v.z);
    *(float4 *)(RGBD + 4 * p) = o;
}

// =================================================================================================
// C ABI
// =================================================================================================
static int planes_args_ok(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, int plane_lo,
                          int plane_hi, const float *pose, int N_prim, int M_prim, int n, int m) {
    if (!ctx || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !pose || n < 1 || m < 1 || N_prim < 2 || M_prim < 2) return RGBDSLAM_E_ARG;
    if (plane_lo < 0 || plane_hi < plane_lo) return RGBDSLAM_E_ARG;
    // 32-bit triangle ids over ALL planes up to plane_hi
    if ((double)plane_hi * (double)(N_prim - 1) * (double)(M_prim - 1) * 2.0 >= 4294967295.0 || plane_hi - plane_lo > 65535) return RGBDSLAM_E_LIMIT;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    return 0;
}

extern "C" int rgbdslam_planes_draw(rgbdslam_ctx *ctx, uint64_t *zbuf, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab,
                                    float **ParamTab, int plane_lo, int plane_hi, const float *pose_world_to_cam16_host, int N_prim,
                                    int M_prim, unsigned short min_conf, float thresh, float znear, float zfar, int n, int m) {
    int rc = planes_args_ok(ctx, BumpTab, RGBTab, MaskTab, ParamTab, plane_lo, plane_hi, pose_world_to_cam16_host, N_prim, M_prim, n, m);
    if (rc) return rc;
    if (!zbuf || min_conf < 1) return RGBDSLAM_E_ARG;         // min_conf 0 would let EMPTY cells (mask 10) into the mesh
    if (plane_hi == plane_lo) return 0;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = pose_world_to_cam16_host[k];
    PlaneTabs T = {(const uint16_t *const *)BumpTab, (const uint8_t *const *)RGBTab, (const uint8_t *const *)MaskTab, (const float *const *)ParamTab};
    // scratch owned by the context (grown on demand; never shrinks): one stamp word per tile, the queue of occupied tiles,
    // two counters (call parity)
    const int planes = plane_hi - plane_lo;
    const int tiles_x = (M_prim - 1 + 63) / 64, bands = (N_prim - 1 + PD_ROWS - 1) / PD_ROWS, pieces = (M_prim + 15) / 16;
    const size_t tiles = (size_t)planes * tiles_x * bands;
    if (tiles > 0x7FFFFFFFull || (size_t)pieces * bands > 0x7FFFFFFFull) return RGBDSLAM_E_LIMIT;
    if (tiles > ctx->pd_cap) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->pd_scratch) { (void)hipFree(ctx->pd_scratch); ctx->pd_scratch = nullptr; ctx->pd_cap = 0; }
        void *ptr = nullptr;
        HIP_TRY(hipMalloc(&ptr, 256 + 8 * tiles));
        HIP_TRY(hipMemsetAsync(ptr, 0, 256 + 8 * tiles, ctx->stream));
        ctx->pd_scratch = (uint32_t *)ptr; ctx->pd_cap = tiles; ctx->pd_stamp = 0;
    }
    if (++ctx->pd_stamp == 0u) {                       // the stamp wrapped: the tile words start again
        HIP_TRY(hipMemsetAsync(ctx->pd_scratch + 64, 0, 4 * ctx->pd_cap, ctx->stream));
        ctx->pd_stamp = 1u;
    }
    uint32_t *cnt = ctx->pd_scratch + 32 * (ctx->pd_stamp & 1u), *cnt_next = ctx->pd_scratch + 32 * ((ctx->pd_stamp + 1u) & 1u);
    uint32_t *flags = ctx->pd_scratch + 64, *queue = flags + ctx->pd_cap;
    KLAUNCH(ctx, k_planes_scan, dim3((unsigned)(((size_t)pieces * bands + 255) / 256), (unsigned)planes), dim3(256), T, plane_lo, N_prim, M_prim,
            (int)min_conf, pieces, bands, tiles_x, flags, queue, cnt, ctx->pd_stamp);
    KLAUNCH(ctx, k_planes_draw, dim3((unsigned)(tiles < PD_DRAW_WGS ? tiles : PD_DRAW_WGS)), dim3(256), (unsigned long long *)zbuf, T, plane_lo, P,
            ctx->calib, N_prim, M_prim, (int)min_conf, thresh, znear, zfar, n, m, bands, tiles_x, queue, cnt, cnt_next);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_planes_resolve(rgbdslam_ctx *ctx, uint64_t *zbuf, float *VMap, float *RGBD, uint16_t **BumpTab, uint8_t **RGBTab,
                                       uint8_t **MaskTab, float **ParamTab, int plane_lo, int plane_hi,
                                       const float *pose_world_to_cam16_host, int N_prim, int M_prim, int reset_zbuf, int n, int m) {
    int rc = planes_args_ok(ctx, BumpTab, RGBTab, MaskTab, ParamTab, plane_lo, plane_hi, pose_world_to_cam16_host, N_prim, M_prim, n, m);
    if (rc) return rc;
    if (!zbuf || !VMap || !RGBD) return RGBDSLAM_E_ARG;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = pose_world_to_cam16_host[k];
    PlaneTabs T = {(const uint16_t *const *)BumpTab, (const uint8_t *const *)RGBTab, (const uint8_t *const *)MaskTab, (const float *const *)ParamTab};
    KLAUNCH(ctx, k_planes_resolve, grid2d(n, m), block2d(), (unsigned long long *)zbuf, VMap, RGBD, T, plane_lo, plane_hi, P, ctx->calib,
            N_prim, M_prim, reset_zbuf ? 1 : 0, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
