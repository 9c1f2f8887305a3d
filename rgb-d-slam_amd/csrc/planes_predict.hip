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
#define PD_TPW 4                       // tiles (of 64 quads, side by side) per workgroup: one wave scans the masks of one
struct PlaneTabs {
    const uint16_t *const *Bump; const uint8_t *const *RGB; const uint8_t *const *Mask; const float *const *Param;
};

__device__ __forceinline__ bool pd_corner_rejected(uint8_t mk, int min_conf) { return mk != 1 && (int)mk - 10 < min_conf; }
// any of the four mask bytes of w a usable corner?  (!pd_corner_rejected: mk == 1 or mk >= min_conf + 10 = thr)
__device__ __forceinline__ int pd_any_accepted4(uint32_t w, int thr) {
    int a = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int b = (int)((w >> (8 * k)) & 255u);
        a |= (int)(b == 1) | (int)(b >= thr);
    }
    return a;
}

// One workgroup = PD_TPW tiles of 64 x PD_ROWS quads of one plane.
// Phase 1, the part that decides the kernel's time (99.5 % of the tiles of a 256-plane model are empty in a given view; every
// mask byte has to be looked at, 67 MB for 256 planes x 512^2): wave w reads the (PD_ROWS+1) x 65 mask bytes of tile w as ONE
// load per lane -- 36 lanes a 16-byte piece of a row each, 9 lanes the 65th byte of a row -- and a ballot says whether the tile
// holds a usable corner at all.  A workgroup whose four tiles are empty leaves after one load, one ballot and one barrier,
// before it has touched the plane's parameters or any LDS tile.  (Round 2: one workgroup per tile, 585 byte loads through a
// division by 65, two barriers: 226 us for this model.)  Rows that are not 16-byte aligned (M_prim % 16 != 0 or an unaligned
// slab) take byte loads.
// Phase 2, per non-empty tile, all four waves together: the (PD_ROWS+1) x 65 cells are decoded ONCE into LDS (world-space
// vertex, camera-space projection, mask), then one lane per quad trims and draws.
__global__ __launch_bounds__(256)
void k_planes_draw(unsigned long long *__restrict__ zbuf, PlaneTabs T, int plane_lo, PoseV P, CalibV c, int N, int M, int min_conf,
                   float thresh, float znear, float zfar, int n, int m) {
    __shared__ float sw[PD_ROWS + 1][65][3];       // world-space vertices (trim test, Primitive.cu:112-133)
    __shared__ RasVert sp[PD_ROWS + 1][65];        // projected camera-space vertices
    __shared__ uint8_t sm[PD_ROWS + 1][65];
    __shared__ int occ[PD_TPW];
    const int plane = plane_lo + blockIdx.z;
    const int i0 = blockIdx.y * PD_ROWS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const GLOBAL_AS uint8_t *__restrict__ Maskb = (const GLOBAL_AS uint8_t *)T.Mask[plane];
    {
        const int thr = min_conf + 10;
        const int j0 = (blockIdx.x * PD_TPW + wave) * 64;
        bool acc = false;
        if (j0 <= M - 2) {                           // the tile exists
            if (((((uintptr_t)Maskb) | (uintptr_t)(unsigned)M) & (uintptr_t)15) == 0) {
                if (lane < 4 * (PD_ROWS + 1)) {
                    const int i = i0 + (lane >> 2), j = j0 + 16 * (lane & 3);
                    if (i < N && j < M) {            // M % 16 == 0: the 16 bytes lie inside the row
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 w = *(const GLOBAL_AS u32x4 *)(Maskb + (size_t)i * M + j);
                        acc = (pd_any_accepted4(w.x, thr) | pd_any_accepted4(w.y, thr) | pd_any_accepted4(w.z, thr) | pd_any_accepted4(w.w, thr)) != 0;
                    }
                } else if (lane < 5 * (PD_ROWS + 1)) {
                    const int i = i0 + lane - 4 * (PD_ROWS + 1), j = j0 + 64;
                    if (i < N && j < M) acc = !pd_corner_rejected(Maskb[(size_t)i * M + j], min_conf);
                }
            } else {
                for (int e = lane; e < (PD_ROWS + 1) * 65; e += 64) {
                    const int i = i0 + e / 65, j = j0 + e % 65;
                    if (i < N && j < M) acc |= !pd_corner_rejected(Maskb[(size_t)i * M + j], min_conf);
                }
            }
        }
        const bool any = __ballot(acc) != 0ull;
        if (lane == 0) occ[wave] = any ? 1 : 0;
    }
    __syncthreads();
    int todo = 0;
#pragma unroll
    for (int w = 0; w < PD_TPW; w++) todo |= occ[w] << w;
    if (!todo) return;                               // four empty tiles of the Bump image: nothing to draw
    const GLOBAL_AS uint16_t *__restrict__ Bump = (const GLOBAL_AS uint16_t *)T.Bump[plane];
    const BumpParam Q = load_param(T.Param[plane]);
    const uint32_t Qn = (uint32_t)(N - 1) * (uint32_t)(M - 1);
    for (int w = 0; w < PD_TPW; w++) {
        if (!((todo >> w) & 1)) continue;            // uniform over the workgroup
        const int j0 = (blockIdx.x * PD_TPW + w) * 64;
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
    dim3 grid(((M_prim - 1 + 63) / 64 + PD_TPW - 1) / PD_TPW, (N_prim - 1 + PD_ROWS - 1) / PD_ROWS, (unsigned)(plane_hi - plane_lo));
    KLAUNCH(ctx, k_planes_draw, grid, dim3(256), (unsigned long long *)zbuf, T, plane_lo, P, ctx->calib, N_prim, M_prim, (int)min_conf,
            thresh, znear, zfar, n, m);
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
