// fusion.hip -- stage 4: multi-plane Bump-image fusion (scatter) and the Bump -> vertex decode.
#include "common.hpp"

// Winner rule replacing the data race of UpdateKernel (D1/Primitive.cu:309-349), see DESIGN.md:
// all threads decide against the SNAPSHOT of the cell mask taken before any write; a gate-passing
// writer beats an eroder; among writers the highest mask_curr wins, ties go to the lowest frame
// pixel index; with only eroders on a cell a single erode step is applied.
// key = stamp(32) | writer(1) | mask_curr(8) | ~idx(23)
__device__ __forceinline__ unsigned long long claim_key(uint32_t stamp, bool writer, uint8_t mc, uint32_t idx) {
    uint32_t lo = ((uint32_t)(writer ? 1u : 0u) << 31) | ((uint32_t)mc << 23) | (0x7FFFFFu - idx);
    return ((unsigned long long)stamp << 32) | lo;
}

struct UpdCell { int L, cell; float sa, sb, err; };

// geometry of D1/Primitive.cu:278-307
__device__ __forceinline__ bool update_cell(const float *__restrict__ VMap, const uint8_t *__restrict__ Label,
                                            const PoseV &P, float *const *__restrict__ ParamTab, size_t idx, int N_prim,
                                            int M_prim, int plane_lo, int plane_hi, UpdCell &o) {
    int L = (int)Label[idx] - 1;
    if (L < 0 || L < plane_lo || L >= plane_hi) return false;
    const float *Q = ParamTab[L];
    F3 p = xf_point(P.v, ld3(VMap, idx));
    o.err = ((p.x * Q[0] + p.y * Q[1]) + p.z * Q[2]) - Q[3];
    float a = (p.x * Q[8] + p.y * Q[9]) + p.z * Q[10];
    float b = (p.x * Q[11] + p.y * Q[12]) + p.z * Q[13];
    a = (float)((double)((a - Q[6]) * Q[4]) / 2.0);
    b = (float)((double)((b - Q[7]) * Q[5]) / 2.0);
    int ii = f2i_rd(a), jj = f2i_rd(b);
    if (ii > N_prim - 1 || jj > M_prim - 1 || ii < 0 || jj < 0) return false;
    o.L = L; o.cell = ii * M_prim + jj; o.sa = a - (float)ii; o.sb = b - (float)jj;
    return true;
}

template <bool APPLY>
__global__ void k_update(uint16_t *const *__restrict__ BumpTab, uint8_t *const *__restrict__ RGBTab,
                         uint8_t *const *__restrict__ MaskTab, float *const *__restrict__ ParamTab,
                         unsigned long long *const *__restrict__ ClaimTab, uint32_t stamp,
                         const float *__restrict__ VMap, const float *__restrict__ RGBD, const uint8_t *__restrict__ Mask,
                         const uint8_t *__restrict__ Label, PoseV P, const rgbdslam_pose_block *__restrict__ pb, int N_prim,
                         int M_prim, int n, int m, int plane_lo, int plane_hi) {
    if (pb) {                                         // fused pipeline: pose from the device block, gated on it
        if (pb->status[0] != 1) return;
#pragma unroll
        for (int k = 0; k < 16; k++) P.v[k] = pb->P_glob[k];
    }
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    size_t idx = (size_t)i * m + j;
    UpdCell u;
    if (!update_cell(VMap, Label, P, ParamTab, idx, N_prim, M_prim, plane_lo, plane_hi, u)) return;
    uint8_t mc = Mask[idx];
    uint8_t *cellmask = MaskTab[u.L] + u.cell;
    uint8_t mref = *cellmask;
    bool erode = (mc < 10 && mref != 1);                                      // Primitive.cu:313
    if (!erode && !(mc > mref)) return;                                       // Primitive.cu:341
    unsigned long long key = claim_key(stamp, !erode, erode ? (uint8_t)0 : mc, (uint32_t)idx);
    unsigned long long *slot = ClaimTab[u.L] + u.cell;
    if (!APPLY) { atomicMax(slot, key); return; }
    if (*slot != key) return;
    uint16_t *B = BumpTab[u.L] + 3 * (size_t)u.cell;
    uint8_t *C = RGBTab[u.L] + 3 * (size_t)u.cell;
    if (erode) {
        if (mref > 11) { *cellmask = (uint8_t)(mref - 1); }
        else { *cellmask = 10; B[0] = 0; B[1] = 0; B[2] = 0; C[0] = 0; C[1] = 0; C[2] = 0; }
    } else {
        float4 col = *(const float4 *)(RGBD + 4 * idx);
        B[0] = (uint16_t)f2i_rn((float)((double)u.sa * 60000.0));             // Primitive.cu:342-344
        B[1] = (uint16_t)f2i_rn((float)((double)u.sb * 60000.0));
        B[2] = (uint16_t)f2i_rn((float)(((double)u.err + 15.0) * 2000.0));
        C[0] = (uint8_t)f2i_rn((float)((double)col.x * 255.0));
        C[1] = (uint8_t)f2i_rn((float)((double)col.y * 255.0));
        C[2] = (uint8_t)f2i_rn((float)((double)col.z * 255.0));
        *cellmask = mc;
    }
}

// ---- D1/Primitive.cu:31-93 VertexFromBumpKernel ----------------------------------------------------
struct BumpCell { float3 v; float4 c, k; };
__device__ __forceinline__ BumpCell decode_cell(uint16_t b0, uint16_t b1, uint16_t b2, uint8_t r, uint8_t g, uint8_t b, uint8_t mk,
                                                const float *__restrict__ param, float fi, float fj) {
    BumpCell o;
    if (mk < 11 && mk != 1) {                                   // Primitive.cu:47
        o.v.x = o.v.y = o.v.z = 0.0f;
        o.c.x = o.c.y = o.c.z = o.c.w = 0.0f;
        o.k = o.c;
        return o;
    }
    // (float)((double)x / C) == x / (float)C for float x: double rounding is innocuous for division
    float s0 = (float)b0 / 60000.0f;
    float s1 = (float)b1 / 60000.0f;
    float x = (float)((double)(fi + s0) * 2.0 / (double)param[4] + (double)param[6]);
    float y = (float)((double)(fj + s1) * 2.0 / (double)param[5] + (double)param[7]);
    float d = (float)((double)param[3] + (((double)(float)b2 / 2000.0) - 15.0));
    o.v.x = (x * param[8] + y * param[11]) + d * param[0];
    o.v.y = (x * param[9] + y * param[12]) + d * param[1];
    o.v.z = (x * param[10] + y * param[13]) + d * param[2];
    o.c.x = (float)r / 255.0f;
    o.c.y = (float)g / 255.0f;
    o.c.z = (float)b / 255.0f;
    o.c.w = 1.0f;
    o.k.x = (float)((10.0 + (double)o.v.x) / 20.0);
    o.k.y = (float)((10.0 + (double)o.v.y) / 20.0);
    o.k.z = (float)((10.0 + (double)o.v.z) / 20.0);
    o.k.w = (float)mk / 255.0f;
    return o;
}

__global__ void k_vertex_from_bump(float *__restrict__ VMap, float *__restrict__ RGBA, float *__restrict__ MaskOut,
                                   const uint16_t *__restrict__ Bump, const uint8_t *__restrict__ RGB_bump,
                                   const uint8_t *__restrict__ Mask, const float *__restrict__ param, int n, int m, int lvl) {
    int no = n / lvl, mo = m / lvl;
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= no || j >= mo) return;
    size_t idx = (size_t)i * mo + j;
    size_t ref = (size_t)(lvl * i) * m + (size_t)(lvl * j);
    BumpCell o = decode_cell(Bump[3 * ref], Bump[3 * ref + 1], Bump[3 * ref + 2], RGB_bump[3 * ref], RGB_bump[3 * ref + 1],
                             RGB_bump[3 * ref + 2], Mask[ref], param, (float)(lvl * i), (float)(lvl * j));
    st3(VMap, idx, o.v.x, o.v.y, o.v.z);
    *(float4 *)(RGBA + 4 * idx) = o.c;
    *(float4 *)(MaskOut + 4 * idx) = o.k;
}

// Whole-model decode in ONE launch (config 3: hundreds of planes): blockIdx.z = plane, each lane
// decodes 4 consecutive cells of a row, so every access is an aligned dword vector -- 24 B of Bump,
// 12 B of RGB, 4 B of Mask in; 48 + 64 + 64 B out -- and a wavefront streams contiguous memory.
// Needs lvl == 1 and m % 4 == 0.
#define DEC_CELLS 4
__global__ __launch_bounds__(256)
void k_vertex_from_bump_batch(float *const *__restrict__ VMapTab, float *const *__restrict__ RGBATab,
                              float *const *__restrict__ MaskOutTab, const uint16_t *const *__restrict__ BumpTab,
                              const uint8_t *const *__restrict__ RGBTab, const uint8_t *const *__restrict__ MaskTab,
                              const float *const *__restrict__ ParamTab, int plane_lo, int n, int m) {
    const int plane = plane_lo + blockIdx.z;
    const size_t groups = (size_t)n * m / DEC_CELLS;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const float *__restrict__ param = ParamTab[plane];
    const size_t cell0 = g * DEC_CELLS;
    const int i = (int)(cell0 / m), j0 = (int)(cell0 % m);
    const uint4 bq = *(const uint4 *)(BumpTab[plane] + 3 * cell0);            // 8 of the 12 u16
    const uint2 bq2 = *(const uint2 *)(BumpTab[plane] + 3 * cell0 + 8);       // the other 4
    const uint32_t bw[6] = {bq.x, bq.y, bq.z, bq.w, bq2.x, bq2.y};
    const uint32_t *cp = (const uint32_t *)(RGBTab[plane] + 3 * cell0);
    const uint32_t cw[3] = {cp[0], cp[1], cp[2]};
    const uint32_t mw = *(const uint32_t *)(MaskTab[plane] + cell0);
    float vo[12];
    float4 co[4], ko[4];
#pragma unroll
    for (int k = 0; k < DEC_CELLS; k++) {
        // u16 #(3k+c) sits in word (3k+c)/2, half (3k+c)%2; byte #(3k+c) in word (3k+c)/4, byte (3k+c)%4
        uint16_t b[3];
        uint8_t c[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            int e = 3 * k + q;
            b[q] = (uint16_t)(bw[e >> 1] >> (16 * (e & 1)));
            c[q] = (uint8_t)(cw[e >> 2] >> (8 * (e & 3)));
        }
        BumpCell o = decode_cell(b[0], b[1], b[2], c[0], c[1], c[2], (uint8_t)(mw >> (8 * k)), param, (float)i, (float)(j0 + k));
        vo[3 * k] = o.v.x; vo[3 * k + 1] = o.v.y; vo[3 * k + 2] = o.v.z;
        co[k] = o.c; ko[k] = o.k;
    }
    float4 *vp = (float4 *)(VMapTab[plane] + 3 * cell0);
#pragma unroll
    for (int k = 0; k < 3; k++) { float4 t; t.x = vo[4 * k]; t.y = vo[4 * k + 1]; t.z = vo[4 * k + 2]; t.w = vo[4 * k + 3]; vp[k] = t; }
    float4 *cpo = (float4 *)(RGBATab[plane] + 4 * cell0), *kpo = (float4 *)(MaskOutTab[plane] + 4 * cell0);
#pragma unroll
    for (int k = 0; k < DEC_CELLS; k++) { cpo[k] = co[k]; kpo[k] = ko[k]; }
}

// =================================================================================================
// C ABI
// =================================================================================================
static int launch_update(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pb, uint16_t **BumpTab, uint8_t **RGBTab,
                         uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp, const float *VMap,
                         const float *RGBD, const uint8_t *Mask, const uint8_t *Label, const float *pose, int N_prim,
                         int M_prim, int n, int m, int plane_lo, int plane_hi) {
    if (!ctx || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !ClaimTab || !VMap || !RGBD || !Mask || !Label || (!pose && !pb))
        return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0 || stamp == 0) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;   // 23-bit pixel index in the claim key
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = pose ? pose[k] : 0.0f;
    KLAUNCH(ctx, k_update<false>, grid2d(n, m), block2d(), BumpTab, RGBTab, MaskTab, ParamTab,
            (unsigned long long *const *)ClaimTab, stamp, VMap, RGBD, Mask, Label, P, pb, N_prim, M_prim, n, m, plane_lo, plane_hi);
    KLAUNCH(ctx, k_update<true>, grid2d(n, m), block2d(), BumpTab, RGBTab, MaskTab, ParamTab,
            (unsigned long long *const *)ClaimTab, stamp, VMap, RGBD, Mask, Label, P, pb, N_prim, M_prim, n, m, plane_lo, plane_hi);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_update(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                               uint64_t **ClaimTab, uint32_t stamp, const float *VMap, const float *RGBD,
                               const uint8_t *Mask, const uint8_t *Label, const float *pose, int N_prim, int M_prim,
                               int n, int m, int plane_lo, int plane_hi) {
    if (!pose) return RGBDSLAM_E_ARG;
    return launch_update(ctx, nullptr, BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, Label, pose, N_prim,
                         M_prim, n, m, plane_lo, plane_hi);
}

extern "C" int rgbdslam_update_tracked(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pb, uint16_t **BumpTab, uint8_t **RGBTab,
                                       uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp,
                                       const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label,
                                       int N_prim, int M_prim, int n, int m, int plane_lo, int plane_hi) {
    if (!pb) return RGBDSLAM_E_ARG;
    return launch_update(ctx, pb, BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, Label, nullptr, N_prim,
                         M_prim, n, m, plane_lo, plane_hi);
}

extern "C" int rgbdslam_vertex_from_bump(rgbdslam_ctx *ctx, float *VMap, float *RGBA, float *MaskOut, const uint16_t *Bump,
                                         const uint8_t *RGB_bump, const uint8_t *Mask, const float *param_dev, int n, int m,
                                         int lvl) {
    if (!ctx || !VMap || !RGBA || !MaskOut || !Bump || !RGB_bump || !Mask || !param_dev) return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || lvl <= 0 || n / lvl <= 0 || m / lvl <= 0) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_vertex_from_bump, grid2d(n / lvl, m / lvl), block2d(), VMap, RGBA, MaskOut, Bump,
                       RGB_bump, Mask, param_dev, n, m, lvl);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_vertex_from_bump_batch(rgbdslam_ctx *ctx, float **VMapTab, float **RGBATab, float **MaskOutTab,
                                               uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                                               int plane_lo, int plane_hi, int n, int m) {
    if (!ctx || !VMapTab || !RGBATab || !MaskOutTab || !BumpTab || !RGBTab || !MaskTab || !ParamTab) return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || plane_lo < 0 || plane_hi < plane_lo) return RGBDSLAM_E_ARG;
    if (m % DEC_CELLS != 0 || plane_hi - plane_lo > 65535) return RGBDSLAM_E_LIMIT;
    if (plane_hi == plane_lo) return 0;
    size_t groups = (size_t)n * m / DEC_CELLS;
    dim3 grid((unsigned)((groups + 255) / 256), 1, (unsigned)(plane_hi - plane_lo));
    KLAUNCH(ctx, k_vertex_from_bump_batch, grid, dim3(256), VMapTab, RGBATab, MaskOutTab, (const uint16_t *const *)BumpTab,
            (const uint8_t *const *)RGBTab, (const uint8_t *const *)MaskTab, (const float *const *)ParamTab, plane_lo, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
