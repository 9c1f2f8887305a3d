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
__global__ void k_vertex_from_bump(float *__restrict__ VMap, float *__restrict__ RGBA, float *__restrict__ MaskOut,
                                   const uint16_t *__restrict__ Bump, const uint8_t *__restrict__ RGB_bump,
                                   const uint8_t *__restrict__ Mask, const float *__restrict__ param, int n, int m, int lvl) {
    int no = n / lvl, mo = m / lvl;
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= no || j >= mo) return;
    size_t idx = (size_t)i * mo + j;
    size_t ref = (size_t)(lvl * i) * m + (size_t)(lvl * j);
    uint8_t mk = Mask[ref];
    float4 zero; zero.x = zero.y = zero.z = zero.w = 0.0f;
    if (mk < 11 && mk != 1) {
        st3(VMap, idx, 0.0f, 0.0f, 0.0f);
        *(float4 *)(RGBA + 4 * idx) = zero;
        *(float4 *)(MaskOut + 4 * idx) = zero;
        return;
    }
    float fi = (float)(lvl * i), fj = (float)(lvl * j);
    float s0 = (float)((double)(float)Bump[3 * ref] / 60000.0);
    float s1 = (float)((double)(float)Bump[3 * ref + 1] / 60000.0);
    float x = (float)((double)(fi + s0) * 2.0 / (double)param[4] + (double)param[6]);
    float y = (float)((double)(fj + s1) * 2.0 / (double)param[5] + (double)param[7]);
    float d = (float)((double)param[3] + (((double)(float)Bump[3 * ref + 2] / 2000.0) - 15.0));
    float v0 = (x * param[8] + y * param[11]) + d * param[0];
    float v1 = (x * param[9] + y * param[12]) + d * param[1];
    float v2 = (x * param[10] + y * param[13]) + d * param[2];
    st3(VMap, idx, v0, v1, v2);
    float4 c;
    c.x = (float)((double)(float)RGB_bump[3 * ref] / 255.0);
    c.y = (float)((double)(float)RGB_bump[3 * ref + 1] / 255.0);
    c.z = (float)((double)(float)RGB_bump[3 * ref + 2] / 255.0);
    c.w = 1.0f;
    *(float4 *)(RGBA + 4 * idx) = c;
    float4 k;
    k.x = (float)((10.0 + (double)v0) / 20.0);
    k.y = (float)((10.0 + (double)v1) / 20.0);
    k.z = (float)((10.0 + (double)v2) / 20.0);
    k.w = (float)((double)(float)mk / 255.0);
    *(float4 *)(MaskOut + 4 * idx) = k;
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
