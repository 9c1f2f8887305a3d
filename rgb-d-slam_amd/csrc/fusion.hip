// fusion.hip -- stage 4: multi-plane Bump-image fusion (scatter) and the Bump -> vertex decode.
#include "common.hpp"
#include "device_fns.hpp"

// UpdateArgs / update_tile: device_fns.hpp (k_predict_draw carries apply tiles too)
template <bool APPLY>
__global__ void k_update(UpdateArgs a) { update_tile<APPLY>(a, blockIdx.x, blockIdx.y); }
template <bool APPLY>
__global__ void k_update_thin(UpdateArgs a) { update_tile<APPLY, true>(a, blockIdx.x, blockIdx.y); }   // frame read from a thin pack

// Apply pass of Graph::Update for frame f AND OffLineFrame::LoadFrame of frame f+1 in one launch: the two are
// independent (every reader of frame f's maps has finished before the apply pass starts), both are latency-bound, and
// the preprocessing tiles fill the issue slots the update leaves idle.
__global__ __launch_bounds__(TILE_W * TILE_H)
void k_update_apply_preprocess(UpdateArgs u, PreprocessArgs p, int gy_update) {
    // tile rows alternate between the two jobs, so both kinds of workgroup are resident from the first wave on
    if (blockIdx.y & 1) frame_preprocess_tile(p, blockIdx.x, (int)(blockIdx.y >> 1));
    else update_tile<true>(u, blockIdx.x, (int)(blockIdx.y >> 1));
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
    BumpCell o = decode_cell(Bump[3 * ref], Bump[3 * ref + 1], Bump[3 * ref + 2], RGB_bump[3 * ref], RGB_bump[3 * ref + 1],
                             RGB_bump[3 * ref + 2], Mask[ref], load_param(param), (float)(lvl * i), (float)(lvl * j), unorm8);
    st3(VMap, idx, o.v.x, o.v.y, o.v.z);
    *(float4 *)(RGBA + 4 * idx) = o.c;
    *(float4 *)(MaskOut + 4 * idx) = o.k;
}

// Whole-model decode in ONE launch (config 3: hundreds of planes): blockIdx.z = plane.  A wavefront
// owns 256 consecutive cells; lane L decodes cells L, L+64, L+128, L+192 of them, so every store
// instruction writes one contiguous run (768 B of VMap, 1 KB of RGBA / MaskOut) -- the outputs are
// 82 % of the traffic; the inputs come in through LDS (below).  x/255 comes from a 256-entry LDS table.  Needs lvl == 1.
#define DEC_CELLS 4
#define DEC_ST(v, p) __builtin_nontemporal_store(v, p)
__global__ __launch_bounds__(256)
void k_vertex_from_bump_batch(float *const *__restrict__ VMapTab, float *const *__restrict__ RGBATab,
                              float *const *__restrict__ MaskOutTab, const uint16_t *const *__restrict__ BumpTab,
                              const uint8_t *const *__restrict__ RGBTab, const uint8_t *const *__restrict__ MaskTab,
                              const float *const *__restrict__ ParamTab, int plane_lo, int n, int m) {
    __shared__ float lut[256];                                  // x / 255 for every byte: one division per thread
    lut[threadIdx.x] = unorm8((uint8_t)threadIdx.x);          // (published by the barrier of either branch below)
    const int plane = plane_lo + blockIdx.z;
    const size_t cells = (size_t)n * m;
    const size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * DEC_CELLS) + (threadIdx.x & 63);
    const BumpParam P = load_param(ParamTab[plane]);
    auto unorm = [&](uint8_t x) { return lut[x]; };
    // (pointers fetched from a table are generic to the compiler: said to be global, they get global_load / global_store
    // instead of flat_* instructions)
    const GLOBAL_AS uint16_t *__restrict__ Bump = (const GLOBAL_AS uint16_t *)BumpTab[plane];
    const GLOBAL_AS uint8_t *__restrict__ RGBb = (const GLOBAL_AS uint8_t *)RGBTab[plane];
    const GLOBAL_AS uint8_t *__restrict__ Maskb = (const GLOBAL_AS uint8_t *)MaskTab[plane];
    GLOBAL_AS float *__restrict__ VMap = (GLOBAL_AS float *)VMapTab[plane];
    GLOBAL_AS float *__restrict__ RGBA = (GLOBAL_AS float *)RGBATab[plane];
    GLOBAL_AS float *__restrict__ MaskOut = (GLOBAL_AS float *)MaskOutTab[plane];
    uint16_t b[DEC_CELLS][3];
    uint8_t c[DEC_CELLS][3], mk[DEC_CELLS];
    // The 10 input bytes of a cell sit in three byte streams (6 + 3 + 1): read cell by cell they cost 7 narrow strided
    // loads each.  A workgroup whose 1024 cells are all there and whose plane pointers are word-aligned fetches its
    // 10 KB as 7 full-width coalesced loads per thread into LDS instead and picks the cells' bytes from there.
    const size_t wg_cell0 = (size_t)blockIdx.x * (256 * DEC_CELLS);
    const bool wide = DEC_CELLS == 4 && wg_cell0 + 256 * DEC_CELLS <= cells &&
                      (((uintptr_t)Bump & 7) | ((uintptr_t)RGBb & 3) | ((uintptr_t)Maskb & 3)) == 0;
    // staging area: the workgroup's 10 KB of inputs first, then (once every thread holds its bytes) 3 KB per wave in
    // which the wave's 256 vertices are laid out as they are in memory, so that they leave as three full 16-byte-per-lane
    // stores instead of twelve 4-byte ones at a 12-byte stride
    __shared__ __align__(16) unsigned char stage[12 * 256 * DEC_CELLS];
    float *vt = nullptr;
    if (wide) {
        typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
        vu2 *sB = (vu2 *)stage;                                   // 6144 B = 768 x 8
        uint32_t *sC = (uint32_t *)(stage + 6 * 256 * DEC_CELLS);     // 3072 B = 768 x 4
        uint32_t *sM = (uint32_t *)(stage + 9 * 256 * DEC_CELLS);     // 1024 B = 256 x 4
        const GLOBAL_AS vu2 *gB = (const GLOBAL_AS vu2 *)(Bump + 3 * wg_cell0);
        const GLOBAL_AS uint32_t *gC = (const GLOBAL_AS uint32_t *)(RGBb + 3 * wg_cell0);
        const GLOBAL_AS uint32_t *gM = (const GLOBAL_AS uint32_t *)(Maskb + wg_cell0);
        const int t = threadIdx.x;
        vu2 rb0 = gB[t], rb1 = gB[t + 256], rb2 = gB[t + 512];
        uint32_t rc0 = gC[t], rc1 = gC[t + 256], rc2 = gC[t + 512], rm = gM[t];
        sB[t] = rb0; sB[t + 256] = rb1; sB[t + 512] = rb2;
        sC[t] = rc0; sC[t + 256] = rc1; sC[t + 512] = rc2;
        sM[t] = rm;
        __syncthreads();
        const uint16_t *lB = (const uint16_t *)stage;
        const uint8_t *lC = stage + 6 * 256 * DEC_CELLS, *lM = stage + 9 * 256 * DEC_CELLS;
#pragma unroll
        for (int k = 0; k < DEC_CELLS; k++) {
            const int lc = (int)(base + 64 * k - wg_cell0);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                b[k][q] = lB[3 * lc + q];
                c[k][q] = lC[3 * lc + q];
            }
            mk[k] = lM[lc];
        }
        if (((uintptr_t)VMap & 15) == 0) {
            __syncthreads();
            vt = (float *)stage + 3 * 64 * DEC_CELLS * (threadIdx.x >> 6);
        }
    } else {
#pragma unroll
        for (int k = 0; k < DEC_CELLS; k++) {                   // all loads first
            size_t cell = base + 64 * k;
            bool in = cell < cells;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                b[k][q] = in ? Bump[3 * cell + q] : (uint16_t)0;
                c[k][q] = in ? RGBb[3 * cell + q] : (uint8_t)0;
            }
            mk[k] = in ? Maskb[cell] : (uint8_t)0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < DEC_CELLS; k++) {
        size_t cell = base + 64 * k;
        if (cell >= cells) break;
        int i = (int)(cell / (size_t)m), j = (int)(cell % (size_t)m);
        BumpCell o = decode_cell(b[k][0], b[k][1], b[k][2], c[k][0], c[k][1], c[k][2], mk[k], P, (float)i, (float)j, unorm);
        // streaming outputs: written once, never re-read by this kernel
        typedef float vfloat4 __attribute__((ext_vector_type(4)));
        if (vt) {
            const int lc = (int)(threadIdx.x & 63) + 64 * k;
            vt[3 * lc] = o.v.x; vt[3 * lc + 1] = o.v.y; vt[3 * lc + 2] = o.v.z;
        } else {
            DEC_ST(o.v.x, VMap + 3 * cell);
            DEC_ST(o.v.y, VMap + 3 * cell + 1);
            DEC_ST(o.v.z, VMap + 3 * cell + 2);
        }
        vfloat4 cv = {o.c.x, o.c.y, o.c.z, o.c.w}, kv = {o.k.x, o.k.y, o.k.z, o.k.w};
        DEC_ST(cv, (GLOBAL_AS vfloat4 *)(RGBA + 4 * cell));
        DEC_ST(kv, (GLOBAL_AS vfloat4 *)(MaskOut + 4 * cell));
    }
    if (vt) {                                                   // workgroup-uniform; the region is this wave's own
        typedef float vfloat4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int lane = threadIdx.x & 63;
        GLOBAL_AS vfloat4 *dst = (GLOBAL_AS vfloat4 *)(VMap + 3 * (base - lane));   // first cell of the wave: a multiple of 256, 3 KB apart
#pragma unroll
        for (int q = 0; q < 3 * DEC_CELLS / 4; q++) DEC_ST(((const vfloat4 *)vt)[lane + 64 * q], dst + lane + 64 * q);
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
static int launch_update(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pb, const float *hist, uint16_t **BumpTab, uint8_t **RGBTab,
                         uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp, const float *VMap,
                         const float *RGBD, const uint8_t *Mask, const uint8_t *Label, const float *pose, int N_prim,
                         int M_prim, int n, int m, int plane_lo, int plane_hi, const uint16_t *Label16 = nullptr) {
    if (!ctx || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !ClaimTab || !VMap || !RGBD || !Mask || (!Label && !Label16) || (!pose && !pb && !hist))
        return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0 || stamp == 0) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;   // 23-bit pixel index in the claim key
    if (Label && labels16(ctx)) { Label16 = (const uint16_t *)Label; Label = nullptr; }   // rgbdslam_set_label_width(ctx, 2)
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = pose ? pose[k] : 0.0f;
    UpdateArgs a = {BumpTab, RGBTab, MaskTab, ParamTab, (unsigned long long *const *)ClaimTab, stamp, VMap, RGBD, Mask, Label, Label16, P, pb, hist,
                    N_prim, M_prim, n, m, plane_lo, plane_hi};
    KLAUNCH(ctx, k_update<false>, grid2d(n, m), block2d(), a);
    KLAUNCH(ctx, k_update<true>, grid2d(n, m), block2d(), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_update(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                               uint64_t **ClaimTab, uint32_t stamp, const float *VMap, const float *RGBD,
                               const uint8_t *Mask, const uint8_t *Label, const float *pose, int N_prim, int M_prim,
                               int n, int m, int plane_lo, int plane_hi) {
    if (!pose) return RGBDSLAM_E_ARG;
    return launch_update(ctx, nullptr, nullptr, BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, Label, pose, N_prim,
                         M_prim, n, m, plane_lo, plane_hi);
}

extern "C" int rgbdslam_update_tracked(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pb, uint16_t **BumpTab, uint8_t **RGBTab,
                                       uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp,
                                       const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label,
                                       int N_prim, int M_prim, int n, int m, int plane_lo, int plane_hi) {
    if (!pb) return RGBDSLAM_E_ARG;
    return launch_update(ctx, pb, nullptr, BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, Label, nullptr, N_prim,
                         M_prim, n, m, plane_lo, plane_hi);
}

// second half of rgbdslam_update_tracked for callers whose claim pass rode on rgbdslam_model_merge_claim
extern "C" int rgbdslam_update_apply_tracked(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pb, uint16_t **BumpTab, uint8_t **RGBTab,
                                             uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp,
                                             const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label,
                                             int N_prim, int M_prim, int n, int m, int plane_lo, int plane_hi) {
    if (!ctx || !pb || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !ClaimTab || !VMap || !RGBD || !Mask || !Label) return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0 || stamp == 0) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = 0.0f;
    const bool l16 = labels16(ctx);
    UpdateArgs a = {BumpTab, RGBTab, MaskTab, ParamTab, (unsigned long long *const *)ClaimTab, stamp, VMap, RGBD, Mask, l16 ? nullptr : Label,
                    l16 ? (const uint16_t *)Label : nullptr, P, pb, nullptr, N_prim, M_prim, n, m, plane_lo, plane_hi};
    KLAUNCH(ctx, k_update<true>, grid2d(n, m), block2d(), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The apply pass alone, posed and gated by the frame's history entry instead of the pose block (which the next frame's
// tracking rewrites): for a caller that deferred the apply pass and has to run it on its own after all.
extern "C" int rgbdslam_update_apply_from_history(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab,
                                                  uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                                  uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                                                  const uint8_t *Label, int N_prim, int M_prim, int n, int m, int plane_lo,
                                                  int plane_hi) {
    if (!ctx || !history_entry_dev || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !ClaimTab || !VMap || !RGBD || !Mask || !Label)
        return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0 || stamp == 0) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = 0.0f;
    const bool l16 = labels16(ctx);
    UpdateArgs a = {BumpTab, RGBTab, MaskTab, ParamTab, (unsigned long long *const *)ClaimTab, stamp, VMap, RGBD, Mask, l16 ? nullptr : Label,
                    l16 ? (const uint16_t *)Label : nullptr, P, nullptr, history_entry_dev, N_prim, M_prim, n, m, plane_lo, plane_hi};
    KLAUNCH(ctx, k_update<true>, grid2d(n, m), block2d(), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// rgbdslam_update_apply_tracked of frame f + rgbdslam_frame_preprocess of frame f+1 (same n x m) in one launch.
extern "C" int rgbdslam_update_apply_tracked_preprocess(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pb, uint16_t **BumpTab,
                                                        uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                                        uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                                                        const uint8_t *Label, int N_prim, int M_prim, int n, int m, int plane_lo,
                                                        int plane_hi, const uint16_t *next_depth, size_t depth_pitch,
                                                        const uint8_t *next_bgr, size_t bgr_pitch, float *fVMap, float *fNMap,
                                                        float *fRGB, float *gx, size_t gx_pitch, float *gy, size_t gy_pitch) {
    if (!ctx || !pb || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !ClaimTab || !VMap || !RGBD || !Mask || !Label) return RGBDSLAM_E_ARG;
    if (!next_depth || !next_bgr || !fVMap || !fNMap || !fRGB || !gx || !gy) return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0 || stamp == 0) return RGBDSLAM_E_ARG;
    if (depth_pitch < 2 * (size_t)m || bgr_pitch < 3 * (size_t)m || gx_pitch < 4 * (size_t)m || gy_pitch < 4 * (size_t)m)
        return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = 0.0f;
    const bool l16 = labels16(ctx);
    UpdateArgs u = {BumpTab, RGBTab, MaskTab, ParamTab, (unsigned long long *const *)ClaimTab, stamp, VMap, RGBD, Mask, l16 ? nullptr : Label,
                    l16 ? (const uint16_t *)Label : nullptr, P, pb, nullptr, N_prim, M_prim, n, m, plane_lo, plane_hi};
    PreprocessArgs p = {next_depth, depth_pitch, next_bgr, bgr_pitch, fVMap, fNMap, fRGB, gx, gx_pitch, gy, gy_pitch, ctx->calib, n, m, ctx->aux_write};
    ctx->aux_write = nullptr;                         // announced for one launch (rgbdslam_set_frame_aux)
    dim3 grid = grid2d(n, m);
    const int gy_update = (int)grid.y;
    grid.y *= 2;
    KLAUNCH(ctx, k_update_apply_preprocess, grid, block2d(), u, p, gy_update);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_update_from_history(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab,
                                            uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                            uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                                            const uint8_t *Label, int N_prim, int M_prim, int n, int m, int plane_lo,
                                            int plane_hi) {
    if (!history_entry_dev) return RGBDSLAM_E_ARG;
    return launch_update(ctx, nullptr, history_entry_dev, BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, Label,
                         nullptr, N_prim, M_prim, n, m, plane_lo, plane_hi);
}

// ---- 16-bit labels: models of more than 255 planes (BASELINE configs[2]: 256 planes; the reference's u8 label image,
// Frame.h:197, caps a model at 255).  Same kernels, same winner rule; label 0 = none, L = plane L-1. ----------------------
extern "C" int rgbdslam_update_l16(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                                   uint64_t **ClaimTab, uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                                   const uint16_t *Label16, const float *pose, int N_prim, int M_prim, int n, int m, int plane_lo,
                                   int plane_hi) {
    if (!pose || !Label16) return RGBDSLAM_E_ARG;
    return launch_update(ctx, nullptr, nullptr, BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, nullptr, pose, N_prim,
                         M_prim, n, m, plane_lo, plane_hi, Label16);
}
extern "C" int rgbdslam_update_from_history_l16(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab, uint8_t **RGBTab,
                                                uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp,
                                                const float *VMap, const float *RGBD, const uint8_t *Mask, const uint16_t *Label16,
                                                int N_prim, int M_prim, int n, int m, int plane_lo, int plane_hi) {
    if (!history_entry_dev || !Label16) return RGBDSLAM_E_ARG;
    return launch_update(ctx, nullptr, history_entry_dev, BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, nullptr,
                         nullptr, N_prim, M_prim, n, m, plane_lo, plane_hi, Label16);
}

// ---- thin frame pack (plane-partitioned multi-GPU; no reference counterpart) ------------------------------------------------
// What the plane owners need of a tracked frame that they cannot derive: the merged model depth RGBD.w (f32), the frame's colour
// bytes, the confidence mask and the labels -- 10 B per pixel instead of the 31 B of VMap + RGBD + Mask + Label.  After Merge
// (Frame.cpp:300-333) VMap is VertexMapGL(RGBD.w) (Kernel.cu:54-72) and RGBD.rgb is the frame's byte / 255 (Kernel.cu:463-465):
// k_export_thin recovers the byte (rn(255 x) inverts x = b / 255 exactly), the update kernels rebuild both from the thin
// fields with the reference's own operations.
// Label16 / hist may be null (round 3: the label image the tracker maintains and the frame's 64-byte history entry travel in the
// same launch instead of two stream-ordered copies of their own)
__global__ void k_export_thin(const float *__restrict__ RGBD, const uint8_t *__restrict__ Mask, float *__restrict__ W,
                              uint8_t *__restrict__ RGB8, uint8_t *__restrict__ MaskOut, const uint16_t *__restrict__ Label16,
                              uint16_t *__restrict__ Label16Out, const float *__restrict__ hist, float *__restrict__ histOut,
                              int n, int m) {
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (hist && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0 && threadIdx.x < 16) histOut[threadIdx.x] = hist[threadIdx.x];
    if (i >= n || j >= m) return;
    const size_t idx = (size_t)i * m + j;
    if (Label16) Label16Out[idx] = Label16[idx];
    const float4 v = *(const float4 *)(RGBD + 4 * idx);
    W[idx] = v.w;
    RGB8[3 * idx] = (uint8_t)f2i_rn((float)((double)v.x * 255.0));
    RGB8[3 * idx + 1] = (uint8_t)f2i_rn((float)((double)v.y * 255.0));
    RGB8[3 * idx + 2] = (uint8_t)f2i_rn((float)((double)v.z * 255.0));
    MaskOut[idx] = Mask[idx];
}
extern "C" int rgbdslam_export_thin(rgbdslam_ctx *ctx, const float *RGBD, const uint8_t *Mask, float *W_out, uint8_t *RGB8_out,
                                    uint8_t *Mask_out, int n, int m) {
    if (!ctx || !RGBD || !Mask || !W_out || !RGB8_out || !Mask_out || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_export_thin, grid2d(n, m), block2d(), RGBD, Mask, W_out, RGB8_out, Mask_out, (const uint16_t *)nullptr, (uint16_t *)nullptr,
            (const float *)nullptr, (float *)nullptr, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
// the same + the frame's u16 label image and its 64-byte history entry (either pair may be NULL), all in the one launch
extern "C" int rgbdslam_export_thin_l16(rgbdslam_ctx *ctx, const float *RGBD, const uint8_t *Mask, const uint16_t *Label16,
                                        const float *history_entry_dev, float *W_out, uint8_t *RGB8_out, uint8_t *Mask_out,
                                        uint16_t *Label16_out, float *history_entry_out, int n, int m) {
    if (!ctx || !RGBD || !Mask || !W_out || !RGB8_out || !Mask_out || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if ((Label16 == nullptr) != (Label16_out == nullptr) || (history_entry_dev == nullptr) != (history_entry_out == nullptr)) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_export_thin, grid2d(n, m), block2d(), RGBD, Mask, W_out, RGB8_out, Mask_out, Label16, Label16_out, history_entry_dev,
            history_entry_out, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
// Graph::Update (claim + apply) from a thin pack, posed and gated by the frame's history entry, 16-bit labels.
extern "C" int rgbdslam_update_from_history_thin_l16(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab,
                                                     uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                                     uint32_t stamp, const float *W, const uint8_t *RGB8, const uint8_t *Mask,
                                                     const uint16_t *Label16, int N_prim, int M_prim, int n, int m, int plane_lo,
                                                     int plane_hi) {
    if (!ctx || !history_entry_dev || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !ClaimTab || !W || !RGB8 || !Mask || !Label16)
        return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0 || stamp == 0) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = 0.0f;
    UpdateArgs a = {BumpTab, RGBTab, MaskTab, ParamTab, (unsigned long long *const *)ClaimTab, stamp, nullptr, nullptr, Mask, nullptr, Label16, P,
                    nullptr, history_entry_dev, N_prim, M_prim, n, m, plane_lo, plane_hi, W, RGB8, ctx->calib};
    KLAUNCH(ctx, k_update_thin<false>, grid2d(n, m), block2d(), a);
    KLAUNCH(ctx, k_update_thin<true>, grid2d(n, m), block2d(), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Labels by plane membership for tiled models (test / bench harness stand-in for plane detection, like rgbdslam_seed_labels):
// a pixel takes the FIRST plane k in [0, nb) whose equation it satisfies (|n.p - d| <= epsilon, n.n_p >= alpha) AND whose
// N_prim x M_prim Bump image contains its cell -- so patches that tile one infinite plane get distinct labels.
// The parameters of 256 planes at a time are staged in LDS (one plane per thread: pointer, then 14 floats, all in flight
// together).  Round 3: the scan no longer evaluates every plane's equation for every pixel.  Patches that tile one plane
// carry the SAME equation, basis and scale and differ in their shift only; consecutive planes whose twelve such words are
// bitwise equal form a run, found while staging (a ballot of "differs from its predecessor" per wave).  The equation
// test and the two in-plane coordinates are evaluated once per run (same operands, same operations: the same bits the
// per-plane evaluation gave), a run no lane of the wave belongs to is jumped over, and inside a run only the cell
// containment is tested per member.  The first match in plane order is unchanged (tests vs orc.label_planes_l16).
#define LBL_CHUNK (TILE_W * TILE_H)
__global__ __launch_bounds__(TILE_W * TILE_H)
void k_label_planes_l16(uint16_t *__restrict__ Label16, const float *__restrict__ VMap, const float *__restrict__ NMap,
                        PoseV P, const float *__restrict__ hist, const float *const *__restrict__ ParamTab, int nb, int N_prim,
                        int M_prim, float epsilon, float alpha, int n, int m) {
    __shared__ float sq[LBL_CHUNK][15];                 // 15: odd stride, the staging writes of neighbouring threads do not collide
    __shared__ unsigned long long lead[LBL_CHUNK / 64]; // bit k: plane k of the chunk starts a run
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    const bool in = i < n && j < m;
    if (hist) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            P.v[4 * c] = hist[3 * c]; P.v[4 * c + 1] = hist[3 * c + 1]; P.v[4 * c + 2] = hist[3 * c + 2]; P.v[4 * c + 3] = 0.0f;
        }
        P.v[12] = hist[9]; P.v[13] = hist[10]; P.v[14] = hist[11]; P.v[15] = 1.0f;
    }
    const size_t idx = in ? (size_t)i * m + j : 0;
    uint16_t res = 0;
    const F3 nl = ld3(NMap, idx);
    bool open = in && !is_zero3(nl);                    // still looking for its plane
    const F3 p = xf_point(P.v, ld3(VMap, idx));
    const F3 q = xf_dir(P.v, nl);
    for (int base = 0; base < nb; base += LBL_CHUNK) {
        const int cnt = min(LBL_CHUNK, nb - base);
        __syncthreads();                                // the previous chunk has been read
        if (tid < cnt) {
            const float *Q = ParamTab[base + tid];
#pragma unroll
            for (int z = 0; z < 14; z++) sq[tid][z] = Q[z];
        }
        __syncthreads();
        {
            bool differs = tid < cnt;                   // planes beyond the chunk: no run starts there
            if (tid > 0 && tid < cnt) {
                differs = false;
#pragma unroll
                for (int z = 0; z < 14; z++)
                    if (z != 6 && z != 7) differs |= __float_as_uint(sq[tid][z]) != __float_as_uint(sq[tid - 1][z]);
            }
            const unsigned long long bits = __ballot(differs);
            if ((tid & 63) == 0) lead[tid >> 6] = bits;
        }
        __syncthreads();
        if (!__syncthreads_or(open)) break;             // every pixel of the tile has its label
        int k = 0;                                      // wave-uniform: always the first plane of a run
        while (k < cnt && __ballot(open)) {
            // end of the run that starts at k: the next set bit of `lead` above k (LBL_CHUNK / 64 words)
            int kend = cnt;
            {
                const int w0 = k >> 6;
                unsigned long long rest = lead[w0] & ~((2ull << (k & 63)) - 1ull);      // bits above k in its word
                if ((k & 63) == 63) rest = 0ull;
                int wq = w0;
                while (rest == 0ull && ++wq < LBL_CHUNK / 64) rest = lead[wq];
                if (rest != 0ull) kend = min(cnt, 64 * wq + __builtin_ctzll(rest));
            }
            const float *Q = sq[k];
            // the two cheap tests first, once per run: the cell geometry (two f64 divisions) only where the equation fits --
            // the same three conditions as the oracle's, evaluated lazily
            const float ea = (q.x * Q[0] + q.y * Q[1]) + q.z * Q[2];
            const float err = ((p.x * Q[0] + p.y * Q[1]) + p.z * Q[2]) - Q[3];
            bool fits = open && !(__builtin_fabsf(err) > epsilon || ea < alpha);
            if (__ballot(fits)) {
                // update_cell's in-plane coordinates (device_fns.hpp), the part that does not depend on the shift
                const float a0 = (p.x * Q[8] + p.y * Q[9]) + p.z * Q[10];
                const float b0 = (p.x * Q[11] + p.y * Q[12]) + p.z * Q[13];
                // Which members of the run can hold a cell of this wave at all?  The wave's 64 pixels are neighbours in a
                // row: their (a0, b0) span a small box.  A member's image spans [shift, shift + 2 N / scale) in each
                // coordinate; 64 members at a time, one per lane, are tested against the box with a generous margin (the
                // exact test below decides; a member skipped here cannot pass it for any lane), and only the candidates --
                // one or two of the ~85 patches that tile a wall -- are walked in plane order.
                float amin = fits ? a0 : __builtin_inff(), amax = fits ? a0 : -__builtin_inff();
                float bmin = fits ? b0 : __builtin_inff(), bmax = fits ? b0 : -__builtin_inff();
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    amin = fminf(amin, __shfl_xor(amin, o)); amax = fmaxf(amax, __shfl_xor(amax, o));
                    bmin = fminf(bmin, __shfl_xor(bmin, o)); bmax = fmaxf(bmax, __shfl_xor(bmax, o));
                }
                const float ext0 = 2.0f * (float)N_prim / Q[4], ext1 = 2.0f * (float)M_prim / Q[5];
                const int lane = tid & 63;
                for (int kb = k; kb < kend && __ballot(fits); kb += 64) {
                    const int mk = min(kb + lane, kend - 1);
                    const float s6 = sq[mk][6], s7 = sq[mk][7];
                    const float mg0 = 1e-5f * (((__builtin_fabsf(s6) + __builtin_fabsf(ext0)) + __builtin_fabsf(amin)) + __builtin_fabsf(amax)) + 1e-7f;
                    const float mg1 = 1e-5f * (((__builtin_fabsf(s7) + __builtin_fabsf(ext1)) + __builtin_fabsf(bmin)) + __builtin_fabsf(bmax)) + 1e-7f;
                    // written so that a NaN anywhere keeps the member (it is then decided by the exact test)
                    // (a negative scale mirrors the extent; a zero scale puts every pixel into cell 0: then nothing is skipped)
                    const float lo0 = fminf(s6, s6 + ext0), hi0 = fmaxf(s6, s6 + ext0), lo1 = fminf(s7, s7 + ext1), hi1 = fmaxf(s7, s7 + ext1);
                    const bool bounded = __builtin_fabsf(ext0) < 3e38f && __builtin_fabsf(ext1) < 3e38f;
                    const bool out = bounded && (amax < lo0 - mg0 || amin > hi0 + mg0 || bmax < lo1 - mg1 || bmin > hi1 + mg1);
                    unsigned long long cand = __ballot(kb + lane < kend && !out);
                    while (cand != 0ull && __ballot(fits)) {
                        const int kk = kb + __builtin_ctzll(cand);
                        cand &= cand - 1ull;
                        if (!fits) continue;
                        const float a = (float)((double)((a0 - sq[kk][6]) * Q[4]) / 2.0);
                        const float b = (float)((double)((b0 - sq[kk][7]) * Q[5]) / 2.0);
                        const int ii = f2i_rd(a), jj = f2i_rd(b);
                        if (ii > N_prim - 1 || jj > M_prim - 1 || ii < 0 || jj < 0) continue;
                        res = (uint16_t)(base + kk + 1);
                        open = false;
                        fits = false;
                    }
                }
            }
            k = kend;
        }
    }
    if (in) Label16[idx] = res;
}
extern "C" int rgbdslam_label_planes_l16(rgbdslam_ctx *ctx, uint16_t *Label16, const float *VMap, const float *NMap,
                                         const float *pose16_host, const float *history_entry_dev, float **ParamTab, int nb_planes,
                                         int N_prim, int M_prim, float epsilon, float alpha, int n, int m) {
    if (!ctx || !Label16 || !VMap || !NMap || (!pose16_host && !history_entry_dev) || !ParamTab || n <= 0 || m <= 0 || N_prim <= 0 || M_prim <= 0)
        return RGBDSLAM_E_ARG;
    if (nb_planes <= 0 || nb_planes > 65535) return RGBDSLAM_E_LIMIT;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = pose16_host ? pose16_host[k] : 0.0f;
    KLAUNCH(ctx, k_label_planes_l16, grid2d(n, m), block2d(), Label16, VMap, NMap, P, pose16_host ? nullptr : history_entry_dev,
            (const float *const *)ParamTab, nb_planes, N_prim, M_prim, epsilon, alpha, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
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

// ---- D2/Primitive.cu:31-177: the decode as DemoStep2's viewer calls it ---------------------------------
// variant 0 = VertexFromBumpKernel / VertexOnlyFromBumpKernel (identical bodies, D2/Primitive.cu:31-121): 16-bit mask (the dump
// stores mask x 200), vertices only, sub-sampled output grid like D1's.  variant 1 = VertexRGBFromBumpKernel (:123-179): 8-bit
// mask, vertex + RGBA, input and output indexed by the SAME i * m + j, lvl only scales the cell index in the position.
// Both: a cell is empty when mask < 11 (no special value 1 as in D1).
__global__ void k_vertex_from_bump_d2(float *__restrict__ VMap, float *__restrict__ RGBA, const uint16_t *__restrict__ Bump,
                                      const uint8_t *__restrict__ RGB_bump, const void *__restrict__ Mask,
                                      const float *__restrict__ param, int n, int m, int lvl, int variant) {
    const int no = variant ? n : n / lvl, mo = variant ? m : m / lvl;
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= no || j >= mo) return;
    const size_t idx = (size_t)i * mo + j;
    const size_t ref = variant ? idx : (size_t)(lvl * i) * m + (size_t)(lvl * j);
    const unsigned int mk = variant ? (unsigned int)((const uint8_t *)Mask)[ref] : (unsigned int)((const uint16_t *)Mask)[ref];
    uint8_t cr = 0, cg = 0, cb = 0;
    if (variant) { cr = RGB_bump[3 * ref]; cg = RGB_bump[3 * ref + 1]; cb = RGB_bump[3 * ref + 2]; }
    // decode_cell with a mask that passes its own emptiness test; emptiness is decided here
    BumpCell o = decode_cell(Bump[3 * ref], Bump[3 * ref + 1], Bump[3 * ref + 2], cr, cg, cb, (uint8_t)255, load_param(param),
                             (float)(lvl * i), (float)(lvl * j), unorm8);
    if (mk < 11u) { o.v.x = o.v.y = o.v.z = 0.0f; o.c.x = o.c.y = o.c.z = o.c.w = 0.0f; }
    VMap[3 * idx] = o.v.x; VMap[3 * idx + 1] = o.v.y; VMap[3 * idx + 2] = o.v.z;
    if (variant) *(float4 *)(RGBA + 4 * idx) = o.c;
}
extern "C" int rgbdslam_vertex_from_bump_d2(rgbdslam_ctx *ctx, float *VMap, float *RGBA, const uint16_t *Bump, const uint8_t *RGB_bump,
                                            const void *Mask, const float *param_dev, int n, int m, int lvl, int variant) {
    if (!ctx || !VMap || !Bump || !Mask || !param_dev || (variant != 0 && variant != 1)) return RGBDSLAM_E_ARG;
    if (variant == 1 && (!RGBA || !RGB_bump)) return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || lvl <= 0 || n / lvl <= 0 || m / lvl <= 0) return RGBDSLAM_E_ARG;
    KLAUNCH(ctx, k_vertex_from_bump_d2, variant ? grid2d(n, m) : grid2d(n / lvl, m / lvl), block2d(), VMap, RGBA, Bump, RGB_bump, Mask,
            param_dev, n, m, lvl, variant);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_vertex_from_bump_batch(rgbdslam_ctx *ctx, float **VMapTab, float **RGBATab, float **MaskOutTab,
                                               uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                                               int plane_lo, int plane_hi, int n, int m) {
    if (!ctx || !VMapTab || !RGBATab || !MaskOutTab || !BumpTab || !RGBTab || !MaskTab || !ParamTab) return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || plane_lo < 0 || plane_hi < plane_lo) return RGBDSLAM_E_ARG;
    if (plane_hi - plane_lo > 65535 || (size_t)n * m > 0x7FFFFFFFu) return RGBDSLAM_E_LIMIT;
    if (plane_hi == plane_lo) return 0;
    size_t cells = (size_t)n * m;
    dim3 grid((unsigned)((cells + 256 * DEC_CELLS - 1) / (256 * DEC_CELLS)), 1, (unsigned)(plane_hi - plane_lo));
    KLAUNCH(ctx, k_vertex_from_bump_batch, grid, dim3(256), VMapTab, RGBATab, MaskOutTab, (const uint16_t *const *)BumpTab,
            (const uint8_t *const *)RGBTab, (const uint8_t *const *)MaskTab, (const float *const *)ParamTab, plane_lo, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
