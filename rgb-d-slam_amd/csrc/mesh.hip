// mesh.hip -- Bump image -> mesh: vertices, colours, normals and the trimmed quad list
// of a plane, i.e. what the reference keeps per Primitive in its VBO triple + index buffer.
//   QuadTrimKernel           D1/Primitive.cu:95-152  (variant 0)   D2/Primitive.cu:367-424 (variant 1)
//   VertexFromBumpKernel     D1/Primitive.cu:31-93   (device_fns.hpp: decode_cell)
//   NormalKernel             D2/Primitive.cu:181-365 == D1/Kernel.cu:123-307 (device_fns.hpp: normal_from_cross)
// Two launch shapes, same per-cell functions: the per-operator k_quad_trim behind the reference's wrapper,
// and k_bump_mesh, which does decode + normals + trim of a whole model (blockIdx.z = plane) in one pass
// over the Bump images: 10 B read and 72 B written per cell instead of three passes that re-read the
// 12 B vertices four more times.
#include "common.hpp"
#include "device_fns.hpp"

#define HALO_W (TILE_W + 2)
#define HALO_H (TILE_H + 2)
#define HALO_N (HALO_W * HALO_H)

// Primitive.cu:135 (D1) / D2 Primitive.cu:407: 1 = the quad's corner disqualifies it
__device__ __forceinline__ bool corner_rejected(uint8_t mk, int min_conf, int variant) {
    return variant ? (mk < 11) : (mk != 1 && (int)mk - 10 < min_conf);
}
__device__ __forceinline__ uint4 quad_record(bool cut, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, int variant) {
    uint4 o;
    if (cut) { o.x = o.y = o.z = o.w = 0u; }
    else if (variant) { o.x = v0; o.y = v3; o.z = v2; o.w = v1; }          // D2 winds the quad the other way round
    else { o.x = v0; o.y = v1; o.z = v2; o.w = v3; }
    return o;
}

// Grid of (n/lvl - 1) x (m/lvl - 1) quads over a vertex map decoded at level lvl.  The reference forms the
// vertex ids as i*m/lvl + j in integer arithmetic and indexes Mask with the SAME ids (Primitive.cu:107-110,135).
__global__ void k_quad_trim(uint32_t *__restrict__ index, const float *__restrict__ VMap, const uint8_t *__restrict__ Mask,
                            int min_conf, uint32_t n, uint32_t m, float thresh, uint32_t lvl, int variant) {
    uint32_t j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i > n / lvl - 2 || j > m / lvl - 2) return;
    size_t q = (size_t)i * (m / lvl - 1) + j;
    uint32_t v0 = i * m / lvl + j, v1 = v0 + 1, v2 = (i + 1) * m / lvl + j + 1, v3 = v2 - 1;
    bool cut = corner_rejected(Mask[v0], min_conf, variant) || corner_rejected(Mask[v1], min_conf, variant) ||
               corner_rejected(Mask[v2], min_conf, variant) || corner_rejected(Mask[v3], min_conf, variant) ||
               quad_trimmed(ld3(VMap, v0), ld3(VMap, v1), ld3(VMap, v2), ld3(VMap, v3), thresh);
    *(uint4 *)(index + 4 * q) = quad_record(cut, v0, v1, v2, v3, variant);
}

// One pass per model: tile of 64 x MESH_ROWS cells + 1-cell halo decoded into LDS (halo cells: vertex + mask
// only; 16 % extra decodes).  A workgroup is 64 x 4 threads; thread (x, y) owns cells (y + 4k, x), so every
// store instruction of a wave covers one contiguous row segment.  Outputs are streaming (non-temporal).
#define MESH_ROWS 16
#define MHALO_H (MESH_ROWS + 2)
#define MHALO_N (HALO_W * MHALO_H)
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef uint32_t vuint4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(TILE_W *TILE_H)
void k_bump_mesh(float *const *__restrict__ VMapTab, float *const *__restrict__ RGBATab, float *const *__restrict__ MaskOutTab,
                 float *const *__restrict__ NMapTab, uint32_t *const *__restrict__ IndexTab,
                 const uint16_t *const *__restrict__ BumpTab, const uint8_t *const *__restrict__ RGBTab,
                 const uint8_t *const *__restrict__ MaskTab, const float *const *__restrict__ ParamTab, int plane_lo, int n, int m,
                 int min_conf, float thresh, int inverse, int variant) {
    __shared__ float sv[MHALO_H][HALO_W][3];
    __shared__ uint8_t sm[MHALO_H][HALO_W];
    const int plane = plane_lo + blockIdx.z;
    const BumpParam P = load_param(ParamTab[plane]);
    const uint16_t *__restrict__ Bump = BumpTab[plane];
    const uint8_t *__restrict__ RGBb = RGBTab[plane];
    const uint8_t *__restrict__ Maskb = MaskTab[plane];
    float *__restrict__ VMap = VMapTab[plane];
    float *__restrict__ RGBA = RGBATab[plane];
    float *__restrict__ MaskOut = MaskOutTab[plane];
    float *__restrict__ NMap = NMapTab[plane];
    uint32_t *__restrict__ Index = IndexTab[plane];
    const int j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * MESH_ROWS;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int e = tid; e < MHALO_N; e += TILE_W * TILE_H) {
        int r = e / HALO_W, cc = e % HALO_W;
        int ii = i0 + r - 1, jj = j0 + cc - 1;
        float3 v; v.x = v.y = v.z = 0.0f;
        uint8_t mk = 0;
        if (ii >= 0 && ii < n && jj >= 0 && jj < m) {
            size_t cell = (size_t)ii * m + jj;
            bool own = r >= 1 && r <= MESH_ROWS && cc >= 1 && cc <= TILE_W;
            mk = Maskb[cell];
            uint8_t cr = 0, cg = 0, cb = 0;
            if (own) { cr = RGBb[3 * cell]; cg = RGBb[3 * cell + 1]; cb = RGBb[3 * cell + 2]; }
            BumpCell o = decode_cell(Bump[3 * cell], Bump[3 * cell + 1], Bump[3 * cell + 2], cr, cg, cb, mk, P, (float)ii, (float)jj, unorm8);
            v = o.v;
            if (own) {
                __builtin_nontemporal_store(v.x, VMap + 3 * cell);
                __builtin_nontemporal_store(v.y, VMap + 3 * cell + 1);
                __builtin_nontemporal_store(v.z, VMap + 3 * cell + 2);
                vfloat4 cv = {o.c.x, o.c.y, o.c.z, o.c.w}, kv = {o.k.x, o.k.y, o.k.z, o.k.w};
                __builtin_nontemporal_store(cv, (vfloat4 *)(RGBA + 4 * cell));
                __builtin_nontemporal_store(kv, (vfloat4 *)(MaskOut + 4 * cell));
            }
        }
        sv[r][cc][0] = v.x; sv[r][cc][1] = v.y; sv[r][cc][2] = v.z;
        sm[r][cc] = mk;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= m) return;
    const int cc = threadIdx.x + 1;
    auto lv = [&](int rr, int c2) { F3 q; q.x = sv[rr][c2][0]; q.y = sv[rr][c2][1]; q.z = sv[rr][c2][2]; return q; };
#pragma unroll
    for (int k = 0; k < MESH_ROWS / TILE_H; k++) {
        const int r = threadIdx.y + TILE_H * k + 1, i = i0 + r - 1;
        if (i >= n) break;
        const F3 p = lv(r, cc);
        F3 nr; nr.x = nr.y = nr.z = 0.0f;
        if (!(i < 1 || i > n - 2 || j < 1 || j > m - 2 || p.z == 0.0f))
            nr = normal_from_cross(p, lv(r + 1, cc), lv(r - 1, cc), lv(r, cc + 1), lv(r, cc - 1), inverse);
        const size_t cell = (size_t)i * m + j;
        __builtin_nontemporal_store(nr.x, NMap + 3 * cell);
        __builtin_nontemporal_store(nr.y, NMap + 3 * cell + 1);
        __builtin_nontemporal_store(nr.z, NMap + 3 * cell + 2);
        if (i <= n - 2 && j <= m - 2) {
            uint32_t v0 = (uint32_t)cell, v1 = v0 + 1, v2 = v0 + m + 1, v3 = v0 + m;
            bool cut = corner_rejected(sm[r][cc], min_conf, variant) || corner_rejected(sm[r][cc + 1], min_conf, variant) ||
                       corner_rejected(sm[r + 1][cc + 1], min_conf, variant) || corner_rejected(sm[r + 1][cc], min_conf, variant) ||
                       quad_trimmed(p, lv(r, cc + 1), lv(r + 1, cc + 1), lv(r + 1, cc), thresh);
            uint4 o = quad_record(cut, v0, v1, v2, v3, variant);
            vuint4 ov = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(ov, (vuint4 *)(Index + 4 * ((size_t)i * (m - 1) + j)));
        }
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int rgbdslam_quad_trim(rgbdslam_ctx *ctx, const float *VMap, const uint8_t *Mask, uint32_t *indices_dev,
                                  unsigned short min_conf, unsigned int n, unsigned int m, float thresh, int lvl, int variant) {
    if (!ctx || !VMap || !Mask || !indices_dev || lvl <= 0 || (variant != 0 && variant != 1)) return RGBDSLAM_E_ARG;
    if (n / (unsigned)lvl < 2 || m / (unsigned)lvl < 2) return RGBDSLAM_E_ARG;   // the reference's guard wraps around here
    if ((size_t)n * m > 0x7FFFFFFFu) return RGBDSLAM_E_LIMIT;
    KLAUNCH(ctx, k_quad_trim, grid2d((int)(n / lvl - 1), (int)(m / lvl - 1)), block2d(), indices_dev, VMap, Mask, (int)min_conf,
            (uint32_t)n, (uint32_t)m, thresh, (uint32_t)lvl, variant);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_bump_mesh_batch(rgbdslam_ctx *ctx, float **VMapTab, float **RGBATab, float **MaskOutTab, float **NMapTab,
                                        uint32_t **IndexTab, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab,
                                        float **ParamTab, int plane_lo, int plane_hi, int n, int m, unsigned short min_conf,
                                        float thresh, int inverse_normals, int variant) {
    if (!ctx || !VMapTab || !RGBATab || !MaskOutTab || !NMapTab || !IndexTab || !BumpTab || !RGBTab || !MaskTab || !ParamTab)
        return RGBDSLAM_E_ARG;
    if (n < 2 || m < 2 || plane_lo < 0 || plane_hi < plane_lo || (variant != 0 && variant != 1)) return RGBDSLAM_E_ARG;
    if (plane_hi - plane_lo > 65535 || (size_t)n * m > 0x7FFFFFFFu) return RGBDSLAM_E_LIMIT;
    if (plane_hi == plane_lo) return 0;
    dim3 grid((m + TILE_W - 1) / TILE_W, (n + MESH_ROWS - 1) / MESH_ROWS, (unsigned)(plane_hi - plane_lo));
    KLAUNCH(ctx, k_bump_mesh, grid, block2d(), VMapTab, RGBATab, MaskOutTab, NMapTab, IndexTab, (const uint16_t *const *)BumpTab,
            (const uint8_t *const *)RGBTab, (const uint8_t *const *)MaskTab, (const float *const *)ParamTab, plane_lo, n, m,
            (int)min_conf, thresh, inverse_normals ? 1 : 0, variant);
    HIP_TRY(hipGetLastError());
    return 0;
}
