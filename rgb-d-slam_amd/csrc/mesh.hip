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
// only; 16 % extra decodes).  A workgroup is 64 x 4 threads = 4 waves; wave y owns rows y + 4k, so every
// store instruction of a wave covers one contiguous row segment.  Outputs are streaming (non-temporal).
// 72 of the 82 bytes per cell are written, so the stores decide: the 12-byte vertices and normals of a row segment
// (768 B) leave as 48 lanes x 16 bytes from LDS -- the vertices straight from the halo tile, whose rows are laid out
// so that the segment starts 16-byte aligned, the normals through 768 B per wave -- instead of three 4-byte stores at
// a 12-byte stride each (MESH_WIDE; needs m % 4 == 0, a full tile column and 16-byte aligned planes, else the
// narrow stores).  x / 255 comes from a 256-entry table, and the table pointers are global ones (fetched from a
// table they are generic to the compiler, and every access a flat_ instruction).
#define MESH_ROWS 16
#define MHALO_H (MESH_ROWS + 2)
#define MHALO_N (HALO_W * MHALO_H)
#define MESH_SV_STRIDE (3 * HALO_W + 2)           // floats per LDS row: 200 = a multiple of 4; cell cc at 1 + 3 cc, so cell 1 at 16 B
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef uint32_t vuint4 __attribute__((ext_vector_type(4)));
static_assert(MESH_SV_STRIDE % 4 == 0 && TILE_W == 64 && TILE_H == 4, "mesh tile layout");
__global__ __launch_bounds__(TILE_W *TILE_H)
void k_bump_mesh(float *const *__restrict__ VMapTab, float *const *__restrict__ RGBATab, float *const *__restrict__ MaskOutTab,
                 float *const *__restrict__ NMapTab, uint32_t *const *__restrict__ IndexTab,
                 const uint16_t *const *__restrict__ BumpTab, const uint8_t *const *__restrict__ RGBTab,
                 const uint8_t *const *__restrict__ MaskTab, const float *const *__restrict__ ParamTab, int plane_lo, int n, int m,
                 int min_conf, float thresh, int inverse, int variant) {
    __shared__ __align__(16) float sv[MHALO_H][MESH_SV_STRIDE];
    __shared__ uint8_t sm[MHALO_H][HALO_W];
    __shared__ float lut[256];                                   // x / 255 for every byte: one division per thread
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    lut[tid] = unorm8((uint8_t)tid);
    const int plane = plane_lo + blockIdx.z;
    const BumpParam P = load_param(ParamTab[plane]);
    auto unorm = [&](uint8_t x) { return lut[x]; };
    const GLOBAL_AS uint16_t *__restrict__ Bump = (const GLOBAL_AS uint16_t *)BumpTab[plane];
    const GLOBAL_AS uint8_t *__restrict__ RGBb = (const GLOBAL_AS uint8_t *)RGBTab[plane];
    const GLOBAL_AS uint8_t *__restrict__ Maskb = (const GLOBAL_AS uint8_t *)MaskTab[plane];
    GLOBAL_AS float *__restrict__ VMap = (GLOBAL_AS float *)VMapTab[plane];
    GLOBAL_AS float *__restrict__ RGBA = (GLOBAL_AS float *)RGBATab[plane];
    GLOBAL_AS float *__restrict__ MaskOut = (GLOBAL_AS float *)MaskOutTab[plane];
    GLOBAL_AS float *__restrict__ NMap = (GLOBAL_AS float *)NMapTab[plane];
    GLOBAL_AS uint32_t *__restrict__ Index = (GLOBAL_AS uint32_t *)IndexTab[plane];
    const int j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * MESH_ROWS;
    // a row segment starts at byte 12 (i m + j0): a multiple of 16 for every i iff m % 4 == 0 (j0 is a multiple of 64)
    const bool seg = (m & 3) == 0 && j0 + TILE_W <= m;
    const bool wide_v = seg && ((uintptr_t)VMap & 15) == 0, wide_n = seg && ((uintptr_t)NMap & 15) == 0;
    // The 10 input bytes of a cell sit in three byte streams (6 + 3 + 1): read cell by cell they are 7 narrow strided loads.
    // A full tile (16 rows of 64 own cells, every row segment word-aligned) is fetched as 7 full-width loads per thread
    // into LDS instead -- 16 x (384 + 192 + 64) bytes -- and the own cells' bytes are picked from there (k_vertex_from_bump_batch
    // does the same); only the 164 cells of the halo ring are read narrow.
    __shared__ __align__(16) unsigned char stage[MESH_ROWS * 10 * TILE_W];
    // per wave, one row segment of normals as it lies in memory: in the staging area, which is dead once the tile is decoded
    // (behind the second barrier) -- 26.9 KB of LDS per workgroup instead of 29.9 KB, i.e. six workgroups per CU instead of five
    float (*sn)[3 * TILE_W] = (float (*)[3 * TILE_W])stage;
    static_assert(sizeof(float) * TILE_H * 3 * TILE_W <= MESH_ROWS * 10 * TILE_W, "normals fit in the staging area");
    const bool wide_in = seg && i0 + MESH_ROWS <= n &&
                         (((uintptr_t)Bump & 7) | ((uintptr_t)RGBb & 3) | ((uintptr_t)Maskb & 3)) == 0;
    typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
    vu2 *sB = (vu2 *)stage;                                              // 16 rows x 48 words of 8 bytes
    uint32_t *sC = (uint32_t *)(stage + MESH_ROWS * 6 * TILE_W);        // 16 rows x 48 words of 4 bytes
    uint32_t *sM = (uint32_t *)(stage + MESH_ROWS * 9 * TILE_W);        // 16 rows x 16 words of 4 bytes
    if (wide_in) {
        vu2 rb[3]; uint32_t rc[3], rm;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int w = tid + 256 * q, row = w / 48, col = w % 48;     // 768 words of either kind
            const size_t c0 = (size_t)(i0 + row) * m + j0;
            rb[q] = ((const GLOBAL_AS vu2 *)(Bump + 3 * c0))[col];
            rc[q] = ((const GLOBAL_AS uint32_t *)(RGBb + 3 * c0))[col];
        }
        rm = ((const GLOBAL_AS uint32_t *)(Maskb + (size_t)(i0 + (tid >> 4)) * m + j0))[tid & 15];
#pragma unroll
        for (int q = 0; q < 3; q++) { sB[tid + 256 * q] = rb[q]; sC[tid + 256 * q] = rc[q]; }
        sM[tid] = rm;
    }
    __syncthreads();                                             // the table, the staged inputs
    auto emit = [&](int r, int cc, int ii, int jj, bool own, uint16_t b0, uint16_t b1, uint16_t b2, uint8_t cr, uint8_t cg, uint8_t cb,
                    uint8_t mk) {
        const size_t cell = (size_t)ii * m + jj;
        BumpCell o = decode_cell(b0, b1, b2, cr, cg, cb, mk, P, (float)ii, (float)jj, unorm);
        if (own) {
            if (!wide_v) {
                __builtin_nontemporal_store(o.v.x, VMap + 3 * cell);
                __builtin_nontemporal_store(o.v.y, VMap + 3 * cell + 1);
                __builtin_nontemporal_store(o.v.z, VMap + 3 * cell + 2);
            }
            vfloat4 cv = {o.c.x, o.c.y, o.c.z, o.c.w}, kv = {o.k.x, o.k.y, o.k.z, o.k.w};
            __builtin_nontemporal_store(cv, (GLOBAL_AS vfloat4 *)(RGBA + 4 * cell));
            __builtin_nontemporal_store(kv, (GLOBAL_AS vfloat4 *)(MaskOut + 4 * cell));
        }
        sv[r][1 + 3 * cc] = o.v.x; sv[r][2 + 3 * cc] = o.v.y; sv[r][3 + 3 * cc] = o.v.z;
        sm[r][cc] = mk;
    };
    if (wide_in) {
        const uint16_t *lB = (const uint16_t *)stage;
        const uint8_t *lC = stage + MESH_ROWS * 6 * TILE_W, *lM = stage + MESH_ROWS * 9 * TILE_W;
#pragma unroll
        for (int k = 0; k < MESH_ROWS / TILE_H; k++) {           // the own cells: thread (x, y) decodes (y + 4k, x)
            const int row = (int)threadIdx.y + TILE_H * k, lc = row * TILE_W + (int)threadIdx.x;
            emit(row + 1, (int)threadIdx.x + 1, i0 + row, j0 + (int)threadIdx.x, true, lB[3 * lc], lB[3 * lc + 1], lB[3 * lc + 2],
                 lC[3 * lc], lC[3 * lc + 1], lC[3 * lc + 2], lM[lc]);
        }
        if (tid < 2 * HALO_W + 2 * MESH_ROWS) {                  // the ring: rows 0 and 17 whole, columns 0 and 65 of the others
            int r, cc;
            if (tid < 2 * HALO_W) { r = tid < HALO_W ? 0 : MHALO_H - 1; cc = tid < HALO_W ? tid : tid - HALO_W; }
            else { const int q = tid - 2 * HALO_W; r = 1 + (q >> 1); cc = (q & 1) ? HALO_W - 1 : 0; }
            const int ii = i0 + r - 1, jj = j0 + cc - 1;
            if (ii >= 0 && ii < n && jj >= 0 && jj < m) {
                const size_t cell = (size_t)ii * m + jj;
                emit(r, cc, ii, jj, false, Bump[3 * cell], Bump[3 * cell + 1], Bump[3 * cell + 2], 0, 0, 0, Maskb[cell]);
            } else {
                sv[r][1 + 3 * cc] = 0.0f; sv[r][2 + 3 * cc] = 0.0f; sv[r][3 + 3 * cc] = 0.0f;
                sm[r][cc] = 0;
            }
        }
    } else {
        for (int e = tid; e < MHALO_N; e += TILE_W * TILE_H) {
            int r = e / HALO_W, cc = e % HALO_W;
            int ii = i0 + r - 1, jj = j0 + cc - 1;
            if (ii >= 0 && ii < n && jj >= 0 && jj < m) {
                size_t cell = (size_t)ii * m + jj;
                bool own = r >= 1 && r <= MESH_ROWS && cc >= 1 && cc <= TILE_W;
                uint8_t cr = 0, cg = 0, cb = 0;
                if (own) { cr = RGBb[3 * cell]; cg = RGBb[3 * cell + 1]; cb = RGBb[3 * cell + 2]; }
                emit(r, cc, ii, jj, own, Bump[3 * cell], Bump[3 * cell + 1], Bump[3 * cell + 2], cr, cg, cb, Maskb[cell]);
            } else {
                sv[r][1 + 3 * cc] = 0.0f; sv[r][2 + 3 * cc] = 0.0f; sv[r][3 + 3 * cc] = 0.0f;
                sm[r][cc] = 0;
            }
        }
    }
    __syncthreads();
    const int j = j0 + threadIdx.x, lane = threadIdx.x, wave = threadIdx.y;
    const bool jin = j < m;
    const int cc = threadIdx.x + 1;
    auto lv = [&](int rr, int c2) { F3 q; q.x = sv[rr][1 + 3 * c2]; q.y = sv[rr][2 + 3 * c2]; q.z = sv[rr][3 + 3 * c2]; return q; };
#pragma unroll
    for (int k = 0; k < MESH_ROWS / TILE_H; k++) {
        const int r = wave + TILE_H * k + 1, i = i0 + r - 1;
        if (i >= n) break;                                       // (the same for the whole wave)
        const size_t cell0 = (size_t)i * m + j0, cell = cell0 + lane;
        if (wide_v && lane < 3 * TILE_W / 4)                     // the row's 64 vertices: floats 4 .. 195 of the LDS row
            __builtin_nontemporal_store(*(const vfloat4 *)&sv[r][4 + 4 * lane], (GLOBAL_AS vfloat4 *)(VMap + 3 * cell0) + lane);
        const F3 p = lv(r, cc);
        F3 nr; nr.x = nr.y = nr.z = 0.0f;
        if (jin && !(i < 1 || i > n - 2 || j < 1 || j > m - 2 || p.z == 0.0f))
            nr = normal_from_cross(p, lv(r + 1, cc), lv(r - 1, cc), lv(r, cc + 1), lv(r, cc - 1), inverse);
        if (wide_n) {
            sn[wave][3 * lane] = nr.x; sn[wave][3 * lane + 1] = nr.y; sn[wave][3 * lane + 2] = nr.z;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 3 * TILE_W / 4)
                __builtin_nontemporal_store(*(const vfloat4 *)&sn[wave][4 * lane], (GLOBAL_AS vfloat4 *)(NMap + 3 * cell0) + lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next row overwrites the segment
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else if (jin) {
            __builtin_nontemporal_store(nr.x, NMap + 3 * cell);
            __builtin_nontemporal_store(nr.y, NMap + 3 * cell + 1);
            __builtin_nontemporal_store(nr.z, NMap + 3 * cell + 2);
        }
        if (jin && i <= n - 2 && j <= m - 2) {
            uint32_t v0 = (uint32_t)cell, v1 = v0 + 1, v2 = v0 + m + 1, v3 = v0 + m;
            bool cut = corner_rejected(sm[r][cc], min_conf, variant) || corner_rejected(sm[r][cc + 1], min_conf, variant) ||
                       corner_rejected(sm[r + 1][cc + 1], min_conf, variant) || corner_rejected(sm[r + 1][cc], min_conf, variant) ||
                       quad_trimmed(p, lv(r, cc + 1), lv(r + 1, cc + 1), lv(r + 1, cc), thresh);
            uint4 o = quad_record(cut, v0, v1, v2, v3, variant);
            vuint4 ov = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(ov, (GLOBAL_AS vuint4 *)(Index + 4 * ((size_t)i * (m - 1) + j)));
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
