// frame.hip -- the fused frame pipeline: each kernel here does the work of one METHOD of the
// reference's L3 classes (OffLineFrame::LoadFrame, PredictedFrame::Transform+ReadFrame+ProjectMask+
// ProjectLabel, PredictedFrame::Merge, PredictedFrame::Cpy+InitMask) instead of one kernel per
// operator, with the per-pixel formulas of device_fns.hpp (bit-identical to the per-operator path).
//
// Why (tools/microbench/latency.hip, profiles/): at 640x480 every operator kernel moves 4-25 MB and
// lasts 3-12 us -- the frame is bound by the NUMBER of dependent launches, not by bandwidth.  Fusing
// removes ~20 launches per frame and the intermediate maps (texture, transformed VMap, index buffer)
// never reach HBM.  Neighbourhood operators (normals, Sobel) read an LDS tile with a one-pixel halo.
//
// The tracker state lives in a device-resident rgbdslam_pose_block: the kernels read the poses that
// the ICP loop produced from device memory and gate themselves on its `tracked` flag, so a frame is
// a fixed launch sequence without any host round trip (the reference syncs after every operator and
// reads the 27 sums back 12 times per frame).
#include "device_fns.hpp"

__device__ __forceinline__ bool pb_tracked(const rgbdslam_pose_block *pb) { return pb->status[0] == 1; }

// ---- OffLineFrame::LoadFrame + grey + 2 x Sobel: frame_preprocess_tile (device_fns.hpp) ---------------
__global__ __launch_bounds__(TILE_W * TILE_H)
void k_frame_preprocess(PreprocessArgs a) { frame_preprocess_tile(a, blockIdx.x, blockIdx.y); }

struct ReseedArgs {                                  // model := input frame (set A), used when the tracker reports a loss
    const float *fV, *fN, *fRGB;
    const uint16_t *depth;
    size_t dpitch;
    float *VMap, *NMap, *RGBD;
    uint8_t *Mask;
};
__device__ __forceinline__ void reseed_pixel(const ReseedArgs &r, const CalibV &c, int i, int j, int n, int m) {
    size_t idx = (size_t)i * m + j;
    F3 v = ld3(r.fV, idx), nn = ld3(r.fN, idx), col = ld3(r.fRGB, idx);
    st3(r.VMap, idx, v.x, v.y, v.z);
    st3(r.NMap, idx, nn.x, nn.y, nn.z);
    float4 o;
    o.x = col.x; o.y = col.y; o.z = col.z;
    o.w = zbuff_w(((const uint16_t *)((const char *)r.depth + (size_t)(n - i - 1) * r.dpitch))[j], c);
    *(float4 *)(r.RGBD + 4 * idx) = o;
    r.Mask[idx] = (o.w == 0.0f) ? 10 : 11;                         // Kernel.cu:624-627
}

// ---- PredictedFrame::Cpy + InitMask (D1/Frame.cpp:335-370): model := input frame --------------------
// runs when forced (first frame) or when the tracker reported a loss (GP-demo.cpp:164-167,189-193)
__global__ void k_model_reseed(const rgbdslam_pose_block *__restrict__ pb, int force, const float *__restrict__ fV,
                               const float *__restrict__ fN, const float *__restrict__ fRGB,
                               const uint16_t *__restrict__ depth, size_t dpitch, float *__restrict__ VMap,
                               float *__restrict__ NMap, float *__restrict__ RGBD, uint8_t *__restrict__ Mask, CalibV c,
                               int n, int m) {
    if (!force && pb_tracked(pb)) return;
    int j = blockIdx.x * TILE_W + threadIdx.x, i = blockIdx.y * TILE_H + threadIdx.y;
    if (i >= n || j >= m) return;
    ReseedArgs r = {fV, fN, fRGB, depth, dpitch, VMap, NMap, RGBD, Mask};
    reseed_pixel(r, c, i, j, n, m);
}

// ---- PredictedFrame::Transform + the GL draw (Frame.cpp:281-298, GP-demo.cpp:98-113) ------------------
// one lane per quad: the four vertices are warped by P_rel in registers (TransformVertexMapKernel),
// trimmed (QuadTrimFrameKernel, on the warped vertices like the reference) and z-buffered.
__device__ __forceinline__ F3 warp_vertex(const float *P, F3 v) {      // selects, not a branch around the nine products
    const F3 w = xf_point(P, v);
    const bool z = is_zero3(v);
    F3 r; r.x = z ? v.x : w.x; r.y = z ? v.y : w.y; r.z = z ? v.z : w.z;
    return r;
}

// grid over n x m pixels: quad (i, j) for i < n-1, j < m-1
// The kernel is VALU-issue bound (822 instructions per wave at 1920 x 1080 = 43 us of issue time, 43.5 us measured; the
// z-buffer atomics are ~1 per pixel and do not show), so a vertex is warped and projected ONCE per workgroup, not once per
// quad that uses it: the 65 x 5 vertices of a 64 x 4 quad tile go through LDS (two f32 divisions, one f64 reciprocal and
// the 3x4 warp per vertex: ~110 instructions, four times per quad before).
#define PD_ROWS (TILE_H + 1)
#define PD_COLS (TILE_W + 1)
struct PdVert { F3 v; RasVert p; };
// What the draw leaves per vertex for the resolve when the caller gives the room (vertex_scratch: 16 + 4 bytes per pixel):
// the projection (1/256-pixel position, reciprocal depth) and the warped depth.  The resolve then shades the winning
// triangle from three of these records instead of warping and projecting its vertices a second time (two f32 divisions and
// one f64 reciprocal each: 207 of its 578 instructions).
struct ProjVert { int X, Y; double inv_d; };
static_assert(sizeof(ProjVert) == 16, "ProjVert");
__device__ __forceinline__ PdVert pd_vertex(const float *P, const CalibV &c, F3 v, bool exists) {
    PdVert o;
    o.v = warp_vertex(P, v);
    o.p = ras_project(o.v, c);
    if (!exists) { o.v.x = o.v.y = o.v.z = 0.0f; o.p.X = 0; o.p.Y = 0; o.p.inv_d = 0.0; o.p.ok = false; }
    return o;
}
__global__ __launch_bounds__(TILE_W * TILE_H)      // (70 registers = 7 waves per SIMD; held to 64 for 8 it spills four and gains nothing)
void k_predict_draw(const rgbdslam_pose_block *__restrict__ pb, const float *__restrict__ VMap,
                    unsigned long long *__restrict__ zbuf, float thresh, CalibV c, float znear, float zfar,
                    UpdateArgs upd, int rows_draw, ProjVert *__restrict__ pv, float *__restrict__ zw, int n, int m) {
    // Tile rows from rows_draw on (if the grid has them): the APPLY pass of the PREVIOUS frame's Graph::Update, deferred by
    // the caller (posed by that frame's history entry).  It reads what this launch reads -- the model the previous frame
    // left -- and the planes; nothing in this launch writes either.
    if ((int)blockIdx.y >= rows_draw) { update_tile<true>(upd, blockIdx.x, (int)blockIdx.y - rows_draw); return; }
    // structure of arrays, one word per lane and field: no bank conflicts
    __shared__ float s_v[3][PD_ROWS][PD_COLS];
    __shared__ int s_xy[2][PD_ROWS][PD_COLS];
    __shared__ double s_inv[PD_ROWS][PD_COLS];
    __shared__ uint8_t s_ok[PD_ROWS][PD_COLS];
    const int j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * TILE_H;
    const int j = j0 + threadIdx.x, i = i0 + threadIdx.y;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    // The tracked flag was written by the launch before this one: reading it is a memory round trip.  The tile's
    // vertices are requested in the same round trip; the flag is only looked at before the first side effect.
    const bool tracked = pb_tracked(pb);
    const float *P = pb->P_rel;
    // vertex (threadIdx.y, threadIdx.x) of the tile, and for the first 69 threads one of the extra row / column
    const bool ex1 = i < n && j < m;
    const bool second = tid < PD_COLS + TILE_H;
    const int r2 = tid < PD_COLS ? TILE_H : tid - PD_COLS, c2 = tid < PD_COLS ? tid : TILE_W;
    const bool ex2 = second && i0 + r2 < n && j0 + c2 < m;
    F3 v1 = ld3(VMap, ex1 ? (size_t)i * m + j : 0);
    F3 v2 = ld3(VMap, ex2 ? (size_t)(i0 + r2) * m + (j0 + c2) : 0);
    pin_loaded(v1); pin_loaded(v2);
    if (!tracked) return;                             // (the resolve launch re-seeds the model)
    {
        const PdVert a = pd_vertex(P, c, v1, ex1);
        if (pv && ex1) {                              // every vertex of the image has exactly one owner
            ProjVert o; o.X = a.p.X; o.Y = a.p.Y; o.inv_d = a.p.inv_d;
            pv[(size_t)i * m + j] = o;
            zw[(size_t)i * m + j] = a.v.z;
        }
        s_v[0][threadIdx.y][threadIdx.x] = a.v.x; s_v[1][threadIdx.y][threadIdx.x] = a.v.y; s_v[2][threadIdx.y][threadIdx.x] = a.v.z;
        s_xy[0][threadIdx.y][threadIdx.x] = a.p.X; s_xy[1][threadIdx.y][threadIdx.x] = a.p.Y;
        s_inv[threadIdx.y][threadIdx.x] = a.p.inv_d; s_ok[threadIdx.y][threadIdx.x] = a.p.ok ? 1 : 0;
    }
    if (second) {
        const PdVert a = pd_vertex(P, c, v2, ex2);
        s_v[0][r2][c2] = a.v.x; s_v[1][r2][c2] = a.v.y; s_v[2][r2][c2] = a.v.z;
        s_xy[0][r2][c2] = a.p.X; s_xy[1][r2][c2] = a.p.Y;
        s_inv[r2][c2] = a.p.inv_d; s_ok[r2][c2] = a.p.ok ? 1 : 0;
    }
    __syncthreads();
    if (i > n - 2 || j > m - 2) return;               // no quad here
    // triangle ids count in PIXEL indices (2 * (i * m + j) + t, not the quad index i * (m - 1) + j): the same order, so the
    // same winner on ties, and the resolve gets the quad's first vertex with a shift instead of a division by m - 1
    const uint32_t q = (uint32_t)(i * m + j);
    // corners in the order of quad_vertex_ids: (i, j) (i, j+1) (i+1, j+1) (i+1, j)
    const int rr[4] = {(int)threadIdx.y, (int)threadIdx.y, (int)threadIdx.y + 1, (int)threadIdx.y + 1};
    const int cc[4] = {(int)threadIdx.x, (int)threadIdx.x + 1, (int)threadIdx.x + 1, (int)threadIdx.x};
    F3 w[4];
    RasVert p[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w[k].x = s_v[0][rr[k]][cc[k]]; w[k].y = s_v[1][rr[k]][cc[k]]; w[k].z = s_v[2][rr[k]][cc[k]];
        p[k].X = s_xy[0][rr[k]][cc[k]]; p[k].Y = s_xy[1][rr[k]][cc[k]];
        p[k].inv_d = s_inv[rr[k]][cc[k]]; p[k].ok = s_ok[rr[k]][cc[k]] != 0;
    }
    if (quad_trimmed(w[0], w[1], w[2], w[3], thresh)) return;
    // triangles (0,1,2) and (0,2,3)
    RasTri T;
    if (ras_setup_projected(p[0], p[1], p[2], T)) ras_draw_tri(zbuf, T, 2 * q, znear, zfar, n, m);
    if (ras_setup_projected(p[0], p[2], p[3], T)) ras_draw_tri(zbuf, T, 2 * q + 1, znear, zfar, n, m);
}

// ---- resolve + PredictedFrame::ReadFrame + ProjectMask + the gather of ProjectLabel -------------------
// (Frame.cpp:229-245,372-393).  Reads model set "in", writes model set "out" (ping-pong): the texel
// never reaches memory.  Also resets the z-buffer word.
// LT = element type of the label images: uint8_t (the reference's, Frame.h:197) or uint16_t (rgbdslam_set_label_width 2)
template <bool PROJ, typename LT>
__global__ void k_predict_resolve(const rgbdslam_pose_block *__restrict__ pb, const float *__restrict__ VMap_in,
                                  const float *__restrict__ RGBD_in, const uint8_t *__restrict__ Mask_in,
                                  const LT *__restrict__ Label_in, unsigned long long *__restrict__ zbuf,
                                  float *__restrict__ VMap_out, float *__restrict__ RGBD_out, uint8_t *__restrict__ Mask_out,
                                  LT *__restrict__ Label_out, CalibV c, ReseedArgs rs, const ProjVert *__restrict__ pv,
                                  const float *__restrict__ zw, int n, int m) {
    const bool tracked = pb_tracked(pb);               // round trip shared with the z-buffer word (see k_predict_draw)
    int px = blockIdx.x * TILE_W + threadIdx.x, py = blockIdx.y * TILE_H + threadIdx.y;
    if (py >= n || px >= m) return;
    size_t p = (size_t)py * m + px;
    unsigned long long key = zbuf[p];
    asm volatile("" : "+v"(key));
    // when the tracker lost the frame this grid re-seeds the model (set "in") instead (GP-demo.cpp:189-193) -- no separate
    // launch that usually does nothing; nothing was drawn, the z-buffer is still clear
    if (!tracked) {
        if (rs.fV) reseed_pixel(rs, c, py, px, n, m);
        return;
    }
    float4 tex; tex.x = tex.y = tex.z = tex.w = 0.0f;
    if (key != ZBUF_EMPTY) {
        zbuf[p] = ZBUF_EMPTY;
        const float *P = pb->P_rel;
        const uint32_t tri_id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull), v0 = tri_id >> 1;   // see k_predict_draw
        uint32_t vid[3];
        vid[0] = v0;
        if (tri_id & 1) { vid[1] = v0 + (uint32_t)m + 1u; vid[2] = v0 + (uint32_t)m; } else { vid[1] = v0 + 1u; vid[2] = v0 + (uint32_t)m + 1u; }
        float4 att[3];
        if (PROJ) {
            int4 pr[3];
            float z[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                pr[k] = *(const int4 *)(pv + vid[k]);
                z[k] = zw[vid[k]];
                att[k] = *(const float4 *)(RGBD_in + 4 * (size_t)vid[k]);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {                          // nine loads in flight
                asm volatile("" : "+v"(pr[k].x), "+v"(pr[k].y), "+v"(pr[k].z), "+v"(pr[k].w), "+v"(z[k]));
                pin_loaded(att[k]);
            }
            RasVert q[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                q[k].X = pr[k].x; q[k].Y = pr[k].y; q[k].ok = true;
                q[k].inv_d = __hiloint2double(pr[k].w, pr[k].z);
                att[k].w = depth_attr(z[k]);                       // Kernel.cu:368-371
            }
            tex = ras_shade_projected(q[0], q[1], q[2], att, px, py);
        } else {
            F3 vtx[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                vtx[k] = ld3(VMap_in, vid[k]);
                att[k] = *(const float4 *)(RGBD_in + 4 * (size_t)vid[k]);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { pin_loaded(vtx[k]); pin_loaded(att[k]); }   // six loads in flight, then the warps
#pragma unroll
            for (int k = 0; k < 3; k++) {
                vtx[k] = warp_vertex(P, vtx[k]);
                att[k].w = depth_attr(vtx[k].z);                   // Kernel.cu:368-371
            }
            tex = ras_shade(vtx, att, c, px, py);
        }
    }
    F3 v; v.x = v.y = v.z = 0.0f;
    float4 o; o.x = o.y = o.z = o.w = 0.0f;
    if (tex.w != 0.0f) { v = vertex_from_texel(tex.w, py, px, c); o = tex; }   // Kernel.cu:321-341
    st3(VMap_out, p, v.x, v.y, v.z);
    *(float4 *)(RGBD_out + 4 * p) = o;
    long src = project_source(pb->P_rel_inv, c, v, n, m);          // Kernel.cu:634-667, for Mask and Label
    Mask_out[p] = (src < 0) ? (uint8_t)0 : Mask_in[src];
    Label_out[p] = (src < 0) ? (LT)0 : Label_in[src];
}

// ---- ComputeNormal of ReadFrame (Frame.cpp:242) + first SegmentKernel pass (RG.cu:252-320) ------------
// Eq == nullptr (no planes yet): normals only, labels copied.  Tile functions, so that launches can carry two jobs.
template <typename LT>
struct SegArgsT {
    const rgbdslam_pose_block *pb;
    float *NMap;                 // written by the first pass, read by the others
    const float *VMap;
    const LT *in; LT *out;
    const float *Eq; float epsilon, alpha;
    int n, m;
    int lazy_normals;            // 1: NMap is scratch of the growth passes only -- written (and computed) where a pixel has no label
    int nb_eq;                   // planes in Eq (k_segment3_merge_claim stages them in LDS)
};
// The normals of the predicted frame feed nothing but the growth passes (the reference computes them in ReadFrame,
// Frame.cpp:242, and Merge overwrites them, Frame.cpp:331), and a growth pass looks at the normal of UNLABELED pixels only --
// about 1 % of the image in the steady state.  When NMap is the passes' private scratch (lazy_normals) the ~330 instructions
// of the four-triangle normal are spent only there; the labels that come out are the same.
template <typename LT>
__device__ __forceinline__ void normal_segment_tile(const SegArgsT<LT> &a, const int bx, const int by) {
    const bool tracked = pb_tracked(a.pb);             // round trip shared with the pixel's own loads
    int j = bx * TILE_W + threadIdx.x, i = by * TILE_H + threadIdx.y;
    if (i >= a.n || j >= a.m) return;
    size_t idx = (size_t)i * a.m + j;
    LT mine = a.in[idx];
    if (a.lazy_normals) {
        if (!tracked) return;
        if (mine == 0) {
            const F3 nl = normal_global(a.VMap, i, j, a.n, a.m, 1);
            st3(a.NMap, idx, nl.x, nl.y, nl.z);
            if (a.Eq) mine = segment_pixel(a.in, a.VMap, nl, a.pb->P_glob, a.Eq, a.epsilon, a.alpha, i, j, a.n, a.m);
        }
        a.out[idx] = mine;
        return;
    }
    F3 nl = normal_global(a.VMap, i, j, a.n, a.m, 1);
    if (!tracked) return;
    st3(a.NMap, idx, nl.x, nl.y, nl.z);
    if (mine == 0 && a.Eq) mine = segment_pixel(a.in, a.VMap, nl, a.pb->P_glob, a.Eq, a.epsilon, a.alpha, i, j, a.n, a.m);
    a.out[idx] = mine;
}
// one later pass; returns the pixel's label (0 outside the image / when not tracked: `done` tells)
template <typename LT>
__device__ __forceinline__ LT segment_pass_tile(const SegArgsT<LT> &a, const int bx, const int by, bool &done, size_t &idx_out) {
    const bool tracked = pb_tracked(a.pb);             // round trip shared with the pixel's own loads
    int j = bx * TILE_W + threadIdx.x, i = by * TILE_H + threadIdx.y;
    done = false; idx_out = 0;
    if (i >= a.n || j >= a.m) return 0;
    size_t idx = (size_t)i * a.m + j;
    int mine_w = a.in[idx];
    asm volatile("" : "+v"(mine_w));                   // the load stays above the flag test
    if (!tracked) return 0;
    LT mine = (LT)mine_w;
    if (mine == 0) mine = segment_pixel(a.in, a.VMap, ld3(a.NMap, idx), a.pb->P_glob, a.Eq, a.epsilon, a.alpha, i, j, a.n, a.m);
    a.out[idx] = mine;
    done = true; idx_out = idx;
    return mine;
}
template <typename LT>
__global__ void k_normal_segment(SegArgsT<LT> a) { normal_segment_tile(a, blockIdx.x, blockIdx.y); }
template <typename LT>
__global__ void k_segment_pass_pb(SegArgsT<LT> a) {
    bool done; size_t idx;
    (void)segment_pass_tile(a, blockIdx.x, blockIdx.y, done, idx);
}

// ---- PredictedFrame::Merge (Frame.cpp:300-333): MergeDepthOff + VertexMapGL + ComputeNormal -----------
// set "in" -> set "out"; the merged vertices of the tile + halo are staged in LDS for the normals.
// Claim pass of Graph::Update (k_update<false>, D1/Primitive.cu:267-357 under the winner rule of DESIGN.md 4.2) for
// the pixel that was just merged: it needs only that pixel's new vertex and mask, so it rides on the merge instead
// of costing a launch of its own.  ParamTab == nullptr: no planes / plain merge.
struct MergeClaimArgs {
    float *const *ParamTab;
    uint8_t *const *MaskTab;
    unsigned long long *const *ClaimTab;
    const uint8_t *Label;
    const uint16_t *Label16;     // non-null: 16-bit labels, Label unused (k_model_merge with a claim)
    uint32_t stamp;
    int N_prim, M_prim, plane_lo, plane_hi;
};
// `in`: this lane has a pixel (lanes outside the image take part in the wave-wide votes only)
// label = the pixel's plane label (0 = none)
__device__ __forceinline__ void merge_claim(const MergeClaimArgs &ca, const float *P_glob, F3 v, uint8_t mc, size_t idx, bool in, int label) {
    const int L = in ? label - 1 : -1;
    const bool valid = L >= 0 && L >= ca.plane_lo && L < ca.plane_hi;
    const unsigned long long vmask = __ballot(valid);
    if (!vmask) return;
    // wave-uniform plane (the usual case: a 64-pixel row segment lies on one plane): table entries and the 14 plane
    // parameters come through uniform addresses (scalar loads) instead of 16 dependent vector loads per lane
    const int L0 = __builtin_amdgcn_readfirstlane(__shfl(L, (int)__builtin_ctzll(vmask)));
    const bool one_plane = __ballot(valid && L != L0) == 0ull;
    bool want = false;
    unsigned long long key = 0ull, *slot = nullptr;
    if (valid) {
        UpdCell u;
        const uint8_t *maskb;
        unsigned long long *claimb;
        const F3 p = xf_point(P_glob, v);
        bool hit;
        if (one_plane) {
            const float *Q = ca.ParamTab[L0];
            float q[14];
#pragma unroll
            for (int k = 0; k < 14; k++) q[k] = Q[k];
            maskb = ca.MaskTab[L0]; claimb = ca.ClaimTab[L0];
            hit = update_cell(p, q, ca.N_prim, ca.M_prim, u);
        } else {
            hit = update_cell(p, ca.ParamTab[L], ca.N_prim, ca.M_prim, u);
            maskb = ca.MaskTab[L]; claimb = ca.ClaimTab[L];
        }
        if (hit) {
            const uint8_t mref = maskb[u.cell];
            const bool erode = (mc < 10 && mref != 1);                        // Primitive.cu:313
            if (erode || mc > mref) {                                         // Primitive.cu:341
                want = true;
                key = claim_key(ca.stamp, !erode, erode ? (uint8_t)0 : mc, (uint32_t)idx);
                slot = claimb + u.cell;
            }
        }
    }
    // Neighbouring pixels of a row often fall into the same cell (always, once the image is denser than the 4 mm
    // cells).  On equal confidence the lower pixel index wins, i.e. the LEFT neighbour: a lane whose left neighbour
    // claims the same cell with a key that is not smaller leaves the atomic to it; the maximum that reaches the
    // cell is unchanged.
    const int lane = threadIdx.x & 63;
    const unsigned long long live = __ballot(want);
    const unsigned long long slot_l = __shfl_up((unsigned long long)slot, 1), key_l = __shfl_up(key, 1);
    const bool want_l = lane > 0 && ((live >> (lane - 1)) & 1ull);
    if (want && !(want_l && slot_l == (unsigned long long)slot && key_l >= key)) atomicMax(slot, key);
}

// ---- the three growth passes of ONE tile out of LDS (RG.cu:905-911 runs SegmentKernel three times over the image) ------
// A pass only looks at the 3x3 neighbourhood, so the labels a tile ends with depend on the labels of the tile + 3 rings
// before the first pass: those 70 x 10 bytes are staged in LDS and the passes run there -- the same function of the same
// inputs as three launches over the image, without two of the launches and without the label / normal images in between.
// 99.5 % of the pixels carry a label in the steady state (tools/label_stats.py) and a pass leaves them alone: the pixels
// WITHOUT one (tile + 2 rings) are queued, one lane per queued pixel computes what does not change from pass to pass
// (normal, the pixel and its normal in the global frame, which neighbours are within 3 mm of depth: ONE memory round
// trip, all loads unconditional) and then only walks the neighbours' labels three times.  A tile without such pixels
// skips all of it.  (Evaluating the passes pixel by pixel over the shrinking regions instead -- 544 + 396 + 256 pixels
// in six rounds -- serialised up to six scans of four dependent round trips each per wave: 25.8 us at 640 x 480, no
// better than the three launches.)
#define SG_RING 3
#define SG_EQ_LDS 64                                 // planes whose equations are staged in LDS (the others are read from memory)
#define SG_W (TILE_W + 2 * SG_RING)
#define SG_H (TILE_H + 2 * SG_RING)
#define SG_WP (SG_W + 2)
#define SG_QMAX ((TILE_W + 4) * (TILE_H + 4))
#define SG_QCHUNKS ((SG_QMAX + TILE_W * TILE_H - 1) / (TILE_W * TILE_H))
template <typename LT>
struct SegTileLdsT {
    LT lab[SG_H][SG_WP];
    unsigned int qn;
    unsigned short qpos[SG_QMAX];                    // r * SG_WP + c of the queued pixel's cell in lab
    float eq[10 * SG_EQ_LDS];                        // the first planes' equations (a pass reads up to nine of them per pixel)
};
// what a lane keeps about a queued pixel from pass to pass (registers: entry tid + 256 * chunk belongs to lane tid)
struct SegQueued {
    F3 p, q;                                         // the pixel and its normal in the global frame
    unsigned int zmask;                              // bit k: neighbour k's depth is within 3 mm (0: the pixel can never get a label)
    int pos;
};
// Labels of the tile + 3 rings: requested before the workgroup's first barrier (seg3_request), staged and queued after it
// (seg3_stage; S.qn was zeroed before that barrier).
template <typename LT>
struct SegTileRegsT { LT l[3]; bool queue[3]; float eq0; };
template <typename LT>
__device__ __forceinline__ SegTileRegsT<LT> seg3_request(const SegArgsT<LT> &a, const int i0, const int j0, const int tid, SegTileLdsT<LT> &S) {
    if (tid == 0) S.qn = 0u;
    SegTileRegsT<LT> R;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int e = tid + k * TILE_W * TILE_H;
        const int r = e / SG_W, c = e % SG_W;
        const int ii = i0 - SG_RING + r, jj = j0 - SG_RING + c;
        const bool ok = e < SG_H * SG_W && ii >= 0 && ii < a.n && jj >= 0 && jj < a.m;
        R.l[k] = ok ? a.in[(size_t)ii * a.m + jj] : (LT)0;
        // tile + 2 rings; a pixel on the image border has a zero normal (ComputeNormal) and can never get a label
        R.queue[k] = ok && r >= 1 && r < SG_H - 1 && c >= 1 && c < SG_W - 1 && ii >= 1 && ii <= a.n - 2 && jj >= 1 && jj <= a.m - 2;
    }
    R.eq0 = a.Eq[tid < 10 * min(a.nb_eq, SG_EQ_LDS) ? tid : 0];
    return R;
}
// returns whether the tile + 3 rings hold any label at all
template <typename LT>
__device__ __forceinline__ bool seg3_stage(const SegArgsT<LT> &a, const SegTileRegsT<LT> &R, const int tid, SegTileLdsT<LT> &S) {
    bool any = false;
    const int neq = 10 * min(a.nb_eq, SG_EQ_LDS);
    if (tid < neq) S.eq[tid] = R.eq0;
    for (int e = tid + TILE_W * TILE_H; e < neq; e += TILE_W * TILE_H) S.eq[e] = a.Eq[e];   // (more than 25 planes)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int e = tid + k * TILE_W * TILE_H;
        const int r = e / SG_W, c = e % SG_W;
        if (e < SG_H * SG_W) S.lab[r][c] = R.l[k];
        // one LDS atomic per wave, not per pixel: a tile without labels queues all its 544
        const bool push = R.queue[k] && R.l[k] == 0;
        const unsigned long long pm = __ballot(push);
        if (pm) {
            const int lane = tid & 63;
            unsigned int base = 0u;
            if (lane == (int)__builtin_ctzll(pm)) base = atomicAdd(&S.qn, (unsigned int)__builtin_popcountll(pm));
            base = (unsigned int)__shfl((int)base, (int)__builtin_ctzll(pm));
            if (push) S.qpos[base + (unsigned int)__builtin_popcountll(pm & ((1ull << lane) - 1ull))] = (unsigned short)(r * SG_WP + c);
        }
        any |= (e < SG_H * SG_W) && R.l[k] != 0;
    }
    // labels only spread from labelled pixels, three steps at most: a tile whose 70 x 10 labels are all 0 stays that way
    return __syncthreads_or(any ? 1 : 0) != 0;
}
// after seg3_stage.  Returns the label of the thread's own pixel after the third pass.
template <typename LT>
__device__ __forceinline__ LT seg3_passes(const SegArgsT<LT> &a, const int i0, const int j0, const int tid, SegTileLdsT<LT> &S,
                                          const bool any_label) {
    const int nq = any_label ? (int)S.qn : 0;        // uniform
    LT *const L = &S.lab[0][0];
    if (nq > 0) {
        const float *__restrict__ V = a.VMap;
        const float *P = a.pb->P_glob;
        const int m = a.m;
        SegQueued Q[SG_QCHUNKS];
#pragma unroll
        for (int ch = 0; ch < SG_QCHUNKS; ch++) {
            const int e = tid + ch * TILE_W * TILE_H;
            Q[ch].zmask = 0u; Q[ch].pos = 0;
            Q[ch].p.x = Q[ch].p.y = Q[ch].p.z = Q[ch].q.x = Q[ch].q.y = Q[ch].q.z = 0.0f;
            if (e < nq) {                                                     // (wave-uniform but for one wave)
                const int pos = S.qpos[e];
                const int ii = i0 - SG_RING + pos / SG_WP, jj = j0 - SG_RING + pos % SG_WP;   // an interior pixel of the image
                const size_t idx = (size_t)ii * m + jj;
                // the 3x3 block around the pixel in one round trip
                const size_t up = idx - m, dn = idx + m;
                F3 pc = ld3(V, idx), pu = ld3(V, up), pd = ld3(V, dn), plf = ld3(V, idx - 1), prt = ld3(V, idx + 1);
                float zc[4] = {V[3 * (up - 1) + 2], V[3 * (up + 1) + 2], V[3 * (dn - 1) + 2], V[3 * (dn + 1) + 2]};
                pin_loaded(pc); pin_loaded(pu); pin_loaded(pd); pin_loaded(plf); pin_loaded(prt);
                asm volatile("" : "+v"(zc[0]), "+v"(zc[1]), "+v"(zc[2]), "+v"(zc[3]));
                F3 nl; nl.x = nl.y = nl.z = 0.0f;
                if (pc.z != 0.0f) nl = normal_from_cross(pc, pd, pu, prt, plf, 1);      // normal_global
                const F3 q = xf_dir(P, nl);
                const float dz[9] = {zc[0], pu.z, zc[1], plf.z, pc.z, prt.z, zc[2], pd.z, zc[3]};
                unsigned int zmask = 0u;
#pragma unroll
                for (int k = 0; k < 9; k++) zmask |= (__builtin_fabsf(dz[k] - pc.z) < 0.003f) ? (1u << k) : 0u;
                Q[ch].zmask = is_zero3(q) ? 0u : zmask;
                Q[ch].p = xf_point(P, pc); Q[ch].q = q; Q[ch].pos = pos;
            }
        }
        for (int pass = 0; pass < 3; pass++) {
            LT got[SG_QCHUNKS];
#pragma unroll
            for (int ch = 0; ch < SG_QCHUNKS; ch++) {
                got[ch] = 0;
                const int pos = Q[ch].pos;
                const unsigned int zmask = Q[ch].zmask;
                if (zmask != 0u && L[pos] == 0) {
                    const F3 p = Q[ch].p, q = Q[ch].q;
                    LT res = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) {                            // row by row, the first neighbour that qualifies (RG.cu:285-311)
                        const LT l = L[pos + (k / 3 - 1) * SG_WP + (k % 3 - 1)];
                        if (res == 0 && l > 0 && ((zmask >> k) & 1u)) {
                            const float *E = (l <= SG_EQ_LDS ? (const float *)S.eq : a.Eq) + 10 * ((int)l - 1);
                            const float ed = ((p.x * E[0] + p.y * E[1]) + p.z * E[2]) - E[9];
                            const float ea = (q.x * E[0] + q.y * E[1]) + q.z * E[2];
                            if (!(__builtin_fabsf(ed) > a.epsilon || ea < a.alpha)) res = l;
                        }
                    }
                    got[ch] = res;
                }
            }
            __syncthreads();                                                  // every lane has read the labels of this pass
#pragma unroll
            for (int ch = 0; ch < SG_QCHUNKS; ch++)
                if (got[ch] != 0) L[Q[ch].pos] = got[ch];
            __syncthreads();
        }
    }
    const int ii = i0 + (int)threadIdx.y, jj = j0 + (int)threadIdx.x;
    return (ii < a.n && jj < a.m) ? S.lab[threadIdx.y + SG_RING][threadIdx.x + SG_RING] : (LT)0;
}

struct MergeArgs {
    const rgbdslam_pose_block *pb;
    const float *RGBD_in; const uint8_t *Mask_in; const uint16_t *depth; size_t dpitch; const float *RGB;
    float *RGBD_out; uint8_t *Mask_out; float *VMap_out, *NMap_out;
    CalibV c;
    MergeClaimArgs ca;
    int n, m;
};
__device__ __forceinline__ void model_merge_tile(const MergeArgs &a, const int bx, const int by) {
    const rgbdslam_pose_block *__restrict__ pb = a.pb;
    const float *__restrict__ RGBD_in = a.RGBD_in; const uint8_t *__restrict__ Mask_in = a.Mask_in;
    const uint16_t *__restrict__ depth = a.depth; const size_t dpitch = a.dpitch; const float *__restrict__ RGB = a.RGB;
    float *__restrict__ RGBD_out = a.RGBD_out; uint8_t *__restrict__ Mask_out = a.Mask_out;
    float *__restrict__ VMap_out = a.VMap_out, *__restrict__ NMap_out = a.NMap_out;
    const CalibV &c = a.c;
    const MergeClaimArgs &ca = a.ca;
    const int n = a.n, m = a.m;
    if (!pb_tracked(pb)) return;
    __shared__ float sv[HALO_H][HALO_W][3];
    __shared__ uint8_t smask[TILE_H][TILE_W];         // merged mask of the tile's own pixels, for the claim pass
    const int j0 = bx * TILE_W, i0 = by * TILE_H;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int e = tid; e < HALO_N; e += TILE_W * TILE_H) {
        int r = e / HALO_W, cc = e % HALO_W;
        int ii = i0 + r - 1, jj = j0 + cc - 1;
        F3 v; v.x = v.y = v.z = 0.0f;
        if (ii >= 0 && ii < n && jj >= 0 && jj < m) {
            size_t idx = (size_t)ii * m + jj;
            uint16_t d = ((const uint16_t *)((const char *)depth + (size_t)(n - ii - 1) * dpitch))[jj];
            uint8_t mk = Mask_in[idx];
            int newmask;
            bool own = r >= 1 && r <= TILE_H && cc >= 1 && cc <= TILE_W;
            float w_old = own ? 0.0f : RGBD_in[4 * idx + 3];
            float4 px;
            if (own) { px = *(const float4 *)(RGBD_in + 4 * idx); w_old = px.w; }
            float w = merge_pixel(w_old, d, mk, c, newmask);
            v = vertex_from_w(w, ii, jj, c);
            if (own) {
                F3 col = ld3(RGB, idx);
                px.x = col.x; px.y = col.y; px.z = col.z; px.w = w;
                *(float4 *)(RGBD_out + 4 * idx) = px;
                const uint8_t mc = (newmask >= 0) ? (uint8_t)newmask : mk;
                Mask_out[idx] = mc;
                st3(VMap_out, idx, v.x, v.y, v.z);
                smask[r - 1][cc - 1] = mc;
            }
        }
        sv[r][cc][0] = v.x; sv[r][cc][1] = v.y; sv[r][cc][2] = v.z;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x, i = i0 + threadIdx.y;
    const bool in = i < n && j < m;
    const int r = threadIdx.y + 1, cc = threadIdx.x + 1;
    F3 p; p.x = sv[r][cc][0]; p.y = sv[r][cc][1]; p.z = sv[r][cc][2];
    if (in) {
        F3 nr; nr.x = nr.y = nr.z = 0.0f;
        if (!(i < 1 || i > n - 2 || j < 1 || j > m - 2 || p.z == 0.0f)) {
            F3 dn, up, rt, lf;
            dn.x = sv[r + 1][cc][0]; dn.y = sv[r + 1][cc][1]; dn.z = sv[r + 1][cc][2];
            up.x = sv[r - 1][cc][0]; up.y = sv[r - 1][cc][1]; up.z = sv[r - 1][cc][2];
            rt.x = sv[r][cc + 1][0]; rt.y = sv[r][cc + 1][1]; rt.z = sv[r][cc + 1][2];
            lf.x = sv[r][cc - 1][0]; lf.y = sv[r][cc - 1][1]; lf.z = sv[r][cc - 1][2];
            nr = normal_from_cross(p, dn, up, rt, lf, 1);
        }
        st3(NMap_out, (size_t)i * m + j, nr.x, nr.y, nr.z);
    }
    // The claim is a chain of dependent reads (label -> plane tables -> cell mask -> atomic): issued before the barrier
    // it held every wave of the tile back; here only this wave waits for it.
    if (ca.ParamTab) {
        const size_t cidx = in ? (size_t)i * m + j : 0;
        merge_claim(ca, pb->P_glob, p, in ? smask[threadIdx.y][threadIdx.x] : (uint8_t)0, cidx, in,
                    in ? (ca.Label16 ? (int)ca.Label16[cidx] : (int)ca.Label[cidx]) : 0);
    }
}
__global__ __launch_bounds__(TILE_W * TILE_H)
void k_model_merge(MergeArgs a) { model_merge_tile(a, blockIdx.x, blockIdx.y); }

// The three growth passes of a tile (seg3_passes) + the claim pass of Graph::Update for its pixels.  The claim needs the
// pixel's MERGED vertex and mask, which the merge tiles of the same launch are writing: this tile recomputes the two from
// the merge's inputs (7 bytes per pixel: .w, depth, mask -- no halo), so that the two kinds of tile stay independent and
// run side by side (one after the other inside one workgroup, their dependent memory round trips added up: 20.6 us
// against 25.2 us for the three launches at 640 x 480).
template <typename LT>
__device__ __forceinline__ void segment3_claim_tile(const SegArgsT<LT> &sg, const MergeArgs &mg, const int bx, const int by) {
    __shared__ SegTileLdsT<LT> S;
    if (!pb_tracked(sg.pb)) return;
    const int n = sg.n, m = sg.m;
    const int j0 = bx * TILE_W, i0 = by * TILE_H;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    const int j = j0 + threadIdx.x, i = i0 + threadIdx.y;
    const bool in = i < n && j < m;
    const size_t idx = in ? (size_t)i * m + j : 0;
    const SegTileRegsT<LT> SR = seg3_request(sg, i0, j0, tid, S);
    float w_old = mg.RGBD_in[4 * idx + 3];
    uint16_t d = ((const uint16_t *)((const char *)mg.depth + (size_t)(in ? n - i - 1 : 0) * mg.dpitch))[in ? j : 0];
    int mk = mg.Mask_in[idx];
    asm volatile("" : "+v"(w_old), "+v"(mk));
    __syncthreads();
    const bool any_label = seg3_stage(sg, SR, tid, S);
    const LT mine = seg3_passes(sg, i0, j0, tid, S, any_label);
    if (in) sg.out[idx] = mine;
    if (!mg.ca.ParamTab) return;                     // a tracker without Bump images of its own (equations only): no claim pass
    int newmask;
    const float w = merge_pixel(w_old, d, (uint8_t)mk, mg.c, newmask);
    const F3 v = vertex_from_w(w, i, j, mg.c);
    const uint8_t mc = (newmask >= 0) ? (uint8_t)newmask : (uint8_t)mk;
    merge_claim(mg.ca, sg.pb->P_glob, v, in ? mc : (uint8_t)0, idx, in, (int)mine);
}
// PredictedFrame::Merge || the three growth passes + claim: tile rows alternate between the two jobs.  rows2 = twice the
// tile rows of the image; rows from there on (if the grid has them) preprocess the NEXT input frame (OffLineFrame::LoadFrame
// of frame f+1 into buffers nothing in this launch reads): the job is independent of everything here and its ~11 us of
// dependent steps fit under this launch's own.
// 65 registers would leave 7 waves per SIMD; held to 64 the kernel runs 8 (a handful of scalar spills to lanes, no vector
// ones): 119.0 -> 108.7 us at 1920x1080, 26.0 -> 25.8 us at 640x480 (same-box, tools/ab_large.sh)
#ifndef SEG3_WAVES_PER_EU
#define SEG3_WAVES_PER_EU 8
#endif
template <typename LT>
__global__ __launch_bounds__(TILE_W * TILE_H)
#if SEG3_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(SEG3_WAVES_PER_EU)))
#endif
void k_segment3_merge_claim(SegArgsT<LT> sg, MergeArgs mg, PreprocessArgs pp, int rows2) {
    if ((int)blockIdx.y >= rows2) {
        frame_preprocess_tile(pp, blockIdx.x, (int)blockIdx.y - rows2);
    } else if (blockIdx.y & 1) {
        mg.ca.ParamTab = nullptr;                    // the merge tiles do not claim here
        model_merge_tile(mg, blockIdx.x, (int)(blockIdx.y >> 1));
    } else {
        segment3_claim_tile(sg, mg, blockIdx.x, (int)(blockIdx.y >> 1));
    }
}
// PredictedFrame::Merge does not read labels and the label growth does not read anything Merge writes (given a normal
// map of its own): the first growth pass and the merge share a launch, tile rows alternating between the two jobs.
template <typename LT>
__global__ __launch_bounds__(TILE_W * TILE_H)
void k_normal_segment_merge(SegArgsT<LT> sg, MergeArgs mg) {
    if (blockIdx.y & 1) model_merge_tile(mg, blockIdx.x, (int)(blockIdx.y >> 1));
    else normal_segment_tile(sg, blockIdx.x, (int)(blockIdx.y >> 1));
}
// last growth pass + the claim pass of Graph::Update for the same pixel (its merged vertex and mask were written by
// the launch above)
template <typename LT>
__global__ __launch_bounds__(TILE_W * TILE_H)
void k_segment_pass_claim(SegArgsT<LT> sg, MergeClaimArgs ca, const float *__restrict__ VMap_merged, const uint8_t *__restrict__ Mask_merged) {
    bool done; size_t idx;
    const LT mine = segment_pass_tile(sg, blockIdx.x, blockIdx.y, done, idx);
    F3 v; v.x = v.y = v.z = 0.0f;
    uint8_t mc = 0;
    if (done) { v = ld3(VMap_merged, idx); mc = Mask_merged[idx]; }
    if (!pb_tracked(sg.pb) || !ca.ParamTab) return;  // wave-uniform
    merge_claim(ca, sg.pb->P_glob, v, mc, idx, done, (int)mine);
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int rgbdslam_frame_preprocess(rgbdslam_ctx *ctx, const uint16_t *depth, size_t depth_pitch, const uint8_t *bgr,
                                         size_t bgr_pitch, float *VMap, float *NMap, float *RGB, float *gx, size_t gx_pitch,
                                         float *gy, size_t gy_pitch, int n, int m) {
    if (!ctx || !depth || !bgr || !VMap || !NMap || !RGB || !gx || !gy || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (depth_pitch < 2 * (size_t)m || bgr_pitch < 3 * (size_t)m || gx_pitch < 4 * (size_t)m || gy_pitch < 4 * (size_t)m)
        return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    PreprocessArgs a = {depth, depth_pitch, bgr, bgr_pitch, VMap, NMap, RGB, gx, gx_pitch, gy, gy_pitch, ctx->calib, n, m, ctx->aux_write};
    ctx->aux_write = nullptr;                         // announced for one launch (rgbdslam_set_frame_aux)
    KLAUNCH(ctx, k_frame_preprocess, grid2d(n, m), block2d(), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_model_reseed(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, int force, const float *fVMap,
                                     const float *fNMap, const float *fRGB, const uint16_t *depth, size_t depth_pitch,
                                     float *VMap, float *NMap, float *RGBD, uint8_t *Mask, int n, int m) {
    if (!ctx || (!pose_block_dev && !force) || !fVMap || !fNMap || !fRGB || !depth || !VMap || !NMap || !RGBD || !Mask)
        return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || depth_pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_model_reseed, grid2d(n, m), block2d(), pose_block_dev, force, fVMap, fNMap, fRGB, depth, depth_pitch, VMap, NMap,
            RGBD, Mask, ctx->calib, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The label images of every entry point below are u8, or u16 after rgbdslam_set_label_width(ctx, 2) (the pointers keep
// their u8 type in the signatures); LT is the element type the launch reads them with.
template <typename LT>
static int model_predict_impl(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, const float *VMap_in,
                              const float *RGBD_in, const uint8_t *Mask_in, const LT *Label_in, uint64_t *zbuf,
                              float *VMap_out, float *RGBD_out, uint8_t *Mask_out, LT *Label_out, float thresh,
                              float znear, float zfar, int n, int m, const float *frame_VMap, const float *frame_NMap,
                              const float *frame_RGB, const uint16_t *depth, size_t depth_pitch, float *NMap_reseed,
                              void *vertex_scratch, const UpdateArgs *apply) {
    if (!ctx || !pose_block_dev || !VMap_in || !RGBD_in || !Mask_in || !Label_in || !zbuf || !VMap_out || !RGBD_out ||
        !Mask_out || !Label_out)
        return RGBDSLAM_E_ARG;
    if (VMap_in == VMap_out || RGBD_in == RGBD_out || Mask_in == Mask_out || Label_in == Label_out || n < 2 || m < 2)
        return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x3FFFFFFFu) return RGBDSLAM_E_LIMIT;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    // lost frame: the draw grid re-seeds set "in" from the input frame (all-or-nothing: NULL frame_VMap = never)
    if (frame_VMap && (!frame_NMap || !frame_RGB || !depth || !NMap_reseed || depth_pitch < 2 * (size_t)m)) return RGBDSLAM_E_ARG;
    ReseedArgs rs = {frame_VMap, frame_NMap, frame_RGB, depth, depth_pitch, const_cast<float *>(VMap_in), NMap_reseed,
                     const_cast<float *>(RGBD_in), const_cast<uint8_t *>(Mask_in)};
    dim3 gd = grid2d(n, m);
    const int rows_draw = (int)gd.y;
    UpdateArgs upd = {};
    if (apply) { upd = *apply; gd.y *= 2; }
    if (vertex_scratch && ((uintptr_t)vertex_scratch & 15)) return RGBDSLAM_E_ARG;
    ProjVert *pv = (ProjVert *)vertex_scratch;
    float *zw = pv ? (float *)(pv + (size_t)n * m) : nullptr;
    KLAUNCH(ctx, k_predict_draw, gd, block2d(), pose_block_dev, VMap_in, (unsigned long long *)zbuf, thresh,
            ctx->calib, znear, zfar, upd, rows_draw, pv, zw, n, m);
    if (pv)
        KLAUNCH_AS(ctx, "k_predict_resolve<true>", (k_predict_resolve<true, LT>), grid2d(n, m), block2d(), pose_block_dev, VMap_in, RGBD_in,
                   Mask_in, Label_in, (unsigned long long *)zbuf, VMap_out, RGBD_out, Mask_out, Label_out, ctx->calib, rs,
                   (const ProjVert *)pv, (const float *)zw, n, m);
    else
        KLAUNCH_AS(ctx, "k_predict_resolve<false>", (k_predict_resolve<false, LT>), grid2d(n, m), block2d(), pose_block_dev, VMap_in, RGBD_in,
                   Mask_in, Label_in, (unsigned long long *)zbuf, VMap_out, RGBD_out, Mask_out, Label_out, ctx->calib, rs,
                   (const ProjVert *)nullptr, (const float *)nullptr, n, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_model_predict(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, const float *VMap_in,
                                      const float *RGBD_in, const uint8_t *Mask_in, const uint8_t *Label_in, uint64_t *zbuf,
                                      float *VMap_out, float *RGBD_out, uint8_t *Mask_out, uint8_t *Label_out, float thresh,
                                      float znear, float zfar, int n, int m, const float *frame_VMap, const float *frame_NMap,
                                      const float *frame_RGB, const uint16_t *depth, size_t depth_pitch, float *NMap_reseed,
                                      void *vertex_scratch) {
    if (ctx && labels16(ctx))
        return model_predict_impl(ctx, pose_block_dev, VMap_in, RGBD_in, Mask_in, (const uint16_t *)Label_in, zbuf, VMap_out, RGBD_out,
                                  Mask_out, (uint16_t *)Label_out, thresh, znear, zfar, n, m, frame_VMap, frame_NMap, frame_RGB, depth,
                                  depth_pitch, NMap_reseed, vertex_scratch, nullptr);
    return model_predict_impl(ctx, pose_block_dev, VMap_in, RGBD_in, Mask_in, Label_in, zbuf, VMap_out, RGBD_out, Mask_out, Label_out,
                              thresh, znear, zfar, n, m, frame_VMap, frame_NMap, frame_RGB, depth, depth_pitch, NMap_reseed,
                              vertex_scratch, nullptr);
}
// The same + the APPLY pass of the PREVIOUS frame's Graph::Update (rgbdslam_update_apply_from_history on VMap_in, RGBD_in,
// Mask_in, Label_in -- the model that frame left), riding on the draw launch.
extern "C" int rgbdslam_model_predict_apply(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, const float *VMap_in,
                                            const float *RGBD_in, const uint8_t *Mask_in, const uint8_t *Label_in, uint64_t *zbuf,
                                            float *VMap_out, float *RGBD_out, uint8_t *Mask_out, uint8_t *Label_out, float thresh,
                                            float znear, float zfar, int n, int m, const float *frame_VMap, const float *frame_NMap,
                                            const float *frame_RGB, const uint16_t *depth, size_t depth_pitch, float *NMap_reseed,
                                            void *vertex_scratch, const float *prev_history_entry_dev, uint16_t **BumpTab, uint8_t **RGBTab,
                                            uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t prev_stamp,
                                            int N_prim, int M_prim, int plane_lo, int plane_hi) {
    if (!prev_history_entry_dev || !BumpTab || !RGBTab || !MaskTab || !ParamTab || !ClaimTab) return RGBDSLAM_E_ARG;
    if (N_prim <= 0 || M_prim <= 0 || prev_stamp == 0) return RGBDSLAM_E_ARG;
    if ((size_t)(n > 0 ? n : 0) * (size_t)(m > 0 ? m : 0) > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;
    PoseV P;
    for (int k = 0; k < 16; k++) P.v[k] = 0.0f;
    const bool l16 = ctx && labels16(ctx);
    UpdateArgs a = {BumpTab, RGBTab, MaskTab, ParamTab, (unsigned long long *const *)ClaimTab, prev_stamp, VMap_in, RGBD_in, Mask_in,
                    l16 ? nullptr : Label_in, l16 ? (const uint16_t *)Label_in : nullptr, P, nullptr, prev_history_entry_dev, N_prim, M_prim,
                    n, m, plane_lo, plane_hi};
    if (l16)
        return model_predict_impl(ctx, pose_block_dev, VMap_in, RGBD_in, Mask_in, (const uint16_t *)Label_in, zbuf, VMap_out, RGBD_out,
                                  Mask_out, (uint16_t *)Label_out, thresh, znear, zfar, n, m, frame_VMap, frame_NMap, frame_RGB, depth,
                                  depth_pitch, NMap_reseed, vertex_scratch, &a);
    return model_predict_impl(ctx, pose_block_dev, VMap_in, RGBD_in, Mask_in, Label_in, zbuf, VMap_out, RGBD_out, Mask_out, Label_out,
                              thresh, znear, zfar, n, m, frame_VMap, frame_NMap, frame_RGB, depth, depth_pitch, NMap_reseed,
                              vertex_scratch, &a);
}

template <typename LT>
static int model_normal_segment_impl(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, float *NMap, const float *VMap,
                                     LT *Label, LT *Label_swap, const float *Equations_dev, int nbPlans, float epsilon, float alpha,
                                     int n, int m) {
    const float *Eq = nbPlans > 0 ? Equations_dev : nullptr;
    SegArgsT<LT> p1 = {pose_block_dev, NMap, VMap, Label, Label_swap, Eq, epsilon, alpha, n, m, 0};
    KLAUNCH_AS(ctx, "k_normal_segment", (k_normal_segment<LT>), grid2d(n, m), block2d(), p1);
    if (Eq) {                                                    // passes 2 and 3 (D1/RG.cu:910-911)
        SegArgsT<LT> p2 = p1, p3 = p1;
        p2.in = Label_swap; p2.out = Label;
        KLAUNCH_AS(ctx, "k_segment_pass_pb", (k_segment_pass_pb<LT>), grid2d(n, m), block2d(), p2);
        KLAUNCH_AS(ctx, "k_segment_pass_pb", (k_segment_pass_pb<LT>), grid2d(n, m), block2d(), p3);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_model_normal_segment(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, float *NMap,
                                             const float *VMap, uint8_t *Label, uint8_t *Label_swap, const float *Equations_dev,
                                             int nbPlans, float epsilon, float alpha, int n, int m) {
    if (!ctx || !pose_block_dev || !NMap || !VMap || !Label || !Label_swap || Label == Label_swap || n <= 0 || m <= 0)
        return RGBDSLAM_E_ARG;
    if (nbPlans < 0 || nbPlans > max_label(ctx) || (nbPlans > 0 && !Equations_dev)) return RGBDSLAM_E_LIMIT;
    if (labels16(ctx))
        return model_normal_segment_impl(ctx, pose_block_dev, NMap, VMap, (uint16_t *)Label, (uint16_t *)Label_swap, Equations_dev, nbPlans,
                                         epsilon, alpha, n, m);
    return model_normal_segment_impl(ctx, pose_block_dev, NMap, VMap, Label, Label_swap, Equations_dev, nbPlans, epsilon, alpha, n, m);
}

extern "C" int rgbdslam_model_merge(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, const float *RGBD_in,
                                    const uint8_t *Mask_in, const uint16_t *depth, size_t depth_pitch, const float *RGB,
                                    float *RGBD_out, uint8_t *Mask_out, float *VMap_out, float *NMap_out, int n, int m) {
    if (!ctx || !pose_block_dev || !RGBD_in || !Mask_in || !depth || !RGB || !RGBD_out || !Mask_out || !VMap_out || !NMap_out)
        return RGBDSLAM_E_ARG;
    if (RGBD_in == RGBD_out || Mask_in == Mask_out || n <= 0 || m <= 0 || depth_pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    MergeClaimArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0, 0, 0, 0};
    MergeArgs a = {pose_block_dev, RGBD_in, Mask_in, depth, depth_pitch, RGB, RGBD_out, Mask_out, VMap_out, NMap_out, ctx->calib, none, n, m};
    KLAUNCH(ctx, k_model_merge, grid2d(n, m), block2d(), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_model_merge_claim(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, const float *RGBD_in,
                                          const uint8_t *Mask_in, const uint16_t *depth, size_t depth_pitch, const float *RGB,
                                          float *RGBD_out, uint8_t *Mask_out, float *VMap_out, float *NMap_out,
                                          const uint8_t *Label, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                          uint32_t stamp, int N_prim, int M_prim, int n, int m, int plane_lo, int plane_hi) {
    if (!ctx || !pose_block_dev || !RGBD_in || !Mask_in || !depth || !RGB || !RGBD_out || !Mask_out || !VMap_out || !NMap_out)
        return RGBDSLAM_E_ARG;
    if (!Label || !MaskTab || !ParamTab || !ClaimTab || stamp == 0 || N_prim <= 0 || M_prim <= 0) return RGBDSLAM_E_ARG;
    if (RGBD_in == RGBD_out || Mask_in == Mask_out || n <= 0 || m <= 0 || depth_pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;                   // 23-bit pixel index in the claim key
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    const bool l16 = labels16(ctx);
    MergeClaimArgs ca = {ParamTab, MaskTab, (unsigned long long *const *)ClaimTab, l16 ? nullptr : Label, l16 ? (const uint16_t *)Label : nullptr,
                         stamp, N_prim, M_prim, plane_lo, plane_hi};
    MergeArgs a = {pose_block_dev, RGBD_in, Mask_in, depth, depth_pitch, RGB, RGBD_out, Mask_out, VMap_out, NMap_out, ctx->calib, ca, n, m};
    KLAUNCH(ctx, k_model_merge, grid2d(n, m), block2d(), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// rgbdslam_model_normal_segment + rgbdslam_model_merge_claim as ONE launch (k_segment3_merge_claim), or, with
// RGBDSLAM_SEG3=0, as the three launches this replaced: [first growth pass + merge] -> [second pass] -> [third pass +
// claim], where NMap_seg = the normal map of the predicted frame that the growth passes use (not the merge's NMap_out).
// Needs planes (nbPlans > 0).  next != nullptr: the next input frame's preprocessing rides along.
template <typename LT>
static int segment_merge_claim_impl(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, float *NMap_seg,
                                    const float *VMap_pred, LT *Label, LT *Label_swap,
                                    const float *Equations_dev, int nbPlans, float epsilon, float alpha,
                                    const float *RGBD_in, const uint8_t *Mask_in, const uint16_t *depth,
                                    size_t depth_pitch, const float *RGB, float *RGBD_out, uint8_t *Mask_out,
                                    float *VMap_out, float *NMap_out, uint8_t **MaskTab, float **ParamTab,
                                    uint64_t **ClaimTab, uint32_t stamp, int N_prim, int M_prim, int n, int m,
                                    int plane_lo, int plane_hi, const PreprocessArgs *next) {
    if (!ctx || !pose_block_dev || !NMap_seg || !VMap_pred || !Label || !Label_swap || Label == Label_swap || !Equations_dev)
        return RGBDSLAM_E_ARG;
    if (!RGBD_in || !Mask_in || !depth || !RGB || !RGBD_out || !Mask_out || !VMap_out || !NMap_out) return RGBDSLAM_E_ARG;
    // all three tables NULL: no claim pass (a tracker that holds the plane EQUATIONS but not the Bump images, e.g. the
    // tracking rank of the plane-partitioned form); otherwise all three and a stamp
    const bool no_claim = !MaskTab && !ParamTab && !ClaimTab;
    if (!no_claim && (!MaskTab || !ParamTab || !ClaimTab || stamp == 0 || N_prim <= 0 || M_prim <= 0)) return RGBDSLAM_E_ARG;
    if (RGBD_in == RGBD_out || Mask_in == Mask_out || NMap_seg == NMap_out || VMap_pred == VMap_out) return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || depth_pitch < 2 * (size_t)m) return RGBDSLAM_E_ARG;
    if (nbPlans <= 0 || nbPlans > (sizeof(LT) == 2 ? 65535 : 255)) return RGBDSLAM_E_LIMIT;
    if ((size_t)n * m > 0x7FFFFFu) return RGBDSLAM_E_LIMIT;                   // 23-bit pixel index in the claim key
    if (!ctx->have_calib) return RGBDSLAM_E_STATE;
    SegArgsT<LT> p1 = {pose_block_dev, NMap_seg, VMap_pred, Label, Label_swap, Equations_dev, epsilon, alpha, n, m, 1};   // NMap_seg: scratch
    SegArgsT<LT> p2 = p1, p3 = p1;
    p2.in = Label_swap; p2.out = Label;
    MergeClaimArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0, 0, 0, 0};
    MergeArgs mg = {pose_block_dev, RGBD_in, Mask_in, depth, depth_pitch, RGB, RGBD_out, Mask_out, VMap_out, NMap_out, ctx->calib, none, n, m};
    MergeClaimArgs ca = {ParamTab, MaskTab, (unsigned long long *const *)ClaimTab, nullptr, nullptr, stamp, N_prim, M_prim, plane_lo, plane_hi};
    if (ctx->seg3_fused) {                                       // labels: Label -> Label_swap, as after the three passes
        SegArgsT<LT> s3 = p1;
        s3.nb_eq = nbPlans;
        mg.ca = ca;
        dim3 g2 = grid2d(n, m);
        const int rows2 = 2 * (int)g2.y;
        g2.y = (unsigned)(next ? 3 * (rows2 / 2) : rows2);
        PreprocessArgs pp = {};
        if (next) pp = *next;
        KLAUNCH_AS(ctx, "k_segment3_merge_claim", (k_segment3_merge_claim<LT>), g2, block2d(), s3, mg, pp, rows2);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    dim3 grid2 = grid2d(n, m);
    grid2.y *= 2;
    KLAUNCH_AS(ctx, "k_normal_segment_merge", (k_normal_segment_merge<LT>), grid2, block2d(), p1, mg);
    KLAUNCH_AS(ctx, "k_segment_pass_pb", (k_segment_pass_pb<LT>), grid2d(n, m), block2d(), p2);
    KLAUNCH_AS(ctx, "k_segment_pass_claim", (k_segment_pass_claim<LT>), grid2d(n, m), block2d(), p3, ca, (const float *)VMap_out,
               (const uint8_t *)Mask_out);
    if (next) KLAUNCH(ctx, k_frame_preprocess, grid2d(n, m), block2d(), *next);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rgbdslam_model_segment_merge_claim(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, float *NMap_seg,
                                                  const float *VMap_pred, uint8_t *Label, uint8_t *Label_swap,
                                                  const float *Equations_dev, int nbPlans, float epsilon, float alpha,
                                                  const float *RGBD_in, const uint8_t *Mask_in, const uint16_t *depth,
                                                  size_t depth_pitch, const float *RGB, float *RGBD_out, uint8_t *Mask_out,
                                                  float *VMap_out, float *NMap_out, uint8_t **MaskTab, float **ParamTab,
                                                  uint64_t **ClaimTab, uint32_t stamp, int N_prim, int M_prim, int n, int m,
                                                  int plane_lo, int plane_hi) {
    if (ctx && labels16(ctx))
        return segment_merge_claim_impl(ctx, pose_block_dev, NMap_seg, VMap_pred, (uint16_t *)Label, (uint16_t *)Label_swap, Equations_dev,
                                        nbPlans, epsilon, alpha, RGBD_in, Mask_in, depth, depth_pitch, RGB, RGBD_out, Mask_out, VMap_out,
                                        NMap_out, MaskTab, ParamTab, ClaimTab, stamp, N_prim, M_prim, n, m, plane_lo, plane_hi, nullptr);
    return segment_merge_claim_impl(ctx, pose_block_dev, NMap_seg, VMap_pred, Label, Label_swap, Equations_dev, nbPlans, epsilon, alpha,
                                    RGBD_in, Mask_in, depth, depth_pitch, RGB, RGBD_out, Mask_out, VMap_out, NMap_out, MaskTab, ParamTab,
                                    ClaimTab, stamp, N_prim, M_prim, n, m, plane_lo, plane_hi, nullptr);
}
// The same + rgbdslam_frame_preprocess of the NEXT input frame (same n x m) into buffers that this frame does not use.
extern "C" int rgbdslam_model_segment_merge_claim_preprocess(
    rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, float *NMap_seg, const float *VMap_pred, uint8_t *Label,
    uint8_t *Label_swap, const float *Equations_dev, int nbPlans, float epsilon, float alpha, const float *RGBD_in,
    const uint8_t *Mask_in, const uint16_t *depth, size_t depth_pitch, const float *RGB, float *RGBD_out, uint8_t *Mask_out,
    float *VMap_out, float *NMap_out, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp, int N_prim, int M_prim,
    int n, int m, int plane_lo, int plane_hi, const uint16_t *next_depth, size_t next_depth_pitch, const uint8_t *next_bgr,
    size_t next_bgr_pitch, float *fVMap, float *fNMap, float *fRGB, float *gx, size_t gx_pitch, float *gy, size_t gy_pitch) {
    if (!ctx || !next_depth || !next_bgr || !fVMap || !fNMap || !fRGB || !gx || !gy || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    if (next_depth_pitch < 2 * (size_t)m || next_bgr_pitch < 3 * (size_t)m || gx_pitch < 4 * (size_t)m || gy_pitch < 4 * (size_t)m)
        return RGBDSLAM_E_ARG;
    if (fRGB == RGB || fVMap == VMap_pred || fVMap == VMap_out || fNMap == NMap_out || fNMap == NMap_seg) return RGBDSLAM_E_ARG;
    PreprocessArgs pp = {next_depth, next_depth_pitch, next_bgr, next_bgr_pitch, fVMap, fNMap, fRGB, gx, gx_pitch, gy, gy_pitch, ctx->calib, n, m, ctx->aux_write};
    ctx->aux_write = nullptr;                         // announced for one launch (rgbdslam_set_frame_aux)
    if (labels16(ctx))
        return segment_merge_claim_impl(ctx, pose_block_dev, NMap_seg, VMap_pred, (uint16_t *)Label, (uint16_t *)Label_swap, Equations_dev,
                                        nbPlans, epsilon, alpha, RGBD_in, Mask_in, depth, depth_pitch, RGB, RGBD_out, Mask_out, VMap_out,
                                        NMap_out, MaskTab, ParamTab, ClaimTab, stamp, N_prim, M_prim, n, m, plane_lo, plane_hi, &pp);
    return segment_merge_claim_impl(ctx, pose_block_dev, NMap_seg, VMap_pred, Label, Label_swap, Equations_dev, nbPlans, epsilon, alpha,
                                    RGBD_in, Mask_in, depth, depth_pitch, RGB, RGBD_out, Mask_out, VMap_out, NMap_out, MaskTab, ParamTab,
                                    ClaimTab, stamp, N_prim, M_prim, n, m, plane_lo, plane_hi, &pp);
}
