// common.hpp -- shared declarations of librgbdslam_hip.so (gfx950 only).
// Arithmetic contract: IEEE round-to-nearest, no FMA contraction (-ffp-contract=off), correctly
// rounded f32 divide/sqrt (hipcc default), C promotion rules of the reference's expressions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/rgbdslam.h"

#define SLAM_MAX_DEPTH 100.0     // D1/cudaTypes.cuh:51
#define SLAM_MAXTOLERANCE 0.2    // D1/cudaTypes.cuh:47

// Launch geometry: one lane per COLUMN (coalesced rows), 64 x 4 threads per workgroup.
#define TILE_W 64
#define TILE_H 4

// v = the reference's 11 calibration floats (fx, fy, cx, cy, ..., [9] depth scale, [10] depth factor).  Derived on the host by
// rgbdslam_set_calibration_matrix: inv_fx / inv_fy = the correctly rounded reciprocals of the focal lengths, s = v[9] / v[10], and
// fast = the per-pixel quotients by fx, fy and the constant 100 may be formed from reciprocals (device_fns.hpp: quot_rcp) --
// every value finite and in the range where no intermediate overflows or underflows; otherwise the kernels divide.
struct CalibV { float v[11]; float inv_fx, inv_fy, s; int fast; };
struct PoseV  { float v[16]; };

// Device-resident state of the Gauss-Newton loop (RG::AlignFrameGaussNewton, D1/RG.cpp:797-932).
struct IcpState {
    float Rinv[9], tinv[3];    // Rcurr_inv, tcurr_inv
    float Rcurr[9], tcurr[3];  // the pair uploaded for the current iteration (RG.cpp:851-855)
    float intr[4];
    int   cvg, fail, iters, lvl_iters;   // lvl_iters: executed iterations per level, 8 bits each
};

struct rgbdslam_ctx {
    int device;
    hipStream_t stream;
    CalibV calib;
    PoseV pose;
    bool have_calib;
    // ICP.cu:5-7 globals
    float *buf_dev;      // f32[27][512] chain sums (first reduction stage, ICP.cu:57-58)
    long long *icp_timing_dev;  // debug (rgbdslam_debug_icp_timing)
    float *outbuf_dev;   // f32[27]
    float *out_host;     // pinned f32[27]
    IcpState *icp_state; // device
    char *icp_block;               // the one allocation behind buf_dev / outbuf_dev / icp_state / icp_epoch_dev
    unsigned int *icp_epoch_dev;   // k_icp_loop: [0] epoch word, [32] fault flag
    unsigned int icp_epoch_base;   // host-side monotonic base of the epoch word
    int icp_loop;                  // rgbdslam_set_icp_loop
    int icp_line_check;            // rgbdslam_set_icp_line_check
    int icp_timeout_ticks;         // rgbdslam_debug_set_icp_timeout
    // rgbdslam_set_frame_aux: where the NEXT fused preprocessing launch leaves the frame's auxiliary ICP records, and the records
    // the NEXT rgbdslam_track_frame reads (each consumed by that call)
    float *aux_write; const float *track_aux;
    int sync_spin_us;              // rgbdslam_synchronize polls the stream this long before it blocks (rgbdslam_set_sync_spin)
    int seg3_fused;                // rgbdslam_model_segment_merge_claim as one launch (RGBDSLAM_SEG3=0: the three-launch form)
    // instrumentation (rgbdslam_profile_*): HIP-event pairs around every launch of one named kernel
    const char *prof_name;
    hipEvent_t *prof_ev;   // 2 * prof_cap events
    int prof_cap, prof_n;
    int prof_launches;     // launches covered by the recorded event pairs
    bool prof_in_group;    // inside prof_group_begin/end: launches are counted, the group carries the one event pair
    // row-sharded ICP: peer-mapped exchange blocks (rgbdslam_exchange_*), slot r of every block = rank r's line of sums
    unsigned int *xchg_peer[8];    // [xchg_rank] = this rank's own block; the others are IPC mappings of the peers' blocks
    int xchg_world, xchg_rank;
    unsigned int xchg_seq;         // advances once per exchanging LAUNCH (16 tags, one slot set each), in lockstep on every rank
    int xchg_delay_ticks;          // rgbdslam_debug_set_exchange_delay (test hook)
    unsigned int **xchg_tab_dev;   // device copy of xchg_peer (the kernels read the block addresses from it)
    // rgbdslam_planes_draw: [0] / [32] the two tile counters (call parity), from word 64 one stamp word per tile, then the queue
    uint32_t *pd_scratch; size_t pd_cap; uint32_t pd_stamp;
    unsigned long long *noalloc_zbuf;   // this device's copy of the static z-buffer of rgbdslam_predict_raster_noalloc
    float *bil_lut; double bil_sigma_color;   // rgbdslam_bilateral_depth: colour-weight table (device) and the sigma it was built for
    // compat.hip: work buffers behind the reference-shaped Update_cu / Segment_cu / Overlap_cu (allocated on first use)
    uint64_t *claim_slab[RGBDSLAM_COMPAT_MAX_PLANES];   // one u64 claim word per Bump cell and plane
    uint64_t **claim_tab_dev;                           // device copy of claim_slab
    size_t claim_cells;
    uint32_t claim_stamp;
    uint8_t *label_swap; size_t label_swap_px;
    int label_width;               // rgbdslam_set_label_width: bytes per element of the label images (1 = u8 as in Frame.h:197, 2 = u16)
    unsigned int *compat_counter;
};
void compat_release(rgbdslam_ctx *ctx);
// label images: u8 (the reference's, Frame.h:197), or u16 after rgbdslam_set_label_width(ctx, 2) -- the entry points keep
// their uint8_t pointer types and read / write the images with the width the context says
static inline bool labels16(const rgbdslam_ctx *ctx) { return ctx->label_width == 2; }
static inline int max_label(const rgbdslam_ctx *ctx) { return labels16(ctx) ? 65535 : 255; }

// Launch on the context's stream; when the kernel's name is the one being profiled, bracket the
// launch with a HIP event pair recorded on that same stream.
bool prof_begin(rgbdslam_ctx *ctx, const char *kname);
void prof_end(rgbdslam_ctx *ctx);
// one event pair around a run of back-to-back launches of `kname` (their average then includes the real
// launch-to-launch behaviour instead of one marker round trip per launch)
bool prof_group_begin(rgbdslam_ctx *ctx, const char *kname);
void prof_group_end(rgbdslam_ctx *ctx);
// predict.hip: resolves the device address of the static z-buffer of rgbdslam_predict_raster_noalloc on `device`
int raster_noalloc_setup(int device, unsigned long long **out);
// ---- D1/ICP.cu:173: weight of a correspondence from the depth component z of the current frame's vertex (shared by the ICP
// kernels and by the preprocessing that leaves it in the frame's auxiliary records) ----------------------------------------
__device__ __forceinline__ float icp_depth_weight(float z) {
    const float sd = -z;
    return (float)(0.0012 / (0.0012 + 0.0019 * ((double)sd - 0.4) * ((double)sd - 0.4)));
}

#define KLAUNCH(ctx, kern, grid, block, ...)                                         \
    do {                                                                             \
        bool _prof = (ctx)->prof_name && prof_begin((ctx), #kern);                   \
        hipLaunchKernelGGL(kern, grid, block, 0, (ctx)->stream, __VA_ARGS__);        \
        if (_prof) prof_end(ctx);                                                    \
    } while (0)

// the same with the profiled name given explicitly (kern may be a parenthesised template instance)
#define KLAUNCH_AS(ctx, name, kern, grid, block, ...)                                \
    do {                                                                             \
        bool _prof = (ctx)->prof_name && prof_begin((ctx), name);                    \
        hipLaunchKernelGGL(kern, grid, block, 0, (ctx)->stream, __VA_ARGS__);        \
        if (_prof) prof_end(ctx);                                                    \
    } while (0)

#define HIP_TRY(expr)                                     \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return -(int)_e;            \
    } while (0)

static inline dim3 grid2d(int n, int m) { return dim3((m + TILE_W - 1) / TILE_W, (n + TILE_H - 1) / TILE_H); }
static inline dim3 block2d() { return dim3(TILE_W, TILE_H); }

// CUDA __float2int_rn / __float2int_rd semantics: saturating, NaN -> 0.
// __float2int_rn / __float2int_rd of the reference: round, then convert with saturation, NaN -> 0.  That is what
// v_cvt_i32_f32 does by itself (CDNA ISA: out-of-range inputs saturate, NaN converts to 0), so the three range tests a C
// cast needs to stay defined -- nested branches in every projection -- are not needed.
__device__ __forceinline__ int f2i_rn(float x) {
    int r;
    asm("v_rndne_f32_e32 %0, %1\n\tv_cvt_i32_f32_e32 %0, %0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int f2i_rd(float x) {
    int r;
    asm("v_floor_f32_e32 %0, %1\n\tv_cvt_i32_f32_e32 %0, %0" : "=v"(r) : "v"(x));
    return r;
}

struct F3 { float x, y, z; };
// address space of global memory, for pointers whose origin the compiler cannot see (fetched from tables)
#define GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ F3 ld3(const float *p, size_t idx) {
    F3 r; r.x = p[3 * idx]; r.y = p[3 * idx + 1]; r.z = p[3 * idx + 2]; return r;
}
__device__ __forceinline__ void st3(float *p, size_t idx, float a, float b, float c) {
    p[3 * idx] = a; p[3 * idx + 1] = b; p[3 * idx + 2] = c;
}
// column-major 4x4 applied to a point / a direction (left-to-right sums, D1/Kernel.cu:365-367)
__device__ __forceinline__ F3 xf_point(const float *P, F3 a) {
    F3 r;
    r.x = ((P[0] * a.x + P[4] * a.y) + P[8] * a.z) + P[12];
    r.y = ((P[1] * a.x + P[5] * a.y) + P[9] * a.z) + P[13];
    r.z = ((P[2] * a.x + P[6] * a.y) + P[10] * a.z) + P[14];
    return r;
}
__device__ __forceinline__ F3 xf_dir(const float *P, F3 a) {
    F3 r;
    r.x = (P[0] * a.x + P[4] * a.y) + P[8] * a.z;
    r.y = (P[1] * a.x + P[5] * a.y) + P[9] * a.z;
    r.z = (P[2] * a.x + P[6] * a.y) + P[10] * a.z;
    return r;
}
__device__ __forceinline__ bool is_zero3(F3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
