// planefit.hip -- the per-blob MEDIAN plane fit of RG::DetectPrimitives (D1/RG.cpp:469-646) on the device.
//
// The reference fits every blob on one CPU core: it copies the blob's normals (rotated into the global frame) into three
// lists, sorts each and takes the middle element (RG.cpp:479-533), normalises that vector, sorts the list of the blob's
// point-plane distances and takes ITS middle element (RG.cpp:557-598), and builds an in-plane basis (RG.cpp:607-646).  Round 2
// kept that on the host (nth_element, one task per blob): four downloads of 9.8 MB and ~2.7 ms per detection -- the bulk
// of RG::Finish's periodic cycle.
//
// The middle element of a sorted list is a SELECTION, and a selection needs no sort and no list: a three-pass radix select
// (11 + 11 + 10 bits of an order-preserving 32-bit key) over the IMAGE finds, for every blob and every list at once, the one
// element `sort` + `[size / 2]` would return -- the same float, so equations stay bit-exact.  Each pass is one sweep of the
// frame (a wave adds its lanes' bins with one atomic per distinct bin: the normals and distances of a planar blob fall
// into a handful of bins) and one small kernel that walks the histograms.  Only the blob centre is a genuinely serial
// quantity: the reference sums the blob's points in raster order in float (RG.cpp:573-578), so one wave per blob and
// coordinate walks the image in that order, the running sum passing from lane to lane (k_fit_center).  What comes back to the
// host is 80 bytes per blob.
//
// The one semantic difference: -0.0f sorts below +0.0f here while `sort` treats them as equal (either may then end up in
// the middle); it can only matter when a blob's median component is exactly zero.
#include "device_fns.hpp"

#define FIT_MAX_BLOBS 64                 // blobs larger than min_size in one frame (the host falls back to its own fit beyond)
#define FIT_BINS 2048
#define FIT_REC 20                       // floats per blob record, see rgbdslam.h

struct FitState {                        // per (blob slot, list): the key bits found so far and the rank still to find below them
    uint32_t prefix, krem;
};

__device__ __forceinline__ uint32_t fit_key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fit_unkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// roots of the blobs that count (RG.cpp:342-347: size > MIN_SIZE_PLAN), in no particular order: the host orders them
__global__ void k_fit_roots(const int32_t *__restrict__ sizes, int min_size, int32_t *__restrict__ root_slot, float *__restrict__ rec,
                            uint32_t *__restrict__ count, int P) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int sz = sizes[p];
    if (sz <= min_size) return;
    const uint32_t s = atomicAdd(count, 1u);
    if (s >= FIT_MAX_BLOBS) return;
    root_slot[p] = (int32_t)s;
    rec[FIT_REC * s] = __int_as_float(p);
    rec[FIT_REC * s + 1] = __int_as_float(sz);
}

// one atomic per distinct histogram word among the lanes of a wave
__device__ __forceinline__ void fit_wave_add(uint32_t *__restrict__ base, bool active, uint32_t word) {
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t w0 = (uint32_t)__shfl((int)word, leader);
        const unsigned long long same = __ballot(active && word == w0);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(base + w0, (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
    }
}

// PHASE 0: the three normal-component lists (RG.cpp:479-497); PHASE 1: the distance list (RG.cpp:557-578).
// pass 0 / 1 / 2 = key bits 31..21 / 20..10 / 9..0.
template <int PHASE>
__global__ __launch_bounds__(256)
void k_fit_hist(const int32_t *__restrict__ blob, const int32_t *__restrict__ root_slot, const float *__restrict__ VMap,
                const float *__restrict__ NMap, PoseV Pz, const float *__restrict__ rec, const FitState *__restrict__ st,
                uint32_t *__restrict__ hist, uint32_t *__restrict__ nonzero, int pass, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const float *pose = Pz.v;
    int s = -1;
    if (p < P) {
        const int r = blob[p];
        if (r >= 0) s = root_slot[r];
    }
    const bool in = s >= 0;
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const uint32_t mask = pass == 2 ? 1023u : 2047u;
    const int lists = PHASE == 0 ? 3 : 1;
    float val[3] = {0.0f, 0.0f, 0.0f};
    bool nz = false;
    if (in) {
        if (PHASE == 0) {
            const float N0 = NMap[3 * (size_t)p], N1 = NMap[3 * (size_t)p + 1], N2 = NMap[3 * (size_t)p + 2];
            nz = !(N0 == 0.0f && N1 == 0.0f && N2 == 0.0f);           // such entries stay 0 in the lists (RG.cpp:483-486)
            if (nz) {
                val[0] = (N0 * pose[0] + N1 * pose[4]) + N2 * pose[8];
                val[1] = (N0 * pose[1] + N1 * pose[5]) + N2 * pose[9];
                val[2] = (N0 * pose[2] + N1 * pose[6]) + N2 * pose[10];
            }
        } else {
            const float V0 = VMap[3 * (size_t)p], V1 = VMap[3 * (size_t)p + 1], V2 = VMap[3 * (size_t)p + 2];
            const float pt0 = ((V0 * pose[0] + V1 * pose[4]) + V2 * pose[8]) + pose[12];
            const float pt1 = ((V0 * pose[1] + V1 * pose[5]) + V2 * pose[9]) + pose[13];
            const float pt2 = ((V0 * pose[2] + V1 * pose[6]) + V2 * pose[10]) + pose[14];
            const float *n = rec + FIT_REC * s + 4;                   // the blob's normalised median normal
            val[0] = (pt0 * n[0] + pt1 * n[1]) + pt2 * n[2];
        }
    }
    if (PHASE == 0 && pass == 0) fit_wave_add(nonzero, in && nz, (uint32_t)(in ? s : 0));
#pragma unroll
    for (int l = 0; l < lists; l++) {
        const int sl = (in ? s : 0) * 3 + l;
        const uint32_t key = fit_key(val[l]);
        bool act = in;
        if (pass > 0 && in) act = (key >> (shift + (pass == 2 ? 10 : 11))) == st[sl].prefix;
        const uint32_t word = (uint32_t)sl * FIT_BINS + ((key >> shift) & mask);
        // first pass: sign, exponent and two mantissa bits -- a planar blob's values share a few bins: one atomic per distinct bin
        // of the wave.  Later passes see mantissa bits: still a handful of bins for the distances of a plane (aggregated:
        // 37 us a pass against 97 us with one atomic per lane), all different for the noisy normals (one atomic per lane: 57 us
        // against 69 us)
        if (PHASE == 1 || pass == 0) fit_wave_add(hist, act, word);
        else if (act) atomicAdd(hist + word, 1u);
    }
}

// One workgroup per (list, slot): the bin that holds rank krem, by a scan of the histogram; clears the histogram behind it.
__global__ __launch_bounds__(256)
void k_fit_pick(uint32_t *__restrict__ hist, FitState *__restrict__ st, const float *__restrict__ rec, const uint32_t *__restrict__ count,
                int pass, int lists) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t found_bin, found_below;
    const int l = blockIdx.x, s = blockIdx.y;
    if ((uint32_t)s >= min(*count, (uint32_t)FIT_MAX_BLOBS) || l >= lists) return;
    const int sl = s * 3 + l;
    uint32_t *h = hist + (size_t)sl * FIT_BINS;
    const int t = threadIdx.x;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = h[8 * t + k]; sum += v[k]; h[8 * t + k] = 0u; }
    part[t] = sum;
    __syncthreads();
    // rank to find: the middle of the blob's list (v[v.size() / 2], RG.cpp:529-533, 598) in the first pass, what is left of it later
    const uint32_t krem = pass == 0 ? (uint32_t)__float_as_int(rec[FIT_REC * s + 1]) / 2u : st[sl].krem;
    if (t == 0) {
        uint32_t below = 0;
        int w = 0;
        while (w < 255 && below + part[w] <= krem) { below += part[w]; w++; }
        found_bin = (uint32_t)w; found_below = below;
    }
    __syncthreads();
    if (t == (int)found_bin) {
        uint32_t below = found_below;
        int k = 0;
        while (k < 7 && below + v[k] <= krem) { below += v[k]; k++; }
        const uint32_t bin = 8u * (uint32_t)t + (uint32_t)k;
        const uint32_t prefix = pass == 0 ? bin : ((st[sl].prefix << (pass == 2 ? 10 : 11)) | bin);
        st[sl].prefix = prefix;
        st[sl].krem = krem - below;
    }
}

// after the three passes of phase 0: the median normal, normalised (RG.cpp:535-540), and the acceptance test (RG.cpp:500-516)
__global__ void k_fit_normal(float *__restrict__ rec, const FitState *__restrict__ st, const uint32_t *__restrict__ nonzero,
                             const uint32_t *__restrict__ count, int min_normals) {
    const int s = threadIdx.x;
    if ((uint32_t)s >= min(*count, (uint32_t)FIT_MAX_BLOBS)) return;
    float nm[3];
#pragma unroll
    for (int l = 0; l < 3; l++) nm[l] = fit_unkey(st[s * 3 + l].prefix);
    const float norm = __builtin_sqrtf((nm[0] * nm[0] + nm[1] * nm[1]) + nm[2] * nm[2]);
    float *r = rec + FIT_REC * s;
    r[2] = __int_as_float((int)nonzero[s]);
    r[3] = __int_as_float((int)nonzero[s] < min_normals ? 0 : 1);
    r[4] = nm[0] / norm; r[5] = nm[1] / norm; r[6] = nm[2] / norm;
}

// The blob centre: the reference adds the blob's points (global frame) one after the other, in raster order, in float
// (RG.cpp:573-578) -- a serial sum by definition: 100 000 dependent additions for a wall of a 640 x 480 frame, and the order
// is part of the result.  One workgroup per blob walks the image in that order, in batches of 3 072 pixels, as a two-stage
// pipeline: waves 1-3 STAGE the next batch (each owns 1 024 consecutive pixels: sixteen loads per lane in flight, the
// blob's pixels transformed into the global frame and written DENSELY, in raster order, into the wave's own LDS region --
// positions from ballots, no cross-wave step) while wave 0 CHAINS the batch before: lanes 0..2 own the x / y / z sums and
// add their coordinate of one staged pixel after the other, four pixels per 16-byte LDS read.  One barrier per batch.
// (Tried first: a systolic DPP chain over the lanes, one wave per coordinate -- exact too, but 8+ cycles per step and a
// memory round trip per 64 pixels: 2.1 ms for three walls, then 0.83 ms batched; the host's nth_element fit took 2.7 ms.)
#define FITC_PX 16                        // pixels per stager lane and batch
#define FITC_SPAN (64 * FITC_PX)          // pixels per stager wave and batch
#define FITC_STAGERS 3
__global__ __launch_bounds__(256)
void k_fit_center(const int32_t *__restrict__ blob, const float *__restrict__ VMap, PoseV Pz, float *__restrict__ rec,
                  const uint32_t *__restrict__ count, int P) {
    __shared__ __align__(16) float dense[2][FITC_STAGERS][3][FITC_SPAN + 4];
    __shared__ int cnt[2][FITC_STAGERS];
    const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((uint32_t)s >= min(*count, (uint32_t)FIT_MAX_BLOBS)) return;
    const float *pose = Pz.v;
    const int root = __float_as_int(rec[FIT_REC * s]);
    const int size = __float_as_int(rec[FIT_REC * s + 1]);
    auto stage = [&](int buf, int b0) {              // stager waves only
        const int v = wave - 1, span = b0 + v * FITC_SPAN;
        int r[FITC_PX];
#pragma unroll
        for (int j = 0; j < FITC_PX; j++) {
            const int p = span + 64 * j + lane;
            r[j] = (p < P) ? blob[p] : -1;
        }
        int base = 0;
#pragma unroll
        for (int j = 0; j < FITC_PX; j++) {
            const int p = span + 64 * j + lane;
            const bool mine = r[j] == root;          // root >= 0, so pixels beyond the image never match
            const unsigned long long bits = __ballot(mine);
            const size_t pc = mine ? (size_t)p : (size_t)root;
            const float V0 = VMap[3 * pc], V1 = VMap[3 * pc + 1], V2 = VMap[3 * pc + 2];
            if (mine) {
                const int pos = base + (int)__builtin_popcountll(bits & ((1ull << lane) - 1ull));
                dense[buf][v][0][pos] = ((V0 * pose[0] + V1 * pose[4]) + V2 * pose[8]) + pose[12];      // RG.cpp:566-571
                dense[buf][v][1][pos] = ((V0 * pose[1] + V1 * pose[5]) + V2 * pose[9]) + pose[13];
                dense[buf][v][2][pos] = ((V0 * pose[2] + V1 * pose[6]) + V2 * pose[10]) + pose[14];
            }
            base += (int)__builtin_popcountll(bits);
        }
        if (lane < 4) { dense[buf][v][0][base + lane] = 0.0f; dense[buf][v][1][base + lane] = 0.0f; dense[buf][v][2][base + lane] = 0.0f; }   // pad: + 0.0f is exact
        if (lane == 0) cnt[buf][v] = base;
    };
    float c = 0.0f;                                  // wave 0, lane 0..2: the running sum of coordinate `lane`
    int seen = 0, b0 = root;                         // the root is the blob's first pixel in raster order
    if (wave > 0) stage(0, b0);
    __syncthreads();
    for (int it = 0;; it++) {
        const int buf = it & 1;
        if (wave > 0) {
            stage(buf ^ 1, b0 + FITC_STAGERS * FITC_SPAN);
        } else {
            const int cl = lane < 3 ? lane : 0;
            for (int v = 0; v < FITC_STAGERS; v++) {
                const int nv = cnt[buf][v];
                seen += nv;
                const float4 *src = (const float4 *)dense[buf][v][cl];
#pragma unroll 4
                for (int k = 0; 4 * k < nv; k++) {
                    const float4 q = src[k];
                    c = c + q.x; c = c + q.y; c = c + q.z; c = c + q.w;
                }
            }
        }
        b0 += FITC_STAGERS * FITC_SPAN;
        // (seen is wave 0's; every wave takes the decision from it through LDS)
        __shared__ int done;
        if (threadIdx.x == 0) done = (seen >= size || b0 >= P) ? 1 : 0;
        __syncthreads();
        const int d = done;
        __syncthreads();                             // nobody overwrites `done` (next round) before everyone has read it
        if (d) break;
    }
    if (wave == 0 && lane < 3) rec[FIT_REC * s + 17 + lane] = c / (float)size;  // RG.cpp:581-583
}

// after the three passes of phase 1: the median distance and the in-plane basis (RG.cpp:598-646) -> equation (n, e1, e2, d)
__global__ void k_fit_finish(float *__restrict__ rec, const FitState *__restrict__ st, const uint32_t *__restrict__ count) {
    const int s = threadIdx.x;
    if ((uint32_t)s >= min(*count, (uint32_t)FIT_MAX_BLOBS)) return;
    float *r = rec + FIT_REC * s;
    const float n0 = r[4], n1 = r[5], n2 = r[6];
    const float dist = fit_unkey(st[s * 3].prefix);
    float e1[3], e2[3];
    if (n0 > n1 && n0 > n2) { e1[0] = -n1; e1[1] = n0; e1[2] = 0.0f; }
    else if (n1 > n0 && n1 > n2) { e1[0] = 0.0f; e1[1] = -n2; e1[2] = n1; }
    else { e1[0] = n2; e1[1] = 0.0f; e1[2] = -n0; }
    const float norm_e = __builtin_sqrtf((e1[0] * e1[0] + e1[1] * e1[1]) + e1[2] * e1[2]);
    e1[0] = e1[0] / norm_e; e1[1] = e1[1] / norm_e; e1[2] = e1[2] / norm_e;
    e2[0] = n1 * e1[2] - n2 * e1[1];
    e2[1] = n2 * e1[0] - n0 * e1[2];
    e2[2] = n0 * e1[1] - n1 * e1[0];
    r[7] = e1[0]; r[8] = e1[1]; r[9] = e1[2];
    r[10] = e2[0]; r[11] = e2[1]; r[12] = e2[2];
    r[13] = dist;
}

// RG.cpp:590: the blob's pixels take the plane's label; label_of_slot[s] = 0 for blobs that were not accepted
template <typename LT>
__global__ void k_fit_labels(LT *__restrict__ Label, const int32_t *__restrict__ blob, const int32_t *__restrict__ root_slot,
                             const uint8_t *__restrict__ label_of_slot, int P) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int r = blob[p];
    LT out = 0;
    if (r >= 0) {
        const int s = root_slot[r];
        if (s >= 0) out = label_of_slot[s];
    }
    Label[p] = out;
}

// =================================================================================================
// C ABI
// =================================================================================================
// workspace layout (bytes): [0] count | [64] nonzero[64] | [512] FitState[64 * 3] | [4096] label_of_slot[64] | [8192] rec[64 * 20] |
//                           [16384] hist[64 * 3 * 2048] | then root_slot[P]
#define FIT_OFF_NONZERO 64
#define FIT_OFF_STATE 512
#define FIT_OFF_LABELS 4096
#define FIT_OFF_REC 8192
#define FIT_OFF_HIST 16384
#define FIT_OFF_SLOT (FIT_OFF_HIST + 4 * FIT_MAX_BLOBS * 3 * FIT_BINS)

extern "C" size_t rgbdslam_fit_planes_workspace_bytes(int n, int m) {
    if (n <= 0 || m <= 0) return 0;
    return (size_t)FIT_OFF_SLOT + 4 * (size_t)n * m;
}

extern "C" int rgbdslam_fit_planes(rgbdslam_ctx *ctx, const int32_t *blob, const int32_t *sizes, const float *VMap, const float *NMap,
                                   const float *pose16_host, int min_size, int min_normals, void *workspace, int n, int m) {
    if (!ctx || !blob || !sizes || !VMap || !NMap || !pose16_host || !workspace || n <= 0 || m <= 0 || min_size < 0) return RGBDSLAM_E_ARG;
    if (((uintptr_t)workspace & 255) != 0) return RGBDSLAM_E_ARG;
    if ((size_t)n * m > 0x7FFFFFFFull) return RGBDSLAM_E_LIMIT;
    const int P = n * m;
    char *ws = (char *)workspace;
    uint32_t *count = (uint32_t *)ws, *nonzero = (uint32_t *)(ws + FIT_OFF_NONZERO), *hist = (uint32_t *)(ws + FIT_OFF_HIST);
    FitState *st = (FitState *)(ws + FIT_OFF_STATE);
    float *rec = (float *)(ws + FIT_OFF_REC);
    int32_t *root_slot = (int32_t *)(ws + FIT_OFF_SLOT);
    HIP_TRY(hipMemsetAsync(ws, 0, FIT_OFF_SLOT, ctx->stream));
    HIP_TRY(hipMemsetAsync(root_slot, 0xFF, 4 * (size_t)P, ctx->stream));
    PoseV Pz;
    for (int k = 0; k < 16; k++) Pz.v[k] = pose16_host[k];
    const dim3 gp((unsigned)((P + 255) / 256));
    KLAUNCH(ctx, k_fit_roots, gp, dim3(256), sizes, min_size, root_slot, rec, count, P);
    KLAUNCH(ctx, k_fit_center, dim3(FIT_MAX_BLOBS), dim3(256), blob, VMap, Pz, rec, (const uint32_t *)count, P);
    for (int pass = 0; pass < 3; pass++) {
        KLAUNCH(ctx, k_fit_hist<0>, gp, dim3(256), blob, (const int32_t *)root_slot, VMap, NMap, Pz, (const float *)rec, (const FitState *)st,
                hist, nonzero, pass, P);
        KLAUNCH(ctx, k_fit_pick, dim3(3, FIT_MAX_BLOBS), dim3(256), hist, st, (const float *)rec, (const uint32_t *)count, pass, 3);
    }
    KLAUNCH(ctx, k_fit_normal, dim3(1), dim3(FIT_MAX_BLOBS), rec, (const FitState *)st, (const uint32_t *)nonzero, (const uint32_t *)count,
            min_normals);
    for (int pass = 0; pass < 3; pass++) {
        KLAUNCH(ctx, k_fit_hist<1>, gp, dim3(256), blob, (const int32_t *)root_slot, VMap, NMap, Pz, (const float *)rec, (const FitState *)st,
                hist, nonzero, pass, P);
        KLAUNCH(ctx, k_fit_pick, dim3(1, FIT_MAX_BLOBS), dim3(256), hist, st, (const float *)rec, (const uint32_t *)count, pass, 1);
    }
    KLAUNCH(ctx, k_fit_finish, dim3(1), dim3(FIT_MAX_BLOBS), rec, (const FitState *)st, (const uint32_t *)count);
    HIP_TRY(hipGetLastError());
    return 0;
}

// blocking: *nb_blobs = blobs larger than min_size (may exceed RGBDSLAM_FIT_MAX_BLOBS: then the records are incomplete and
// the caller has to fit on the host); records_host: min(*nb_blobs, 64) x 20 floats
extern "C" int rgbdslam_fit_planes_read(rgbdslam_ctx *ctx, const void *workspace, int *nb_blobs, float *records_host) {
    if (!ctx || !workspace || !nb_blobs || !records_host) return RGBDSLAM_E_ARG;
    const char *ws = (const char *)workspace;
    uint32_t cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, ws, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(records_host, ws + FIT_OFF_REC, 4 * FIT_REC * FIT_MAX_BLOBS, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *nb_blobs = (int)cnt;
    return 0;
}

// labels of the blobs' pixels: label_of_slot_host[s] for the blob of record s (0 = not a plane); every other pixel 0
extern "C" int rgbdslam_fit_planes_label(rgbdslam_ctx *ctx, uint8_t *Label, const int32_t *blob, void *workspace,
                                         const uint8_t *label_of_slot_host, int n, int m) {
    if (!ctx || !Label || !blob || !workspace || !label_of_slot_host || n <= 0 || m <= 0) return RGBDSLAM_E_ARG;
    char *ws = (char *)workspace;
    HIP_TRY(hipMemcpyAsync(ws + FIT_OFF_LABELS, label_of_slot_host, FIT_MAX_BLOBS, hipMemcpyHostToDevice, ctx->stream));
    const int P = n * m;
    if (labels16(ctx))                                          // rgbdslam_set_label_width(ctx, 2): Label is a u16 image
        KLAUNCH_AS(ctx, "k_fit_labels", (k_fit_labels<uint16_t>), dim3((unsigned)((P + 255) / 256)), dim3(256), (uint16_t *)Label, blob,
                   (const int32_t *)(ws + FIT_OFF_SLOT), (const uint8_t *)(ws + FIT_OFF_LABELS), P);
    else
        KLAUNCH_AS(ctx, "k_fit_labels", (k_fit_labels<uint8_t>), dim3((unsigned)((P + 255) / 256)), dim3(256), Label, blob,
                   (const int32_t *)(ws + FIT_OFF_SLOT), (const uint8_t *)(ws + FIT_OFF_LABELS), P);
    HIP_TRY(hipGetLastError());
    return 0;
}
