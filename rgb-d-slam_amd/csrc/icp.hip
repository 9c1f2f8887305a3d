// icp.hip -- stages 2+3: projective data association, Gauss-Newton rows, the 27-term JtJ/Jtr
// reduction, and the device-resident coarse-to-fine loop.
//
// Mapping: ONE WAVEFRONT = one 8x8 pixel block of the reference launch
// (D1/ICP.cu:258-265).  Lane L handles block-local (row L/8, col L%8), so each wave reads eight
// 8-pixel row segments.  The 64-value sum of each of the 27 terms is folded with cross-lane
// shuffles in exactly the pairing of the reference's shared-memory tree (reduce<64>,
// D1/ICP.cu:28-48), and the second stage repeats ReduceKernel's 512-strided chains + tree
// (D1/ICP.cu:50-70): the 27 sums are bit-identical to a serial restatement of the reference.
#include "common.hpp"
#include <string.h>
#include <math.h>

#define ICP_CHAIN_WAVES 6  // waves per chain: 12 waves per workgroup = 3 on every SIMD (10 left two SIMDs with 3 and two with 2, and the loaded pair set the pace); 640x480 has 9.4 blocks per chain -> rounds of 6 + 3.4
#define ICP_WG_CHAINS 2    // a workgroup owns chains t and t+256 and adds their sums itself (first level of reduce<512>):
                           // half as many sums to publish, to count and -- the expensive part -- to read back in the tail
#define ICP_MAX_WAVES (ICP_CHAIN_WAVES * ICP_WG_CHAINS)
#define ICP_WGS (ICP_CHAINS / ICP_WG_CHAINS)
#define ICP_CHAINS 512     // STRIDE (D1/cudaTypes.cuh:46): chains of ReduceKernel

struct IcpRow { float g[7]; float c[7]; };

// D1/ICP.cu:72-146 searchGauss + 164-205 row construction, split so that every memory access of a
// stage is issued before anything waits on it (the kernel is a chain of dependent memory round
// trips; the arithmetic and its order are untouched).
struct IcpModelRaw { F3 np, vp; float cx, cy, cz; };   // a model sample as loaded
struct IcpModelPx { F3 np, vp; float ip; };   // ip = intensity of the model pixel (ICP.cu:100), the only use of its colour
// what the rows need of the current frame at the projected pixel: normal, vertex, and four values that are functions of that
// pixel alone -- the scaled gradients g0, g1 (ICP.cu:168-169), the intensity ic (ICP.cu:167) and the depth weight w
// (ICP.cu:173) -- either formed here from the gathered colour / gradients / vertex, or read ready-made from the frame's
// auxiliary record (icp_gather_aux)
struct IcpCurrPx { F3 nc, s; float g0, g1, ic, w; };
struct IcpProj { float d0, d1, d2, n0, n1, n2, az; int pu, pv; unsigned int pidx; };   // pidx < 2^28: launch_iter checks the image size

__device__ __forceinline__ IcpModelPx icp_load_model(size_t indx, const float *__restrict__ VMap_prev,
                                                     const float *__restrict__ NMap_prev, const float *__restrict__ RGBD_prev) {
    IcpModelPx mp;
    mp.np = ld3(NMap_prev, indx);
    mp.vp = ld3(VMap_prev, indx);
    const float4 cp = *(const float4 *)(RGBD_prev + 4 * indx);
    // (float)((double)x / 3.0) == x / 3.0f for float x: double rounding is innocuous for division (53 >= 2*24 + 2)
    mp.ip = ((cp.x + cp.y) + cp.z) / 3.0f;                                       // ICP.cu:100
    return mp;
}
__device__ __forceinline__ IcpModelRaw icp_load_model_raw(unsigned int indx, const float *__restrict__ VMap_prev,
                                                          const float *__restrict__ NMap_prev, const float *__restrict__ RGBD_prev) {
    IcpModelRaw mr;
    mr.np = *(const F3 *)((const char *)NMap_prev + 12u * indx);
    mr.vp = *(const F3 *)((const char *)VMap_prev + 12u * indx);
    const F3 c = *(const F3 *)((const char *)RGBD_prev + 16u * indx);
    mr.cx = c.x; mr.cy = c.y; mr.cz = c.z;
    return mr;
}
__device__ __forceinline__ IcpModelPx icp_model_from_raw(const IcpModelRaw &mr) {
    IcpModelPx mp;
    mp.np = mr.np; mp.vp = mr.vp;
    mp.ip = ((mr.cx + mr.cy) + mr.cz) / 3.0f;                                    // as icp_load_model
    return mp;
}
__device__ __forceinline__ IcpProj icp_project(const IcpModelPx &mp, const float *R, const float *t, const float *intr, int n, int m) {
    IcpProj p;
    const F3 &vp = mp.vp, &np = mp.np;
    p.d0 = ((R[0] * vp.x + R[3] * vp.y) + R[6] * vp.z) + t[0];
    p.d1 = ((R[1] * vp.x + R[4] * vp.y) + R[7] * vp.z) + t[1];
    p.d2 = ((R[2] * vp.x + R[5] * vp.y) + R[8] * vp.z) + t[2];
    p.n0 = (R[0] * np.x + R[3] * np.y) + R[6] * np.z;
    p.n1 = (R[1] * np.x + R[4] * np.y) + R[7] * np.z;
    p.n2 = (R[2] * np.x + R[5] * np.y) + R[8] * np.z;
    p.az = __builtin_fabsf(p.d2);
    int pu = f2i_rn((p.d0 / p.az) * intr[0] + intr[2]);
    int pv = f2i_rn((p.d1 / p.az) * intr[1] + intr[3]);
    p.pu = min(m - 1, max(0, pu));
    p.pv = min(n - 1, max(0, pv));
    p.pidx = (unsigned int)p.pv * (unsigned int)m + (unsigned int)p.pu;
    return p;
}
// The consumers of the gathered values sit behind early exits (zero normal, distance, angle); left alone, the
// compiler sinks each load behind the exit that precedes its first use and the gather becomes three dependent
// memory round trips.  Passing the values through an empty asm pins all the loads before the first test.
__device__ __forceinline__ IcpCurrPx icp_gather(const IcpProj &p, const float *__restrict__ VMap, const float *__restrict__ NMap,
                                                const float *__restrict__ RGB, const float *__restrict__ Gx, size_t gxp,
                                                const float *__restrict__ Gy, size_t gyp) {
    // 32-bit byte offsets from the (scalar) base pointers: one add per address instead of 64-bit multiply-adds
    IcpCurrPx c;
    const unsigned int o12 = 12u * p.pidx;
    c.nc = *(const F3 *)((const char *)NMap + o12);
    c.s = *(const F3 *)((const char *)VMap + o12);
    F3 cc = *(const F3 *)((const char *)RGB + o12);
    float gx = *(const float *)((const char *)Gx + ((unsigned int)p.pv * (unsigned int)gxp + 4u * (unsigned int)p.pu));
    float gy = *(const float *)((const char *)Gy + ((unsigned int)p.pv * (unsigned int)gyp + 4u * (unsigned int)p.pu));
    asm volatile("" : "+v"(c.nc.x), "+v"(c.nc.y), "+v"(c.nc.z), "+v"(c.s.x), "+v"(c.s.y), "+v"(c.s.z), "+v"(cc.x), "+v"(cc.y),
                 "+v"(cc.z), "+v"(gx), "+v"(gy));
    c.ic = ((cc.x + cc.y) + cc.z) / 3.0f;
    c.g0 = (float)((double)gx / 8.0);
    c.g1 = (float)((double)gy / 8.0);
    c.w = icp_depth_weight(c.s.z);
    return c;
}
// The same from the frame's auxiliary records (16 bytes per pixel: g0, g1, ic, w as the fused preprocessing left them,
// rgbdslam_set_frame_aux): one 16-byte load instead of five words from three maps, and the division, the two conversions and
// the f64 quotient of w (~60 issue slots of a ~500-slot block) are paid once per pixel and frame by the preprocessing instead
// of once per sample and iteration here.  The values are formed by the same expressions from the same stored maps.
__device__ __forceinline__ IcpCurrPx icp_gather_aux(const IcpProj &p, const float *__restrict__ VMap, const float *__restrict__ NMap,
                                                    const float *__restrict__ Aux) {
    IcpCurrPx c;
    const unsigned int o12 = 12u * p.pidx;
    c.nc = *(const F3 *)((const char *)NMap + o12);
    c.s = *(const F3 *)((const char *)VMap + o12);
    float4 x = *(const float4 *)((const char *)Aux + 16u * p.pidx);
    asm volatile("; aux gather" : "+v"(c.nc.x), "+v"(c.nc.y), "+v"(c.nc.z), "+v"(c.s.x), "+v"(c.s.y), "+v"(c.s.z), "+v"(x.x), "+v"(x.y),
                 "+v"(x.z), "+v"(x.w));
    c.g0 = x.x; c.g1 = x.y; c.ic = x.z; c.w = x.w;
    return c;
}

// Returns false -> o untouched (the caller passes zeroed rows and needs no copy).  The four rejection tests of
// searchGauss are evaluated side by side and the rows sit behind ONE branch: as a cascade of early exits the compiler
// re-materialised the fourteen zeros at every nesting level (5 x 14 moves per pixel), and the exits never save anything
// unless all 64 pixels of a block fail the same test.
__device__ __forceinline__ bool icp_rows(const IcpModelPx &mp, const IcpProj &pr, const IcpCurrPx &cu, const float *intr,
                                         float distThres, float angleThres, bool inside, IcpRow &o, int &proj) {   // thresholds: icp_sq_threshold of the reference's
    const float d0 = pr.d0, d1 = pr.d1, d2 = pr.d2, n0 = pr.n0, n1 = pr.n1, n2 = pr.n2, az = pr.az;
    const unsigned int pidx = pr.pidx;
    const float i_prev = mp.ip;
    const F3 nc = cu.nc;
    const F3 s = cu.s;
    // The two tests of searchGauss are sqrt(x) > threshold (ICP.cu:118-131).  A correctly rounded square root is monotonic, so
    // each is the same predicate as x > T with T = the largest float whose root does not exceed the threshold -- found once
    // on the host (icp_sq_threshold) -- and the two roots (14 instructions each) are never taken.  NaN and infinity behave
    // alike on both sides (NaN fails both forms of the comparison, i.e. passes the test, as in the reference).
    float ex = d0 - s.x, ey = d1 - s.y, ez = d2 - s.z;
    const float dist2 = (ex * ex + ey * ey) + ez * ez;
    float x0 = nc.y * n2 - nc.z * n1;
    float x1 = (-nc.x) * n2 + nc.z * n0;
    float x2 = nc.x * n1 - nc.y * n0;
    const float ang2 = (x0 * x0 + x1 * x1) + x2 * x2;
    // ICP.cu:96-131: zero model normal, zero current normal, distance, angle; and the sample position beyond the image (ICP.cu:222)
    const bool reject = (int)!inside | (int)is_zero3(mp.np) | (int)is_zero3(nc) | (int)(dist2 > distThres) | (int)(ang2 > angleThres);
    if (reject) return false;

    const float g0 = cu.g0, g1 = cu.g1;
    float g2 = cu.ic - i_prev;
    proj = (int)pidx;

    // ---- rows (D1/ICP.cu:175-205).  Products with the literal 0/1 entries of JD/JRot are
    // dropped: they are exact for finite inputs, so the remaining operations round identically.
    const float w = cu.w;
    float e0 = s.x - d0, e1 = s.y - d1, e2 = s.z - d2;
    float tdx = 2.0f * d0, tdy = 2.0f * d1, tdz = 2.0f * d2;
    float tnx = 2.0f * n0, tny = 2.0f * n1, tnz = 2.0f * n2;
    o.g[0] = w * (-n0);
    o.g[1] = w * (-n1);
    o.g[2] = w * (-n2);
    {
        float a = -(n1 * (-tdz) + n2 * tdy);
        a = a + (-tnz) * e1; a = a + tny * e2;
        o.g[3] = w * a;
        a = -(n0 * tdz + n2 * (-tdx));
        a = a + tnz * e0; a = a + (-tnx) * e2;
        o.g[4] = w * a;
        a = -(n0 * (-tdy) + n1 * tdx);
        a = a + (-tny) * e0; a = a + tnx * e1;
        o.g[5] = w * a;
    }
    o.g[6] = (-w) * ((n0 * e0 + n1 * e1) + n2 * e2);

    const float lambda = 0.03f;
    float JP0 = intr[0] / az, JP3 = intr[1] / az;
    float JP4 = ((-d0) * intr[0]) / (d2 * d2);
    float JP5 = ((-d1) * intr[1]) / (d2 * d2);
    float ga = g0 * JP0, gb = g1 * JP3, gc = g0 * JP4 + g1 * JP5;
    float lw = lambda * w;
    o.c[0] = lw * ga;
    o.c[1] = lw * gb;
    o.c[2] = lw * gc;
    o.c[3] = lw * (gb * (-tdz) + gc * tdy);
    o.c[4] = lw * (ga * tdz + gc * (-tdx));
    o.c[5] = lw * (ga * (-tdy) + gb * tdx);
    o.c[6] = ((-lambda) * w) * g2;
    return true;
}

// ---- 27-term block sum in the pairing of reduce<64> (D1/ICP.cu:28-48) -----------------------------
// Reference tid = col*8 + row, lane = row*8 + col, so the tree's tid offsets 32,16,8 pair COLUMNS
// c|c+4, c|c+2, c|c+1 of one image row (lanes of one 8-lane group) and the offsets 4,2,1 pair ROWS
// r|r+4, r|r+2, r|r+1.  Column levels: three DPP row-shift adds per term (v_add_f32 with a lane
// shift, no LDS, no select); the lanes with col == 0 then hold 8 row partials per term, which are
// transposed through LDS so that lanes 0..26 finish one term each.  IEEE addition is commutative
// and the pairing is the reference's, so the result has the bits of the serial tree.
// row_shl:N -- lane i receives lane i+N of its 16-lane row (0 beyond the row)
template <int N>
__device__ __forceinline__ float add_shl(float v) {
    int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xF, 0xF, true);
    return v + __int_as_float(moved);
}
// ---- integrity word of a published 128-byte line (chain sums, loop state) ---------------------------------------------
// Slot 30 of a line carries the XOR of its payload words (slots 0..29 without the tag slots 7, 15, 23).  A reader that has
// accepted a line by its four sequence words can verify it (rgbdslam_set_icp_line_check): a line whose sectors were NOT
// written or read indivisibly -- the hardware behaviour the sequence-word protocol relies on -- fails the check, is counted
// (rgbdslam_icp_line_rejections) and is read again.  row_shr:N -- lane i receives lane i-N of its 16-lane row (0 before it).
template <int N>
__device__ __forceinline__ unsigned int xor_shr(unsigned int v) {
    return v ^ (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ bool icp_line_payload_slot(int slot) { return slot < 30 && (slot & 7) != 7; }
// XOR of the payload words of the line held one word per lane in lanes 0..31 (wave-uniform result)
__device__ __forceinline__ unsigned int icp_line_xor_lo(unsigned int word, int lane) {
    unsigned int p = (lane < 32 && icp_line_payload_slot(lane)) ? word : 0u;
    p = xor_shr<1>(p); p = xor_shr<2>(p); p = xor_shr<4>(p); p = xor_shr<8>(p);
    return (unsigned int)__builtin_amdgcn_readlane((int)p, 15) ^ (unsigned int)__builtin_amdgcn_readlane((int)p, 31);
}
// both halves of the wave hold a line each (lanes 0..31 and 32..63): true when both match their slot-30 word
__device__ __forceinline__ bool icp_lines_intact(unsigned int word, int lane) {
    unsigned int p = icp_line_payload_slot(lane & 31) ? word : 0u;
    p = xor_shr<1>(p); p = xor_shr<2>(p); p = xor_shr<4>(p); p = xor_shr<8>(p);
    const unsigned int x0 = (unsigned int)__builtin_amdgcn_readlane((int)p, 15) ^ (unsigned int)__builtin_amdgcn_readlane((int)p, 31);
    const unsigned int x1 = (unsigned int)__builtin_amdgcn_readlane((int)p, 47) ^ (unsigned int)__builtin_amdgcn_readlane((int)p, 63);
    return x0 == (unsigned int)__builtin_amdgcn_readlane((int)word, 30) && x1 == (unsigned int)__builtin_amdgcn_readlane((int)word, 62);
}

// tr = this wave's LDS scratch [27][8]; returns the finished sum of term `lane` in lanes 0..26
__device__ __forceinline__ float fold27(const float *v, int lane, float *tr) {
    const int col = lane & 7;
    const int r = lane >> 3;
    float s[27];
#pragma unroll
    for (int k = 0; k < 27; k++) s[k] = add_shl<4>(v[k]);      // level by level: independent DPP chains interleave
    // After the first level only columns 0..3 of a term matter: two terms share a register from here on, term 2j in
    // columns 0..3 and term 2j+1 moved into columns 4..7 (one masked DPP move), so the remaining two levels and the
    // transposing stores run on 14 registers instead of 27.  The pairs added are the same (c | c+2, then c | c+1, for
    // c = 0, 1 of either half), so are the bits.
    float x[14];
#pragma unroll
    for (int j = 0; j < 13; j++)      // row_shr:4 into banks 1 and 3 (lanes 4..7 and 12..15 of a row); the other lanes keep s[2j]
        x[j] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(s[2 * j]), __float_as_int(s[2 * j + 1]), 0x110 + 4, 0xF, 0xA, false));
    x[13] = s[26];
#pragma unroll
    for (int j = 0; j < 14; j++) x[j] = add_shl<2>(x[j]);
#pragma unroll
    for (int j = 0; j < 14; j++) {
        x[j] = add_shl<1>(x[j]);
        asm volatile("" : "+v"(x[j]));                          // or the add is sunk into the branch below and its DPP operand becomes a separate move
    }
    if ((col & 3) == 0) {                                       // column 0 holds term 2j, column 4 term 2j+1
        float *dst = tr + (col >> 2) * 8 + r;
#pragma unroll
        for (int j = 0; j < 13; j++) dst[2 * j * 8] = x[j];
        if (col == 0) tr[26 * 8 + r] = x[13];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float out = 0.0f;
    if (lane < 27) {
        const float4 lo = *(const float4 *)(tr + lane * 8), hi = *(const float4 *)(tr + lane * 8 + 4);
        float y0 = lo.x + hi.x, y1 = lo.y + hi.y, y2 = lo.z + hi.z, y3 = lo.w + hi.w;   // rows r | r+4
        float z0 = y0 + y2, z1 = y1 + y3;                                              // rows r | r+2
        out = z0 + z1;                                                                 // rows 0 | 1
    }
    return out;
}

// ---- one Gauss-Newton iteration: match + both reduction stages (+ solve) in ONE launch -----------
// Workgroup t owns "chain" t of ReduceKernel (D1/ICP.cu:57-58): the 8x8 blocks t, t+512, t+1024...
// Its waves compute those blocks' 27 partial sums (one wave = one block), wave 0 adds them in chain
// order, and the last workgroup of the grid (it waits until every line carries the launch's sequence words) folds the sums per term in the
// pairing of reduce<512> and runs the 6x6 solve -- so one launch replaces the reference's
// memset + MatchKernel + memset + ReduceKernel + D2H + host solve + H2D (D1/ICP.cu:254-286,
// RG.cpp:851-901) with bit-identical sums.
// Row-sharded ICP across GPUs: every rank's tail publishes its band's 27 sums as one 128-byte line into slot
// `rank` of EVERY peer's exchange block (peer-mapped, fine-grained device memory) and then waits until all `world` slots of
// its own block carry this iteration's sequence.  Line = four 32-byte sectors of 7 payload words + 1 tag word, tag =
// seq ^ xor(payload words of the sector): a sector is accepted only when tag ^ xor(payload) == seq, so neither a stale nor
// a torn sector (whatever granularity the fabric writes at) can pass for this iteration's.
// The slots are MULTI-BUFFERED: a rank that has gathered exchange e may publish e + 1 at once, while a slower peer is still
// polling for the lines of e -- with one slot per rank that store would overwrite the line the slow peer has not seen yet
// (with three or more ranks: C's line reaches A before B's does, A runs ahead), and the slow rank would wait for the full
// timeout and lose the frame.  Rule: a rank overwrites a line only two or more exchanges after it wrote it (it publishes e + 2
// after it has gathered every peer's line of e + 1, and a peer publishes e + 1 only after it has consumed e).  A launch owns
// 16 slots per rank -- slot q = 1..15 for its q-th step (the persistent loop may jump over steps, so consecutive EXECUTED
// steps differ by 1..15 slots, never 0; the multi-launch form uses q = 1) -- and consecutive launches alternate between two such
// sets (parity), so the first exchange of a launch never lands on the last one of the launch before.
#define ICP_XCHG_MAX_WORLD 8
#define ICP_XCHG_STEPS 16                            // slots per rank and launch parity (steps 1..15 of a launch)
#define ICP_XCHG_LINE_WORDS(rank, q, parity) (32u * ((unsigned)(rank) + ICP_XCHG_MAX_WORLD * ((unsigned)(q) + ICP_XCHG_STEPS * (unsigned)(parity))))
// tab: DEVICE table of the world block addresses ([rank] = this rank's own); tag = tag_base + q is unique per exchange
struct IcpXchg { unsigned int *const *tab; int world, rank; unsigned int tag_base; int parity; int delay_ticks; };   // delay_ticks: test hook, see rgbdslam_debug_set_exchange_delay
struct IcpIterArgs {
    int nx_lvl, nx_rows, nx_cols, nx_br_lo;          // persistent loop: the level after this one (-1: none), its block grid (band)
    IcpXchg xchg;                                    // world == 0: no exchange
    const float *intr, *R, *t;                       // device scalars (user buffers or &IcpState fields)
    const float *VMap, *NMap, *RGB, *Gx, *Gy;
    size_t gxp, gyp;
    const float *Aux;                                // optional: the frame's auxiliary records (icp_gather_aux), else NULL
    const float *VMap_prev, *NMap_prev, *RGBD_prev;
    float distThres, angleThres;                     // icp_sq_threshold(distThres), icp_sq_threshold(angleThres): see icp_rows
    int n, m, fact;
    int grid_rows, grid_cols;                        // full 8x8-block grid (D1/ICP.cu:260-261)
    int br_lo, rows_local;                           // block-row band handled by this launch (row-sharded ICP)
    int nchain, nwaves;
    float *chains;                                   // 256 lines of 32 words: 27 sums + 4 sequence words (icp_chain_line)
    unsigned int seq;                                // sequence word of this launch / step
    float *sums;                                     // f32[27] out
    int32_t *corr;                                   // optional debug map
    IcpState *state;                                 // optional: gates the launch / receives the solve
    const float *pose_state;
    int do_solve, lvl, iter, max_iter;
    float thr_alpha, thr_tr;                         // icp_sq_threshold_lt of the caller's (icp_pose_update_wave)
    float *trace;
    long long *timing;                               // optional debug: phase time stamps (wall_clock64, 10 ns ticks)
    int *fault;                                      // optional: set to 1 when a wait gives up (rgbdslam_icp_loop_faults)
    int line_check;                                  // verify the integrity word of every accepted line (rgbdslam_set_icp_line_check)
    unsigned int *rejections;                        // lines that carried the right sequence words but failed the check
    int timeout_ticks;                               // give-up time of the tail's wait (10 ns ticks)
    // track_frame path: the first launch builds the loop state itself (k_icp_begin), the last one
    // appends the track-loss test, the commit and the display() pose algebra (k_icp_end)
    int first_launch, final_launch;
    float intr4[4];
    float *pose_state_rw;
    int32_t *status;
    rgbdslam_pose_block *pb;
    float *history;
    int frame_index;
};

// Collective over the calling workgroup (>= 2 waves, every thread must call it): wave 1 evaluates the
// determinant test, thread 0 the Cholesky solve and the pose update.  st = working copy of the state
// (LDS), out = device-resident state it is committed to, sh = 2 floats of LDS.
struct IcpEndArgs { float *pose_state; int32_t *status; rgbdslam_pose_block *pb; float *history; int frame_index; };
// persistent loop: the tail publishes the state in ONE 128-byte line: 25 data words (Rinv 9, tinv 3, Rcurr 9, tcurr 3,
// flags = lvl_iters | cvg << 24 | fail << 25) in slots k + k/7, and the epoch word = base + index of the next step
// that executes (>= nsteps: finished) in the last slot of each 32-byte sector (slots 7, 15, 23, 31), stored after
// the data has landed.  A reader takes the line with one wave-wide load and accepts it when all four copies
// carry the new epoch: whatever way the memory system splits the line, every piece was then read after the data.
struct IcpLoopCtl { unsigned int *pub; unsigned int base; int step, next_level, nsteps, check; };
#define ICP_OFF_REJECT_WORD 33   // unsigned words after the epoch word: [32] fault flag, [33] rejected lines
#define ICP_PUB_DATA 25
__device__ __forceinline__ constexpr int icp_pub_slot(int k) { return k + k / 7; }
__device__ __forceinline__ void icp_solve_step(IcpState *st, IcpState *out, const float *sums, const float *pose_state, int lvl,
                                               int iter, int max_iter, float thr_alpha, float thr_tr, float *trace, float *sh,
                                               bool with_end, IcpEndArgs end, IcpLoopCtl lc, long long *timing = nullptr);
__device__ __forceinline__ void icp_begin_step(IcpState *st, const float *pose_state, const float *intr4);
__device__ __forceinline__ void icp_end_step(const IcpState *st, const IcpEndArgs &e);

// LDS of one workgroup of the iteration kernels
struct IcpShared {
    float (&part)[2][ICP_MAX_WAVES][27];   // block sums of a round; two sets, so that a round needs ONE barrier (below)
    float (&tr)[ICP_MAX_WAVES][27 * 8];
    float (&fin)[27];
    float (&solve_sh)[2];
    IcpState &sst;         // snapshot of the loop state (or of the caller's R, t, intr)
    float (&spose)[24];    // pose_state snapshot for the solve
};
#define ICP_SHARED_DECL(S)                                                       \
    __shared__ float part_[2][ICP_MAX_WAVES][27];                                \
    __shared__ __attribute__((aligned(16))) float tr_[ICP_MAX_WAVES][27 * 8];    \
    __shared__ float fin_[27];                                                   \
    __shared__ float solve_sh_[2];                                               \
    __shared__ IcpState sst_;                                                    \
    __shared__ float spose_[24];                                                 \
    IcpShared S = {part_, tr_, fin_, solve_sh_, sst_, spose_}

// One iteration as seen by workgroup blockIdx.x.  LOOP = called from the persistent k_icp_loop: the
// state changes while the kernel runs, so it is read and written with agent-scope atomics (plain
// loads could be served from a stale scalar or vector L1 line), and the tail publishes the next step.
// persistent loop: the model pixels a wave samples in its first two rounds are the same for every iteration of a
// level (only the pose changes), so they stay in registers from one step to the next
// in*: the sample lies inside the image (locate).  n0 / n1: the NEXT level's samples, requested during the first step of
// this level (icp_iter_body) and still raw -- their intensity is formed when the level starts, so nothing waits on them
struct IcpModelCache { IcpModelPx r0, r1; IcpModelRaw n0, n1; int lvl, nlvl; bool in0, in1, nin0, nin1; };
// Line of the chain buffer that holds workgroup t's sums: the 32 workgroups c = 16j + 2v + h that wave v of the tail
// folds (j = 0..15, h = 0..1) are stored back to back, in the order (j, h).
__device__ __forceinline__ int icp_chain_line(int t) { return ((t >> 1) & 7) * 32 + ((t >> 4) << 1) + (t & 1); }

// field-by-field copy: a struct assignment is one memcpy, and when the optimiser cannot split it the private copy
// ends up in scratch -- or is "promoted" to LDS, which costs a read of the dispatch packet per launch
__device__ __forceinline__ void icp_state_copy(IcpState *d, const IcpState *s) {
#pragma unroll
    for (int k = 0; k < 9; k++) { d->Rinv[k] = s->Rinv[k]; d->Rcurr[k] = s->Rcurr[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { d->tinv[k] = s->tinv[k]; d->tcurr[k] = s->tcurr[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) d->intr[k] = s->intr[k];
    d->cvg = s->cvg; d->fail = s->fail; d->iters = s->iters; d->lvl_iters = s->lvl_iters;
}

// XCHG: the row-sharded form -- the tail exchanges its band's 27 sums with the peers' before the solve
template <bool LOOP, bool XCHG>
__device__ __forceinline__ void icp_iter_body(const IcpIterArgs &a, const IcpShared &S, IcpLoopCtl lc, const int tid, const int t,
                                              IcpModelCache &mc) {
    float (&part)[2][ICP_MAX_WAVES][27] = S.part;
    float (&tr)[ICP_MAX_WAVES][27 * 8] = S.tr;
    float (&fin)[27] = S.fin;
    float (&solve_sh)[2] = S.solve_sh;
    IcpState &sst = S.sst;
    float (&spose)[24] = S.spose;
    const int wave = LOOP ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), lane = tid & 63;   // scalar in the loop: see want_next
    const int nblk = a.rows_local * a.grid_cols;
    const int nwg = min(a.nchain, ICP_WGS);             // workgroups of this launch / step
    const int half = wave / ICP_CHAIN_WAVES, wv = wave - half * ICP_CHAIN_WAVES;   // chain of this wave: t + 256 * half
    const int tc = t + ICP_WGS * half;
    const int chain_len = max(0, (nblk - tc + ICP_CHAINS - 1) / ICP_CHAINS);
    const int rounds = ((nblk - t + ICP_CHAINS - 1) / ICP_CHAINS + ICP_CHAIN_WAVES - 1) / ICP_CHAIN_WAVES;   // of the longer chain, t
    long long ts0 = a.timing ? wall_clock64() : 0;
    if (a.timing && t == 0 && tid == 0) a.timing[0] = ts0;

    // Launches that follow a converged / failed iteration leave at once: the two flags are read through a uniform
    // address before any LDS traffic or barrier (the host cannot know the iteration count, so it always enqueues
    // max_iter + 2 launches per level; at [3,4,3] executed iterations two of the twelve are such no-ops).
    if (!LOOP && a.state && !a.first_launch) {
        const int cvg = a.state->cvg, fail = a.state->fail;
        if (fail || (a.iter > 0 && cvg)) {
            if (a.final_launch && t == 0 && tid == 0) {
                IcpState loc = *a.state;
                IcpEndArgs e = {a.pose_state_rw, a.status, a.pb, a.history, a.frame_index};
                icp_end_step(&loc, e);
            }
            return;
        }
    }

    // First memory round trip: the state snapshot (last wave) and round 0's model pixels (all waves)
    // are requested together; nothing below waits on global memory until the gathers.
    if (wave == ICP_MAX_WAVES - 1) {
        float *dst = (float *)&sst;
        if (a.state) {
            const float *src = (const float *)a.state;
            // persistent loop after step 0: the hand-over at the end of the previous step left the published state in
            // sst, and spose is the kernel prologue's
            if ((!LOOP || a.first_launch) && a.pose_state && lane < 24) spose[lane] = a.pose_state[lane];
            if (!a.first_launch) {
                if (!LOOP && lane < (int)(sizeof(IcpState) / 4)) dst[lane] = src[lane];
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane == 0) icp_begin_step(&sst, spose, LOOP ? sst.intr : a.intr4);   // RG.cpp:832-835,851-852 (the loop's prologue has stored the intrinsics)
            }
        } else {
            if (lane < 9) sst.Rcurr[lane] = a.R[lane];
            else if (lane < 12) sst.tcurr[lane - 9] = a.t[lane - 9];
            else if (lane < 16) sst.intr[lane - 12] = a.intr[lane - 12];
            else if (lane == 16) { sst.cvg = 0; sst.fail = 0; }
        }
    }
    int k = wv;
    bool active = k < chain_len, inside = false;
    size_t idx = 0;
    IcpModelPx mp;
    auto locate = [&](int kk) {
        int blk = tc + ICP_CHAINS * kk;                // band-local block number, blk = brow + bcol*rows (ICP.cu:238)
        int br = a.br_lo + blk % a.rows_local, bc = blk / a.rows_local;
        int i = br * 8 + (lane >> 3), j = bc * 8 + (lane & 7);
        inside = (i * a.fact < a.n && j * a.fact < a.m);
        idx = inside ? (size_t)(i * a.fact) * a.m + (size_t)(j * a.fact) : 0;
    };
    bool want_next = false;                           // this step requests the next level's samples (after its own gather)
    if (LOOP) {
        // The level tags are the same in every lane, but live in vector registers: read back as scalars, so that `want_next`
        // is uniform to the compiler and the branch round request_next below is a SCALAR branch.  As a lane-mask branch its
        // if / else had a third path in the control-flow graph (both bodies skipped by their execz jumps) on which the
        // gather was still pending, so the wait counts behind the join were those of a wave WITHOUT the six requests --
        // vmcnt(3) .. vmcnt(0) -- and the first step of a level waited for the next level's samples after all.
        const int have_lvl = __builtin_amdgcn_readfirstlane(mc.lvl), have_next = __builtin_amdgcn_readfirstlane(mc.nlvl);
        if (have_lvl != a.lvl) {                      // first step of a level
            if (have_next == a.lvl) {                 // its samples were requested a level ago
                mc.r0 = icp_model_from_raw(mc.n0); mc.r1 = icp_model_from_raw(mc.n1);
                mc.in0 = mc.nin0; mc.in1 = mc.nin1;
            } else {                                  // first level of the frame: both rounds' pixels are requested at once
                mc.in0 = false; mc.in1 = false;
                if (wv + ICP_CHAIN_WAVES < chain_len) { locate(wv + ICP_CHAIN_WAVES); mc.r1 = icp_load_model(idx, a.VMap_prev, a.NMap_prev, a.RGBD_prev); mc.in1 = inside; }
                if (active) { locate(k); mc.r0 = icp_load_model(idx, a.VMap_prev, a.NMap_prev, a.RGBD_prev); mc.in0 = inside; }
            }
            mc.lvl = a.lvl;
            want_next = a.nx_lvl >= 0;
        }
        // later steps of the level: neither the block's position (two integer divisions) nor its model sample is recomputed
        // (idx is only needed for the correspondence map, which the persistent loop does not write)
        inside = mc.in0;
        mp = mc.r0;
    } else if (active) { locate(k); mp = icp_load_model(idx, a.VMap_prev, a.NMap_prev, a.RGBD_prev); }
    // persistent loop after step 0: nothing above has written to LDS, and the barrier behind the hand-over poll (k_icp_loop)
    // already stands between the new state in sst and its readers
    if (!LOOP || a.first_launch) __syncthreads();
    if (!LOOP && (sst.fail || (a.iter > 0 && sst.cvg))) {   // converged / failed earlier: the whole grid leaves
        if (a.final_launch && t == 0 && tid == 0) {
            IcpEndArgs e = {a.pose_state_rw, a.status, a.pb, a.history, a.frame_index};
            icp_end_step(&sst, e);
        }
        return;
    }

    // The pose and the intrinsics of this step, held in VECTOR registers: left to itself the compiler moves these sixteen
    // uniform LDS values to scalar registers, runs out of those (the loop's arguments fill them) and re-reads each from
    // a spill lane once per use -- two v_readlane per value per round.
    float Rc[9], tc3[3], in4[4];
#pragma unroll
    for (int z = 0; z < 9; z++) Rc[z] = sst.Rcurr[z];
#pragma unroll
    for (int z = 0; z < 3; z++) tc3[z] = sst.tcurr[z];
#pragma unroll
    for (int z = 0; z < 4; z++) in4[z] = sst.intr[z];
    // (one statement: pinned one by one, every LDS read was waited for before the next was issued)
    asm volatile("" : "+v"(Rc[0]), "+v"(Rc[1]), "+v"(Rc[2]), "+v"(Rc[3]), "+v"(Rc[4]), "+v"(Rc[5]), "+v"(Rc[6]), "+v"(Rc[7]), "+v"(Rc[8]),
                 "+v"(tc3[0]), "+v"(tc3[1]), "+v"(tc3[2]), "+v"(in4[0]), "+v"(in4[1]), "+v"(in4[2]), "+v"(in4[3]));

    // The model maps are 12.3 MB at full resolution: read when a level starts, they cost its first step ~3 us of memory
    // time on top of the gather's.  Instead every wave requests its samples of the NEXT level during the first step of
    // this one, AFTER its own gather (loads return in order) and without a branch around any single load (a wait behind a
    // join has to assume the fewest requests): six unconditional loads, from pixel 0 where the wave has no block.
    auto request_next = [&]() {
        int nx_rows = a.nx_rows;
        asm volatile("" : "+s"(nx_rows));             // opaque here: otherwise the index arithmetic (two integer divisions) is
                                                      // hoisted out of the round loop and paid by every step
        const int nfact = 1 << a.nx_lvl, nblk_n = nx_rows * a.nx_cols;
        const int len_n = max(0, (nblk_n - tc + ICP_CHAINS - 1) / ICP_CHAINS);
        unsigned int id[2];
        bool in[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int kk = wv + h * ICP_CHAIN_WAVES;
            const int blk = tc + ICP_CHAINS * kk;
            const int i = ((XCHG ? a.nx_br_lo : 0) + blk % nx_rows) * 8 + (lane >> 3), j = (blk / nx_rows) * 8 + (lane & 7);
            in[h] = kk < len_n && i * nfact < a.n && j * nfact < a.m;
            id[h] = in[h] ? (unsigned int)(i * nfact) * (unsigned int)a.m + (unsigned int)(j * nfact) : 0u;
        }
        mc.n0 = icp_load_model_raw(id[0], a.VMap_prev, a.NMap_prev, a.RGBD_prev);
        mc.n1 = icp_load_model_raw(id[1], a.VMap_prev, a.NMap_prev, a.RGBD_prev);
        mc.nin0 = in[0]; mc.nin1 = in[1];
        mc.nlvl = a.nx_lvl;
    };

    float acc = 0.0f;                                 // wave 0, lanes 0..26: running chain sum (ICP.cu:56-58)
    for (int r = 0; r < rounds; r++) {
        if (r > 0) {
            k = r * ICP_CHAIN_WAVES + wv;
            active = k < chain_len;
            if (active) {
                if (LOOP && r == 1) { mp = mc.r1; inside = mc.in1; }
                else { locate(k); mp = icp_load_model(idx, a.VMap_prev, a.NMap_prev, a.RGBD_prev); }
            }
        }
        if (active) {
            IcpRow q;
#pragma unroll
            for (int z = 0; z < 7; z++) { q.g[z] = 0.0f; q.c[z] = 0.0f; }
            if (LOOP && r == 0 && want_next && t == nwg - 1) request_next();   // the tail: see below
            IcpProj pr = icp_project(mp, Rc, tc3, in4, a.n, a.m);
            IcpCurrPx cu;
            if (a.Aux) cu = icp_gather_aux(pr, a.VMap, a.NMap, a.Aux);      // (uniform)
            else cu = icp_gather(pr, a.VMap, a.NMap, a.RGB, a.Gx, a.gxp, a.Gy, a.gyp);
            int proj = -1;
            const bool ok = icp_rows(mp, pr, cu, in4, a.distThres, a.angleThres, inside, q, proj);
            if (!LOOP && a.corr && inside) a.corr[idx] = ok ? proj : -1;
            // the 27 products of ICP.cu:229 two at a time (v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations per element,
            // half the instructions): row kk times the aligned pairs (0,1) (2,3) (4,5) (6,-) that reach l >= kk
            float v[27];
            {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 G[4], C[4];
#pragma unroll
                for (int z = 0; z < 3; z++) { G[z] = f2{q.g[2 * z], q.g[2 * z + 1]}; C[z] = f2{q.c[2 * z], q.c[2 * z + 1]}; }
                G[3] = f2{q.g[6], q.g[6]}; C[3] = f2{q.c[6], q.c[6]};
                int term = 0;
#pragma unroll
                for (int kk = 0; kk < 6; kk++) {
                    const f2 gk = f2{q.g[kk], q.g[kk]}, ck = f2{q.c[kk], q.c[kk]};
#pragma unroll
                    for (int pi = kk / 2; pi < 4; pi++) {
                        const f2 pr2 = gk * G[pi] + ck * C[pi];
                        if (2 * pi >= kk) v[term++] = pr2.x;
                        if (2 * pi + 1 <= 6) v[term++] = pr2.y;
                    }
                }
            }
            float sum = fold27(v, lane, tr[wave]);
            if (lane < 27) part[r & 1][wave][lane] = sum;
        }
        else if (LOOP && r == 0 && want_next && t == nwg - 1) request_next();
        // ONE barrier per round: the block sums alternate between two sets, and the two accumulating waves pass the next
        // round's barrier only after they have read this round's set -- nobody can overwrite it before (round r + 2)
        __syncthreads();
        // wave 0 adds BOTH chains' blocks in chain order, chain t in lanes 0..26 and chain t + 256 in lanes 32..58: the two
        // sums meet through a lane exchange below instead of an LDS word and one more workgroup barrier per step
        if (wave == 0 && (lane & 31) < 27) {
            const int hb = lane >> 5;
            const int len_h = max(0, (nblk - (t + ICP_WGS * hb) + ICP_CHAINS - 1) / ICP_CHAINS);
            const int cnt = min(ICP_CHAIN_WAVES, len_h - r * ICP_CHAIN_WAVES);
            // all six slots are read at once (one LDS round trip instead of one per block, in a loop whose trip count differs
            // between the lanes) and added in chain order under selects; a slot beyond the chain's end holds an older round's sum
            float pv[ICP_CHAIN_WAVES];
#pragma unroll
            for (int w = 0; w < ICP_CHAIN_WAVES; w++) pv[w] = part[r & 1][hb * ICP_CHAIN_WAVES + w][lane & 31];
#pragma unroll
            for (int w = 0; w < ICP_CHAIN_WAVES; w++) { const float nx = acc + pv[w]; acc = w < cnt ? nx : acc; }
        }
    }
    long long ts1 = a.timing ? wall_clock64() : 0;
    // offset 256 of reduce<512> (ICP.cu:35): chain t + chain t+256, added here instead of in the tail (a chain that does
    // not exist contributes the 0 the reference's memset leaves in its place)
    // Publish the 27 sums to the other XCDs without an L2 write-back and without waiting for an acknowledgement: ONE
    // wave-wide agent-scope (write-through) store puts them into line icp_chain_line(t) of the chain buffer, laid out
    // like the state line of the persistent loop -- term k in slot k + k/7, and a.seq (unique per launch / step) in the
    // last slot of every 32-byte sector.  The reader checks those sequence words (below) instead of anyone waiting.
    if (wave == 0) {
        const int k = lane - (lane >> 3);             // term held by slot `lane`
        const int kq = k < 27 ? k : 0;
        const float v = __shfl(acc, kq) + __shfl(acc, kq + 32);   // chain t + chain t + 256 (a chain that does not exist has added nothing)
        unsigned int word = ((lane & 7) == 7) ? a.seq : __float_as_uint(v);
        if (a.line_check) {                               // slot 30: integrity word of the 27 sums
            const unsigned int chk = icp_line_xor_lo(word, lane);
            if (lane == 30) word = chk;
        }
        if (lane < 32)
            __hip_atomic_store((unsigned int *)a.chains + 32 * icp_chain_line(t) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The next level's samples are requested HERE, behind the step's last use of a loaded value, not inside the match round:
    // there the branch round the six loads joined paths with different numbers of requests in flight, and the wait counts
    // the compiler places behind such a join are those of the path with the fewest -- the first step of a level waited for
    // the next level's samples (12 MB of cold model maps at level 1) in the middle of its match phase.  The tail is the
    // exception: it is about to poll the lines, and loads return in order, so it has made its requests BEFORE its gather
    // (above) -- half a microsecond ahead of everybody else's step, which the hand-over it does not need pays for.
    if (LOOP && want_next && t != nwg - 1) request_next();
    long long ts2 = a.timing ? wall_clock64() : 0;
    // No arrival counter: the LAST workgroup of the step (it is dispatched last and owns the shortest chains) is the tail.
    // It re-reads the published lines until every one carries this step's sequence words -- one round trip after the
    // last line has landed, instead of two dependent counter round trips before the first read.
    if (t != nwg - 1) return;
    long long ts3 = a.timing ? wall_clock64() : 0;

    // second stage, tree part (reduce<512>, ICP.cu:28-48; offset 256 is already in the published sums).  The arrival
    // counter only says that every workgroup's store has been ISSUED: a line is taken when the sequence word of each of
    // its sectors is this step's, and the lines are re-read otherwise -- so nobody ever waits for a store
    // acknowledgement.  Lane l reads slot l%32 of lines 2i + l/32; wave v (of eight) owns the 32 workgroups
    // c = 16j + 2v + l/32, j = 0..15, stored back to back (icp_chain_line): offsets 128..16 are adds between the 16
    // registers of a lane, offsets 8, 4, 2 go through LDS, offset 1 is one lane shuffle.  All loads are issued before the
    // first check; workgroups beyond nwg count as 0.
    if (wave < 8) {
        const unsigned int *src = (const unsigned int *)a.chains + 32 * (32 * wave + (lane >> 5)) + (lane & 31);
        const int c0 = 2 * wave + (lane >> 5);
        const bool is_seq = (lane & 7) == 7;
        float r[16];
        const long long t_poll = wall_clock64();
        for (;;) {                                    // until every workgroup's line of this step has landed
            bool stale = false;
#pragma unroll
            for (int j = 0; j < 16; j++)
                r[j] = __uint_as_float(__hip_atomic_load(src + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const bool live = c0 + 16 * j < nwg;
                stale |= live && __float_as_uint(r[j]) != a.seq;
                r[j] = live ? r[j] : 0.0f;
            }
            if (__ballot(stale && is_seq) == 0ull) {
                if (!a.line_check) break;
                bool intact = true;
#pragma unroll
                for (int j = 0; j < 16; j++) intact &= icp_lines_intact(__float_as_uint(r[j]), lane);
                if (intact) break;
                if (lane == 0 && a.rejections) atomicAdd(a.rejections, 1u);
            }
            if (wall_clock64() - t_poll > (long long)a.timeout_ticks) {  // 50 ms: a workgroup of this step never ran (the device is not ours):
#pragma unroll
                for (int j = 0; j < 16; j++) r[j] = __builtin_nanf("");   // the solve rejects the system, the frame counts as lost
                if (a.fault && lane == 0) __hip_atomic_store(a.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
            for (int j = 0; j < off; j++) r[j] = r[j] + r[j + off];   // offsets 128, 64, 32, 16
        tr[wave][lane] = r[0];
    }
    __syncthreads();
    if (wave == 0) {
        const float q0 = tr[0][lane] + tr[4][lane], q1 = tr[1][lane] + tr[5][lane];   // offset 8
        const float q2 = tr[2][lane] + tr[6][lane], q3 = tr[3][lane] + tr[7][lane];
        const float p0 = q0 + q2, p1 = q1 + q3;                                       // offset 4
        float q = p0 + p1;                                                            // offset 2
        q = q + __shfl_down(q, 32);                                                   // offset 1 (lines 2i | 2i+1)
        const int k = lane - (lane >> 3);
        if (lane < 32 && (lane & 7) != 7 && k < 27) { fin[k] = q; a.sums[k] = q; }
    }
    long long ts4 = a.timing ? wall_clock64() : 0;
    __syncthreads();
    if (XCHG && a.xchg.world > 0) {                   // uniform over the grid; the tail is the only workgroup still here
        if (wave == 0) {
            const unsigned int q = LOOP ? (unsigned int)lc.step + 1u : 1u;         // slot of this exchange within its launch
            const unsigned int tag = a.xchg.tag_base + q;
            // publish: slot l of the line = term l - l/8 (as in the chain lines), tag words in slots 7, 15, 23, 31
            const int k = lane - (lane >> 3);
            unsigned int word = ((lane & 7) != 7 && k < 27 && lane < 32) ? __float_as_uint(fin[k]) : 0u;
            unsigned int x = word;
            x ^= (unsigned int)__shfl_xor((int)x, 1); x ^= (unsigned int)__shfl_xor((int)x, 2); x ^= (unsigned int)__shfl_xor((int)x, 4);
            if ((lane & 7) == 7) word = tag ^ x;
            for (int p = 0; p < a.xchg.world; p++) {
                unsigned int *peer = a.xchg.tab[p];
                if (lane < 32)
                    __hip_atomic_store(peer + ICP_XCHG_LINE_WORDS(a.xchg.rank, q, a.xchg.parity) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (a.xchg.delay_ticks > 0) {            // test hook: this rank falls behind between its publication and its first poll
                const long long t_d = wall_clock64();
                while (wall_clock64() - t_d < (long long)a.xchg.delay_ticks) __builtin_amdgcn_s_sleep(32);
            }
            // gather: lane reads word lane % 32 of lines 2i + lane / 32 of the own block
            const unsigned int *mine = a.xchg.tab[a.xchg.rank] + ICP_XCHG_LINE_WORDS(0, q, a.xchg.parity);
            unsigned int v[ICP_XCHG_MAX_WORLD / 2];
            const long long t_poll = wall_clock64();
            bool gave_up = false;
            for (;;) {
                bool bad = false;
#pragma unroll
                for (int i = 0; i < ICP_XCHG_MAX_WORLD / 2; i++) {
                    const int line = 2 * i + (lane >> 5);
                    const bool live = line < a.xchg.world;
                    v[i] = live ? __hip_atomic_load(mine + 32 * line + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                    unsigned int y = v[i];
                    y ^= (unsigned int)__shfl_xor((int)y, 1); y ^= (unsigned int)__shfl_xor((int)y, 2); y ^= (unsigned int)__shfl_xor((int)y, 4);
                    bad |= live && y != tag;
                }
                if (__ballot(bad) == 0ull) break;
                if (wall_clock64() - t_poll > (long long)a.timeout_ticks) { gave_up = true; break; }   // 50 ms: a peer never published
                __builtin_amdgcn_s_sleep(2);
            }
            // rank-ordered total (the order of k_icp_solve), through this wave's LDS scratch
            unsigned int *lds = (unsigned int *)tr[0];
#pragma unroll
            for (int i = 0; i < ICP_XCHG_MAX_WORLD / 2; i++) lds[32 * (2 * i + (lane >> 5)) + (lane & 31)] = v[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 27) {
                const int slot = lane + lane / 7;
                float tot = __uint_as_float(lds[slot]);
                for (int p = 1; p < a.xchg.world; p++) tot = tot + __uint_as_float(lds[32 * p + slot]);
                if (gave_up) tot = __builtin_nanf("");
                fin[lane] = tot;
                a.sums[lane] = tot;
            }
            if (gave_up && lane == 0 && a.fault) __hip_atomic_store(a.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (a.do_solve) {
        IcpEndArgs e = {a.pose_state_rw, a.status, a.pb, a.history, a.frame_index};
        icp_solve_step(&sst, a.state, fin, spose, a.lvl, a.iter, a.max_iter, a.thr_alpha, a.thr_tr, a.trace, solve_sh,
                       a.final_launch != 0, e, lc, a.timing);    // by value: a pointer to the local would put it in scratch memory
    }
    if (a.timing && tid == 0) {
        a.timing[1] = ts0; a.timing[2] = ts1; a.timing[3] = ts2; a.timing[4] = ts3; a.timing[5] = ts4; a.timing[6] = wall_clock64();
    }
}

// 256 workgroups of 10 waves: one per CU, all 512 chains resident at once
__global__ __launch_bounds__(64 * ICP_MAX_WAVES)
void k_icp_iter(IcpIterArgs a) {
    ICP_SHARED_DECL(S);
    IcpModelCache mc;
    icp_iter_body<false, true>(a, S, IcpLoopCtl{nullptr, 0u, 0, 0, 0, 0}, (int)threadIdx.x, (int)blockIdx.x, mc);
}

// ---- persistent coarse-to-fine loop: ALL iterations of a frame in one launch -----------------------
// Workgroup t keeps chain t for every step.  Between two Gauss-Newton iterations the grid meets at an
// epoch word: the tail workgroup of step s (the last workgroup of the step's grid) solves, commits
// the state and stores base + (next step that executes) -- the steps the multi-launch loop would have
// started as no-ops (after convergence) are jumped over; >= nsteps ends the kernel.  The host only
// launches this kernel when every workgroup fits on the device at once (rgbdslam_set_icp_loop), and a
// waiting workgroup gives up after timeout_ticks so that every wave reaches an exit even if that
// assumption is broken by another process sharing the device.
// steps are numbered through the levels, coarse to fine; level k owns [start, start + max_iter + 2)
// packed (two scalar registers per level instead of five -- the loop is short of them): w0 = start | max_iter << 16 | lvl << 28,
// w1 = grid_rows | grid_cols << 16
struct IcpLoopLevel { unsigned int w0, w1, w2; };    // w2 (row-sharded form only) = first block row of the band | rows of the band << 16
#define ICP_LV_UNUSED 0xffffu
// The argument block is kept small on purpose: every field stays in a scalar register for the whole
// loop, and what does not fit is spilled into vector registers and scratch memory.  The five ICP work
// buffers are therefore one allocation with fixed offsets, and pose_state / status come from pb.
#define ICP_OFF_CHAINS 0
#define ICP_OFF_OUT (ICP_WGS * 128)              // after the 256 chain lines
#define ICP_OFF_STATE (ICP_OFF_OUT + 128)
#define ICP_OFF_EPOCH (ICP_OFF_STATE + 128)      // fault flag of k_icp_loop at + 128
#define ICP_OFF_PUB (ICP_OFF_EPOCH + 256)        // published loop state: one 128-byte line
#define ICP_BLOCK_BYTES (ICP_OFF_PUB + 128)
static_assert(ICP_OFF_PUB % 128 == 0 && offsetof(IcpState, tcurr) == 21 * 4, "published state layout");
static_assert(sizeof(IcpState) <= 128, "state slot");
struct IcpLoopArgs {
    const float *VMap, *NMap, *RGB, *Gx, *Gy, *VMap_prev, *NMap_prev, *RGBD_prev;
    const float *Aux;                                // the frame's auxiliary records or NULL (icp_gather_aux)
    unsigned int gxp, gyp;
    char *block;                                     // the ICP work block (offsets above)
    rgbdslam_pose_block *pb;
    float *history;
    int frame_index;
    float distThres, angleThres, thr_alpha, thr_tr;  // the first two: icp_sq_threshold of the caller's (icp_rows)
    int n, m;
    float intr4[4];
    IcpLoopLevel lv[3];                              // unused entries: start = ICP_LV_UNUSED
    int nsteps;
    unsigned int epoch_base;
    int timeout_ticks;
    // row-sharded form (k_icp_loop<.., true>): the peers' exchange blocks and this launch's tags / slot set
    unsigned int *const *xtab; unsigned int xtag_base; int xworld, xrank, xparity, xdelay;
#ifdef ICP_LOOP_TIMING
    long long *timing;                               // debug build only: 8 time stamps per step (rgbdslam_debug_icp_timing)
#endif
};

// CHECK = rgbdslam_set_icp_line_check, a template parameter so that the default kernel carries none of the checking code
template <bool CHECK, bool XCHG = false>
__global__ __launch_bounds__(64 * ICP_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(3)))
void k_icp_loop(IcpLoopArgs L) {
    ICP_SHARED_DECL(S);
    __shared__ int next_step;
    const int t = blockIdx.x;
    unsigned int *const pub = (unsigned int *)(L.block + ICP_OFF_PUB);
    int s = 0;
    IcpModelCache mc;
    mc.lvl = -1; mc.nlvl = -1;
    // what no later step refreshes: the pose pair of the frame and the intrinsics (a workgroup whose chain does not
    // exist at the coarsest level never runs step 0's begin)
    if (threadIdx.x < 24) spose_[threadIdx.x] = L.pb->pose_state[threadIdx.x];
    else if (threadIdx.x >= 32 && threadIdx.x < 36) sst_.intr[threadIdx.x - 32] = L.intr4[threadIdx.x - 32];
    for (;;) {
        // the level's constants are picked with uniform selects (a dynamically indexed kernarg array would be
        // copied to scratch memory and turn every one of them into a vector register)
        // fresh (opaque) thread and workgroup ids every step: otherwise the compiler hoists every address it can
        // form from them out of the loop and then spills those registers to scratch memory
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const bool in2 = s >= (int)(L.lv[2].w0 & 0xffffu), in1 = s >= (int)(L.lv[1].w0 & 0xffffu);
        const unsigned int lw0 = in2 ? L.lv[2].w0 : (in1 ? L.lv[1].w0 : L.lv[0].w0);
        const unsigned int lw1 = in2 ? L.lv[2].w1 : (in1 ? L.lv[1].w1 : L.lv[0].w1);
        const int grid_rows = (int)(lw1 & 0xffffu), grid_cols = (int)(lw1 >> 16);
        int br_lo = 0, rows_local = grid_rows;
        if (XCHG) {                                   // this rank's band of the level's block rows
            const unsigned int lw2 = in2 ? L.lv[2].w2 : (in1 ? L.lv[1].w2 : L.lv[0].w2);
            br_lo = (int)(lw2 & 0xffffu); rows_local = (int)(lw2 >> 16);
        }
        const int nchain = min(rows_local * grid_cols, ICP_CHAINS);
        if (t < min(nchain, ICP_WGS)) {
            IcpIterArgs a;
            if (XCHG) {
                a.xchg.tab = L.xtab; a.xchg.world = L.xworld; a.xchg.rank = L.xrank; a.xchg.tag_base = L.xtag_base;
                a.xchg.parity = L.xparity; a.xchg.delay_ticks = L.xdelay;
            } else a.xchg.world = 0;
            a.fault = (int *)(L.block + ICP_OFF_EPOCH + 128);
            a.line_check = CHECK; a.rejections = (unsigned int *)(L.block + ICP_OFF_EPOCH) + ICP_OFF_REJECT_WORD;
            a.timeout_ticks = L.timeout_ticks;
            a.intr = nullptr; a.R = nullptr; a.t = nullptr;
            a.VMap = L.VMap; a.NMap = L.NMap; a.RGB = L.RGB; a.Gx = L.Gx; a.Gy = L.Gy; a.gxp = L.gxp; a.gyp = L.gyp; a.Aux = L.Aux;
            a.VMap_prev = L.VMap_prev; a.NMap_prev = L.NMap_prev; a.RGBD_prev = L.RGBD_prev;
            a.distThres = L.distThres; a.angleThres = L.angleThres;
            a.n = L.n; a.m = L.m;
            a.lvl = (int)(lw0 >> 28); a.max_iter = (int)((lw0 >> 16) & 0xfffu);
            {
                const unsigned int nw0 = in2 ? ICP_LV_UNUSED : (in1 ? L.lv[2].w0 : L.lv[1].w0);
                const unsigned int nw1 = in1 ? L.lv[2].w1 : L.lv[1].w1;
                a.nx_lvl = (nw0 & 0xffffu) == ICP_LV_UNUSED ? -1 : (int)(nw0 >> 28);
                a.nx_rows = (int)(nw1 & 0xffffu); a.nx_cols = (int)(nw1 >> 16); a.nx_br_lo = 0;
                if (XCHG) {
                    const unsigned int nw2 = in1 ? L.lv[2].w2 : L.lv[1].w2;
                    a.nx_br_lo = (int)(nw2 & 0xffffu); a.nx_rows = (int)(nw2 >> 16);
                }
            }
            a.fact = 1 << a.lvl; a.grid_rows = grid_rows; a.grid_cols = grid_cols;
            a.br_lo = br_lo; a.rows_local = rows_local; a.nchain = nchain; a.nwaves = ICP_MAX_WAVES;
            a.chains = (float *)(L.block + ICP_OFF_CHAINS);
            a.sums = (float *)(L.block + ICP_OFF_OUT); a.corr = nullptr; a.seq = L.epoch_base + (unsigned int)s + 1u;
            a.state = (IcpState *)(L.block + ICP_OFF_STATE); a.pose_state = L.pb->pose_state;
            a.do_solve = 1;
            const int start = (int)(lw0 & 0xffffu);
            a.iter = s - start;
            a.thr_alpha = L.thr_alpha; a.thr_tr = L.thr_tr; a.trace = nullptr; a.timing = nullptr;
#ifdef ICP_LOOP_TIMING
            if (L.timing) a.timing = L.timing + 8 * s;
#endif
            a.first_launch = (s == 0); a.final_launch = 0;
            a.pose_state_rw = L.pb->pose_state; a.status = L.pb->status; a.pb = L.pb; a.history = L.history;
            a.frame_index = L.frame_index;
            icp_iter_body<true, XCHG>(a, S, IcpLoopCtl{pub, L.epoch_base, s, start + a.max_iter + 2, L.nsteps, CHECK}, tid, t, mc);
        }
        // No barrier in front of the poll: wave 0 writes the new state into sst_, whose readers (the pose reads at the top of
        // the step) all lie in front of the step's round barriers, which wave 0 has passed too; the tail's other LDS traffic
        // (tree, solve) ends at barriers inside icp_iter_body.
        if (tid < 64) {                                // wave 0 polls the published line, one word per lane
            const long long t0 = wall_clock64();
            int nxt;
            unsigned int w = 0;
            for (;;) {
                w = __hip_atomic_load(pub + (tid & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int e0 = __builtin_amdgcn_readlane(w, 7), e1 = __builtin_amdgcn_readlane(w, 15),
                                   e2 = __builtin_amdgcn_readlane(w, 23), e3 = __builtin_amdgcn_readlane(w, 31);
                const int d = (int)(e0 - L.epoch_base);
                if (d > s && e1 == e0 && e2 == e0 && e3 == e0) {
                    if (!CHECK || icp_line_xor_lo(w, tid) == (unsigned int)__builtin_amdgcn_readlane((int)w, 30)) { nxt = d; break; }
                    if (tid == 0) atomicAdd((unsigned int *)(L.block + ICP_OFF_EPOCH) + ICP_OFF_REJECT_WORD, 1u);
                }
                if (wall_clock64() - t0 > (long long)L.timeout_ticks) { nxt = -1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (nxt >= 0 && tid < 32 && (tid & 7) != 7) {
                const int k = tid - (tid >> 3);        // inverse of icp_pub_slot
                float *dst = (float *)&sst_;           // Rinv, tinv, Rcurr, tcurr are the first 24 words of IcpState
                if (k < 24) dst[k] = __uint_as_float(w);
                else if (k == 24) {
                    const int li = (int)(w & 0xffffff);
                    sst_.lvl_iters = li; sst_.iters = (li & 255) + ((li >> 8) & 255) + ((li >> 16) & 255);
                    sst_.cvg = (w >> 24) & 1; sst_.fail = (w >> 25) & 1;
                }
            }
            if (tid == 0) next_step = nxt;
        }
        __syncthreads();
        const int nxt = next_step;
        if (nxt < 0) {                                 // broken residency assumption: report and leave
            if (tid == 0) {
                __hip_atomic_store((int *)(L.block + ICP_OFF_EPOCH + 128), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == 0) L.pb->status[0] = 0;
            }
            return;
        }
        if (nxt >= L.nsteps) return;
        s = nxt;
    }
}

// ---- device-side 6x6 solve and pose update (D1/RG.cpp:851-901) ---------------------------------
#define M3(a, r, c) (a)[(c) * 3 + (r)]

__device__ __forceinline__ void d_mat3_mul(const float *a, const float *b, float *o) {
    float r[9];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            r[c * 3 + i] = (M3(a, i, 0) * M3(b, 0, c) + M3(a, i, 1) * M3(b, 1, c)) + M3(a, i, 2) * M3(b, 2, c);
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = r[k];
}
__device__ __forceinline__ void d_mat3_vec(const float *a, const float *v, float *o) {
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; i++) r[i] = (M3(a, i, 0) * v[0] + M3(a, i, 1) * v[1]) + M3(a, i, 2) * v[2];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
__device__ __forceinline__ void d_inverse3(const float *a, float *o) {
    float cof[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            cof[i][j] = M3(a, i1, j1) * M3(a, i2, j2) - M3(a, i1, j2) * M3(a, i2, j1);
        }
    float det = (cof[0][0] * M3(a, 0, 0) + cof[1][0] * M3(a, 1, 0)) + cof[2][0] * M3(a, 2, 0);
    float invdet = 1.0f / det;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M3(o, i, j) = cof[j][i] * invdet;
}
__device__ __forceinline__ float d_frob_minus_I(const float *a) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float v = M3(a, r, c) - (r == c ? 1.0f : 0.0f);
            s = s + v * v;
        }
    return __builtin_sqrtf(s);
}
__device__ __forceinline__ float d_norm3(const float *v) { return __builtin_sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
// the arguments of the two square roots above, for comparisons against icp_sq_threshold_lt thresholds
__device__ __forceinline__ float d_frob_minus_I_sq(const float *a) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float v = M3(a, r, c) - (r == c ? 1.0f : 0.0f);
            s = s + v * v;
        }
    return s;
}
__device__ __forceinline__ float d_norm3_sq(const float *v) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; }

// ---- 6x6 helpers (stand-ins for Eigen's determinant() and llt().solve(), same operation order as
// the oracle's orc_det6 / orc_llt_solve6) ----------------------------------------------------------
// A and b come straight from the 27 packed sums (ICP.cu:273-283): (i,j), i <= j <= 6, row-major.
__device__ __forceinline__ constexpr int tri_idx(int i, int j) { return i * 7 - (i * (i - 1)) / 2 + (j - i); }
#define SYM_A(S, i, j) ((S)[tri_idx((i) < (j) ? (i) : (j), (i) < (j) ? (j) : (i))])

// value of v in lane src (src must be wave-uniform): v_readlane, a few cycles -- unlike __shfl, which goes through
// the LDS crossbar (ds_bpermute) even for a constant lane
__device__ __forceinline__ float lane_read(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// Partial-pivot LU determinant, one matrix ROW per lane (lanes 0..5), columns in registers: no
// dynamically indexed private arrays (which would live in scratch memory) and no LDS round trips.
// Pivot search and row exchange are the serial algorithm's, done with lane reads (columns left of the
// pivot are never looked at again and are not exchanged).
__device__ __forceinline__ float wave_det6(const float *sums, int lane) {
    float a[6];
    // the 21 matrix entries in ONE LDS round trip (left alone the reads are issued and waited for in five or six batches,
    // each next to its first use), then the rows are picked by lane
    float sv[21];
#pragma unroll
    for (int z = 0; z < 21; z++) sv[z] = sums[z < 6 ? z : z + (z >= 11) + (z >= 15) + (z >= 18) + (z >= 20) + 1];   // tri_idx(i, j) for j < 6: skips the b entries
    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]), "+v"(sv[8]),
                 "+v"(sv[9]), "+v"(sv[10]), "+v"(sv[11]), "+v"(sv[12]), "+v"(sv[13]), "+v"(sv[14]), "+v"(sv[15]), "+v"(sv[16]), "+v"(sv[17]),
                 "+v"(sv[18]), "+v"(sv[19]), "+v"(sv[20]));
    auto at = [&](int i, int j) { const int lo = i < j ? i : j, hi = i < j ? j : i; return sv[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)]; };
#pragma unroll
    for (int k = 0; k < 6; k++) {
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < 6; r++) v = (lane == r) ? at(r, k) : v;
        a[k] = v;
    }
    float det = 1.0f;
    bool done = false;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        int p = c;
        float best = __builtin_fabsf(lane_read(a[c], c));
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
            float v = __builtin_fabsf(lane_read(a[c], r));
            if (v > best) { best = v; p = r; }
        }
        p = __builtin_amdgcn_readfirstlane(p);
        float rowp[6];
#pragma unroll
        for (int k = c; k < 6; k++) {
            float vp = lane_read(a[k], p), vc = lane_read(a[k], c);
            rowp[k] = vp;
            a[k] = (lane == c) ? vp : ((lane == p) ? vc : a[k]);   // exchange rows c and p (no-op when p == c)
        }
        float piv = rowp[c];
        // (no exit inside the elimination either: the value the serial algorithm returns at a zero / NaN pivot is frozen)
        const float dn = (p != c ? -det : det) * piv;
        det = done ? det : dn;
        done |= (piv == 0.0f || piv != piv);
        if (lane > c && lane < 6) {
            float f = a[c] / piv;
#pragma unroll
            for (int k = c + 1; k < 6; k++) a[k] = a[k] - f * rowp[k];
        }
    }
    return det;
}

// Row Cholesky + forward/backward substitution, fully unrolled in registers.  Returns 1 on a
// non-positive pivot.
__device__ __forceinline__ int llt_solve6_reg(const float *sums, float *x) {
    float L[6][6], y[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        float s = SYM_A(sums, k, k);
#pragma unroll
        for (int p = 0; p < k; p++) s = s - L[k][p] * L[k][p];
        if (!(s > 0.0f)) return 1;
        float lkk = __builtin_sqrtf(s);
        L[k][k] = lkk;
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            float v = SYM_A(sums, r, k);
#pragma unroll
            for (int p = 0; p < k; p++) v = v - L[r][p] * L[k][p];
            L[r][k] = v / lkk;
        }
    }
#pragma unroll
    for (int r = 0; r < 6; r++) {
        float v = sums[tri_idx(r, 6)];
#pragma unroll
        for (int p = 0; p < r; p++) v = v - L[r][p] * y[p];
        y[r] = v / L[r][r];
    }
#pragma unroll
    for (int r = 5; r >= 0; r--) {
        float v = y[r];
#pragma unroll
        for (int p = r + 1; p < 6; p++) v = v - L[p][r] * x[p];
        x[r] = v / L[r][r];
    }
    return 0;
}

// The same solve spread over the lanes of one wave: lane r (0..5) owns ROW r of the factor.  Column k of the factor is
// independent across rows, so its divisions and dot products run side by side.  The forward substitution rides along as
// a SEVENTH row, b (lane 6): y[k] = (b[k] - sum_{p<k} L[k][p] y[p]) / L[k][k] is exactly the recurrence of a factor row
// (the products commute, the subtractions come in the same ascending order), so it costs no instruction of its own.  The
// backward substitution, whose rows must subtract in ascending order too, runs on the gathered factor.  Every element
// sees exactly the operations of llt_solve6_reg in the same order: the bits are the same, the instruction count on the
// critical path of the persistent loop drops from ~900 to ~400.  sums = 27 floats in LDS; x is valid in EVERY lane.
__device__ __forceinline__ int wave_llt_solve6(const float *sums, int lane, float *x) {
    const int r = lane < 6 ? lane : 6;
    float Arow[6], Lr[6];
    bool fail = false;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int i = r < k ? r : k, j = r < k ? k : r;
        Arow[k] = sums[i * 7 - (i * (i - 1)) / 2 + (j - i)];     // SYM_A(sums, r, k) for r < 6, b[k] for r == 6
        Lr[k] = 0.0f;
    }
    // all six reads in flight before the first column: left alone, each is sunk to the column that uses it and becomes an
    // LDS round trip of its own on the serial path
    asm volatile("" : "+v"(Arow[0]), "+v"(Arow[1]), "+v"(Arow[2]), "+v"(Arow[3]), "+v"(Arow[4]), "+v"(Arow[5]));
#pragma unroll
    for (int k = 0; k < 6; k++) {
        float s = Arow[k], w = Arow[k], Lk[6];
#pragma unroll
        for (int p = 0; p < k; p++) {
            Lk[p] = lane_read(Lr[p], k);
            s = s - Lr[p] * Lr[p];                               // meaningful in lane k
            w = w - Lr[p] * Lk[p];                               // meaningful in lanes r > k (lane 6: the forward substitution)
        }
        const float sk = lane_read(s, k);
        // no exit inside the factorisation: a non-positive pivot is remembered and the arithmetic goes on (on values nobody
        // will use) -- six early exits cut the unrolled code into seven basic blocks, and the scheduler could not move a
        // column's products with the columns before the last one into the shadow of that one's square root and division
        fail |= !(sk > 0.0f);
        const float lkk = __builtin_sqrtf(sk);
        Lr[k] = (r == k) ? lkk : w / lkk;
    }
    float y[6];
#pragma unroll
    for (int p = 0; p < 6; p++) y[p] = lane_read(Lr[p], 6);
    float L[6][6];
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
        for (int c = 0; c <= p; c++) L[p][c] = lane_read(Lr[c], p);
#pragma unroll
    for (int q = 5; q >= 0; q--) {
        float u = y[q];
#pragma unroll
        for (int p = q + 1; p < 6; p++) u = u - L[p][q] * x[p];
        x[q] = u / L[q][q];
    }
    return fail ? 1 : 0;
}

// pose_state = poseR(9) poset(3) poseRPrev(9) posetPrev(3)
__device__ __forceinline__ void d_set_rt_curr(IcpState *st, const float *ps) {
    float tmp[3];
    d_mat3_mul(st->Rinv, ps + 12, st->Rcurr);                 // RG.cpp:851
    d_mat3_vec(st->Rinv, ps + 21, tmp);                       // RG.cpp:852
    st->tcurr[0] = tmp[0] + st->tinv[0]; st->tcurr[1] = tmp[1] + st->tinv[1]; st->tcurr[2] = tmp[2] + st->tinv[2];
}

__device__ __forceinline__ void icp_begin_step(IcpState *st, const float *pose_state, const float *intr4) {
    float tmp[3];
    d_inverse3(pose_state, st->Rinv);                         // RG.cpp:834
    d_mat3_vec(st->Rinv, pose_state + 9, tmp);                // RG.cpp:835
    st->tinv[0] = -tmp[0]; st->tinv[1] = -tmp[1]; st->tinv[2] = -tmp[2];
    for (int k = 0; k < 4; k++) st->intr[k] = intr4[k];
    st->cvg = 0; st->fail = 0; st->iters = 0; st->lvl_iters = 0;
    d_set_rt_curr(st, pose_state);
}
__global__ void k_icp_begin(IcpState *st, const float *pose_state, CalibV c, int32_t *status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    icp_begin_step(st, pose_state, c.v);
    status[0] = 0; status[1] = 0;
}

// Pose update of RG.cpp:874-889 for a solved increment x, by ONE WAVE, in the lane layout of a published 128-byte line:
// lane l < 32 (l % 8 != 7) owns word k = l - l/8 of IcpState -- k < 9 Rinv[k], 9..11 tinv, 12..20 Rcurr, 21..23 tcurr.  x is
// valid in every lane; the quaternion -> matrix part (f64) and the convergence test run redundantly in all lanes, the
// 3x3 products element-wise, each element with the operations and the order of d_mat3_mul / d_mat3_vec; the new Rinv /
// tinv reach the lanes that form Rcurr / tcurr through lane reads.  The result is the lane's word of the NEW state: it is
// stored to the line (and to LDS) as it is -- no copy of the state, no LDS round trip between update and publication.
// As one thread on a private copy this was ~250 dependent-issue instructions plus 56 LDS accesses per iteration.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ constexpr int icp_slot_of_word(int k) { return k + k / 7; }
// old_word: word k of the current state (what the lane keeps where nothing changes).  Returns the new word k.
__device__ __forceinline__ float icp_pose_update_wave(const IcpState *st, const float *x, const float *pose_state, int lvl, int iter,
                                                      int max_iter, float thr_alpha, float thr_tr, int k, float old_word, int &cvg_out) {
    // everything the update reads from LDS, requested before the arithmetic
    const bool role_b = k >= 12;                                   // this lane forms a word of Rcurr / tcurr
    const int kk = role_b ? k - 12 : k;
    const bool is_t = kk >= 9;                                     // ... of tinv / tcurr
    const int e = is_t ? (kk - 9) % 3 : kk % 9, i = e % 3, c = is_t ? 0 : e / 3;
    const float *col = is_t ? st->tinv : st->Rinv + 3 * c;         // old tinv | column c of the old Rinv
    const float b0 = col[0], b1 = col[1], b2 = col[2];
    const float *pc = is_t ? pose_state + 21 : pose_state + 12 + 3 * c;   // tPrev | column c of RPrev
    const float p0 = pc[0], p1 = pc[1], p2 = pc[2];

    double q[4];
    double nrm = (double)((x[3] * x[3] + x[4] * x[4]) + x[5] * x[5]);   // RG.cpp:875
    q[1] = x[3]; q[2] = x[4]; q[3] = x[5]; q[0] = __builtin_sqrt(1 - nrm);
    // D1/Utilities.cpp:1145-1167 quaternion2matrix
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q03 = q[0] * q[3], q13 = q[1] * q[3], q23 = q[2] * q[3];
    double q02 = q[0] * q[2], q12 = q[1] * q[2], q01 = q[0] * q[1];
    float Rinc[9];
    M3(Rinc, 0, 0) = (float)(q00 + q11 - q22 - q33);
    M3(Rinc, 1, 1) = (float)(q00 - q11 + q22 - q33);
    M3(Rinc, 2, 2) = (float)(q00 - q11 - q22 + q33);
    M3(Rinc, 0, 1) = (float)(2.0 * (q12 - q03));
    M3(Rinc, 1, 0) = (float)(2.0 * (q12 + q03));
    M3(Rinc, 0, 2) = (float)(2.0 * (q13 + q02));
    M3(Rinc, 2, 0) = (float)(2.0 * (q13 - q02));
    M3(Rinc, 1, 2) = (float)(2.0 * (q23 - q01));
    M3(Rinc, 2, 1) = (float)(2.0 * (q23 + q01));
    const float tinc[3] = {x[0], x[1], x[2]};
    // RG.cpp:889, sqrt(x) < threshold as x < icp_sq_threshold_lt(threshold): the same predicate (monotonic sqrt), no root on the
    // serial path of the iteration
    const int cvg = (iter > max_iter) || (d_frob_minus_I_sq(Rinc) < thr_alpha && d_norm3_sq(tinc) < thr_tr);
    cvg_out = cvg;
    // row i of Rinc
    const float a0 = i == 0 ? Rinc[0] : (i == 1 ? Rinc[1] : Rinc[2]);
    const float a1 = i == 0 ? Rinc[3] : (i == 1 ? Rinc[4] : Rinc[5]);
    const float a2 = i == 0 ? Rinc[6] : (i == 1 ? Rinc[7] : Rinc[8]);
    float nv = (a0 * b0 + a1 * b1) + a2 * b2;                      // RG.cpp:887 element (i, c) | RG.cpp:886 row i
    if (is_t) nv = nv + (i == 0 ? tinc[0] : (i == 1 ? tinc[1] : tinc[2]));
    // (meaningful in the lanes of words 0..11; the others computed the same element a second time)
    // Refresh the uploaded pair (RG.cpp:851-852) for whichever iteration runs next: the next one of
    // this level (iff !cvg) or iteration 0 of the next level (always).  After the final executed
    // iteration of level 0 the pair is left untouched: the track-loss test reads the LAST
    // uploaded pair, not the final estimate (RG.cpp:910).
    float cv = old_word;
    {
        float nR[9], nt[3];
#pragma unroll
        for (int z = 0; z < 9; z++) nR[z] = lane_read(nv, icp_slot_of_word(z));          // the new Rinv
#pragma unroll
        for (int z = 0; z < 3; z++) nt[z] = lane_read(nv, icp_slot_of_word(9 + z));      // the new tinv
        const float r0 = i == 0 ? nR[0] : (i == 1 ? nR[1] : nR[2]);                      // its row i
        const float r1 = i == 0 ? nR[3] : (i == 1 ? nR[4] : nR[5]);
        const float r2 = i == 0 ? nR[6] : (i == 1 ? nR[7] : nR[8]);
        float cn = (r0 * p0 + r1 * p1) + r2 * p2;                                        // RG.cpp:851 | 852
        if (is_t) cn = cn + (i == 0 ? nt[0] : (i == 1 ? nt[1] : nt[2]));
        cv = (!cvg || lvl > 0) ? cn : old_word;                                          // (a select, not a branch round twelve lane reads)
    }
    return role_b ? cv : nv;
}

// st: the state in LDS (updated in place); out: where the multi-launch forms keep it between launches
__device__ __forceinline__ void icp_solve_step(IcpState *st, IcpState *out, const float *sums, const float *pose_state, int lvl,
                                               int iter, int max_iter, float thr_alpha, float thr_tr, float *trace, float *sh,
                                               bool with_end, IcpEndArgs end, IcpLoopCtl lc, long long *timing) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = lane - (lane >> 3);                          // word of the state this lane owns (inverse of icp_pub_slot)
    const bool owner = lane < 32 && (lane & 7) != 7 && k < 24;
    const int kr = owner ? k : 0;
    float x[6], word = 0.0f, new_word = 0.0f;
    int llt_fail = 0, iters = 0, lvl_iters = 0, cvg = 0, new_cvg = 0;
    // The determinant (wave 1) is only consulted for the singularity test, and it takes longer than the solve: wave 0 does
    // not wait for it but goes on to the pose update, in registers; the two meet at the barrier and the test then picks
    // the new or the old words.
    if (wave == 1) {
        float det = wave_det6(sums, lane);                     // RG.cpp:862, concurrently with the solve
        if (lane == 0) sh[0] = det;
    } else if (wave == 0) {
        word = ((const float *)st)[kr];
        iters = st->iters; lvl_iters = st->lvl_iters; cvg = st->cvg;
        llt_fail = wave_llt_solve6(sums, lane, x);             // RG.cpp:873 (wave-uniform result)
        // unconditional (a failed factorisation's update is computed on garbage and dropped below): no branch between the
        // solve and the update, so the quaternion's f64 chain -- it only needs x[3..5], the FIRST results of the backward
        // substitution -- is scheduled beside the rest of the substitution
        new_word = icp_pose_update_wave(st, x, pose_state, lvl, iter, max_iter, thr_alpha, thr_tr, kr, word, new_cvg);
    }
    __syncthreads();
    if (wave != 0) return;
    if (timing && lane == 0) timing[7] = wall_clock64();      // debug build: determinant, solve and update are done
    int fail = 0;
    const float det = sh[0];
    if ((double)__builtin_fabsf(det) < 1e-15 || det != det || llt_fail) {   // RG.cpp:864-871
        fail = 1;
    } else {
        if (trace && lane == 0) {
            float *tr = trace + 33 * iters;
            for (int z = 0; z < 27; z++) tr[z] = sums[z];
            for (int z = 0; z < 6; z++) tr[27 + z] = x[z];
        }
        word = new_word; cvg = new_cvg;
        iters = iters + 1;
        lvl_iters = lvl_iters + (1 << (8 * lvl));
    }
    if (owner) ((float *)st)[k] = word;
    if (lane == 0) { st->iters = iters; st->lvl_iters = lvl_iters; st->cvg = cvg; st->fail = fail; }
    if (lc.pub) {
        // persistent loop: a failed solve ends the loop, a converged level continues at the next level
        const int nxt = fail ? lc.nsteps : (cvg ? lc.next_level : lc.step + 1);
        // ONE wave-wide store publishes the line (a single 128-byte write instead of 25 word stores to a line that 512
        // workgroups are polling: those were acknowledged slowly and the ordering fence waited for them)
        unsigned int w = owner ? __float_as_uint(word) : 0u;
        if (lane < 32 && k == 24) w = (unsigned int)(lvl_iters & 0xffffff) | ((unsigned int)(cvg != 0) << 24) | ((unsigned int)(fail != 0) << 25);
        if ((lane & 7) == 7) w = lc.base + (unsigned int)nxt;
        if (lc.check) {                                        // slot 30: integrity word of the 25 data words
            const unsigned int chk = icp_line_xor_lo(w, lane);
            if (lane == 30) w = chk;
        }
        if (lane < 32) __hip_atomic_store(lc.pub + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nxt >= lc.nsteps) {
            wave_lds_sync();
            if (lane == 0) {
                IcpState loc;
                icp_state_copy(&loc, st);
                icp_end_step(&loc, end);
            }
        }
        return;
    }
    wave_lds_sync();
    if (lane == 0) {
        IcpState loc;
        icp_state_copy(&loc, st);
        icp_state_copy(out, &loc);
        if (with_end) icp_end_step(&loc, end);
    }
}

// stand-alone solve for the row-sharded path (launch with 128 threads): sums = sum over `parts`
// gathered 27-float partial systems, in rank order
__global__ __launch_bounds__(128)
void k_icp_solve(IcpState *st, const float *__restrict__ gathered, int parts, const float *pose_state, int lvl,
                 int iter, int max_iter, float thr_alpha, float thr_tr, float *trace) {
    __shared__ float sums[27];
    __shared__ float sh[2];
    __shared__ IcpState sst;
    __shared__ float spose[24];
    if (threadIdx.x < sizeof(IcpState) / 4) ((int *)&sst)[threadIdx.x] = ((const int *)st)[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 24) spose[threadIdx.x - 64] = pose_state[threadIdx.x - 64];
    if (threadIdx.x < 27) {
        float s = gathered[threadIdx.x];
        for (int p = 1; p < parts; p++) s = s + gathered[27 * p + threadIdx.x];
        sums[threadIdx.x] = s;
    }
    __syncthreads();
    if (sst.fail || (iter > 0 && sst.cvg)) return;
    icp_solve_step(&sst, st, sums, spose, lvl, iter, max_iter, thr_alpha, thr_tr, trace, sh, false, IcpEndArgs{nullptr, nullptr, nullptr, nullptr, 0},
                   IcpLoopCtl{nullptr, 0u, 0, 0, 0, 0});
}

__device__ __forceinline__ void d_pack_pose(const float *R, const float *t, float *P) {   // GP-demo.cpp:206-211
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) P[j * 4 + i] = M3(R, i, j);
    for (int i = 0; i < 3; i++) P[12 + i] = t[i];
    P[3] = 0.0f; P[7] = 0.0f; P[11] = 0.0f; P[15] = 1.0f;
}

// pb (optional): pose_state/status are the block's own fields; the display() pose algebra is
// appended (GP-demo.cpp:201-211,225-236) and the previous pose advanced (244-245 / 192-193).
__device__ __forceinline__ void icp_end_step(const IcpState *st, const IcpEndArgs &e) {
    float *pose_state = e.pose_state;
    int32_t *status = e.status;
    rgbdslam_pose_block *pb = e.pb;
    float *history = e.history;
    const int frame_index = e.frame_index;
    status[1] = st->iters;
    status[2] = st->lvl_iters;
    int ok = !st->fail;
    if (ok && ((double)d_frob_minus_I(st->Rcurr) > 0.3 || (double)d_norm3(st->tcurr) > 0.3)) ok = 0;   // RG.cpp:910
    if (ok) {
        float R[9], neg[9], t[3];
        d_inverse3(st->Rinv, R);                               // RG.cpp:923
        for (int k = 0; k < 9; k++) neg[k] = -R[k];
        d_mat3_vec(neg, st->tinv, t);                          // RG.cpp:924
        for (int k = 0; k < 9; k++) pose_state[k] = R[k];
        for (int k = 0; k < 3; k++) pose_state[9 + k] = t[k];
    }
    status[0] = ok;
    if (pb) {
        float R[9], t[3], Rp[9], tp[3];
        for (int k = 0; k < 9; k++) { R[k] = pose_state[k]; Rp[k] = pose_state[12 + k]; }
        for (int k = 0; k < 3; k++) { t[k] = pose_state[9 + k]; tp[k] = pose_state[21 + k]; }
        if (ok) {
            float Rinv[9], neg[9], tinv[3], Rrel[9], trel[3], tmp[3];
            d_inverse3(R, Rinv);                               // GP-demo.cpp:201-205
            for (int k = 0; k < 9; k++) neg[k] = -Rinv[k];
            d_mat3_vec(neg, t, tinv);
            d_mat3_mul(Rinv, Rp, Rrel);
            d_mat3_vec(Rinv, tp, tmp);
            trel[0] = tmp[0] + tinv[0]; trel[1] = tmp[1] + tinv[1]; trel[2] = tmp[2] + tinv[2];
            d_pack_pose(Rrel, trel, pb->P_rel);
            float RrelInv[9], trelInv[3];
            d_inverse3(Rrel, RrelInv);                         // GP-demo.cpp:225-236
            for (int k = 0; k < 9; k++) neg[k] = -RrelInv[k];
            d_mat3_vec(neg, trel, trelInv);
            d_pack_pose(RrelInv, trelInv, pb->P_rel_inv);
            d_pack_pose(R, t, pb->P_glob);
            status[3] = status[3] + 1;
        }
        for (int k = 0; k < 9; k++) pose_state[12 + k] = R[k]; // _poseRPrev = _poseR (both branches)
        for (int k = 0; k < 3; k++) pose_state[21 + k] = t[k];
        if (history) {
            float *h = history + 16 * (size_t)frame_index;
            for (int k = 0; k < 9; k++) h[k] = R[k];
            for (int k = 0; k < 3; k++) h[9 + k] = t[k];
            h[12] = (float)ok; h[13] = (float)st->iters; h[14] = (float)st->lvl_iters; h[15] = 0.0f;
        }
    }
}
__global__ void k_icp_end(IcpState *st, float *pose_state, int32_t *status, rgbdslam_pose_block *pb, float *history,
                          int frame_index) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    IcpEndArgs e = {pose_state, status, pb, history, frame_index};
    icp_end_step(st, e);
}

// =================================================================================================
// C ABI
// =================================================================================================
static int div_up(int x, int y) { return (x % y) ? ((x + y - 1) / y) : (x / y); }

extern "C" int rgbdslam_alloc_buffers(rgbdslam_ctx *ctx, int grid_rows, int grid_cols) {
    // The reference sizes buf_dev by the 8x8-block grid (27*n*m floats, ICP.cu:10); here the first
    // reduction stage never leaves the chip, so the scratch is 27 x 512 chain sums whatever the image size.
    if (!ctx || grid_rows <= 0 || grid_cols <= 0) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->icp_block) {                                     // one allocation: k_icp_loop addresses it by offset
        HIP_TRY(hipMalloc((void **)&ctx->icp_block, ICP_BLOCK_BYTES));
        HIP_TRY(hipMemset(ctx->icp_block, 0, ICP_BLOCK_BYTES));
        ctx->buf_dev = (float *)(ctx->icp_block + ICP_OFF_CHAINS);
        ctx->outbuf_dev = (float *)(ctx->icp_block + ICP_OFF_OUT);
        ctx->icp_state = (IcpState *)(ctx->icp_block + ICP_OFF_STATE);
        ctx->icp_epoch_dev = (unsigned int *)(ctx->icp_block + ICP_OFF_EPOCH);
        ctx->icp_epoch_base = 0;
    }
    if (!ctx->out_host) HIP_TRY(hipHostMalloc((void **)&ctx->out_host, 27 * sizeof(float), hipHostMallocDefault));
    const char *env = getenv("RGBDSLAM_ICP_LOOP");            // opt-in without touching the caller's code
    if (env && atoi(env) > 0) (void)rgbdslam_set_icp_loop(ctx, 1);
    env = getenv("RGBDSLAM_ICP_LINE_CHECK");
    if (env) ctx->icp_line_check = atoi(env) > 0;
    return 0;
}

extern "C" int rgbdslam_free_buffers(rgbdslam_ctx *ctx) {
    if (!ctx) return RGBDSLAM_E_ARG;
    if (ctx->icp_block) { (void)hipFree(ctx->icp_block); ctx->icp_block = nullptr; }
    ctx->buf_dev = nullptr; ctx->outbuf_dev = nullptr; ctx->icp_state = nullptr; ctx->icp_epoch_dev = nullptr;
    if (ctx->out_host) { (void)hipHostFree(ctx->out_host); ctx->out_host = nullptr; }
    return 0;
}

// Persistent-loop mode of rgbdslam_track_frame (k_icp_loop).  It needs every one of its 512 workgroups on
// the device at the same time, which holds when this process has the GPU to itself and the frame loop
// runs on one stream (the one-process-per-GPU deployment): off by default, the caller opts in.
extern "C" int rgbdslam_set_icp_loop(rgbdslam_ctx *ctx, int on) {
    if (!ctx) return RGBDSLAM_E_ARG;
    if (!on) { ctx->icp_loop = 0; return 0; }
    HIP_TRY(hipSetDevice(ctx->device));
    int per_cu = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_icp_loop<true>, 64 * ICP_MAX_WAVES, 0));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    if (cus <= 0 || (long)per_cu * cus < ICP_WGS) return RGBDSLAM_E_STATE;
    // The occupancy query packs waves freely over the four SIMDs of a CU; the dispatcher deals a workgroup's waves
    // to the SIMDs starting from the first one (measured with 5-wave workgroups: at 144 registers = 3 waves per SIMD
    // only ONE was resident per CU and the loop starved), so SIMD 0 must have room for every workgroup's share.
    hipFuncAttributes fa, fx;
    HIP_TRY(hipFuncGetAttributes(&fa, (const void *)k_icp_loop<true>));
    HIP_TRY(hipFuncGetAttributes(&fx, (const void *)k_icp_loop<false, true>));      // the row-sharded instance
    const int nregs = fa.numRegs > fx.numRegs ? fa.numRegs : fx.numRegs;
    const int regs = ((nregs > 0 ? nregs : 512) + 7) & ~7;
    const int waves_per_simd = 512 / regs < 8 ? 512 / regs : 8;
    const int wgs_per_cu = (ICP_WGS + cus - 1) / cus;
    if (wgs_per_cu * ((ICP_MAX_WAVES + 3) / 4) > waves_per_simd) return RGBDSLAM_E_STATE;
    ctx->icp_loop = 1;
    return 0;
}

// faults[0] = number of k_icp_loop launches that gave up waiting since the last call (synchronises the
// stream).  After a fault the mode is switched off.
extern "C" int rgbdslam_icp_loop_faults(rgbdslam_ctx *ctx, int *faults) {
    if (!ctx || !faults) return RGBDSLAM_E_ARG;
    *faults = 0;
    if (!ctx->icp_epoch_dev) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    int f = 0;
    HIP_TRY(hipMemcpyAsync(&f, ctx->icp_epoch_dev + 32, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (f) {
        HIP_TRY(hipMemsetAsync(ctx->icp_epoch_dev + 32, 0, sizeof(int), ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->icp_loop = 0;
    }
    *faults = f;
    return 0;
}

// Integrity check of the published lines (chain sums, loop state): see icp_lines_intact.  With the check on, correctness no
// longer rests on gfx950 writing and reading a wave instruction's 32-byte sector indivisibly; a rejected line is read again.
extern "C" int rgbdslam_set_icp_line_check(rgbdslam_ctx *ctx, int on) {
    if (!ctx) return RGBDSLAM_E_ARG;
    ctx->icp_line_check = on != 0;
    return 0;
}
// rejections[0] = lines that carried this step's sequence words but failed the integrity check, since the last call
// (synchronises the stream).  Anything but 0 means the sector-atomicity assumption does not hold on this device.
extern "C" int rgbdslam_icp_line_rejections(rgbdslam_ctx *ctx, unsigned int *rejections) {
    if (!ctx || !rejections) return RGBDSLAM_E_ARG;
    *rejections = 0;
    if (!ctx->icp_epoch_dev) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned int r = 0;
    HIP_TRY(hipMemcpyAsync(&r, ctx->icp_epoch_dev + ICP_OFF_REJECT_WORD, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (r) {
        HIP_TRY(hipMemsetAsync(ctx->icp_epoch_dev + ICP_OFF_REJECT_WORD, 0, sizeof(r), ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *rejections = r;
    return 0;
}
// Debug: give-up time of every device-side wait of the ICP kernels in 10 ns ticks (default 5 000 000 = 50 ms); 0 makes every
// waiter leave at once, which drives the give-up paths deterministically (tests).
extern "C" int rgbdslam_debug_set_exchange_delay(rgbdslam_ctx *ctx, int ticks) {
    if (!ctx || ticks < 0 || ticks > 1000000) return RGBDSLAM_E_ARG;         // at most 10 ms of the 100 MHz clock
    ctx->xchg_delay_ticks = ticks;
    return 0;
}

extern "C" int rgbdslam_debug_set_icp_timeout(rgbdslam_ctx *ctx, int ticks) {
    if (!ctx || ticks < 0) return RGBDSLAM_E_ARG;
    ctx->icp_timeout_ticks = ticks;
    return 0;
}

struct IcpMaps {
    const float *VMap, *NMap, *RGB, *Gx, *Gy;
    size_t gxp, gyp;
    const float *VMap_prev, *NMap_prev, *RGB_prev;
    const float *Aux;                                // the current frame's auxiliary records or NULL (rgbdslam_set_frame_aux)
};

// One launch of k_icp_iter over block rows [br_lo, br_hi) of the level's 8x8-block grid.
// Largest float T with sqrtf(T) <= t, so that "sqrtf(x) > t" == "x > T" for every x >= 0, NaN and infinity included
// (icp_rows).  t < 0: every x >= 0 exceeds it (T = -1); t NaN: nothing does (T = +inf, x > inf is never true).
static float icp_sq_threshold(float t) {
    if (t != t) return INFINITY;
    if (t < 0.0f) return -1.0f;
    if (t == INFINITY) return INFINITY;
    float T = t * t;
    if (T == INFINITY) T = 3.402823466e+38f;
    while (T > 0.0f && sqrtf(T) > t) T = nextafterf(T, 0.0f);
    for (;;) {
        const float up = nextafterf(T, INFINITY);
        if (up == INFINITY || !(sqrtf(up) <= t)) break;
        T = up;
    }
    return T;
}
// Smallest float L with sqrtf(L) >= t, so that "sqrtf(x) < t" == "x < L" for every x >= 0 (and NaN: false on both sides).
// t <= 0 or NaN: never true (L = 0: no sum of squares is below it); t = +inf: true for every finite x (L = +inf).
static float icp_sq_threshold_lt(float t) {
    if (t != t || t <= 0.0f) return 0.0f;
    if (t == INFINITY) return INFINITY;
    float L = t * t;
    if (L == INFINITY) L = 3.402823466e+38f;
    while (sqrtf(L) < t) {                                     // up to the first float whose root reaches t
        if (L == 3.402823466e+38f) return INFINITY;
        L = nextafterf(L, INFINITY);
    }
    for (;;) {                                                 // and down while the one below still does
        const float dn = nextafterf(L, 0.0f);
        if (L == 0.0f || sqrtf(dn) < t) break;
        L = dn;
    }
    return L;
}
// the gathers address the maps with 32-bit byte offsets (icp_gather): 16 bytes per pixel and the pitched gradient images
// must stay below 4 GB -- 268 M pixels, far beyond any depth camera
static bool icp_offsets_fit(int n, int m, size_t gxp, size_t gyp) {
    return (unsigned long long)n * (unsigned long long)m * 16ull <= 0xffffffffull && (unsigned long long)n * gxp <= 0xffffffffull &&
           (unsigned long long)n * gyp <= 0xffffffffull;
}
static int launch_iter(rgbdslam_ctx *ctx, const float *intr, const float *R, const float *t, const IcpMaps &mp,
                       float distThres, float angleThres, int n, int m, int fact, int br_lo, int br_hi, float *sums,
                       int32_t *corr, IcpState *state, const float *pose_state, int do_solve, int lvl, int iter, int max_iter,
                       float thr_alpha, float thr_tr, float *trace, int first_launch = 0, int final_launch = 0,
                       const IcpEndArgs *end = nullptr, const IcpXchg *xchg = nullptr) {
    int gr = div_up(n / fact, 8), gc = div_up(m / fact, 8);   // D1/ICP.cu:260-261
    if (gr <= 0 || gc <= 0) return RGBDSLAM_E_ARG;
    if (!icp_offsets_fit(n, m, mp.gxp, mp.gyp)) return RGBDSLAM_E_ARG;
    if (br_lo < 0) br_lo = 0;
    if (br_hi > gr) br_hi = gr;
    if (br_hi <= br_lo) return RGBDSLAM_E_ARG;
    if (!ctx->buf_dev) return RGBDSLAM_E_STATE;
    IcpIterArgs a;
    a.nx_lvl = -1; a.nx_rows = 1; a.nx_cols = 1; a.nx_br_lo = 0;
    if (xchg) a.xchg = *xchg; else a.xchg.world = 0;
    a.fault = ctx->icp_epoch_dev ? (int *)(ctx->icp_epoch_dev + 32) : nullptr;
    a.line_check = ctx->icp_line_check; a.rejections = ctx->icp_epoch_dev ? ctx->icp_epoch_dev + ICP_OFF_REJECT_WORD : nullptr;
    a.timeout_ticks = ctx->icp_timeout_ticks;
    a.intr = intr; a.R = R; a.t = t;
    a.VMap = mp.VMap; a.NMap = mp.NMap; a.RGB = mp.RGB; a.Gx = mp.Gx; a.Gy = mp.Gy; a.gxp = mp.gxp; a.gyp = mp.gyp;
    a.Aux = (size_t)n * m * 16 <= 0xffffffffull ? mp.Aux : nullptr;
    a.VMap_prev = mp.VMap_prev; a.NMap_prev = mp.NMap_prev; a.RGBD_prev = mp.RGB_prev;
    a.distThres = icp_sq_threshold(distThres); a.angleThres = icp_sq_threshold(angleThres);
    a.n = n; a.m = m; a.fact = fact; a.grid_rows = gr; a.grid_cols = gc;
    a.br_lo = br_lo; a.rows_local = br_hi - br_lo;
    long nblk = (long)a.rows_local * gc;
    a.nchain = (int)(nblk < ICP_CHAINS ? nblk : ICP_CHAINS);
    a.nwaves = ICP_MAX_WAVES;   // always full workgroups: the last one needs the waves for the 27-term tail
    a.chains = ctx->buf_dev; a.sums = sums; a.corr = corr;
    a.seq = ++ctx->icp_epoch_base;                        // one counter for both launch forms: a stale line never matches
    a.state = state; a.pose_state = pose_state; a.do_solve = do_solve; a.lvl = lvl; a.iter = iter; a.max_iter = max_iter;
    a.thr_alpha = icp_sq_threshold_lt(thr_alpha); a.thr_tr = icp_sq_threshold_lt(thr_tr); a.trace = trace;
    a.timing = ctx->icp_timing_dev;
    a.first_launch = first_launch; a.final_launch = final_launch;
    for (int k = 0; k < 4; k++) a.intr4[k] = ctx->calib.v[k];
    a.pose_state_rw = end ? end->pose_state : nullptr; a.status = end ? end->status : nullptr;
    a.pb = end ? end->pb : nullptr; a.history = end ? end->history : nullptr; a.frame_index = end ? end->frame_index : 0;
    KLAUNCH(ctx, k_icp_iter, dim3(a.nchain < ICP_WGS ? a.nchain : ICP_WGS), dim3(64 * a.nwaves), a);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_estimate_system_gauss_newton(rgbdslam_ctx *ctx, const float *Rcurr_dev, const float *tcurr_dev,
                                                     const float *VMap, const float *NMap, const float *RGB,
                                                     const float *Gx, size_t gxp, const float *Gy, size_t gyp,
                                                     const float *VMap_prev, const float *NMap_prev,
                                                     const float *RGB_prev, const float *intr_dev, float distThres,
                                                     float angleThres, int n, int m, float *A, float *b, int fact,
                                                     int32_t *corr_dev) {
    if (!ctx || !Rcurr_dev || !tcurr_dev || !VMap || !NMap || !RGB || !Gx || !Gy || !VMap_prev || !NMap_prev ||
        !RGB_prev || !intr_dev || !A || !b)
        return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || fact <= 0 || gxp < 4 * (size_t)m || gyp < 4 * (size_t)m) return RGBDSLAM_E_ARG;
    if (corr_dev) HIP_TRY(hipMemsetAsync(corr_dev, 0xFF, sizeof(int32_t) * (size_t)n * m, ctx->stream));
    IcpMaps mp = {VMap, NMap, RGB, Gx, Gy, gxp, gyp, VMap_prev, NMap_prev, RGB_prev};
    int rc = launch_iter(ctx, intr_dev, Rcurr_dev, tcurr_dev, mp, distThres, angleThres, n, m, fact, 0, 1 << 30,
                         ctx->outbuf_dev, corr_dev, nullptr, nullptr, 0, 0, 0, 0, 0.0f, 0.0f, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->out_host, ctx->outbuf_dev, 27 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int sh = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 7; j++) {                          // D1/ICP.cu:273-283
            float v = ctx->out_host[sh++];
            if (j == 6) b[i] = v;
            else A[j * 6 + i] = A[i * 6 + j] = v;
        }
    return 0;
}

static int xchg_next(rgbdslam_ctx *ctx, IcpXchg *x);
static int check_align_args(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, int n, int m, size_t gxp, size_t gyp) {
    if (n <= 0 || m <= 0 || prm->lvl < 1 || prm->lvl > 3 || gxp < 4 * (size_t)m || gyp < 4 * (size_t)m) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib || !ctx->icp_state) return RGBDSLAM_E_STATE;
    return 0;
}

static int align_impl(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap, const float *NMap,
                      const float *RGB, const float *Gx, size_t gxp, const float *Gy, size_t gyp, const float *VMap_prev,
                      const float *NMap_prev, const float *RGB_prev, int n, int m, float *pose_state_dev,
                      int32_t *status_dev, float *trace_dev, rgbdslam_pose_block *pose_block, float *history,
                      int frame_index, const int *band_lo = nullptr, const int *band_hi = nullptr) {
    if (!ctx || !prm || !VMap || !NMap || !RGB || !Gx || !Gy || !VMap_prev || !NMap_prev || !RGB_prev ||
        !pose_state_dev || !status_dev)
        return RGBDSLAM_E_ARG;
    const float *aux = ctx->track_aux;                         // announced for this call only (rgbdslam_set_frame_aux)
    ctx->track_aux = nullptr;
    int rc = check_align_args(ctx, prm, n, m, gxp, gyp);
    if (rc) return rc;
    IcpState *st = ctx->icp_state;
    IcpMaps mp = {VMap, NMap, RGB, Gx, Gy, gxp, gyp, VMap_prev, NMap_prev, RGB_prev, aux};
    // track_frame: begin and end ride on the first and the last iteration launch
    const bool ride = pose_block != nullptr;
    IcpEndArgs end = {pose_state_dev, status_dev, pose_block, history, frame_index};
    if (!ride) KLAUNCH(ctx, k_icp_begin, dim3(1), dim3(64), st, pose_state_dev, ctx->calib, status_dev);
    int fact = 1 << (prm->lvl - 1);                            // RG.cpp:832
    const bool sharded = band_lo != nullptr;
    if (sharded && !(ride && ctx->icp_loop && ctx->icp_block)) return RGBDSLAM_E_STATE;   // the sharded form IS the persistent loop
    if (ride && ctx->icp_loop && ctx->icp_block && !trace_dev && pose_state_dev == pose_block->pose_state &&
        status_dev == pose_block->status) {
        IcpLoopArgs L;
        int ns = 0, nl = 0;
        for (int lvl = prm->lvl - 1, f = fact; lvl > -1; lvl--, f /= 2, nl++) {
            if (prm->max_iter[lvl] < 0 || prm->max_iter[lvl] > 1000) return RGBDSLAM_E_ARG;
            const int gr = div_up(n / f, 8), gc = div_up(m / f, 8);   // D1/ICP.cu:260-261
            if (gr <= 0 || gc <= 0) return RGBDSLAM_E_ARG;
            if (gr > 0xffff || gc > 0xffff || lvl > 3) return RGBDSLAM_E_ARG;
            L.lv[nl].w0 = (unsigned int)ns | ((unsigned int)prm->max_iter[lvl] << 16) | ((unsigned int)lvl << 28);
            L.lv[nl].w1 = (unsigned int)gr | ((unsigned int)gc << 16);
            L.lv[nl].w2 = (unsigned int)gr << 16;
            if (sharded) {                                     // every rank needs a tail at every level: no empty bands
                const int lo = band_lo[lvl] < 0 ? 0 : band_lo[lvl], hi = band_hi[lvl] > gr ? gr : band_hi[lvl];
                if (hi <= lo) return RGBDSLAM_E_ARG;
                L.lv[nl].w2 = (unsigned int)lo | ((unsigned int)(hi - lo) << 16);
            }
            ns += prm->max_iter[lvl] + 2;                      // iter = 0 .. max_iter+1 (RG.cpp:889)
        }
        if (sharded && ns > ICP_XCHG_STEPS - 1) return RGBDSLAM_E_LIMIT;   // one exchange slot per step of a launch
        for (int k = nl; k < 3; k++) { L.lv[k] = L.lv[0]; L.lv[k].w0 |= ICP_LV_UNUSED; }
        if (!icp_offsets_fit(n, m, gxp, gyp)) return RGBDSLAM_E_ARG;
        L.VMap = VMap; L.NMap = NMap; L.RGB = RGB; L.Gx = Gx; L.Gy = Gy; L.gxp = (unsigned int)gxp; L.gyp = (unsigned int)gyp;
        // the frame's auxiliary records, if the caller announced them for THIS call (rgbdslam_set_frame_aux; 32-bit byte offsets)
        L.Aux = (size_t)n * m * 16 <= 0xffffffffull ? aux : nullptr;
        L.VMap_prev = VMap_prev; L.NMap_prev = NMap_prev; L.RGBD_prev = RGB_prev;
        L.block = ctx->icp_block; L.pb = pose_block; L.history = history; L.frame_index = frame_index;
        L.distThres = icp_sq_threshold(prm->distThres); L.angleThres = icp_sq_threshold(prm->angleThres);
        L.thr_alpha = icp_sq_threshold_lt(prm->thres_cvg_alpha); L.thr_tr = icp_sq_threshold_lt(prm->thres_cvg_tr);
        L.n = n; L.m = m;
        for (int k = 0; k < 4; k++) L.intr4[k] = ctx->calib.v[k];
        L.nsteps = ns;
        L.epoch_base = ctx->icp_epoch_base;                    // this launch owns base+1 .. base+4095 (epochs and sequence
        ctx->icp_epoch_base += 4096u;                          // words): a stale word never releases a waiter or passes a check
        L.timeout_ticks = ctx->icp_timeout_ticks;              // default 50 ms of the 100 MHz wall clock
#ifdef ICP_LOOP_TIMING
        L.timing = ctx->icp_timing_dev;
#endif
        L.xtab = nullptr; L.xtag_base = 0; L.xworld = 0; L.xrank = 0; L.xparity = 0; L.xdelay = 0;
        if (sharded) {
            IcpXchg x;
            rc = xchg_next(ctx, &x);
            if (rc) return rc;
            L.xtab = x.tab; L.xtag_base = x.tag_base; L.xworld = x.world; L.xrank = x.rank; L.xparity = x.parity; L.xdelay = x.delay_ticks;
            void (*const k_icp_loop_sharded)(IcpLoopArgs) = k_icp_loop<false, true>;      // (the profiling hook's name for it)
            KLAUNCH(ctx, k_icp_loop_sharded, dim3(ICP_WGS), dim3(64 * ICP_MAX_WAVES), L);
        }
        else if (ctx->icp_line_check) KLAUNCH(ctx, k_icp_loop<true>, dim3(ICP_WGS), dim3(64 * ICP_MAX_WAVES), L);
        else KLAUNCH(ctx, k_icp_loop<false>, dim3(ICP_WGS), dim3(64 * ICP_MAX_WAVES), L);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // instrumentation: the loop's launches are back to back, so one event pair brackets all of them
    const bool prof_group = prof_group_begin(ctx, "k_icp_iter");
    for (int lvl = prm->lvl - 1; lvl > -1; lvl--) {
        int n_it = prm->max_iter[lvl] + 2;                     // iter = 0 .. max_iter+1 (RG.cpp:889)
        for (int iter = 0; iter < n_it; iter++) {
            int first = ride && lvl == prm->lvl - 1 && iter == 0, last = ride && lvl == 0 && iter == n_it - 1;
            rc = launch_iter(ctx, st->intr, st->Rcurr, st->tcurr, mp, prm->distThres, prm->angleThres, n, m, fact, 0, 1 << 30,
                             ctx->outbuf_dev, nullptr, st, pose_state_dev, 1, lvl, iter, prm->max_iter[lvl],
                             prm->thres_cvg_alpha, prm->thres_cvg_tr, trace_dev, first, last, ride ? &end : nullptr);
            if (rc) { if (prof_group) prof_group_end(ctx); return rc; }
        }
        fact /= 2;
    }
    if (prof_group) prof_group_end(ctx);
    if (!ride) KLAUNCH(ctx, k_icp_end, dim3(1), dim3(64), st, pose_state_dev, status_dev, pose_block, history, frame_index);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_align_gauss_newton(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap,
                                           const float *NMap, const float *RGB, const float *Gx, size_t gxp,
                                           const float *Gy, size_t gyp, const float *VMap_prev, const float *NMap_prev,
                                           const float *RGB_prev, int n, int m, float *pose_state_dev,
                                           int32_t *status_dev, float *trace_dev) {
    return align_impl(ctx, prm, VMap, NMap, RGB, Gx, gxp, Gy, gyp, VMap_prev, NMap_prev, RGB_prev, n, m, pose_state_dev, status_dev,
                      trace_dev, nullptr, nullptr, 0);
}

extern "C" int rgbdslam_track_frame(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap, const float *NMap,
                                    const float *RGB, const float *Gx, size_t gxp, const float *Gy, size_t gyp,
                                    const float *VMap_prev, const float *NMap_prev, const float *RGB_prev, int n, int m,
                                    rgbdslam_pose_block *pb, float *history_dev, int frame_index) {
    if (!pb || frame_index < 0) return RGBDSLAM_E_ARG;
    return align_impl(ctx, prm, VMap, NMap, RGB, Gx, gxp, Gy, gyp, VMap_prev, NMap_prev, RGB_prev, n, m, pb->pose_state, pb->status,
                      nullptr, pb, history_dev, frame_index);
}

// Row-sharded form of rgbdslam_track_frame, the whole coarse-to-fine loop in ONE persistent launch per rank with the 27-float
// exchange inside it (RG.cpp:837-904 is one dependent chain per frame, and the multi-launch sharded form paid
// the twelve launch boundaries the single-GPU path had got rid of).  band_lo / band_hi [lvl]: this rank's block rows of
// level lvl's grid.  Needs rgbdslam_set_icp_loop (the GPU to itself) and a connected exchange; every rank makes the same calls.
extern "C" int rgbdslam_track_frame_sharded(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap, const float *NMap,
                                            const float *RGB, const float *Gx, size_t gxp, const float *Gy, size_t gyp,
                                            const float *VMap_prev, const float *NMap_prev, const float *RGB_prev, int n, int m,
                                            rgbdslam_pose_block *pb, float *history_dev, int frame_index, const int *band_lo3,
                                            const int *band_hi3) {
    if (!pb || frame_index < 0 || !band_lo3 || !band_hi3) return RGBDSLAM_E_ARG;
    return align_impl(ctx, prm, VMap, NMap, RGB, Gx, gxp, Gy, gyp, VMap_prev, NMap_prev, RGB_prev, n, m, pb->pose_state, pb->status,
                      nullptr, pb, history_dev, frame_index, band_lo3, band_hi3);
}

// ---- row-sharded Gauss-Newton loop (BASELINE configs[3]): the caller interleaves a
// collective between band_system and solve_gathered; everything stays stream-ordered. ------------
extern "C" int rgbdslam_align_begin(rgbdslam_ctx *ctx, const float *pose_state_dev, int32_t *status_dev) {
    if (!ctx || !pose_state_dev || !status_dev) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib || !ctx->icp_state) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_icp_begin, dim3(1), dim3(64), ctx->icp_state, pose_state_dev, ctx->calib, status_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_align_band_system(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap,
                                          const float *NMap, const float *RGB, const float *Gx, size_t gxp, const float *Gy,
                                          size_t gyp, const float *VMap_prev, const float *NMap_prev, const float *RGB_prev,
                                          int n, int m, int lvl, int iter, int block_row_lo, int block_row_hi,
                                          float *sums27_dev) {
    if (!ctx || !prm || !VMap || !NMap || !RGB || !Gx || !Gy || !VMap_prev || !NMap_prev || !RGB_prev || !sums27_dev)
        return RGBDSLAM_E_ARG;
    int rc = check_align_args(ctx, prm, n, m, gxp, gyp);
    if (rc) return rc;
    if (lvl < 0 || lvl >= prm->lvl || iter < 0) return RGBDSLAM_E_ARG;
    IcpState *st = ctx->icp_state;
    IcpMaps mp = {VMap, NMap, RGB, Gx, Gy, gxp, gyp, VMap_prev, NMap_prev, RGB_prev};
    return launch_iter(ctx, st->intr, st->Rcurr, st->tcurr, mp, prm->distThres, prm->angleThres, n, m, 1 << lvl, block_row_lo,
                       block_row_hi, sums27_dev, nullptr, st, nullptr, 0, lvl, iter, prm->max_iter[lvl], 0.0f, 0.0f, nullptr);
}

extern "C" int rgbdslam_align_solve_gathered(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *gathered_dev,
                                             int parts, const float *pose_state_dev, int lvl, int iter, float *trace_dev) {
    if (!ctx || !prm || !gathered_dev || parts <= 0 || !pose_state_dev || lvl < 0 || lvl >= prm->lvl || iter < 0)
        return RGBDSLAM_E_ARG;
    if (!ctx->icp_state) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_icp_solve, dim3(1), dim3(128), ctx->icp_state, gathered_dev, parts, pose_state_dev, lvl, iter,
            prm->max_iter[lvl], icp_sq_threshold_lt(prm->thres_cvg_alpha), icp_sq_threshold_lt(prm->thres_cvg_tr), trace_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rgbdslam_align_end(rgbdslam_ctx *ctx, float *pose_state_dev, int32_t *status_dev) {
    if (!ctx || !pose_state_dev || !status_dev) return RGBDSLAM_E_ARG;
    if (!ctx->icp_state) return RGBDSLAM_E_STATE;
    KLAUNCH(ctx, k_icp_end, dim3(1), dim3(64), ctx->icp_state, pose_state_dev, status_dev, (rgbdslam_pose_block *)nullptr,
            (float *)nullptr, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- peer-mapped exchange of the 27 band sums (row-sharded ICP over several GPUs; no reference counterpart) ----------
#define XCHG_BLOCK_BYTES (2 * ICP_XCHG_STEPS * ICP_XCHG_MAX_WORLD * 128)   // 32 slots per rank: 16 steps x 2 launch parities = 32 KB
static_assert(sizeof(hipIpcMemHandle_t) <= RGBDSLAM_EXCHANGE_HANDLE_BYTES, "IPC handle size");

extern "C" int rgbdslam_exchange_create(rgbdslam_ctx *ctx, int world, int rank, void *handle_out) {
    if (!ctx || world < 1 || world > ICP_XCHG_MAX_WORLD || rank < 0 || rank >= world) return RGBDSLAM_E_ARG;
    if (ctx->xchg_world) return RGBDSLAM_E_STATE;
    HIP_TRY(hipSetDevice(ctx->device));
    // fine-grained device memory: stores arriving over xGMI and the local polls meet at the memory, not in a stale L2 line
    void *blk = nullptr;
    if (hipExtMallocWithFlags(&blk, XCHG_BLOCK_BYTES, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipMalloc(&blk, XCHG_BLOCK_BYTES));
    }
    HIP_TRY(hipMemset(blk, 0, XCHG_BLOCK_BYTES));
    if (handle_out) {
        memset(handle_out, 0, RGBDSLAM_EXCHANGE_HANDLE_BYTES);
        if (world > 1) {
            hipIpcMemHandle_t h;
            hipError_t e = hipIpcGetMemHandle(&h, blk);
            if (e != hipSuccess) { (void)hipFree(blk); return -(int)e; }
            memcpy(handle_out, &h, sizeof(h));
        }
    }
    ctx->xchg_peer[rank] = (unsigned int *)blk;
    ctx->xchg_world = world; ctx->xchg_rank = rank; ctx->xchg_seq = 0;
    return 0;
}

// the exchange descriptor of the next launch: every launch (or multi-launch iteration) owns 16 tags and one of the two slot sets
static int xchg_next(rgbdslam_ctx *ctx, IcpXchg *x) {
    if (!ctx->xchg_world) return RGBDSLAM_E_STATE;
    if (!ctx->xchg_tab_dev) return RGBDSLAM_E_STATE;          // rgbdslam_exchange_connect has not run
    const unsigned int launch = ++ctx->xchg_seq;
    x->tab = (unsigned int *const *)ctx->xchg_tab_dev; x->world = ctx->xchg_world; x->rank = ctx->xchg_rank;
    x->tag_base = launch * (unsigned int)ICP_XCHG_STEPS; x->parity = (int)(launch & 1u); x->delay_ticks = ctx->xchg_delay_ticks;
    return 0;
}

// all_handles: world x RGBDSLAM_EXCHANGE_HANDLE_BYTES, entry r = what rank r's rgbdslam_exchange_create returned
extern "C" int rgbdslam_exchange_connect(rgbdslam_ctx *ctx, const void *all_handles) {
    if (!ctx) return RGBDSLAM_E_ARG;
    if (!ctx->xchg_world) return RGBDSLAM_E_STATE;
    if (ctx->xchg_world > 1 && !all_handles) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    for (int r = 0; r < ctx->xchg_world; r++) {
        if (r == ctx->xchg_rank || ctx->xchg_peer[r]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)all_handles + (size_t)r * RGBDSLAM_EXCHANGE_HANDLE_BYTES, sizeof(h));
        void *p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        ctx->xchg_peer[r] = (unsigned int *)p;
    }
    // the kernels read the block addresses from a device table (eight pointers as kernel arguments would cost the
    // persistent loop sixteen scalar registers it does not have)
    if (!ctx->xchg_tab_dev) HIP_TRY(hipMalloc((void **)&ctx->xchg_tab_dev, sizeof(void *) * ICP_XCHG_MAX_WORLD));
    HIP_TRY(hipMemcpy(ctx->xchg_tab_dev, ctx->xchg_peer, sizeof(void *) * ICP_XCHG_MAX_WORLD, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int rgbdslam_exchange_destroy(rgbdslam_ctx *ctx) {
    if (!ctx) return RGBDSLAM_E_ARG;
    if (!ctx->xchg_world) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int r = 0; r < ctx->xchg_world; r++) {
        if (!ctx->xchg_peer[r]) continue;
        if (r == ctx->xchg_rank) (void)hipFree(ctx->xchg_peer[r]);
        else (void)hipIpcCloseMemHandle(ctx->xchg_peer[r]);
        ctx->xchg_peer[r] = nullptr;
    }
    if (ctx->xchg_tab_dev) { (void)hipFree(ctx->xchg_tab_dev); ctx->xchg_tab_dev = nullptr; }
    ctx->xchg_world = 0;
    return 0;
}

// One Gauss-Newton iteration of the row-sharded loop in ONE launch per rank: band sums -> line into every peer's slot ->
// wait for all ranks' lines -> rank-ordered total -> the identical solve.  Replaces align_band_system + all-gather +
// align_solve_gathered (same arithmetic, same summation order: bit-identical poses).  Every rank must make the same
// sequence of calls (the sequence word advances per call), and every rank's band must be non-empty.
extern "C" int rgbdslam_align_band_exchange(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap, const float *NMap,
                                            const float *RGB, const float *Gx, size_t gxp, const float *Gy, size_t gyp,
                                            const float *VMap_prev, const float *NMap_prev, const float *RGB_prev, int n, int m,
                                            int lvl, int iter, int block_row_lo, int block_row_hi, const float *pose_state_dev,
                                            float *trace_dev) {
    if (!ctx || !prm || !VMap || !NMap || !RGB || !Gx || !Gy || !VMap_prev || !NMap_prev || !RGB_prev || !pose_state_dev)
        return RGBDSLAM_E_ARG;
    int rc = check_align_args(ctx, prm, n, m, gxp, gyp);
    if (rc) return rc;
    if (lvl < 0 || lvl >= prm->lvl || iter < 0 || block_row_hi <= block_row_lo) return RGBDSLAM_E_ARG;
    IcpXchg x;
    rc = xchg_next(ctx, &x);
    if (rc) return rc;
    IcpState *st = ctx->icp_state;
    IcpMaps mp = {VMap, NMap, RGB, Gx, Gy, gxp, gyp, VMap_prev, NMap_prev, RGB_prev};
    return launch_iter(ctx, st->intr, st->Rcurr, st->tcurr, mp, prm->distThres, prm->angleThres, n, m, 1 << lvl, block_row_lo,
                       block_row_hi, ctx->outbuf_dev, nullptr, st, pose_state_dev, 1, lvl, iter, prm->max_iter[lvl],
                       prm->thres_cvg_alpha, prm->thres_cvg_tr, trace_dev, 0, 0, nullptr, &x);
}
