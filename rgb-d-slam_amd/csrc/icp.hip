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

#define ICP_WAVES 4   // waves (= 8x8 blocks) per workgroup

struct IcpRow { float g[7]; float c[7]; };

// D1/ICP.cu:72-146 searchGauss + 164-205 row construction.  Returns false -> zero rows.
__device__ __forceinline__ bool icp_rows(size_t indx, const float *__restrict__ intr, const float *__restrict__ R,
                                         const float *__restrict__ t, const float *__restrict__ VMap,
                                         const float *__restrict__ NMap, const float *__restrict__ RGB,
                                         const float *__restrict__ Gx, size_t gxp, const float *__restrict__ Gy, size_t gyp,
                                         const float *__restrict__ VMap_prev, const float *__restrict__ NMap_prev,
                                         const float *__restrict__ RGBD_prev, float distThres, float angleThres,
                                         int n, int m, IcpRow &o, int &proj) {
    F3 np = ld3(NMap_prev, indx);
    if (is_zero3(np)) return false;
    F3 vp = ld3(VMap_prev, indx);
    float4 cp = *(const float4 *)(RGBD_prev + 4 * indx);

    float d0 = ((R[0] * vp.x + R[3] * vp.y) + R[6] * vp.z) + t[0];
    float d1 = ((R[1] * vp.x + R[4] * vp.y) + R[7] * vp.z) + t[1];
    float d2 = ((R[2] * vp.x + R[5] * vp.y) + R[8] * vp.z) + t[2];
    float n0 = (R[0] * np.x + R[3] * np.y) + R[6] * np.z;
    float n1 = (R[1] * np.x + R[4] * np.y) + R[7] * np.z;
    float n2 = (R[2] * np.x + R[5] * np.y) + R[8] * np.z;
    float i_prev = (float)((double)((cp.x + cp.y) + cp.z) / 3.0);

    float az = __builtin_fabsf(d2);
    int pu = f2i_rn((d0 / az) * intr[0] + intr[2]);
    int pv = f2i_rn((d1 / az) * intr[1] + intr[3]);
    pu = min(m - 1, max(0, pu));
    pv = min(n - 1, max(0, pv));
    size_t pidx = (size_t)pv * m + pu;

    F3 nc = ld3(NMap, pidx);
    if (is_zero3(nc)) return false;
    F3 s = ld3(VMap, pidx);
    float ex = d0 - s.x, ey = d1 - s.y, ez = d2 - s.z;
    float dist = __builtin_sqrtf((ex * ex + ey * ey) + ez * ez);
    if (dist > distThres) return false;
    float x0 = nc.y * n2 - nc.z * n1;
    float x1 = (-nc.x) * n2 + nc.z * n0;
    float x2 = nc.x * n1 - nc.y * n0;
    float ang = __builtin_sqrtf((x0 * x0 + x1 * x1) + x2 * x2);
    if (ang > angleThres) return false;

    F3 cc = ld3(RGB, pidx);
    float i_curr = (float)((double)((cc.x + cc.y) + cc.z) / 3.0);
    float g0 = (float)((double)*(const float *)((const char *)Gx + (size_t)pv * gxp + 4 * (size_t)pu) / 8.0);
    float g1 = (float)((double)*(const float *)((const char *)Gy + (size_t)pv * gyp + 4 * (size_t)pu) / 8.0);
    float g2 = i_curr - i_prev;
    proj = (int)pidx;

    // ---- rows (D1/ICP.cu:175-205).  Products with the literal 0/1 entries of JD/JRot are
    // dropped: they are exact for finite inputs, so the remaining operations round identically.
    float sd = -s.z;
    float w = (float)(0.0012 / (0.0012 + 0.0019 * ((double)sd - 0.4) * ((double)sd - 0.4)));
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

// fold 64 lane values in the pairing of reduce<64>: reference tid = col*8 + row, lane = row*8 + col,
// so tid offsets 32,16,8,4,2,1 are lane offsets 4,2,1,32,16,8.  Result valid in lane 0.
__device__ __forceinline__ float fold64(float v) {
    v = v + __shfl_down(v, 4);
    v = v + __shfl_down(v, 2);
    v = v + __shfl_down(v, 1);
    v = v + __shfl_down(v, 32);
    v = v + __shfl_down(v, 16);
    v = v + __shfl_down(v, 8);
    return v;
}

// Grid: x over groups of ICP_WAVES block-columns, y over block-rows.  state (optional) gates the
// launch for the device-resident loop: skip when the level already converged or tracking failed.
__global__ __launch_bounds__(64 * ICP_WAVES)
void k_icp_match(const float *__restrict__ intr, const float *__restrict__ Rcurr, const float *__restrict__ tcurr,
                 const float *__restrict__ VMap, const float *__restrict__ NMap, const float *__restrict__ RGB,
                 const float *__restrict__ Gx, size_t gxp, const float *__restrict__ Gy, size_t gyp,
                 const float *__restrict__ VMap_prev, const float *__restrict__ NMap_prev,
                 const float *__restrict__ RGBD_prev, float distThres, float angleThres, int n, int m, int fact,
                 int grid_rows, int grid_cols, float *__restrict__ buf, int32_t *__restrict__ corr,
                 const IcpState *__restrict__ state, int iter) {
    if (state && (state->fail || (iter > 0 && state->cvg))) return;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int bc = blockIdx.x * ICP_WAVES + wave, br = blockIdx.y;
    if (bc >= grid_cols) return;   // whole wave exits together
    int i = br * 8 + (lane >> 3), j = bc * 8 + (lane & 7);
    IcpRow r;
#pragma unroll
    for (int k = 0; k < 7; k++) { r.g[k] = 0.0f; r.c[k] = 0.0f; }
    if (i * fact < n && j * fact < m) {
        size_t idx = (size_t)(i * fact) * m + (size_t)(j * fact);
        int proj = -1;
        IcpRow q;
        bool ok = icp_rows(idx, intr, Rcurr, tcurr, VMap, NMap, RGB, Gx, gxp, Gy, gyp, VMap_prev, NMap_prev, RGBD_prev,
                           distThres, angleThres, n, m, q, proj);
        if (ok) r = q;
        if (corr) corr[idx] = ok ? proj : -1;
    }
    int nblk = grid_rows * grid_cols;
    int blk = br + bc * grid_rows;                       // D1/ICP.cu:238
    int term = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
#pragma unroll
        for (int l = k; l < 7; l++) {
            float v = fold64(r.g[k] * r.g[l] + r.c[k] * r.c[l]);
            if (lane == 0) buf[blk + (size_t)term * nblk] = v;
            term++;
        }
    }
}

// D1/ICP.cu:50-70 ReduceKernel<<<27, 512>>>
__global__ __launch_bounds__(512)
void k_icp_reduce(const float *__restrict__ buf, float *__restrict__ out, int length,
                  const IcpState *__restrict__ state, int iter) {
    if (state && (state->fail || (iter > 0 && state->cvg))) return;
    __shared__ float sm[512];
    const float *beg = buf + (size_t)blockIdx.x * length;
    int tid = threadIdx.x;
    float sum = 0.0f;
    for (int k = tid; k < length; k += 512) sum += beg[k];
    sm[tid] = sum;
    __syncthreads();
    for (int off = 256; off >= 1; off >>= 1) {
        if (tid < off) sm[tid] = sm[tid] + sm[tid + off];
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = sm[0];
}

// ---- device-side 6x6 solve and pose update (D1/RG.cpp:851-901) ---------------------------------
#define M3(a, r, c) (a)[(c) * 3 + (r)]

__device__ void d_mat3_mul(const float *a, const float *b, float *o) {
    float r[9];
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 3; i++)
            r[c * 3 + i] = (M3(a, i, 0) * M3(b, 0, c) + M3(a, i, 1) * M3(b, 1, c)) + M3(a, i, 2) * M3(b, 2, c);
    for (int k = 0; k < 9; k++) o[k] = r[k];
}
__device__ void d_mat3_vec(const float *a, const float *v, float *o) {
    float r[3];
    for (int i = 0; i < 3; i++) r[i] = (M3(a, i, 0) * v[0] + M3(a, i, 1) * v[1]) + M3(a, i, 2) * v[2];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
__device__ void d_inverse3(const float *a, float *o) {
    float cof[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            cof[i][j] = M3(a, i1, j1) * M3(a, i2, j2) - M3(a, i1, j2) * M3(a, i2, j1);
        }
    float det = (cof[0][0] * M3(a, 0, 0) + cof[1][0] * M3(a, 1, 0)) + cof[2][0] * M3(a, 2, 0);
    float invdet = 1.0f / det;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M3(o, i, j) = cof[j][i] * invdet;
}
__device__ float d_frob_minus_I(const float *a) {
    float s = 0.0f;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) {
            float v = M3(a, r, c) - (r == c ? 1.0f : 0.0f);
            s = s + v * v;
        }
    return __builtin_sqrtf(s);
}
__device__ float d_norm3(const float *v) { return __builtin_sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

__device__ float d_det6(const float *Ain) {
    float a[36];
    for (int k = 0; k < 36; k++) a[k] = Ain[k];
    float det = 1.0f;
    for (int c = 0; c < 6; c++) {
        int p = c;
        float best = __builtin_fabsf(a[c * 6 + c]);
        for (int r = c + 1; r < 6; r++) {
            float v = __builtin_fabsf(a[r * 6 + c]);
            if (v > best) { best = v; p = r; }
        }
        if (p != c) {
            for (int k = 0; k < 6; k++) { float tmp = a[c * 6 + k]; a[c * 6 + k] = a[p * 6 + k]; a[p * 6 + k] = tmp; }
            det = -det;
        }
        float piv = a[c * 6 + c];
        det = det * piv;
        if (piv == 0.0f || piv != piv) return det;
        for (int r = c + 1; r < 6; r++) {
            float f = a[r * 6 + c] / piv;
            for (int k = c + 1; k < 6; k++) a[r * 6 + k] = a[r * 6 + k] - f * a[c * 6 + k];
        }
    }
    return det;
}
__device__ int d_llt_solve6(const float *A, const float *b, float *x) {
    float L[36];
    for (int k = 0; k < 36; k++) L[k] = 0.0f;
    for (int k = 0; k < 6; k++) {
        float s = A[k * 6 + k];
        for (int p = 0; p < k; p++) s = s - L[k * 6 + p] * L[k * 6 + p];
        if (!(s > 0.0f)) return 1;
        float lkk = __builtin_sqrtf(s);
        L[k * 6 + k] = lkk;
        for (int r = k + 1; r < 6; r++) {
            float v = A[r * 6 + k];
            for (int p = 0; p < k; p++) v = v - L[r * 6 + p] * L[k * 6 + p];
            L[r * 6 + k] = v / lkk;
        }
    }
    float y[6];
    for (int r = 0; r < 6; r++) {
        float v = b[r];
        for (int p = 0; p < r; p++) v = v - L[r * 6 + p] * y[p];
        y[r] = v / L[r * 6 + r];
    }
    for (int r = 5; r >= 0; r--) {
        float v = y[r];
        for (int p = r + 1; p < 6; p++) v = v - L[p * 6 + r] * x[p];
        x[r] = v / L[r * 6 + r];
    }
    return 0;
}

// pose_state = poseR(9) poset(3) poseRPrev(9) posetPrev(3)
__device__ void d_set_rt_curr(IcpState *st, const float *ps) {
    float tmp[3];
    d_mat3_mul(st->Rinv, ps + 12, st->Rcurr);                 // RG.cpp:851
    d_mat3_vec(st->Rinv, ps + 21, tmp);                       // RG.cpp:852
    st->tcurr[0] = tmp[0] + st->tinv[0]; st->tcurr[1] = tmp[1] + st->tinv[1]; st->tcurr[2] = tmp[2] + st->tinv[2];
}

__global__ void k_icp_begin(IcpState *st, const float *pose_state, CalibV c, int32_t *status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float tmp[3];
    d_inverse3(pose_state, st->Rinv);                         // RG.cpp:834
    d_mat3_vec(st->Rinv, pose_state + 9, tmp);                // RG.cpp:835
    st->tinv[0] = -tmp[0]; st->tinv[1] = -tmp[1]; st->tinv[2] = -tmp[2];
    for (int k = 0; k < 4; k++) st->intr[k] = c.v[k];
    st->cvg = 0; st->fail = 0; st->iters = 0; st->lvl_iters = 0;
    d_set_rt_curr(st, pose_state);
    status[0] = 0; status[1] = 0;
}

__global__ void k_icp_solve(IcpState *st, const float *__restrict__ sums, const float *pose_state, int lvl, int iter,
                            int max_iter, float thr_alpha, float thr_tr, float *trace) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->fail || (iter > 0 && st->cvg)) return;
    float A[36], b[6], x[6];
    int sh = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 7; j++) {                          // ICP.cu:273-283
            float v = sums[sh++];
            if (j == 6) b[i] = v; else { A[j * 6 + i] = v; A[i * 6 + j] = v; }
        }
    float det = d_det6(A);                                     // RG.cpp:862
    if ((double)__builtin_fabsf(det) < 1e-15 || det != det) { st->fail = 1; return; }
    if (d_llt_solve6(A, b, x)) { st->fail = 1; return; }
    if (trace) {
        float *tr = trace + 33 * st->iters;
        for (int k = 0; k < 27; k++) tr[k] = sums[k];
        for (int k = 0; k < 6; k++) tr[27 + k] = x[k];
    }
    st->iters = st->iters + 1;
    st->lvl_iters = st->lvl_iters + (1 << (8 * lvl));
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
    float tinc[3] = {x[0], x[1], x[2]}, tmp[3];
    d_mat3_vec(Rinc, st->tinv, tmp);                           // RG.cpp:886
    st->tinv[0] = tmp[0] + tinc[0]; st->tinv[1] = tmp[1] + tinc[1]; st->tinv[2] = tmp[2] + tinc[2];
    d_mat3_mul(Rinc, st->Rinv, st->Rinv);                      // RG.cpp:887
    int cvg = (iter > max_iter) || (d_frob_minus_I(Rinc) < thr_alpha && d_norm3(tinc) < thr_tr);  // RG.cpp:889
    st->cvg = cvg;
    // Refresh the uploaded pair (RG.cpp:851-852) for whichever iteration runs next: the next one of
    // this level (iff !cvg) or iteration 0 of the next level (always).  After the final executed
    // iteration of level 0 the pair is left untouched: the track-loss test reads the LAST
    // uploaded pair, not the final estimate (RG.cpp:910).
    if (!cvg || lvl > 0) d_set_rt_curr(st, pose_state);
}

__global__ void k_icp_end(IcpState *st, float *pose_state, int32_t *status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    status[1] = st->iters;
    status[2] = st->lvl_iters;
    if (st->fail) { status[0] = 0; return; }
    if ((double)d_frob_minus_I(st->Rcurr) > 0.3 || (double)d_norm3(st->tcurr) > 0.3) { status[0] = 0; return; }  // RG.cpp:910
    float R[9], neg[9], t[3];
    d_inverse3(st->Rinv, R);                                   // RG.cpp:923
    for (int k = 0; k < 9; k++) neg[k] = -R[k];
    d_mat3_vec(neg, st->tinv, t);                              // RG.cpp:924
    for (int k = 0; k < 9; k++) pose_state[k] = R[k];
    for (int k = 0; k < 3; k++) pose_state[9 + k] = t[k];
    status[0] = 1;
}

// =================================================================================================
// C ABI
// =================================================================================================
static int div_up(int x, int y) { return (x % y) ? ((x + y - 1) / y) : (x / y); }

extern "C" int rgbdslam_alloc_buffers(rgbdslam_ctx *ctx, int grid_rows, int grid_cols) {
    if (!ctx || grid_rows <= 0 || grid_cols <= 0) return RGBDSLAM_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    size_t blocks = (size_t)grid_rows * grid_cols;
    if (ctx->buf_dev && ctx->buf_blocks >= blocks) return 0;
    if (ctx->buf_dev) { HIP_TRY(hipFree(ctx->buf_dev)); ctx->buf_dev = nullptr; }
    HIP_TRY(hipMalloc((void **)&ctx->buf_dev, 27 * blocks * sizeof(float)));
    ctx->buf_blocks = blocks;
    if (!ctx->outbuf_dev) HIP_TRY(hipMalloc((void **)&ctx->outbuf_dev, 27 * sizeof(float)));
    if (!ctx->out_host) HIP_TRY(hipHostMalloc((void **)&ctx->out_host, 27 * sizeof(float), hipHostMallocDefault));
    if (!ctx->icp_state) HIP_TRY(hipMalloc((void **)&ctx->icp_state, sizeof(IcpState)));
    return 0;
}

extern "C" int rgbdslam_free_buffers(rgbdslam_ctx *ctx) {
    if (!ctx) return RGBDSLAM_E_ARG;
    if (ctx->buf_dev) { (void)hipFree(ctx->buf_dev); ctx->buf_dev = nullptr; ctx->buf_blocks = 0; }
    if (ctx->outbuf_dev) { (void)hipFree(ctx->outbuf_dev); ctx->outbuf_dev = nullptr; }
    if (ctx->out_host) { (void)hipHostFree(ctx->out_host); ctx->out_host = nullptr; }
    if (ctx->icp_state) { (void)hipFree(ctx->icp_state); ctx->icp_state = nullptr; }
    return 0;
}

static int launch_system(rgbdslam_ctx *ctx, const float *intr, const float *R, const float *t, const float *VMap,
                         const float *NMap, const float *RGB, const float *Gx, size_t gxp, const float *Gy, size_t gyp,
                         const float *VMap_prev, const float *NMap_prev, const float *RGB_prev, float distThres,
                         float angleThres, int n, int m, int fact, int32_t *corr, const IcpState *state, int iter) {
    int gr = div_up(n / fact, 8), gc = div_up(m / fact, 8);   // D1/ICP.cu:260-261
    if (gr <= 0 || gc <= 0) return RGBDSLAM_E_ARG;
    if (!ctx->buf_dev || ctx->buf_blocks < (size_t)gr * gc) return RGBDSLAM_E_STATE;
    dim3 grid((gc + ICP_WAVES - 1) / ICP_WAVES, gr);
    KLAUNCH(ctx, k_icp_match, grid, dim3(64 * ICP_WAVES), intr, R, t, VMap, NMap, RGB, Gx, gxp,
                       Gy, gyp, VMap_prev, NMap_prev, RGB_prev, distThres, angleThres, n, m, fact, gr, gc, ctx->buf_dev,
                       corr, state, iter);
    KLAUNCH(ctx, k_icp_reduce, dim3(27), dim3(512), ctx->buf_dev, ctx->outbuf_dev, gr * gc, state, iter);
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
    int rc = launch_system(ctx, intr_dev, Rcurr_dev, tcurr_dev, VMap, NMap, RGB, Gx, gxp, Gy, gyp, VMap_prev, NMap_prev,
                           RGB_prev, distThres, angleThres, n, m, fact, corr_dev, nullptr, 0);
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

extern "C" int rgbdslam_align_gauss_newton(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap,
                                           const float *NMap, const float *RGB, const float *Gx, size_t gxp,
                                           const float *Gy, size_t gyp, const float *VMap_prev, const float *NMap_prev,
                                           const float *RGB_prev, int n, int m, float *pose_state_dev,
                                           int32_t *status_dev, float *trace_dev) {
    if (!ctx || !prm || !VMap || !NMap || !RGB || !Gx || !Gy || !VMap_prev || !NMap_prev || !RGB_prev ||
        !pose_state_dev || !status_dev)
        return RGBDSLAM_E_ARG;
    if (n <= 0 || m <= 0 || prm->lvl < 1 || prm->lvl > 3 || gxp < 4 * (size_t)m || gyp < 4 * (size_t)m) return RGBDSLAM_E_ARG;
    if (!ctx->have_calib || !ctx->icp_state) return RGBDSLAM_E_STATE;
    IcpState *st = ctx->icp_state;
    KLAUNCH(ctx, k_icp_begin, dim3(1), dim3(64), st, pose_state_dev, ctx->calib, status_dev);
    int fact = 1 << (prm->lvl - 1);                            // RG.cpp:832
    for (int lvl = prm->lvl - 1; lvl > -1; lvl--) {
        int n_it = prm->max_iter[lvl] + 2;                     // iter = 0 .. max_iter+1 (RG.cpp:889)
        for (int iter = 0; iter < n_it; iter++) {
            int rc = launch_system(ctx, st->intr, st->Rcurr, st->tcurr, VMap, NMap, RGB, Gx, gxp, Gy, gyp, VMap_prev,
                                   NMap_prev, RGB_prev, prm->distThres, prm->angleThres, n, m, fact, nullptr, st, iter);
            if (rc) return rc;
            KLAUNCH(ctx, k_icp_solve, dim3(1), dim3(64), st, ctx->outbuf_dev, pose_state_dev, lvl, iter,
                               prm->max_iter[lvl], prm->thres_cvg_alpha, prm->thres_cvg_tr, trace_dev);
        }
        fact /= 2;
    }
    KLAUNCH(ctx, k_icp_end, dim3(1), dim3(64), st, pose_state_dev, status_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}
