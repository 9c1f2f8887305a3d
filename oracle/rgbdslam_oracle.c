/*
 * rgbdslam_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (see rgbdslam_oracle.h).  Build: gcc -O2 -ffp-contract=off -fPIC -shared.
 *
 * Each function cites the reference lines (D1 = DemoStep1/DemoStep1) whose arithmetic it follows,
 * including float/double promotion caused by un-suffixed literals.
 */
#include "rgbdslam_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* The per-row / per-pixel / per-block loops carry `#pragma omp parallel for`.  The default library is built WITHOUT
 * -fopenmp: the pragmas are ignored and everything runs on one thread.  The Makefile's second target compiles this same
 * file with -fopenmp into librgbdslam_oracle_omp.so, the multi-core CPU baseline SURVEY 8(d)(ii) asks for.  Every
 * output element is still produced by one thread with the same operations in the same order, and the two places where
 * iterations meet (z-buffer minimum, claim maximum) are commutative: results are bit-identical (tests/test_oracle_omp.py). */
#ifdef _OPENMP
#include <omp.h>
int orc_threads(void) { return omp_get_max_threads(); }
#else
int orc_threads(void) { return 1; }
#endif
#ifdef _OPENMP
#define ORC_MIN_U64(p, key) do { uint64_t k_ = (key), o_ = __atomic_load_n((p), __ATOMIC_RELAXED); \
        while (k_ < o_ && !__atomic_compare_exchange_n((p), &o_, k_, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} } while (0)
#define ORC_MAX_U64(p, key) do { uint64_t k_ = (key), o_ = __atomic_load_n((p), __ATOMIC_RELAXED); \
        while (k_ > o_ && !__atomic_compare_exchange_n((p), &o_, k_, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} } while (0)
#else
#define ORC_MIN_U64(p, key) do { uint64_t k_ = (key); if (k_ < *(p)) *(p) = k_; } while (0)
#define ORC_MAX_U64(p, key) do { uint64_t k_ = (key); if (k_ > *(p)) *(p) = k_; } while (0)
#endif

#define MAX_DEPTH 100.0    /* D1/cudaTypes.cuh:51 */
#define MAXTOLERANCE 0.2   /* D1/cudaTypes.cuh:47 */

/* __float2int_rn: round half to even, saturating, NaN -> 0 (CUDA semantics). */
static int f2i_rn(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT_MAX;
    if (x <= -2147483648.0f) return INT_MIN;
    return (int)rintf(x);
}
/* __float2int_rd: floor, saturating, NaN -> 0. */
static int f2i_rd(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT_MAX;
    if (x <= -2147483648.0f) return INT_MIN;
    return (int)floorf(x);
}

/* ===================================================================================== */
/* Stage 1                                                                               */
/* ===================================================================================== */

void orc_vertex_map_kinect(float *VMap, const uint16_t *depth, size_t pitch,
                           const float *calib, int n, int m) {
    /* D1/Kernel.cu:42-44 */
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            uint16_t d = depth[(size_t)(n - i - 1) * pitch + j];
            if (d == 0) {
                VMap[3 * idx] = 0.0f; VMap[3 * idx + 1] = 0.0f; VMap[3 * idx + 2] = 0.0f;
                continue;
            }
            float s = calib[9] / calib[10];
            float fd = (float)d;
            VMap[3 * idx]     = (s * fd) * (((float)j - calib[2]) / calib[0]);
            VMap[3 * idx + 1] = (s * fd) * (((float)i - calib[3]) / calib[1]);
            VMap[3 * idx + 2] = (-s) * fd;
        }
}

void orc_cpy_char3(const uint8_t *bgr, size_t pitch, float *rgb, int n, int m) {
    /* D1/Kernel.cu:747-749: float(u8)/255.0 is a double divide */
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            const uint8_t *px = bgr + (size_t)(n - i - 1) * pitch + 3 * (size_t)j;
            rgb[3 * idx]     = (float)((double)(float)px[2] / 255.0);
            rgb[3 * idx + 1] = (float)((double)(float)px[1] / 255.0);
            rgb[3 * idx + 2] = (float)((double)(float)px[0] / 255.0);
        }
}

/* one triangle of D1/Kernel.cu:162-272: normalised cross((p2-p1),(p3-p1)), 1 if it counted */
static int tri_normal(const float *p1, const float *p2, const float *p3, float *out) {
    out[0] = out[1] = out[2] = 0.0f;
    if (p2[2] != 0.0f && p3[2] != 0.0f) {
        float a0 = p2[0] - p1[0], a1 = p2[1] - p1[1], a2 = p2[2] - p1[2];
        float b0 = p3[0] - p1[0], b1 = p3[1] - p1[1], b2 = p3[2] - p1[2];
        float c0 = a1 * b2 - a2 * b1;
        float c1 = a2 * b0 - a0 * b2;
        float c2 = a0 * b1 - a1 * b0;
        float nn = (c0 * c0 + c1 * c1 + c2 * c2);
        out[0] = c0; out[1] = c1; out[2] = c2;
        if (nn != 0.0f) {
            float r = sqrtf(nn);
            out[0] = c0 / r; out[1] = c1 / r; out[2] = c2 / r;
            return 1;
        }
        /* zero cross product: the reference keeps the (all-zero) vector and does not count it */
    }
    return 0;
}

void orc_compute_normal(float *NMap, const float *VMap, int n, int m, int inverse) {
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            float *o = NMap + 3 * idx;
            if (i < 1 || i > n - 2 || j < 1 || j > m - 2 || VMap[3 * idx + 2] == 0.0f) {
                o[0] = o[1] = o[2] = 0.0f;
                continue;
            }
            const float *p  = VMap + 3 * idx;
            const float *dn = VMap + 3 * ((size_t)(i + 1) * m + j);   /* (i+1, j) */
            const float *up = VMap + 3 * ((size_t)(i - 1) * m + j);   /* (i-1, j) */
            const float *rt = VMap + 3 * ((size_t)i * m + j + 1);     /* (i, j+1) */
            const float *lf = VMap + 3 * ((size_t)i * m + j - 1);     /* (i, j-1) */
            float n1[3], n2[3], n3[3], n4[3];
            int tot = 0;
            tot += tri_normal(p, dn, rt, n1);   /* Kernel.cu:164-188 */
            tot += tri_normal(p, rt, up, n2);   /* Kernel.cu:192-216 */
            tot += tri_normal(p, up, lf, n3);   /* Kernel.cu:220-244 */
            tot += tri_normal(p, lf, dn, n4);   /* Kernel.cu:248-272 */
            if (tot == 0) { o[0] = o[1] = o[2] = 0.0f; continue; }
            float ft = (float)tot;
            float a0 = (((n1[0] + n2[0]) + n3[0]) + n4[0]) / ft;
            float a1 = (((n1[1] + n2[1]) + n3[1]) + n4[1]) / ft;
            float a2 = (((n1[2] + n2[2]) + n3[2]) + n4[2]) / ft;
            float nn = sqrtf((a0 * a0 + a1 * a1) + a2 * a2);
            if (nn != 0.0f) {
                if (inverse) { o[0] = (-a0) / nn; o[1] = (-a1) / nn; o[2] = (-a2) / nn; }
                else         { o[0] = a0 / nn;    o[1] = a1 / nn;    o[2] = a2 / nn; }
            } else {
                o[0] = o[1] = o[2] = 0.0f;
            }
        }
}

void orc_intensity(float *img, const float *rgb, int n, int m) {
    /* D1/Kernel.cu:678: (rgb[2] + rgb[1] + rgb[0]) / 3.0 */
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < (size_t)n * m; idx++) {
        float s = (rgb[3 * idx + 2] + rgb[3 * idx + 1]) + rgb[3 * idx];
        img[idx] = (float)((double)s / 3.0);
    }
}

static int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

void orc_sobel(const float *img, float *gx, float *gy, int n, int m) {
    /* OpenCV 2.4 cv::gpu::Sobel(ksize 3, scale 1, BORDER_DEFAULT): separable row filter then
     * column filter, accumulated in float in tap order (third-party, parity unpinned).
     * dx: row taps (-1,0,1), column taps (1,2,1);  dy: row taps (1,2,1), column taps (-1,0,1). */
    float *rowd = (float *)malloc(sizeof(float) * (size_t)n * m);
    float *rows = (float *)malloc(sizeof(float) * (size_t)n * m);
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            float a = img[(size_t)i * m + reflect101(j - 1, m)];
            float b = img[(size_t)i * m + j];
            float c = img[(size_t)i * m + reflect101(j + 1, m)];
            float s = 0.0f;
            s = s + a * -1.0f; s = s + b * 0.0f; s = s + c * 1.0f;
            rowd[(size_t)i * m + j] = s;
            s = 0.0f;
            s = s + a * 1.0f; s = s + b * 2.0f; s = s + c * 1.0f;
            rows[(size_t)i * m + j] = s;
        }
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        int iu = reflect101(i - 1, n), id = reflect101(i + 1, n);
        for (int j = 0; j < m; j++) {
            float s = 0.0f;
            s = s + rowd[(size_t)iu * m + j] * 1.0f;
            s = s + rowd[(size_t)i * m + j] * 2.0f;
            s = s + rowd[(size_t)id * m + j] * 1.0f;
            gx[(size_t)i * m + j] = s;
            s = 0.0f;
            s = s + rows[(size_t)iu * m + j] * -1.0f;
            s = s + rows[(size_t)i * m + j] * 0.0f;
            s = s + rows[(size_t)id * m + j] * 1.0f;
            gy[(size_t)i * m + j] = s;
        }
    }
    free(rowd); free(rows);
}

/* ===================================================================================== */
/* Stage 2+3: ICP                                                                        */
/* ===================================================================================== */

static int div_up(int x, int y) { return (x % y) ? ((x + y - 1) / y) : (x / y); } /* cudaTypes.cuh:53 */

void orc_icp_grid(int n, int m, int fact, int *gr, int *gc) {
    *gr = div_up(n / fact, 8);   /* D1/ICP.cu:260 */
    *gc = div_up(m / fact, 8);   /* D1/ICP.cu:261 */
}

/* D1/ICP.cu:72-146.  Returns 1 when a correspondence was found. */
static int search_gauss(size_t indx, const float *intr, const float *R, const float *t,
                        const float *VMap, const float *NMap, const float *RGB,
                        const float *Gx, const float *Gy,
                        const float *VMap_prev, const float *NMap_prev, const float *RGBD_prev,
                        float distThres, float angleThres, int n, int m,
                        float *nn, float *d, float *s, float *rgb_val, int32_t *proj) {
    float np0 = NMap_prev[3 * indx], np1 = NMap_prev[3 * indx + 1], np2 = NMap_prev[3 * indx + 2];
    if (np0 == 0.0f && np1 == 0.0f && np2 == 0.0f) return 0;
    float vp0 = VMap_prev[3 * indx], vp1 = VMap_prev[3 * indx + 1], vp2 = VMap_prev[3 * indx + 2];

    float v0 = ((R[0] * vp0 + R[3] * vp1) + R[6] * vp2) + t[0];
    float v1 = ((R[1] * vp0 + R[4] * vp1) + R[7] * vp2) + t[1];
    float v2 = ((R[2] * vp0 + R[5] * vp1) + R[8] * vp2) + t[2];
    float q0 = (R[0] * np0 + R[3] * np1) + R[6] * np2;
    float q1 = (R[1] * np0 + R[4] * np1) + R[7] * np2;
    float q2 = (R[2] * np0 + R[5] * np1) + R[8] * np2;

    float sp = (RGBD_prev[4 * indx] + RGBD_prev[4 * indx + 1]) + RGBD_prev[4 * indx + 2];
    float intsy_prev = (float)((double)sp / 3.0);                                  /* ICP.cu:106 */

    int pu = f2i_rn((v0 / fabsf(v2)) * intr[0] + intr[2]);                         /* ICP.cu:108 */
    int pv = f2i_rn((v1 / fabsf(v2)) * intr[1] + intr[3]);                         /* ICP.cu:109 */
    if (pu < 0) pu = 0; if (pu > m - 1) pu = m - 1;
    if (pv < 0) pv = 0; if (pv > n - 1) pv = n - 1;
    size_t pidx = (size_t)pv * m + pu;

    float c0 = NMap[3 * pidx], c1 = NMap[3 * pidx + 1], c2 = NMap[3 * pidx + 2];
    if (c0 == 0.0f && c1 == 0.0f && c2 == 0.0f) return 0;
    float w0 = VMap[3 * pidx], w1 = VMap[3 * pidx + 1], w2 = VMap[3 * pidx + 2];

    float e0 = v0 - w0, e1 = v1 - w1, e2 = v2 - w2;
    float dist = sqrtf((e0 * e0 + e1 * e1) + e2 * e2);                             /* ICP.cu:123 */
    if (dist > distThres) return 0;

    float x0 = c1 * q2 - c2 * q1;                                                  /* ICP.cu:127 */
    float x1 = (-c0) * q2 + c2 * q0;
    float x2 = c0 * q1 - c1 * q0;
    float angle = sqrtf((x0 * x0 + x1 * x1) + x2 * x2);
    if (angle > angleThres) return 0;

    float sc = (RGB[3 * pidx] + RGB[3 * pidx + 1]) + RGB[3 * pidx + 2];
    float intsy_curr = (float)((double)sc / 3.0);                                  /* ICP.cu:135 */

    nn[0] = q0; nn[1] = q1; nn[2] = q2;
    d[0] = v0; d[1] = v1; d[2] = v2;
    s[0] = w0; s[1] = w1; s[2] = w2; s[3] = -w2;
    rgb_val[0] = (float)((double)Gx[pidx] / 8.0);
    rgb_val[1] = (float)((double)Gy[pidx] / 8.0);
    rgb_val[2] = (intsy_curr - intsy_prev);
    *proj = (int32_t)pidx;
    return 1;
}

/* D1/ICP.cu:164-205: the two 7-vectors of one correspondence */
static void jacobian_rows(const float *intr, const float *n, const float *d, const float *s,
                          const float *g, float *row, float *row_rgb) {
    float lambda = 0.03f;                                                          /* ICP.cu:159 */
    float sd = s[3];
    float weight = (float)(0.0012 / (0.0012 + 0.0019 * ((double)sd - 0.4) * ((double)sd - 0.4)));
    /* columns of JD (ICP.cu:177-179) and JRot (ICP.cu:181-183); 2.0*x is exact */
    float JD[6][3] = {
        {1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f},
        {0.0f, -2.0f * d[2], 2.0f * d[1]},
        {2.0f * d[2], 0.0f, -2.0f * d[0]},
        {-2.0f * d[1], 2.0f * d[0], 0.0f}};
    float JR[6][3] = {
        {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f},
        {0.0f, -2.0f * n[2], 2.0f * n[1]},
        {2.0f * n[2], 0.0f, -2.0f * n[0]},
        {-2.0f * n[1], 2.0f * n[0], 0.0f}};
    float e0 = s[0] - d[0], e1 = s[1] - d[1], e2 = s[2] - d[2];
    for (int k = 0; k < 6; k++) {                                                  /* ICP.cu:185-190 */
        float acc = -((n[0] * JD[k][0] + n[1] * JD[k][1]) + n[2] * JD[k][2]);
        acc = acc + JR[k][0] * e0;
        acc = acc + JR[k][1] * e1;
        acc = acc + JR[k][2] * e2;
        row[k] = weight * acc;
    }
    row[6] = (-weight) * ((n[0] * e0 + n[1] * e1) + n[2] * e2);                    /* ICP.cu:192 */

    float JP0 = intr[0] / fabsf(d[2]);                                             /* ICP.cu:194 */
    float JP3 = intr[1] / fabsf(d[2]);
    float JP4 = ((-d[0]) * intr[0]) / (d[2] * d[2]);
    float JP5 = ((-d[1]) * intr[1]) / (d[2] * d[2]);
    float ga = g[0] * JP0 + g[1] * 0.0f;
    float gb = g[0] * 0.0f + g[1] * JP3;
    float gc = g[0] * JP4 + g[1] * JP5;
    float lw = lambda * weight;
    for (int k = 0; k < 6; k++)                                                    /* ICP.cu:198-203 */
        row_rgb[k] = lw * ((ga * JD[k][0] + gb * JD[k][1]) + gc * JD[k][2]);
    row_rgb[6] = ((-lambda) * weight) * g[2];                                      /* ICP.cu:205 */
}

/* D1/ICP.cu:28-48 for CTA_SIZE 64: tid pairs (t,t+32),(t,t+16),...,(t,t+1) */
static float tree64(float *v) {
    for (int off = 32; off >= 1; off >>= 1)
        for (int t = 0; t < off; t++) v[t] = v[t] + v[t + off];
    return v[0];
}

void orc_icp_match(const float *intr, const float *R, const float *t,
                   const float *VMap, const float *NMap, const float *RGB,
                   const float *Gx, const float *Gy,
                   const float *VMap_prev, const float *NMap_prev, const float *RGBD_prev,
                   float distThres, float angleThres, int n, int m, int fact,
                   float *buf, int32_t *corr) {
    int gr, gc;
    orc_icp_grid(n, m, fact, &gr, &gc);
    int nblk = gr * gc;
    if (corr) for (size_t k = 0; k < (size_t)n * m; k++) corr[k] = -1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bc = 0; bc < gc; bc++)
        for (int br = 0; br < gr; br++) {
            float rows[64][7], rgbs[64][7], tmp[64];
            int blk = br + bc * gr;                                /* ICP.cu:238 */
            for (int ty = 0; ty < 8; ty++)
                for (int tx = 0; tx < 8; tx++) {
                    int tid = ty * 8 + tx;                         /* ICP.cu:25: y*blockDim.x + x */
                    int i = tx + br * 8, j = ty + bc * 8;          /* ICP.cu:152-153 */
                    for (int k = 0; k < 7; k++) rows[tid][k] = rgbs[tid][k] = 0.0f;
                    if ((i * fact) < n && (j * fact) < m) {
                        size_t idx = (size_t)(i * fact) * m + (size_t)(j * fact);
                        float nn[3], d[3], s[4], g[3];
                        int32_t proj;
                        if (search_gauss(idx, intr, R, t, VMap, NMap, RGB, Gx, Gy, VMap_prev,
                                         NMap_prev, RGBD_prev, distThres, angleThres, n, m,
                                         nn, d, s, g, &proj)) {
                            jacobian_rows(intr, nn, d, s, g, rows[tid], rgbs[tid]);
                            if (corr) corr[idx] = proj;
                        }
                    }
                }
            int term = 0;
            for (int k = 0; k < 6; k++)
                for (int l = k; l < 7; l++) {                      /* ICP.cu:226-241 */
                    for (int tid = 0; tid < 64; tid++)
                        tmp[tid] = rows[tid][k] * rows[tid][l] + rgbs[tid][k] * rgbs[tid][l];
                    buf[blk + (size_t)(term++) * nblk] = tree64(tmp);
                }
        }
}

void orc_icp_reduce(const float *buf, int nblk, float *out27) {
    /* D1/ICP.cu:50-70: thread tid sums beg[tid], beg[tid+512], ... then reduce<512> */
    #pragma omp parallel for schedule(static)
    for (int b = 0; b < 27; b++) {
        float sm[512];
        const float *beg = buf + (size_t)b * nblk;
        for (int tid = 0; tid < 512; tid++) {
            float sum = 0.0f;
            for (int k = tid; k < nblk; k += 512) sum += beg[k];
            sm[tid] = sum;
        }
        for (int off = 256; off >= 1; off >>= 1)
            for (int t = 0; t < off; t++) sm[t] = sm[t] + sm[t + off];
        out27[b] = sm[0];
    }
}

void orc_estimate_system(const float *intr, const float *R, const float *t,
                         const float *VMap, const float *NMap, const float *RGB,
                         const float *Gx, const float *Gy,
                         const float *VMap_prev, const float *NMap_prev, const float *RGBD_prev,
                         float distThres, float angleThres, int n, int m, int fact,
                         float *A, float *b, float *sums27) {
    int gr, gc;
    orc_icp_grid(n, m, fact, &gr, &gc);
    int nblk = gr * gc;
    float *buf = (float *)calloc((size_t)27 * nblk, sizeof(float));
    float out[27];
    orc_icp_match(intr, R, t, VMap, NMap, RGB, Gx, Gy, VMap_prev, NMap_prev, RGBD_prev,
                  distThres, angleThres, n, m, fact, buf, NULL);
    orc_icp_reduce(buf, nblk, out);
    free(buf);
    int sh = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 7; j++) {                              /* ICP.cu:273-283 */
            float v = out[sh++];
            if (j == 6) b[i] = v;
            else A[j * 6 + i] = A[i * 6 + j] = v;
        }
    if (sums27) memcpy(sums27, out, sizeof(out));
}

/* ---- dense helpers (stand-ins for Eigen; parity unpinned) ---- */

float orc_det6(const float *Ain) {
    float a[36];
    memcpy(a, Ain, sizeof(a));
    float det = 1.0f;
    for (int c = 0; c < 6; c++) {
        int p = c;
        float best = fabsf(a[c * 6 + c]);
        for (int r = c + 1; r < 6; r++) {
            float v = fabsf(a[r * 6 + c]);
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

int orc_llt_solve6(const float *A, const float *b, float *x) {
    float L[36];
    memset(L, 0, sizeof(L));
    for (int k = 0; k < 6; k++) {
        float s = A[k * 6 + k];
        for (int p = 0; p < k; p++) s = s - L[k * 6 + p] * L[k * 6 + p];
        if (!(s > 0.0f)) return 1;
        float lkk = sqrtf(s);
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

#define M3(a, r, c) (a)[(c) * 3 + (r)]   /* column-major 3x3 */

void orc_inverse3(const float *a, float *o) {
    /* cofactor inverse (Eigen compute_inverse_size3 structure) */
    /* cofactor<i,j> = m(i+1,j+1)*m(i+2,j+2) - m(i+1,j+2)*m(i+2,j+1), indices mod 3 */
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

void orc_quaternion2matrix(const double *q, double m[3][3]) {
    /* D1/Utilities.cpp:1145-1167 */
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q03 = q[0] * q[3], q13 = q[1] * q[3], q23 = q[2] * q[3];
    double q02 = q[0] * q[2], q12 = q[1] * q[2], q01 = q[0] * q[1];
    m[0][0] = q00 + q11 - q22 - q33;
    m[1][1] = q00 - q11 + q22 - q33;
    m[2][2] = q00 - q11 - q22 + q33;
    m[0][1] = 2.0 * (q12 - q03);
    m[1][0] = 2.0 * (q12 + q03);
    m[0][2] = 2.0 * (q13 + q02);
    m[2][0] = 2.0 * (q13 - q02);
    m[1][2] = 2.0 * (q23 - q01);
    m[2][1] = 2.0 * (q23 + q01);
}

static void mat3_mul(const float *a, const float *b, float *o) {   /* o = a*b, column-major */
    float r[9];
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 3; i++)
            r[c * 3 + i] = (M3(a, i, 0) * M3(b, 0, c) + M3(a, i, 1) * M3(b, 1, c)) + M3(a, i, 2) * M3(b, 2, c);
    memcpy(o, r, sizeof(r));
}
static void mat3_vec(const float *a, const float *v, float *o) {
    float r[3];
    for (int i = 0; i < 3; i++) r[i] = (M3(a, i, 0) * v[0] + M3(a, i, 1) * v[1]) + M3(a, i, 2) * v[2];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
static float frob_minus_I(const float *a) {
    float s = 0.0f;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) {
            float v = M3(a, r, c) - (r == c ? 1.0f : 0.0f);
            s = s + v * v;
        }
    return sqrtf(s);
}
static float norm3(const float *v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

int orc_align_gauss_newton(const orc_icp_params *prm, const float *intr,
                           const float *VMap, const float *NMap, const float *RGB,
                           const float *VMap_prev, const float *NMap_prev, const float *RGBD_prev,
                           int n, int m, float *poseR, float *poset,
                           const float *poseRPrev, const float *posetPrev,
                           float *trace, int *n_iters) {
    size_t P = (size_t)n * m;
    float *img = (float *)malloc(P * sizeof(float));
    float *gx = (float *)malloc(P * sizeof(float));
    float *gy = (float *)malloc(P * sizeof(float));
    orc_intensity(img, RGB, n, m);                                   /* RG.cpp:807 */
    orc_sobel(img, gx, gy, n, m);                                    /* RG.cpp:809-810 */

    float Rcurr[9], tcurr[3], Rinv[9], tinv[3];
    memcpy(Rcurr, poseR, sizeof(Rcurr));
    memcpy(tcurr, poset, sizeof(tcurr));
    int fact = 1;
    for (int k = 0; k < prm->lvl - 1; k++) fact *= 2;                /* RG.cpp:832 */
    orc_inverse3(Rcurr, Rinv);                                       /* RG.cpp:834 */
    { float tmp[3]; mat3_vec(Rinv, tcurr, tmp); tinv[0] = -tmp[0]; tinv[1] = -tmp[1]; tinv[2] = -tmp[2]; }

    int it_total = 0, ok = 1;
    for (int lvl = prm->lvl - 1; lvl > -1 && ok; lvl--) {
        int cvg = 0, iter = 0;
        while (!cvg) {
            float tmpv[3];
            mat3_mul(Rinv, poseRPrev, Rcurr);                        /* RG.cpp:851 */
            mat3_vec(Rinv, posetPrev, tmpv);                         /* RG.cpp:852 */
            tcurr[0] = tmpv[0] + tinv[0]; tcurr[1] = tmpv[1] + tinv[1]; tcurr[2] = tmpv[2] + tinv[2];

            float A[36], b[6], sums[27], x[6];
            orc_estimate_system(intr, Rcurr, tcurr, VMap, NMap, RGB, gx, gy, VMap_prev, NMap_prev,
                                RGBD_prev, prm->distThres, prm->angleThres, n, m, fact, A, b, sums);
            float det = orc_det6(A);                                 /* RG.cpp:862 */
            if (fabsf(det) < 1e-15 || det != det) { ok = 0; break; } /* RG.cpp:864-871 */
            if (orc_llt_solve6(A, b, x)) { ok = 0; break; }          /* non-positive pivot => lost */
            if (trace) { memcpy(trace + 33 * it_total, sums, sizeof(sums)); memcpy(trace + 33 * it_total + 27, x, sizeof(x)); }
            it_total++;

            double q[4];
            double nrm = (double)((x[3] * x[3] + x[4] * x[4]) + x[5] * x[5]);   /* RG.cpp:875 (float expr) */
            q[1] = x[3]; q[2] = x[4]; q[3] = x[5]; q[0] = sqrt(1 - nrm);
            double tm[3][3];
            orc_quaternion2matrix(q, tm);
            float Rinc[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M3(Rinc, r, c) = (float)tm[r][c];
            float tinc[3] = {x[0], x[1], x[2]};
            mat3_vec(Rinc, tinv, tmpv);                              /* RG.cpp:886 */
            tinv[0] = tmpv[0] + tinc[0]; tinv[1] = tmpv[1] + tinc[1]; tinv[2] = tmpv[2] + tinc[2];
            mat3_mul(Rinc, Rinv, Rinv);                              /* RG.cpp:887 */
            if (iter > prm->max_iter[lvl] ||
                (frob_minus_I(Rinc) < prm->thres_cvg_alpha && norm3(tinc) < prm->thres_cvg_tr))
                cvg = 1;                                             /* RG.cpp:889 */
            iter++;
        }
        fact /= 2;
    }
    free(img); free(gx); free(gy);
    if (n_iters) *n_iters = it_total;
    if (!ok) return 0;
    if (frob_minus_I(Rcurr) > 0.3 || norm3(tcurr) > 0.3) return 0;  /* RG.cpp:910 (double literal) */
    orc_inverse3(Rinv, Rcurr);                                       /* RG.cpp:923 */
    { float neg[9]; for (int k = 0; k < 9; k++) neg[k] = -Rcurr[k]; mat3_vec(neg, tinv, tcurr); }
    memcpy(poseR, Rcurr, sizeof(Rcurr));
    memcpy(poset, tcurr, sizeof(tcurr));
    return 1;
}

/* ---- optional stage: the bilateral filter the reference carries commented out (Frame.cpp:50-54) -----------------------
 * cv::bilateralFilter on the depth converted to float, d = 5, sigmaColor = sigmaSpace = 100, converted back to u16.  OpenCV 2.4
 * is third-party and absent: restated from its published float algorithm (circular window, reflect-101 border, interpolated
 * 4096-bin exponential table for the range weights); the table spans the full u16 range, not the image's min..max. */
static int reflect101_o(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
    return p;
}
void orc_bilateral_depth(uint16_t *dst, const uint16_t *src, size_t pitch, int d, double sigma_color, double sigma_space, int n, int m) {
    if (sigma_color <= 0.0) sigma_color = 1.0;
    if (sigma_space <= 0.0) sigma_space = 1.0;
    int radius = d <= 0 ? (int)rint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    const double gcc = -0.5 / (sigma_color * sigma_color), gsc = -0.5 / (sigma_space * sigma_space);
    const int bins = 4096;
    const float len = 65535.0f, scale_index = (float)bins / len;
    float *lut = (float *)malloc(sizeof(float) * (bins + 2));
    for (int i = 0; i < bins + 2; i++) { double val = (double)i / (double)scale_index; lut[i] = (float)exp(val * val * gcc); }
    int cnt = 0, dy[81], dx[81];
    float sw[81];
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            dy[cnt] = i; dx[cnt] = j; sw[cnt++] = (float)exp(r * r * gsc);
        }
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            float val0 = (float)src[(size_t)i * pitch + j], sum = 0.0f, wsum = 0.0f;
            for (int k = 0; k < cnt; k++) {
                float val = (float)src[(size_t)reflect101_o(i + dy[k], n) * pitch + reflect101_o(j + dx[k], m)];
                float alpha = fabsf(val - val0) * scale_index;
                int idx = (int)floorf(alpha);
                alpha = alpha - (float)idx;
                float w = sw[k] * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]));
                sum = sum + val * w;
                wsum = wsum + w;
            }
            float o = rintf(sum / wsum);
            o = o < 0.0f ? 0.0f : (o > 65535.0f ? 65535.0f : o);
            dst[(size_t)i * pitch + j] = (uint16_t)o;
        }
    free(lut);
}

/* ===================================================================================== */
/* Stage 4: fusion                                                                       */
/* ===================================================================================== */

/* one pixel of MergeDepthOffProcess / MergeDepthKinectProcess (Kernel.cu:440-526, 528-611): the two differ only in
 * where the raw depth sample d comes from */
static void merge_one(float *px, uint16_t d, const float *rgb, uint8_t *mask, float ds, float factor) {
    float dv = (ds / factor) * (float)d;                                             /* Kernel.cu:453 / 541 */
    float dp = (float)((double)px[3] * MAX_DEPTH);                                   /* :454 */
    float mv = (float)((double)(float)*mask - 10.0);                                 /* :455 */
    float mt = 1.0f;
    px[0] = rgb[0]; px[1] = rgb[1]; px[2] = rgb[2];
    if ((double)dv < 0.1) return;                                                    /* :467 */
    if ((double)dp < 0.1 || fabs((double)mv + 9.0) < 0.01) {                         /* :477 */
        px[3] = (float)((double)dv / MAX_DEPTH);
        *mask = (uint8_t)((double)mt + 10.0);
        return;
    }
    if ((double)dv < ((double)dp - MAXTOLERANCE)) {                                  /* :489 */
        px[3] = (float)((double)dv / MAX_DEPTH);
        *mask = (uint8_t)((double)mt + 10.0);
        return;
    }
    double mx = 1.0 > 1.0 * (double)dv ? 1.0 : 1.0 * (double)dv;
    if ((double)dv > ((double)dp + mx * MAXTOLERANCE)) {                             /* :502 */
        *mask = 10;
        return;
    }
    float am = fabsf(mv);
    float avg = (dv * mt + am * dp) / (mt + am);                                     /* :512 float */
    px[3] = (float)((double)avg / MAX_DEPTH);
    double mk = (double)(mt + am) + 10.0;
    if (mk > 255.0) mk = 255.0;
    *mask = (uint8_t)mk;                                                             /* :520 */
}

void orc_merge_depth_off(float *RGBD, const uint16_t *depth, size_t pitch, const float *RGB,
                         uint8_t *mask, const float *calib, int n, int m) {
    float ds = calib[9], factor = calib[10];
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            merge_one(RGBD + 4 * idx, depth[(size_t)(n - i - 1) * pitch + j], RGB + 3 * idx, mask + idx, ds, factor);
        }
}

/* Kernel.cu:528-614 MergeDepthKinectKernel: live-Kinect depth layout, rows AND columns flipped, no pitch (:541) */
void orc_merge_depth_kinect(float *RGBD, const uint16_t *depth, const float *RGB, uint8_t *mask,
                            const float *calib, int n, int m) {
    float ds = calib[9], factor = calib[10];
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            merge_one(RGBD + 4 * idx, depth[(size_t)(n - 1 - i) * m + (m - 1 - j)], RGB + 3 * idx, mask + idx, ds, factor);
        }
}

void orc_vertex_map_gl(float *VMap, const float *RGBD, const float *calib, int n, int m) {
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            float w = RGBD[4 * idx + 3];
            double z = (double)w * MAX_DEPTH;
            VMap[3 * idx]     = (w == 0.0f) ? 0.0f : (float)(z * (double)(((float)j - calib[2]) / calib[0]));
            VMap[3 * idx + 1] = (w == 0.0f) ? 0.0f : (float)(z * (double)(((float)i - calib[3]) / calib[1]));
            VMap[3 * idx + 2] = (float)((double)(-w) * MAX_DEPTH);                       /* Kernel.cu:66 */
        }
}

/* claim key: stamp | kind(1 = writer) | mask_curr | ~idx(24 bit) */
static uint64_t claim_key(uint32_t stamp, int writer, uint8_t mask_curr, size_t idx) {
    uint32_t lo = ((uint32_t)(writer ? 1 : 0) << 31) | ((uint32_t)mask_curr << 23) |
                  (uint32_t)(0x7FFFFFu - (uint32_t)idx);
    return ((uint64_t)stamp << 32) | lo;
}

/* geometry part of D1/Primitive.cu:278-307; returns 0 when the pixel takes no part */
static int update_cell(const float *VMap, const uint8_t *Label, const uint16_t *Label16, const float *pose, float **ParamTab,
                       size_t idx, int N_prim, int M_prim, int plane_lo, int plane_hi,
                       int *lbl, int *cell, float *pa, float *pb, float *err) {
    int L = (Label16 ? (int)Label16[idx] : (int)Label[idx]) - 1;
    if (L < 0) return 0;
    if (L < plane_lo || L >= plane_hi) return 0;
    const float *P = ParamTab[L];
    float l0 = VMap[3 * idx], l1 = VMap[3 * idx + 1], l2 = VMap[3 * idx + 2];
    float p0 = ((pose[0] * l0 + pose[4] * l1) + pose[8] * l2) + pose[12];
    float p1 = ((pose[1] * l0 + pose[5] * l1) + pose[9] * l2) + pose[13];
    float p2 = ((pose[2] * l0 + pose[6] * l1) + pose[10] * l2) + pose[14];
    *err = ((p0 * P[0] + p1 * P[1]) + p2 * P[2]) - P[3];                        /* Primitive.cu:293 */
    float a = (p0 * P[8] + p1 * P[9]) + p2 * P[10];
    float b = (p0 * P[11] + p1 * P[12]) + p2 * P[13];
    a = (float)((double)((a - P[6]) * P[4]) / 2.0);                             /* :298 */
    b = (float)((double)((b - P[7]) * P[5]) / 2.0);
    int ii = f2i_rd(a), jj = f2i_rd(b);
    if (ii > N_prim - 1 || jj > M_prim - 1 || ii < 0 || jj < 0) return 0;
    *lbl = L; *cell = ii * M_prim + jj; *pa = a - (float)ii; *pb = b - (float)jj;
    return 1;
}

static void update_impl(uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                uint64_t **ClaimTab, uint32_t stamp,
                const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label, const uint16_t *Label16,
                const float *pose, int N_prim, int M_prim, int n, int m,
                int plane_lo, int plane_hi) {
    size_t P = (size_t)n * m;
    /* pass 1: claims against the snapshot of the cell masks */
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < P; idx++) {
        int L, cell; float sa, sb, err;
        if (!update_cell(VMap, Label, Label16, pose, ParamTab, idx, N_prim, M_prim, plane_lo, plane_hi, &L, &cell, &sa, &sb, &err)) continue;
        uint8_t mc = Mask[idx], mref = MaskTab[L][cell];
        uint64_t key;
        if (mc < 10 && mref != 1) key = claim_key(stamp, 0, 0, idx);            /* eroder, Primitive.cu:313 */
        else if (mc > mref)       key = claim_key(stamp, 1, mc, idx);           /* writer, :341 */
        else continue;
        ORC_MAX_U64(&ClaimTab[L][cell], key);
    }
    /* pass 2: the winner of each cell acts */
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < P; idx++) {
        int L, cell; float sa, sb, err;
        if (!update_cell(VMap, Label, Label16, pose, ParamTab, idx, N_prim, M_prim, plane_lo, plane_hi, &L, &cell, &sa, &sb, &err)) continue;
        uint8_t mc = Mask[idx], mref = MaskTab[L][cell];
        uint64_t key;
        int erode = (mc < 10 && mref != 1);
        if (erode) key = claim_key(stamp, 0, 0, idx);
        else if (mc > mref) key = claim_key(stamp, 1, mc, idx);
        else continue;
        if (ClaimTab[L][cell] != key) continue;
        if (erode) {
            if (mref > 11) MaskTab[L][cell] = (uint8_t)(mref - 1);              /* :315 */
            else {
                MaskTab[L][cell] = 10;
                BumpTab[L][3 * cell] = BumpTab[L][3 * cell + 1] = BumpTab[L][3 * cell + 2] = 0;
                RGBTab[L][3 * cell] = RGBTab[L][3 * cell + 1] = RGBTab[L][3 * cell + 2] = 0;
            }
        } else {
            BumpTab[L][3 * cell]     = (uint16_t)f2i_rn((float)((double)sa * 60000.0)); /* :342 */
            BumpTab[L][3 * cell + 1] = (uint16_t)f2i_rn((float)((double)sb * 60000.0));
            BumpTab[L][3 * cell + 2] = (uint16_t)f2i_rn((float)(((double)err + 15.0) * 2000.0));
            RGBTab[L][3 * cell]     = (uint8_t)f2i_rn((float)((double)RGBD[4 * idx] * 255.0));
            RGBTab[L][3 * cell + 1] = (uint8_t)f2i_rn((float)((double)RGBD[4 * idx + 1] * 255.0));
            RGBTab[L][3 * cell + 2] = (uint8_t)f2i_rn((float)((double)RGBD[4 * idx + 2] * 255.0));
            MaskTab[L][cell] = mc;
        }
    }
}

void orc_update(uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                uint64_t **ClaimTab, uint32_t stamp,
                const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label,
                const float *pose, int N_prim, int M_prim, int n, int m,
                int plane_lo, int plane_hi) {
    update_impl(BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, Label, NULL, pose, N_prim, M_prim, n, m, plane_lo, plane_hi);
}
/* the same with 16-bit labels (models of more than 255 planes; no reference counterpart: Frame.h:197 keeps u8 labels) */
void orc_update_l16(uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                    uint64_t **ClaimTab, uint32_t stamp,
                    const float *VMap, const float *RGBD, const uint8_t *Mask, const uint16_t *Label16,
                    const float *pose, int N_prim, int M_prim, int n, int m,
                    int plane_lo, int plane_hi) {
    update_impl(BumpTab, RGBTab, MaskTab, ParamTab, ClaimTab, stamp, VMap, RGBD, Mask, NULL, Label16, pose, N_prim, M_prim, n, m, plane_lo, plane_hi);
}

/* harness stand-in for plane detection on tiled models: first plane whose equation the pixel satisfies AND whose Bump
 * image contains its cell (geometry of Primitive.cu:293-307) */
void orc_label_planes_l16(uint16_t *Label16, const float *VMap, const float *NMap, const float *pose, float **ParamTab,
                          int nb, int N_prim, int M_prim, float epsilon, float alpha, int n, int m) {
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < (size_t)n * m; idx++) {
        Label16[idx] = 0;
        float n0 = NMap[3 * idx], n1 = NMap[3 * idx + 1], n2 = NMap[3 * idx + 2];
        if (n0 == 0.0f && n1 == 0.0f && n2 == 0.0f) continue;
        float l0 = VMap[3 * idx], l1 = VMap[3 * idx + 1], l2 = VMap[3 * idx + 2];
        float p0 = ((pose[0] * l0 + pose[4] * l1) + pose[8] * l2) + pose[12];
        float p1 = ((pose[1] * l0 + pose[5] * l1) + pose[9] * l2) + pose[13];
        float p2 = ((pose[2] * l0 + pose[6] * l1) + pose[10] * l2) + pose[14];
        float q0 = (pose[0] * n0 + pose[4] * n1) + pose[8] * n2;
        float q1 = (pose[1] * n0 + pose[5] * n1) + pose[9] * n2;
        float q2 = (pose[2] * n0 + pose[6] * n1) + pose[10] * n2;
        for (int k = 0; k < nb; k++) {
            const float *P = ParamTab[k];
            float ea = (q0 * P[0] + q1 * P[1]) + q2 * P[2];
            float err = ((p0 * P[0] + p1 * P[1]) + p2 * P[2]) - P[3];
            float a = (p0 * P[8] + p1 * P[9]) + p2 * P[10];
            float b = (p0 * P[11] + p1 * P[12]) + p2 * P[13];
            a = (float)((double)((a - P[6]) * P[4]) / 2.0);
            b = (float)((double)((b - P[7]) * P[5]) / 2.0);
            int ii = f2i_rd(a), jj = f2i_rd(b);
            int inside = !(ii > N_prim - 1 || jj > M_prim - 1 || ii < 0 || jj < 0);
            if (fabsf(err) > epsilon || ea < alpha || !inside) continue;
            Label16[idx] = (uint16_t)(k + 1);
            break;
        }
    }
}

void orc_vertex_from_bump(float *VMap, float *RGBA, float *MaskOut, const uint16_t *Bump,
                          const uint8_t *RGB_bump, const uint8_t *Mask, const float *param,
                          int n, int m, int lvl) {
    int no = n / lvl, mo = m / lvl;   /* bound fixed to the output size (reference over-runs for lvl>1) */
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < no; i++)
        for (int j = 0; j < mo; j++) {
            size_t idx = (size_t)i * mo + j;
            size_t ref = (size_t)(lvl * i) * m + (size_t)(lvl * j);
            if (Mask[ref] < 11 && Mask[ref] != 1) {                              /* Primitive.cu:47 */
                for (int k = 0; k < 3; k++) VMap[3 * idx + k] = 0.0f;
                for (int k = 0; k < 4; k++) { RGBA[4 * idx + k] = 0.0f; MaskOut[4 * idx + k] = 0.0f; }
                continue;
            }
            float fi = (float)(lvl * i), fj = (float)(lvl * j);
            float s0 = (float)((double)(float)Bump[3 * ref] / 60000.0);
            float s1 = (float)((double)(float)Bump[3 * ref + 1] / 60000.0);
            float x = (float)((double)(fi + s0) * 2.0 / (double)param[4] + (double)param[6]);  /* :70 */
            float y = (float)((double)(fj + s1) * 2.0 / (double)param[5] + (double)param[7]);
            float d = (float)((double)param[3] + (((double)(float)Bump[3 * ref + 2] / 2000.0) - 15.0));
            float v0 = (x * param[8] + y * param[11]) + d * param[0];
            float v1 = (x * param[9] + y * param[12]) + d * param[1];
            float v2 = (x * param[10] + y * param[13]) + d * param[2];
            VMap[3 * idx] = v0; VMap[3 * idx + 1] = v1; VMap[3 * idx + 2] = v2;
            RGBA[4 * idx]     = (float)((double)(float)RGB_bump[3 * ref] / 255.0);
            RGBA[4 * idx + 1] = (float)((double)(float)RGB_bump[3 * ref + 1] / 255.0);
            RGBA[4 * idx + 2] = (float)((double)(float)RGB_bump[3 * ref + 2] / 255.0);
            RGBA[4 * idx + 3] = 1.0f;
            MaskOut[4 * idx]     = (float)((10.0 + (double)v0) / 20.0);
            MaskOut[4 * idx + 1] = (float)((10.0 + (double)v1) / 20.0);
            MaskOut[4 * idx + 2] = (float)((10.0 + (double)v2) / 20.0);
            MaskOut[4 * idx + 3] = (float)((double)(float)Mask[ref] / 255.0);
        }
}

/* DemoStep2's decode kernels (D2/Primitive.cu:31-179; variant 0 = VertexFromBump / VertexOnlyFromBump with a 16-bit mask,
 * variant 1 = VertexRGBFromBump with an 8-bit mask, same-index input and output).  Output size is the bound (see the D1 form). */
void orc_vertex_from_bump_d2(float *VMap, float *RGBA, const uint16_t *Bump, const uint8_t *RGB_bump, const void *Mask,
                             const float *param, int n, int m, int lvl, int variant) {
    int no = variant ? n : n / lvl, mo = variant ? m : m / lvl;
    for (int i = 0; i < no; i++)
        for (int j = 0; j < mo; j++) {
            size_t idx = (size_t)i * mo + j;
            size_t ref = variant ? idx : (size_t)(lvl * i) * m + (size_t)(lvl * j);
            unsigned mk = variant ? ((const uint8_t *)Mask)[ref] : ((const uint16_t *)Mask)[ref];
            if (mk < 11) {                                                       /* D2/Primitive.cu:47,140 */
                for (int k = 0; k < 3; k++) VMap[3 * idx + k] = 0.0f;
                if (variant) for (int k = 0; k < 4; k++) RGBA[4 * idx + k] = 0.0f;
                continue;
            }
            float fi = (float)(lvl * i), fj = (float)(lvl * j);                  /* :55-56 */
            float s0 = (float)((double)(float)Bump[3 * ref] / 60000.0);          /* :59-60 */
            float s1 = (float)((double)(float)Bump[3 * ref + 1] / 60000.0);
            float x = (float)((double)(fi + s0) * 2.0 / (double)param[4] + (double)param[6]);   /* :62-63 */
            float y = (float)((double)(fj + s1) * 2.0 / (double)param[5] + (double)param[7]);
            float d = (float)((double)param[3] + (((double)(float)Bump[3 * ref + 2] / 2000.0) - 15.0));   /* :65 */
            VMap[3 * idx] = (x * param[8] + y * param[11]) + d * param[0];       /* :67-69 */
            VMap[3 * idx + 1] = (x * param[9] + y * param[12]) + d * param[1];
            VMap[3 * idx + 2] = (x * param[10] + y * param[13]) + d * param[2];
            if (variant) {                                                       /* :170-173 */
                for (int k = 0; k < 3; k++) RGBA[4 * idx + k] = (float)((double)(float)RGB_bump[3 * ref + k] / 255.0);
                RGBA[4 * idx + 3] = 1.0f;
            }
        }
}

/* ===================================================================================== */
/* Stage 5: prediction                                                                   */
/* ===================================================================================== */

void orc_transform_vertex_map(const float *pose, float *V, float *RGBD, int n, int m) {
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < (size_t)n * m; idx++) {
        float a = V[3 * idx], b = V[3 * idx + 1], c = V[3 * idx + 2];
        if (a == 0.0f && b == 0.0f && c == 0.0f) {
            V[3 * idx] = 0.0f; V[3 * idx + 1] = 0.0f; V[3 * idx + 2] = 0.0f;
            RGBD[4 * idx + 3] = 0.0f;
            continue;
        }
        float x = ((pose[0] * a + pose[4] * b) + pose[8] * c) + pose[12];
        float y = ((pose[1] * a + pose[5] * b) + pose[9] * c) + pose[13];
        float z = ((pose[2] * a + pose[6] * b) + pose[10] * c) + pose[14];
        V[3 * idx] = x; V[3 * idx + 1] = y; V[3 * idx + 2] = z;
        RGBD[4 * idx + 3] = ((double)(-z) > MAX_DEPTH || (double)(-z) < 0.0) ? 0.0f
                                                                            : (float)((double)(-z) / MAX_DEPTH);
    }
}

static float dist3(const float *a, const float *b) {
    float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return sqrtf((x * x + y * y) + z * z);
}

/* D1/Kernel.cu:398-421: 1 when the quad must be dropped */
static int quad_trimmed(const float *VMap, const uint32_t *id, float thresh) {
    const float *a = VMap + 3 * (size_t)id[0], *b = VMap + 3 * (size_t)id[1];
    const float *c = VMap + 3 * (size_t)id[2], *d = VMap + 3 * (size_t)id[3];
    float mx = fmaxf(dist3(a, b), dist3(a, c));
    mx = fmaxf(dist3(a, d), mx);
    mx = fmaxf(dist3(b, c), mx);
    mx = fmaxf(dist3(b, d), mx);
    mx = fmaxf(dist3(c, d), mx);
    return mx > thresh;
}

static void quad_ids(int i, int j, int m, uint32_t *id) {
    id[0] = (uint32_t)(i * m + j); id[1] = (uint32_t)(i * m + j + 1);
    id[2] = (uint32_t)((i + 1) * m + j + 1); id[3] = (uint32_t)((i + 1) * m + j);
}

void orc_quad_trim_frame(uint32_t *index, const float *VMap, int n, int m, float thresh) {
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n - 1; i++)
        for (int j = 0; j < m - 1; j++) {
            size_t q = (size_t)i * (m - 1) + j;
            uint32_t id[4];
            quad_ids(i, j, m, id);
            if (quad_trimmed(VMap, id, thresh)) { index[4 * q] = index[4 * q + 1] = index[4 * q + 2] = index[4 * q + 3] = 0; }
            else { index[4 * q] = id[0]; index[4 * q + 1] = id[1]; index[4 * q + 2] = id[2]; index[4 * q + 3] = id[3]; }
        }
}

/* D1/Primitive.cu:95-147 (variant 0) and D2/Primitive.cu:367-419 (variant 1): quad list of a plane mesh.
 * Vertex ids are formed as i*m/lvl + j in integer arithmetic and Mask is indexed with the same ids. */
static int corner_rejected(uint8_t mk, int min_conf, int variant) {
    return variant ? (mk < 11) : (mk != 1 && (int)mk - 10 < min_conf);
}
void orc_quad_trim(uint32_t *index, const float *VMap, const uint8_t *Mask, unsigned short min_conf,
                   unsigned n, unsigned m, float thresh, int lvl, int variant) {
    unsigned L = (unsigned)lvl;
    for (unsigned i = 0; i + 2 <= n / L; i++)
        for (unsigned j = 0; j + 2 <= m / L; j++) {
            size_t q = (size_t)i * (m / L - 1) + j;
            uint32_t id[4];
            id[0] = i * m / L + j; id[1] = i * m / L + j + 1;
            id[2] = (i + 1) * m / L + j + 1; id[3] = (i + 1) * m / L + j;
            int cut = quad_trimmed(VMap, id, thresh);
            for (int k = 0; k < 4; k++) cut |= corner_rejected(Mask[id[k]], (int)min_conf, variant);
            if (cut) { index[4 * q] = index[4 * q + 1] = index[4 * q + 2] = index[4 * q + 3] = 0; }
            else if (variant) { index[4 * q] = id[0]; index[4 * q + 1] = id[3]; index[4 * q + 2] = id[2]; index[4 * q + 3] = id[1]; }
            else { index[4 * q] = id[0]; index[4 * q + 1] = id[1]; index[4 * q + 2] = id[2]; index[4 * q + 3] = id[3]; }
        }
}

/* ---- compute rasteriser: fixed-point 1/256 px snapping, integer edge functions,
 *      top-left rule, perspective-correct interpolation in double, u64 min z-buffer ---------- */
#define RASTER_MIN_DEPTH 1e-3f
#define RASTER_GUARD 16384.0f
#define ZBUF_EMPTY 0xFFFFFFFFFFFFFFFFull

typedef struct { int32_t X[3], Y[3]; double inv_d[3]; int64_t area; } ras_tri;

static int ras_setup(const float *VMap, const uint32_t *vid, const float *calib, ras_tri *T, int *order) {
    for (int k = 0; k < 3; k++) {
        const float *v = VMap + 3 * (size_t)vid[k];
        float dk = -v[2];
        if (!(dk >= RASTER_MIN_DEPTH)) return 0;
        float u = (v[0] / dk) * calib[0] + calib[2];
        float w = (v[1] / dk) * calib[1] + calib[3];
        if (!(fabsf(u) <= RASTER_GUARD) || !(fabsf(w) <= RASTER_GUARD)) return 0;
        T->X[k] = f2i_rn(u * 256.0f);
        T->Y[k] = f2i_rn(w * 256.0f);
        T->inv_d[k] = 1.0 / (double)dk;
        order[k] = k;
    }
    int64_t area = (int64_t)(T->X[1] - T->X[0]) * (T->Y[2] - T->Y[0]) - (int64_t)(T->Y[1] - T->Y[0]) * (T->X[2] - T->X[0]);
    if (area == 0) return 0;
    if (area < 0) {
        int32_t t;
        t = T->X[1]; T->X[1] = T->X[2]; T->X[2] = t;
        t = T->Y[1]; T->Y[1] = T->Y[2]; T->Y[2] = t;
        double dd = T->inv_d[1]; T->inv_d[1] = T->inv_d[2]; T->inv_d[2] = dd;
        order[1] = 2; order[2] = 1;
        area = -area;
    }
    T->area = area;
    return 1;
}

static int64_t ras_edge(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py) {
    return (int64_t)(bx - ax) * (py - ay) - (int64_t)(by - ay) * (px - ax);
}
static int ras_topleft(int32_t ax, int32_t ay, int32_t bx, int32_t by) {
    int32_t dx = bx - ax, dy = by - ay;
    return (dy < 0) || (dy == 0 && dx < 0);
}
/* edge weights at pixel (px,py); returns 1 when covered */
static int ras_cover(const ras_tri *T, int px, int py, int64_t *w) {
    int32_t PX = px * 256, PY = py * 256;
    w[0] = ras_edge(T->X[1], T->Y[1], T->X[2], T->Y[2], PX, PY);
    w[1] = ras_edge(T->X[2], T->Y[2], T->X[0], T->Y[0], PX, PY);
    w[2] = ras_edge(T->X[0], T->Y[0], T->X[1], T->Y[1], PX, PY);
    if (w[0] < 0 || w[1] < 0 || w[2] < 0) return 0;
    if (w[0] == 0 && !ras_topleft(T->X[1], T->Y[1], T->X[2], T->Y[2])) return 0;
    if (w[1] == 0 && !ras_topleft(T->X[2], T->Y[2], T->X[0], T->Y[0])) return 0;
    if (w[2] == 0 && !ras_topleft(T->X[0], T->Y[0], T->X[1], T->Y[1])) return 0;
    return 1;
}
static int floor_div256(int32_t v) { return (int)(v >> 8); }           /* arithmetic shift = floor */
static int ceil_div256(int32_t v) { return (int)((v + 255) >> 8); }

static void ras_draw(uint64_t *zbuf, const float *VMap, const uint32_t *vid, uint32_t tri_id,
                     const float *calib, float znear, float zfar, int n, int m) {
    ras_tri T; int order[3];
    if (!ras_setup(VMap, vid, calib, &T, order)) return;
    int32_t xmn = T.X[0], xmx = T.X[0], ymn = T.Y[0], ymx = T.Y[0];
    for (int k = 1; k < 3; k++) {
        if (T.X[k] < xmn) xmn = T.X[k]; if (T.X[k] > xmx) xmx = T.X[k];
        if (T.Y[k] < ymn) ymn = T.Y[k]; if (T.Y[k] > ymx) ymx = T.Y[k];
    }
    int x0 = ceil_div256(xmn), x1 = floor_div256(xmx), y0 = ceil_div256(ymn), y1 = floor_div256(ymx);
    if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0;
    if (x1 > m - 1) x1 = m - 1; if (y1 > n - 1) y1 = n - 1;
    for (int py = y0; py <= y1; py++)
        for (int px = x0; px <= x1; px++) {
            int64_t w[3];
            if (!ras_cover(&T, px, py, w)) continue;
            double W = ((double)w[0] * T.inv_d[0] + (double)w[1] * T.inv_d[1]) + (double)w[2] * T.inv_d[2];
            float dpx = (float)((double)T.area / W);
            if (!(dpx >= znear) || !(dpx <= zfar)) continue;
            uint32_t bits; memcpy(&bits, &dpx, 4);
            uint64_t key = ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - tri_id);
            size_t p = (size_t)py * m + px;
            ORC_MIN_U64(&zbuf[p], key);
        }
}

static void tri_vertex_ids(uint32_t tri_id, const uint32_t *index, int m, uint32_t *vid) {
    uint32_t q = tri_id >> 1, id[4];
    if (index) { id[0] = index[4 * (size_t)q]; id[1] = index[4 * (size_t)q + 1]; id[2] = index[4 * (size_t)q + 2]; id[3] = index[4 * (size_t)q + 3]; }
    else quad_ids((int)(q / (uint32_t)(m - 1)), (int)(q % (uint32_t)(m - 1)), m, id);
    if (tri_id & 1) { vid[0] = id[0]; vid[1] = id[2]; vid[2] = id[3]; }
    else            { vid[0] = id[0]; vid[1] = id[1]; vid[2] = id[2]; }
}

void orc_raster(float *tex, uint64_t *zbuf, const float *VMap, const float *RGBD,
                const uint32_t *index, float thresh, const float *calib,
                float znear, float zfar, int n, int m) {
    size_t P = (size_t)n * m;
    #pragma omp parallel for schedule(static)
    for (size_t p = 0; p < P; p++) zbuf[p] = ZBUF_EMPTY;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n - 1; i++)
        for (int j = 0; j < m - 1; j++) {
            uint32_t q = (uint32_t)(i * (m - 1) + j), id[4];
            if (index) { for (int k = 0; k < 4; k++) id[k] = index[4 * (size_t)q + k]; }
            else { quad_ids(i, j, m, id); if (quad_trimmed(VMap, id, thresh)) continue; }
            uint32_t t0[3] = {id[0], id[1], id[2]}, t1[3] = {id[0], id[2], id[3]};
            ras_draw(zbuf, VMap, t0, 2 * q, calib, znear, zfar, n, m);
            ras_draw(zbuf, VMap, t1, 2 * q + 1, calib, znear, zfar, n, m);
        }
    #pragma omp parallel for schedule(static)
    for (int py = 0; py < n; py++)
        for (int px = 0; px < m; px++) {
            size_t p = (size_t)py * m + px;
            uint64_t key = zbuf[p];
            if (key == ZBUF_EMPTY) { tex[4 * p] = tex[4 * p + 1] = tex[4 * p + 2] = tex[4 * p + 3] = 0.0f; continue; }
            uint32_t tri_id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu), vid[3];
            tri_vertex_ids(tri_id, index, m, vid);
            ras_tri T; int order[3]; int64_t w[3];
            ras_setup(VMap, vid, calib, &T, order);
            ras_cover(&T, px, py, w);
            double b0 = (double)w[0] * T.inv_d[0], b1 = (double)w[1] * T.inv_d[1], b2 = (double)w[2] * T.inv_d[2];
            double W = (b0 + b1) + b2;
            for (int c = 0; c < 4; c++) {
                double a0 = (double)RGBD[4 * (size_t)vid[order[0]] + c];
                double a1 = (double)RGBD[4 * (size_t)vid[order[1]] + c];
                double a2 = (double)RGBD[4 * (size_t)vid[order[2]] + c];
                tex[4 * p + c] = (float)(((b0 * a0 + b1 * a1) + b2 * a2) / W);
            }
        }
}

/* ---- prediction from the planes (no D1 call site: the Bump meshes -- VertexFromBump,
 * Primitive.cu:31-93, + QuadTrim, Primitive.cu:95-152 -- drawn by the same rasteriser rules as the frame mesh) ------------
 * decode of ONE cell: the arithmetic of orc_vertex_from_bump at lvl 1 */
static void decode_one(const uint16_t *Bump, const uint8_t *RGBb, const uint8_t *Mask, const float *param, int M, int i, int j,
                       float *v, float *rgb) {
    size_t ref = (size_t)i * M + j;
    if (Mask[ref] < 11 && Mask[ref] != 1) { v[0] = v[1] = v[2] = 0.0f; rgb[0] = rgb[1] = rgb[2] = 0.0f; return; }
    float s0 = (float)((double)(float)Bump[3 * ref] / 60000.0), s1 = (float)((double)(float)Bump[3 * ref + 1] / 60000.0);
    float x = (float)((double)((float)i + s0) * 2.0 / (double)param[4] + (double)param[6]);
    float y = (float)((double)((float)j + s1) * 2.0 / (double)param[5] + (double)param[7]);
    float d = (float)((double)param[3] + (((double)(float)Bump[3 * ref + 2] / 2000.0) - 15.0));
    v[0] = (x * param[8] + y * param[11]) + d * param[0];
    v[1] = (x * param[9] + y * param[12]) + d * param[1];
    v[2] = (x * param[10] + y * param[13]) + d * param[2];
    for (int c = 0; c < 3; c++) rgb[c] = (float)((double)(float)RGBb[3 * ref + c] / 255.0);
}
static void xf16(const float *P, const float *a, float *o) {
    o[0] = ((P[0] * a[0] + P[4] * a[1]) + P[8] * a[2]) + P[12];
    o[1] = ((P[1] * a[0] + P[5] * a[1]) + P[9] * a[2]) + P[13];
    o[2] = ((P[2] * a[0] + P[6] * a[1]) + P[10] * a[2]) + P[14];
}
/* the four corner cells of quad (i, j): (i,j) (i,j+1) (i+1,j+1) (i+1,j), the order of QuadTrim's record */
static const int QC_I[4] = {0, 0, 1, 1}, QC_J[4] = {0, 1, 1, 0};

/* z-buffer pass over planes [plane_lo, plane_hi); zbuf is NOT cleared (callers start from ZBUF_EMPTY and may draw plane
 * ranges one after the other: the minimum is commutative).  pose = world -> camera, column-major 4x4.
 * triangle id = ((plane * (N-1)*(M-1) + quad) * 2 + t), t = 0: corners (0,1,2), t = 1: corners (0,2,3). */
void orc_planes_draw(uint64_t *zbuf, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, int plane_lo,
                     int plane_hi, const float *pose, int N, int M, int min_conf, float thresh, const float *calib, float znear,
                     float zfar, int n, int m) {
    const uint32_t Q = (uint32_t)(N - 1) * (uint32_t)(M - 1);
    for (int k = plane_lo; k < plane_hi; k++) {
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < N - 1; i++)
            for (int j = 0; j < M - 1; j++) {
                float vw[12], vc[12], rgb[3];
                int cut = 0;
                for (int c = 0; c < 4; c++) {
                    int ci = i + QC_I[c], cj = j + QC_J[c];
                    cut |= corner_rejected(MaskTab[k][(size_t)ci * M + cj], min_conf, 0);
                    decode_one(BumpTab[k], RGBTab[k], MaskTab[k], ParamTab[k], M, ci, cj, vw + 3 * c, rgb);
                }
                uint32_t id4[4] = {0, 1, 2, 3};
                cut |= quad_trimmed(vw, id4, thresh);
                if (cut) continue;
                for (int c = 0; c < 4; c++) xf16(pose, vw + 3 * c, vc + 3 * c);
                uint32_t q = (uint32_t)i * (uint32_t)(M - 1) + (uint32_t)j;
                uint32_t base = ((uint32_t)k * Q + q) * 2u;
                uint32_t t0[3] = {0, 1, 2}, t1[3] = {0, 2, 3};
                ras_draw(zbuf, vc, t0, base, calib, znear, zfar, n, m);
                ras_draw(zbuf, vc, t1, base + 1, calib, znear, zfar, n, m);
            }
    }
}

/* resolve + the ReadKernel arithmetic (Kernel.cu:309-343): pixels whose winning triangle belongs to a plane in
 * [plane_lo, plane_hi) get VMap / RGBD (r, g, b, depth/100), empty pixels get zeros, other pixels are left alone */
void orc_planes_resolve(const uint64_t *zbuf, float *VMap, float *RGBD, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab,
                        float **ParamTab, int plane_lo, int plane_hi, const float *pose, int N, int M, const float *calib, int n, int m) {
    const uint32_t Q = (uint32_t)(N - 1) * (uint32_t)(M - 1);
    #pragma omp parallel for schedule(static)
    for (int py = 0; py < n; py++)
        for (int px = 0; px < m; px++) {
            size_t p = (size_t)py * m + px;
            uint64_t key = zbuf[p];
            if (key == ZBUF_EMPTY) {                                              /* written by the owner of plane 0 only */
                if (plane_lo == 0) {
                    VMap[3 * p] = VMap[3 * p + 1] = VMap[3 * p + 2] = 0.0f;
                    RGBD[4 * p] = RGBD[4 * p + 1] = RGBD[4 * p + 2] = RGBD[4 * p + 3] = 0.0f;
                }
                continue;
            }
            uint32_t tri_id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
            uint32_t k = (tri_id >> 1) / Q, q = (tri_id >> 1) % Q;
            if ((int)k < plane_lo || (int)k >= plane_hi) continue;
            int i = (int)(q / (uint32_t)(M - 1)), j = (int)(q % (uint32_t)(M - 1));
            const int c3[2][3] = {{0, 1, 2}, {0, 2, 3}};
            float vc[9], att[3][4];
            for (int e = 0; e < 3; e++) {
                int c = c3[tri_id & 1][e];
                float vw[3];
                decode_one(BumpTab[k], RGBTab[k], MaskTab[k], ParamTab[k], M, i + QC_I[c], j + QC_J[c], vw, att[e]);
                xf16(pose, vw, vc + 3 * e);
                float z = vc[3 * e + 2];
                att[e][3] = ((double)(-z) > MAX_DEPTH || (double)(-z) < 0.0) ? 0.0f : (float)((double)(-z) / MAX_DEPTH);   /* Kernel.cu:371 */
            }
            uint32_t vid[3] = {0, 1, 2};
            ras_tri T; int order[3]; int64_t w[3];
            ras_setup(vc, vid, calib, &T, order);
            ras_cover(&T, px, py, w);
            double b0 = (double)w[0] * T.inv_d[0], b1 = (double)w[1] * T.inv_d[1], b2 = (double)w[2] * T.inv_d[2];
            double W = (b0 + b1) + b2;
            float tx[4];
            for (int c = 0; c < 4; c++)
                tx[c] = (float)(((b0 * (double)att[order[0]][c] + b1 * (double)att[order[1]][c]) + b2 * (double)att[order[2]][c]) / W);
            if (tx[3] == 0.0f) {
                VMap[3 * p] = VMap[3 * p + 1] = VMap[3 * p + 2] = 0.0f;
                RGBD[4 * p] = RGBD[4 * p + 1] = RGBD[4 * p + 2] = RGBD[4 * p + 3] = 0.0f;
                continue;
            }
            float z = (float)((double)tx[3] * MAX_DEPTH);
            VMap[3 * p]     = z * (((float)px - calib[2]) / calib[0]);
            VMap[3 * p + 1] = z * (((float)py - calib[3]) / calib[1]);
            VMap[3 * p + 2] = -z;
            for (int c = 0; c < 4; c++) RGBD[4 * p + c] = tx[c];
        }
}

void orc_read_frame(float *VMap, float *RGBD, const float *tex, const float *calib, int n, int m) {
    memset(RGBD, 0, sizeof(float) * 4 * (size_t)n * m);                          /* Kernel.cu:900 */
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            const float *tx = tex + 4 * idx;
            if (tx[3] == 0.0f) {
                VMap[3 * idx] = VMap[3 * idx + 1] = VMap[3 * idx + 2] = 0.0f;
                RGBD[4 * idx + 3] = 0.0f;
                continue;
            }
            float z = (float)((double)tx[3] * MAX_DEPTH);
            VMap[3 * idx]     = z * (((float)j - calib[2]) / calib[0]);
            VMap[3 * idx + 1] = z * (((float)i - calib[3]) / calib[1]);
            VMap[3 * idx + 2] = -z;
            RGBD[4 * idx] = tx[0]; RGBD[4 * idx + 1] = tx[1]; RGBD[4 * idx + 2] = tx[2]; RGBD[4 * idx + 3] = tx[3];
        }
}

/* label images: unsigned char in the reference (Frame.h:197); `wide` = the build's 16-bit extension for models of more than
 * 255 planes (same statements, wider element) */
#define LGET(p, wide, i) ((wide) ? (int)((const uint16_t *)(p))[i] : (int)((const uint8_t *)(p))[i])
#define LSET(p, wide, i, v) do { if (wide) ((uint16_t *)(p))[i] = (uint16_t)(v); else ((uint8_t *)(p))[i] = (uint8_t)(v); } while (0)

static void project_mask_impl(const float *pose, const float *calib, const void *mask,
                              void *mask_swap, int wide, const float *V, int n, int m) {
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < (size_t)n * m; idx++) {
        float a = V[3 * idx], b = V[3 * idx + 1], c = V[3 * idx + 2];
        if (a == 0.0f && b == 0.0f && c == 0.0f) { LSET(mask_swap, wide, idx, 0); continue; }
        float x = ((pose[0] * a + pose[4] * b) + pose[8] * c) + pose[12];
        float y = ((pose[1] * a + pose[5] * b) + pose[9] * c) + pose[13];
        float z = ((pose[2] * a + pose[6] * b) + pose[10] * c) + pose[14];
        int pu = f2i_rn((x / fabsf(z)) * calib[0] + calib[2]);                    /* Kernel.cu:653 */
        int pv = f2i_rn((y / fabsf(z)) * calib[1] + calib[3]);
        if (pv > n - 1 || pu > m - 1 || pv < 0 || pu < 0) { LSET(mask_swap, wide, idx, 0); continue; }
        LSET(mask_swap, wide, idx, LGET(mask, wide, (size_t)pv * m + pu));
    }
}
void orc_project_mask(const float *pose, const float *calib, const uint8_t *mask,
                      uint8_t *mask_swap, const float *V, int n, int m) {
    project_mask_impl(pose, calib, mask, mask_swap, 0, V, n, m);
}
void orc_project_label_l16(const float *pose, const float *calib, const uint16_t *label,
                           uint16_t *label_swap, const float *V, int n, int m) {
    project_mask_impl(pose, calib, label, label_swap, 1, V, n, m);
}

static void segment_pass(const void *in, void *out, int wide, const float *VMap, const float *NMap,
                         const float *pose, const float *Eq, float epsilon, float alpha, int n, int m) {
    float thresh_depth = 0.003f;                                                 /* RG.cu:269 */
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            LSET(out, wide, idx, LGET(in, wide, idx));
            if (LGET(in, wide, idx) > 0) continue;
            int lb = i - 1 < 0 ? 0 : i - 1, ub = i + 2 > n ? n : i + 2;
            int lr = j - 1 < 0 ? 0 : j - 1, ur = j + 2 > m ? m : j + 2;
            float depth = VMap[3 * idx + 2];
            float l0 = VMap[3 * idx], l1 = VMap[3 * idx + 1], l2 = VMap[3 * idx + 2];
            float p0 = ((pose[0] * l0 + pose[4] * l1) + pose[8] * l2) + pose[12];
            float p1 = ((pose[1] * l0 + pose[5] * l1) + pose[9] * l2) + pose[13];
            float p2 = ((pose[2] * l0 + pose[6] * l1) + pose[10] * l2) + pose[14];
            float n0 = NMap[3 * idx], n1 = NMap[3 * idx + 1], n2 = NMap[3 * idx + 2];
            float q0 = (pose[0] * n0 + pose[4] * n1) + pose[8] * n2;
            float q1 = (pose[1] * n0 + pose[5] * n1) + pose[9] * n2;
            float q2 = (pose[2] * n0 + pose[6] * n1) + pose[10] * n2;
            if (q0 == 0.0f && q1 == 0.0f && q2 == 0.0f) continue;
            int done = 0;
            for (int ki = lb; ki < ub && !done; ki++)
                for (int kj = lr; kj < ur; kj++) {
                    size_t nb = (size_t)ki * m + kj;
                    if (LGET(in, wide, nb) > 0 && fabsf(VMap[3 * nb + 2] - depth) < thresh_depth) {
                        const float *E = Eq + 10 * (LGET(in, wide, nb) - 1);
                        float ed = (((p0) * E[0] + (p1) * E[1]) + (p2) * E[2]) - E[9];
                        float ea = (q0 * E[0] + q1 * E[1]) + q2 * E[2];
                        if (fabsf(ed) > epsilon || ea < alpha) continue;
                        LSET(out, wide, idx, LGET(in, wide, nb));
                        done = 1;
                        break;
                    }
                }
        }
}

void orc_segment(uint8_t *Label, uint8_t *swap, const float *VMap, const float *NMap,
                 const float *pose, const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    (void)nbPlans;
    segment_pass(Label, swap, 0, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
    segment_pass(swap, Label, 0, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
    segment_pass(Label, swap, 0, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
    memcpy(Label, swap, (size_t)n * m);
}
void orc_segment_l16(uint16_t *Label, uint16_t *swap, const float *VMap, const float *NMap,
                     const float *pose, const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    (void)nbPlans;
    segment_pass(Label, swap, 1, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
    segment_pass(swap, Label, 1, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
    segment_pass(Label, swap, 1, VMap, NMap, pose, Eq, epsilon, alpha, n, m);
    memcpy(Label, swap, 2 * (size_t)n * m);
}

void orc_add_zbuff_off(const uint16_t *depth, size_t pitch, const float *rgb_in, float *RGBD,
                       const float *calib, int n, int m) {
    float ds = calib[9], factor = calib[10];
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            float z = (ds / factor) * (float)depth[(size_t)(n - i - 1) * pitch + j];
            RGBD[4 * idx] = rgb_in[3 * idx]; RGBD[4 * idx + 1] = rgb_in[3 * idx + 1]; RGBD[4 * idx + 2] = rgb_in[3 * idx + 2];
            RGBD[4 * idx + 3] = ((double)z > MAX_DEPTH || (double)z < 0.0) ? 0.0f : (float)((double)z / MAX_DEPTH);
        }
}

/* Kernel.cu:804-823 AddZBUFFKinectKernel: depth[idx], neither flipped nor pitched */
void orc_add_zbuff_kinect(const uint16_t *depth, const float *rgb_in, float *RGBD, const float *calib, int n, int m) {
    float ds = calib[9], factor = calib[10];
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < (size_t)n * m; idx++) {
        float z = (ds / factor) * (float)depth[idx];
        RGBD[4 * idx] = rgb_in[3 * idx]; RGBD[4 * idx + 1] = rgb_in[3 * idx + 1]; RGBD[4 * idx + 2] = rgb_in[3 * idx + 2];
        RGBD[4 * idx + 3] = ((double)z > MAX_DEPTH || (double)z < 0.0) ? 0.0f : (float)((double)z / MAX_DEPTH);
    }
}

/* Kernel.cu:670-682 CpyKernel: grey image into a pitched float matrix (pitch in floats) */
void orc_gpu_cpy(float *img, size_t pitch, const float *rgb, int n, int m) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            img[(size_t)i * pitch + j] = (float)((double)((rgb[3 * idx + 2] + rgb[3 * idx + 1]) + rgb[3 * idx]) / 3.0);
        }
}

/* Kernel.cu:74-121 MapData2VBOKernel (live Kinect frame -> VMap / RGB, both flipped; LONG is 32 bit on the reference's
 * platform).  VMap and RGB must be zeroed by the caller (Frame.cpp:95-97). */
void orc_map_data_2_vbo(const uint8_t *color_bgra, const uint16_t *depth, const int32_t *color_coord, float *VMap,
                        float *RGB, const float *calib, int n, int m) {
    float fx = calib[0], fy = calib[1], cx = calib[2], cy = calib[3];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j, idx_new = (size_t)(n - 1 - i) * m + (m - 1 - j);
            float z = ((float)(-calib[9]) * (float)depth[idx]) / calib[10];         /* :95 */
            float x = (((float)j - cx) * z) / fx;                                   /* :96 */
            float y = (((float)i - cy) * z) / fy;
            int32_t u = color_coord[2 * idx], v = color_coord[2 * idx + 1];
            if (u >= 0 && u < m && v >= 0 && v < n && (double)z > -MAX_DEPTH) {     /* :104 */
                size_t ci = (size_t)v * ((size_t)m * 4) + (size_t)u * 4;
                VMap[3 * idx_new] = x; VMap[3 * idx_new + 1] = y; VMap[3 * idx_new + 2] = z;
                RGB[3 * idx_new]     = (float)((double)(float)color_bgra[ci + 2] / 255.0);
                RGB[3 * idx_new + 1] = (float)((double)(float)color_bgra[ci + 1] / 255.0);
                RGB[3 * idx_new + 2] = (float)((double)(float)color_bgra[ci] / 255.0);
            }
        }
}

/* RG.cu:508-537 VotePlanKernel.  The reference adds with unordered float atomics, so its sums are scheduler dependent;
 * defined order here (and in the HIP kernel): per label, a row's contributions are added left to right starting from 0,
 * and the row sums are added top to bottom starting from 0.  count is exact either way. */
void orc_vote_plan(const uint8_t *Label, const float *VMap, float *centers, float *count, const float *pose,
                   int nbPlans, int n, int m) {
    for (int L = 0; L < nbPlans; L++) {
        float s[3] = {0.0f, 0.0f, 0.0f}, cnt = 0.0f;
        for (int i = 0; i < n; i++) {
            float r[3] = {0.0f, 0.0f, 0.0f}, rc = 0.0f;
            int any = 0;
            for (int j = 0; j < m; j++) {
                size_t idx = (size_t)i * m + j;
                if (Label[idx] != L + 1) continue;
                float l0 = VMap[3 * idx], l1 = VMap[3 * idx + 1], l2 = VMap[3 * idx + 2];
                r[0] = r[0] + (((pose[0] * l0 + pose[4] * l1) + pose[8] * l2) + pose[12]);   /* :526-528 */
                r[1] = r[1] + (((pose[1] * l0 + pose[5] * l1) + pose[9] * l2) + pose[13]);
                r[2] = r[2] + (((pose[2] * l0 + pose[6] * l1) + pose[10] * l2) + pose[14]);
                rc = rc + 1.0f;
                any = 1;
            }
            if (any) { s[0] = s[0] + r[0]; s[1] = s[1] + r[1]; s[2] = s[2] + r[2]; cnt = cnt + rc; }
        }
        if (cnt > 0.0f) { centers[3 * L] += s[0]; centers[3 * L + 1] += s[1]; centers[3 * L + 2] += s[2]; count[L] += cnt; }
    }
}

/* RG.cu:539-555 AffectPlanKernel: label remap through a LUT */
void orc_affect_plan(uint8_t *Label, const int32_t *Buff, int n, int m) {
    for (size_t idx = 0; idx < (size_t)n * m; idx++)
        if (Label[idx] != 0) Label[idx] = (uint8_t)Buff[Label[idx] - 1];
}

/* Primitive.cu:502-556,795-... Overlap_cu: share of plane `param`'s occupied cells that fall on plane `param_in`
 * (distance <= EPSILON = 0.1, cudaTypes.cuh:48) inside its N_prim x M_prim image; 1 when nothing is occupied */
float orc_overlap(const uint16_t *Bump, const uint8_t *Mask, const float *param, const float *param_in, int n, int m,
                  int N_prim, int M_prim, uint32_t *counter2) {
    uint32_t c0 = 0, c1 = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t idx = (size_t)i * m + j;
            if (Mask[idx] < 11) continue;
            float s0 = (float)((double)(float)Bump[3 * idx] / 60000.0), s1 = (float)((double)(float)Bump[3 * idx + 1] / 60000.0);
            float x = (float)((double)((float)i + s0) * 2.0 / (double)param[4] + (double)param[6]);
            float y = (float)((double)((float)j + s1) * 2.0 / (double)param[5] + (double)param[7]);
            float d = (float)((double)param[3] + (((double)(float)Bump[3 * idx + 2] / 2000.0) - 15.0));
            float pt[3];
            pt[0] = (x * param[8] + y * param[11]) + d * param[0];
            pt[1] = (x * param[9] + y * param[12]) + d * param[1];
            pt[2] = (x * param[10] + y * param[13]) + d * param[2];
            if (pt[0] == 0.0f && pt[1] == 0.0f && pt[2] == 0.0f) continue;
            c1++;
            float ed = ((pt[0] * param_in[0] + pt[1] * param_in[1]) + pt[2] * param_in[2]) - param_in[3];
            if ((double)fabsf(ed) > 0.1) continue;
            float pa = (pt[0] * param_in[8] + pt[1] * param_in[9]) + pt[2] * param_in[10];
            float pb = (pt[0] * param_in[11] + pt[1] * param_in[12]) + pt[2] * param_in[13];
            pa = (float)((double)((pa - param_in[6]) * param_in[4]) / 2.0);
            pb = (float)((double)((pb - param_in[7]) * param_in[5]) / 2.0);
            int ii = f2i_rd(pa), jj = f2i_rd(pb);
            if (ii > N_prim - 1 || jj > M_prim - 1 || ii < 0 || jj < 0) continue;
            c0++;
        }
    if (counter2) { counter2[0] = c0; counter2[1] = c1; }
    if (c1 == 0) return 1.0f;
    return (float)c0 / (float)c1;
}

void orc_init_mask(uint8_t *mask, const float *RGBD, int n, int m) {
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < (size_t)n * m; idx++) mask[idx] = (RGBD[4 * idx + 3] == 0.0f) ? 10 : 11;
}

static void seed_labels_impl(void *Label, int wide, const float *VMap, const float *NMap, const float *pose,
                             const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    #pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < (size_t)n * m; idx++) {
        LSET(Label, wide, idx, 0);
        float n0 = NMap[3 * idx], n1 = NMap[3 * idx + 1], n2 = NMap[3 * idx + 2];
        if (n0 == 0.0f && n1 == 0.0f && n2 == 0.0f) continue;
        float l0 = VMap[3 * idx], l1 = VMap[3 * idx + 1], l2 = VMap[3 * idx + 2];
        float p0 = ((pose[0] * l0 + pose[4] * l1) + pose[8] * l2) + pose[12];
        float p1 = ((pose[1] * l0 + pose[5] * l1) + pose[9] * l2) + pose[13];
        float p2 = ((pose[2] * l0 + pose[6] * l1) + pose[10] * l2) + pose[14];
        float q0 = (pose[0] * n0 + pose[4] * n1) + pose[8] * n2;
        float q1 = (pose[1] * n0 + pose[5] * n1) + pose[9] * n2;
        float q2 = (pose[2] * n0 + pose[6] * n1) + pose[10] * n2;
        for (int k = 0; k < nbPlans && k < (wide ? 65535 : 255); k++) {
            const float *E = Eq + 10 * k;
            float ed = ((p0 * E[0] + p1 * E[1]) + p2 * E[2]) - E[9];
            float ea = (q0 * E[0] + q1 * E[1]) + q2 * E[2];
            if (fabsf(ed) > epsilon || ea < alpha) continue;
            LSET(Label, wide, idx, k + 1);
            break;
        }
    }
}
void orc_seed_labels(uint8_t *Label, const float *VMap, const float *NMap, const float *pose,
                     const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    seed_labels_impl(Label, 0, VMap, NMap, pose, Eq, nbPlans, epsilon, alpha, n, m);
}
void orc_seed_labels_l16(uint16_t *Label, const float *VMap, const float *NMap, const float *pose,
                         const float *Eq, int nbPlans, float epsilon, float alpha, int n, int m) {
    seed_labels_impl(Label, 1, VMap, NMap, pose, Eq, nbPlans, epsilon, alpha, n, m);
}

/* ===================================================================================== */
/* Whole frame loop (D1/GP-demo.cpp:138-296)                                             */
/* ===================================================================================== */

struct orc_slam {
    int n, m, max_planes, N_prim, M_prim, nb_planes, count;
    float calib[11], intr[4], znear, zfar;
    orc_icp_params prm;
    /* input frame (D1/Frame.h:37-134) */
    float *fV, *fN, *fRGB;
    uint16_t *depth;
    /* predicted frame (D1/Frame.h:188-282) */
    float *V, *N, *RGBD, *tex;
    uint8_t *Mask, *MaskSwap, *Label, *LabelSwap;   /* Label / LabelSwap: 2 bytes per pixel allocated; u16 images when label16 */
    int label16;                    /* 16-bit label images (extension: models of more than 255 planes; Frame.h:197 keeps u8) */
    uint64_t *zbuf;
    /* planes (D1/Graph.h:54-116, D1/Primitive.h:180-190) */
    uint16_t **Bump; uint8_t **RGBb; uint8_t **Maskb; float **Param; uint64_t **Claim;
    float *Eq;
    uint32_t stamp;
    float poseR[9], poset[3], poseRPrev[9], posetPrev[3];
    /* RG::Finish schedule (RG.cpp:149-271) */
    int bilateral;                  /* optional stage of LoadFrame (Frame.cpp:50-54), default off */
    int finish_enabled, finish_first_after, finish_period, finish_min_size, finish_min_normals, first_in, indx, finish_detections;
};

orc_slam *orc_slam_create(int n, int m, const float *calib, const orc_icp_params *prm,
                          int max_planes, int N_prim, int M_prim, float znear, float zfar) {
    orc_slam *s = (orc_slam *)calloc(1, sizeof(orc_slam));
    size_t P = (size_t)n * m;
    s->n = n; s->m = m; s->max_planes = max_planes; s->N_prim = N_prim; s->M_prim = M_prim;
    s->znear = znear; s->zfar = zfar; s->count = 1;                              /* RG.h:112 */
    s->finish_enabled = 1; s->finish_first_after = 30; s->finish_period = 100;   /* RG.cpp:154,187 */
    s->finish_min_size = 5000; s->finish_min_normals = 1000; s->first_in = 1;    /* Utilities.h:85, RG.cpp:500, RG.h:194 */
    memcpy(s->calib, calib, sizeof(s->calib));
    memcpy(s->intr, calib, sizeof(s->intr));
    s->prm = *prm;
    s->fV = (float *)calloc(3 * P, 4); s->fN = (float *)calloc(3 * P, 4); s->fRGB = (float *)calloc(3 * P, 4);
    s->depth = (uint16_t *)calloc(P, 2);
    s->V = (float *)calloc(3 * P, 4); s->N = (float *)calloc(3 * P, 4); s->RGBD = (float *)calloc(4 * P, 4);
    s->tex = (float *)calloc(4 * P, 4);
    s->Mask = (uint8_t *)calloc(P, 1); s->MaskSwap = (uint8_t *)calloc(P, 1);
    s->Label = (uint8_t *)calloc(P, 2); s->LabelSwap = (uint8_t *)calloc(P, 2);
    s->zbuf = (uint64_t *)calloc(P, 8);
    s->Bump = (uint16_t **)calloc(max_planes, sizeof(void *)); s->RGBb = (uint8_t **)calloc(max_planes, sizeof(void *));
    s->Maskb = (uint8_t **)calloc(max_planes, sizeof(void *)); s->Param = (float **)calloc(max_planes, sizeof(void *));
    s->Claim = (uint64_t **)calloc(max_planes, sizeof(void *));
    s->Eq = (float *)calloc((size_t)10 * max_planes, 4);
    for (int k = 0; k < 9; k++) s->poseR[k] = s->poseRPrev[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    return s;
}

void orc_slam_destroy(orc_slam *s) {
    if (!s) return;
    free(s->fV); free(s->fN); free(s->fRGB); free(s->depth); free(s->V); free(s->N); free(s->RGBD); free(s->tex);
    free(s->Mask); free(s->MaskSwap); free(s->Label); free(s->LabelSwap); free(s->zbuf);
    for (int k = 0; k < s->max_planes; k++) { free(s->Bump[k]); free(s->RGBb[k]); free(s->Maskb[k]); free(s->Param[k]); free(s->Claim[k]); }
    free(s->Bump); free(s->RGBb); free(s->Maskb); free(s->Param); free(s->Claim); free(s->Eq);
    free(s);
}

static void pack_pose(const float *R, const float *t, float *P) {               /* GP-demo.cpp:254-259 */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) P[j * 4 + i] = M3(R, i, j);
    for (int i = 0; i < 3; i++) P[12 + i] = t[i];
    P[3] = 0.0f; P[7] = 0.0f; P[11] = 0.0f; P[15] = 1.0f;
}

/* planes from equations + centres: the Primitive part of CreatePrimitives (RG.cpp:703-741) */
static int install_planes(orc_slam *s, const float *eq, const float *centers, int nb) {
    if (nb > s->max_planes) nb = s->max_planes;
    size_t C = (size_t)s->N_prim * s->M_prim;
    for (int k = 0; k < nb; k++) {
        if (!s->Bump[k]) {
            s->Bump[k] = (uint16_t *)calloc(3 * C, 2); s->RGBb[k] = (uint8_t *)calloc(3 * C, 1);
            s->Maskb[k] = (uint8_t *)calloc(C, 1); s->Param[k] = (float *)calloc(14, 4);
            s->Claim[k] = (uint64_t *)calloc(C, 8);
        } else {
            memset(s->Bump[k], 0, 3 * C * 2); memset(s->RGBb[k], 0, 3 * C); memset(s->Maskb[k], 0, C);
        }
        const float *E = eq + 10 * k, *c = centers + 3 * k;
        memcpy(s->Eq + 10 * k, E, 40);
        /* D1/RG.cpp:706-723 and D1/Primitive.h:180-190 */
        float scale = (float)(2.0 / 0.004);
        float pt0 = (c[0] * E[3] + c[1] * E[4]) + c[2] * E[5];
        float pt1 = (c[0] * E[6] + c[1] * E[7]) + c[2] * E[8];
        float sh0 = (float)((double)pt0 - ((double)((float)s->N_prim) * 0.004) / 2.0);
        float sh1 = (float)((double)pt1 - ((double)((float)s->M_prim) * 0.004) / 2.0);
        float *Pm = s->Param[k];
        Pm[0] = E[0]; Pm[1] = E[1]; Pm[2] = E[2]; Pm[3] = E[9];
        Pm[4] = scale; Pm[5] = scale; Pm[6] = sh0; Pm[7] = sh1;
        Pm[8] = E[3]; Pm[9] = E[4]; Pm[10] = E[5]; Pm[11] = E[6]; Pm[12] = E[7]; Pm[13] = E[8];
    }
    s->nb_planes = nb;
    return nb;
}

void orc_slam_set_planes(orc_slam *s, const float *eq, const float *centers, int nb) {
    nb = install_planes(s, eq, centers, nb);
    float pose[16];
    pack_pose(s->poseR, s->poset, pose);
    seed_labels_impl(s->Label, s->label16, s->V, s->N, pose, s->Eq, nb, 0.1f, 0.8f, s->n, s->m);
}

/* RG::Finish's first branch (RG.cpp:154-175): DetectPrimitives (277-694) + CreatePrimitives (699-793) on the
 * current model frame.  equations_out / centers_out (may be NULL) receive the fitted planes.  Returns the count. */
int orc_slam_detect_planes(orc_slam *s, int min_size, int min_normals, float *equations_out, float *centers_out) {
    int n = s->n, m = s->m;
    size_t P = (size_t)n * m;
    float pose[16];
    pack_pose(s->poseR, s->poset, pose);
    uint8_t *ch = (uint8_t *)malloc(P), *e[3], *img = (uint8_t *)malloc(P), *img2 = (uint8_t *)malloc(P), *er = (uint8_t *)malloc(P);
    for (int c = 0; c < 3; c++) {
        e[c] = (uint8_t *)malloc(P);
        orc_cpy_n(ch, s->N, n, m, c);                                             /* RG.cpp:295-302 */
        orc_canny(ch, e[c], 120.0, 30.0, n, m);
    }
    orc_sum(img, e[0], e[1], e[2], n, m);                                          /* RG.cpp:304 */
    orc_discont(img, s->V, n, m, 0.03f);                                           /* RG.cpp:307 */
    memset(s->Label, 0, 2 * P);                                                    /* RG.cpp:313 */
    orc_thresh(img2, img, 20, s->Label, n, m);                                     /* RG.cpp:315 */
    orc_erode_ellipse5(er, img2, n, m);                                            /* RG.cpp:320-326 */
    int32_t *blob = (int32_t *)malloc(P * 4), *sizes = (int32_t *)malloc(P * 4);
    orc_find_blobs(er, blob, sizes, n, m);                                         /* RG.cpp:336-338 (binary = er > 0) */
    float *eq = (float *)calloc((size_t)10 * s->max_planes, 4), *ce = (float *)calloc((size_t)3 * s->max_planes, 4);
    int nb = orc_fit_planes(blob, sizes, s->V, s->N, pose, s->Label, eq, ce, n, m, min_size, min_normals, s->max_planes);
    nb = install_planes(s, eq, ce, nb);
    if (s->label16) {                                   /* the fit numbers its planes 1..nb <= 255: widen the blob labels */
        uint16_t *w = (uint16_t *)s->Label;
        for (size_t p = P; p-- > 0;) w[p] = s->Label[p];
    }
    if (nb > 0 && s->label16) orc_segment_l16((uint16_t *)s->Label, (uint16_t *)s->LabelSwap, s->V, s->N, pose, s->Eq, nb, 0.1f, 0.8f, n, m);
    else if (nb > 0) orc_segment(s->Label, s->LabelSwap, s->V, s->N, pose, s->Eq, nb, 0.1f, 0.8f, n, m);   /* RG.cpp:786 */
    if (equations_out) memcpy(equations_out, eq, (size_t)40 * nb);
    if (centers_out) memcpy(centers_out, ce, (size_t)12 * nb);
    free(ch); free(img); free(img2); free(er); free(blob); free(sizes); free(eq); free(ce);
    for (int c = 0; c < 3; c++) free(e[c]);
    return nb;
}

static void model_from_frame(orc_slam *s) {                                      /* Frame.cpp:335-370 */
    size_t P = (size_t)s->n * s->m;
    memcpy(s->V, s->fV, 12 * P); memcpy(s->N, s->fN, 12 * P);
    orc_add_zbuff_off(s->depth, s->m, s->fRGB, s->RGBD, s->calib, s->n, s->m);
    orc_init_mask(s->Mask, s->RGBD, s->n, s->m);
}

int orc_slam_step(orc_slam *s, const uint16_t *depth, const uint8_t *bgr) {
    int n = s->n, m = s->m;
    size_t P = (size_t)n * m;
    /* 1. LoadFrame (Frame.cpp:43-77) */
    if (s->bilateral) orc_bilateral_depth(s->depth, depth, m, 5, 100.0, 100.0, n, m);   /* Frame.cpp:50-54, when switched on */
    else memcpy(s->depth, depth, 2 * P);
    orc_cpy_char3(bgr, 3 * (size_t)m, s->fRGB, n, m);
    orc_vertex_map_kinect(s->fV, s->depth, m, s->calib, n, m);
    orc_compute_normal(s->fN, s->fV, n, m, 1);
    s->count++;
    int first = 0;
    if (s->count == 2) { model_from_frame(s); first = 1; }                      /* GP-demo.cpp:164-167 */
    /* 4. align */
    int ok = orc_align_gauss_newton(&s->prm, s->intr, s->fV, s->fN, s->fRGB, s->V, s->N, s->RGBD, n, m,
                                    s->poseR, s->poset, s->poseRPrev, s->posetPrev, NULL, NULL);
    if (!ok) {                                                                   /* GP-demo.cpp:189-196 */
        model_from_frame(s);
        memcpy(s->poseRPrev, s->poseR, 36); memcpy(s->posetPrev, s->poset, 12);
        return 0;
    }
    /* 5. relative pose (GP-demo.cpp:201-211) */
    float Rinv[9], tinv[3], Rrel[9], trel[3], tmp[3], Pm[16], Pinv[16], Pg[16];
    orc_inverse3(s->poseR, Rinv);
    { float neg[9]; for (int k = 0; k < 9; k++) neg[k] = -Rinv[k]; mat3_vec(neg, s->poset, tinv); }
    mat3_mul(Rinv, s->poseRPrev, Rrel);
    mat3_vec(Rinv, s->posetPrev, tmp);
    trel[0] = tmp[0] + tinv[0]; trel[1] = tmp[1] + tinv[1]; trel[2] = tmp[2] + tinv[2];
    pack_pose(Rrel, trel, Pm);
    /* 6. Transform (Frame.cpp:281-298); quad trim is folded into the raster (thresh 0.1) */
    orc_transform_vertex_map(Pm, s->V, s->RGBD, n, m);
    /* 7. raster + ReadFrame (Frame.cpp:229-245) */
    orc_raster(s->tex, s->zbuf, s->V, s->RGBD, NULL, 0.1f, s->calib, s->znear, s->zfar, n, m);
    orc_read_frame(s->V, s->RGBD, s->tex, s->calib, n, m);
    orc_compute_normal(s->N, s->V, n, m, 1);
    /* 8. inverse relative pose and global pose (GP-demo.cpp:225-236) */
    float RrelInv[9], trelInv[3];
    orc_inverse3(Rrel, RrelInv);
    { float neg[9]; for (int k = 0; k < 9; k++) neg[k] = -RrelInv[k]; mat3_vec(neg, trel, trelInv); }
    pack_pose(RrelInv, trelInv, Pinv);
    pack_pose(s->poseR, s->poset, Pg);
    orc_project_mask(Pinv, s->calib, s->Mask, s->MaskSwap, s->V, n, m);
    { uint8_t *t = s->Mask; s->Mask = s->MaskSwap; s->MaskSwap = t; }
    project_mask_impl(Pinv, s->calib, s->Label, s->LabelSwap, s->label16, s->V, n, m);
    { uint8_t *t = s->Label; s->Label = s->LabelSwap; s->LabelSwap = t; }
    if (s->nb_planes > 0 && s->label16)
        orc_segment_l16((uint16_t *)s->Label, (uint16_t *)s->LabelSwap, s->V, s->N, Pg, s->Eq, s->nb_planes, 0.1f, 0.9f, n, m);
    else if (s->nb_planes > 0)
        orc_segment(s->Label, s->LabelSwap, s->V, s->N, Pg, s->Eq, s->nb_planes, 0.1f, 0.9f, n, m);
    /* 9. */
    memcpy(s->poseRPrev, s->poseR, 36); memcpy(s->posetPrev, s->poset, 12);
    /* 10. Merge (Frame.cpp:300-333) */
    orc_merge_depth_off(s->RGBD, s->depth, m, s->fRGB, s->Mask, s->calib, n, m);
    orc_vertex_map_gl(s->V, s->RGBD, s->calib, n, m);
    orc_compute_normal(s->N, s->V, n, m, 1);
    /* 11. Graph::Update (Graph.cpp:50-95) */
    if (s->nb_planes > 0) {
        s->stamp++;
        if (s->label16)
            orc_update_l16(s->Bump, s->RGBb, s->Maskb, s->Param, s->Claim, s->stamp, s->V, s->RGBD, s->Mask,
                           (const uint16_t *)s->Label, Pg, s->N_prim, s->M_prim, n, m, 0, s->nb_planes);
        else
        orc_update(s->Bump, s->RGBb, s->Maskb, s->Param, s->Claim, s->stamp, s->V, s->RGBD, s->Mask,
                   s->Label, Pg, s->N_prim, s->M_prim, n, m, 0, s->nb_planes);
    }
    (void)P;
    /* 12. RG::Finish (GP-demo.cpp:274, RG.cpp:149-271; the dump is file I/O and lives in the host library, ReorganizeEqua's
     * output is overwritten by the detection that follows it) */
    if (s->finish_enabled) {
        if (s->first_in && s->count > s->finish_first_after) {                   /* RG.cpp:154-175 */
            orc_slam_detect_planes(s, s->finish_min_size, s->finish_min_normals, NULL, NULL);
            s->first_in = 0;
            s->finish_detections++;
        } else if (s->finish_period > 0 && (s->count % s->finish_period) == 0) { /* RG.cpp:187-270 */
            memcpy(s->poseRPrev, s->poseR, 36); memcpy(s->posetPrev, s->poset, 12);
            s->indx++;
            orc_slam_detect_planes(s, s->finish_min_size, s->finish_min_normals, NULL, NULL);
            s->finish_detections++;
        }
    }
    return first ? 2 : 1;
}

void orc_slam_set_bilateral(orc_slam *s, int on) { s->bilateral = on; }
/* before any plane exists: the label images become u16 (cleared) */
void orc_slam_set_label16(orc_slam *s, int on) {
    s->label16 = on ? 1 : 0;
    memset(s->Label, 0, 2 * (size_t)s->n * s->m); memset(s->LabelSwap, 0, 2 * (size_t)s->n * s->m);
}
void orc_slam_set_finish(orc_slam *s, int enabled, int first_after, int period, int min_size, int min_normals) {
    s->finish_enabled = enabled; s->finish_first_after = first_after; s->finish_period = period;
    if (min_size >= 0) s->finish_min_size = min_size;
    if (min_normals >= 0) s->finish_min_normals = min_normals;
}
int orc_slam_finish_detections(const orc_slam *s) { return s->finish_detections; }
int orc_slam_nb_planes(const orc_slam *s) { return s->nb_planes; }

void orc_slam_get_pose(const orc_slam *s, float *R, float *t) { memcpy(R, s->poseR, 36); memcpy(t, s->poset, 12); }

const void *orc_slam_buffer(const orc_slam *s, const char *name, size_t *bytes) {
    size_t P = (size_t)s->n * s->m, C = (size_t)s->N_prim * s->M_prim, b = 0;
    const void *p = NULL;
    if (!strcmp(name, "VMap")) { p = s->V; b = 12 * P; }
    else if (!strcmp(name, "NMap")) { p = s->N; b = 12 * P; }
    else if (!strcmp(name, "RGBD")) { p = s->RGBD; b = 16 * P; }
    else if (!strcmp(name, "Mask")) { p = s->Mask; b = P; }
    else if (!strcmp(name, "Label")) { p = s->Label; b = s->label16 ? 2 * P : P; }
    else if (!strcmp(name, "frame.VMap")) { p = s->fV; b = 12 * P; }
    else if (!strcmp(name, "frame.NMap")) { p = s->fN; b = 12 * P; }
    else if (!strcmp(name, "frame.RGB")) { p = s->fRGB; b = 12 * P; }
    else if (!strncmp(name, "Bump", 4)) { int k = atoi(name + 4); if (k < s->nb_planes) { p = s->Bump[k]; b = 6 * C; } }
    else if (!strncmp(name, "RGBb", 4)) { int k = atoi(name + 4); if (k < s->nb_planes) { p = s->RGBb[k]; b = 3 * C; } }
    else if (!strncmp(name, "Maskb", 5)) { int k = atoi(name + 5); if (k < s->nb_planes) { p = s->Maskb[k]; b = C; } }
    if (bytes) *bytes = b;
    return p;
}
