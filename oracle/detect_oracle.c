/* ORACLE (test infrastructure only -- never linked or loaded by the product path).
 *
 * CPU restatement of the reference's plane detection, parity unpinned: the
 * reference holds no fixtures for it and three of its stages are third-party OpenCV 2.4 functions
 * (cv::gpu::Canny, cv::erode, cv::floodFill; version unpinned, no build files) restated here from
 * their published algorithms:
 *   Canny   3x3 Sobel with replicated borders, L1 magnitude, thresholds swapped so that low <= high,
 *           non-maximum suppression with the fixed-point tangent test (tan 22.5 = 13573 / 2^15),
 *           magnitude 0 outside the image, hysteresis = 8-connected closure from the strong pixels;
 *   erode   5x5 MORPH_ELLIPSE (rows 00100 / 11111 / 11111 / 11111 / 00100), pixels outside the
 *           image do not constrain (default border of cv::erode);
 *   blobs   4-connected components, numbered in raster order of their first pixel.
 * Reference call sites: D1/RG.cpp:277-694 (DetectPrimitives), D1/Kernel.cu:684-781 (CpyKernelN,
 * DiscontKernel, SumKernel, ThreshKernel), D1/Utilities.cpp:1052-1092 (FindBlobs).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* D1/Kernel.cu:684-692 */
void orc_cpy_n(uint8_t *mat, const float *NMap, int n, int m, int chnl) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++)
            mat[(size_t)i * m + j] = (uint8_t)(int)(125.0 * (1.0 + (double)NMap[3 * ((size_t)i * m + j) + chnl]));
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv::gpu::Canny(img, edges, low, high, 3) -- RG.cpp:296-302 */
void orc_canny(const uint8_t *img, uint8_t *edges, double low_thresh, double high_thresh, int n, int m) {
    if (low_thresh > high_thresh) { double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
    const int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
    size_t P = (size_t)n * m;
    int *dx = (int *)malloc(P * sizeof(int)), *dy = (int *)malloc(P * sizeof(int)), *mag = (int *)malloc(P * sizeof(int));
    uint8_t *map = (uint8_t *)calloc(P, 1);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            int iu = clampi(i - 1, 0, n - 1), id = clampi(i + 1, 0, n - 1), jl = clampi(j - 1, 0, m - 1), jr = clampi(j + 1, 0, m - 1);
#define PX(a, b) ((int)img[(size_t)(a) * m + (b)])
            int gx = (PX(iu, jr) + 2 * PX(i, jr) + PX(id, jr)) - (PX(iu, jl) + 2 * PX(i, jl) + PX(id, jl));
            int gy = (PX(id, jl) + 2 * PX(id, j) + PX(id, jr)) - (PX(iu, jl) + 2 * PX(iu, j) + PX(iu, jr));
#undef PX
            dx[(size_t)i * m + j] = gx; dy[(size_t)i * m + j] = gy;
            mag[(size_t)i * m + j] = abs(gx) + abs(gy);
        }
#define MAG(a, b) (((a) < 0 || (a) >= n || (b) < 0 || (b) >= m) ? 0 : mag[(size_t)(a) * m + (b)])
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            size_t p = (size_t)i * m + j;
            int mm = mag[p];
            if (mm <= low) continue;
            int xs = dx[p], ys = dy[p];
            long long x = llabs((long long)xs), y = llabs((long long)ys) << 15;
            long long tg22x = x * 13573, tg67x = tg22x + (x << 16);
            int keep;
            if (y < tg22x) keep = mm > MAG(i, j - 1) && mm >= MAG(i, j + 1);
            else if (y > tg67x) keep = mm > MAG(i - 1, j) && mm >= MAG(i + 1, j);
            else { int s = ((xs ^ ys) < 0) ? -1 : 1; keep = mm > MAG(i - 1, j - s) && mm > MAG(i + 1, j + s); }
            if (keep) map[p] = (uint8_t)(mm > high ? 2 : 1);
        }
#undef MAG
    /* hysteresis: grow from the strong pixels through 8-connected candidates */
    memset(edges, 0, P);
    size_t *stack = (size_t *)malloc(P * sizeof(size_t));
    size_t top = 0;
    for (size_t p = 0; p < P; p++)
        if (map[p] == 2) { edges[p] = 255; stack[top++] = p; }
    while (top) {
        size_t p = stack[--top];
        int i = (int)(p / m), j = (int)(p % m);
        for (int a = -1; a <= 1; a++)
            for (int b = -1; b <= 1; b++) {
                int ii = i + a, jj = j + b;
                if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
                size_t q = (size_t)ii * m + jj;
                if (map[q] && !edges[q]) { edges[q] = 255; stack[top++] = q; }
            }
    }
    free(stack); free(map); free(dx); free(dy); free(mag);
}

/* D1/Kernel.cu:755-763 */
void orc_sum(uint8_t *out, const uint8_t *r, const uint8_t *g, const uint8_t *b, int n, int m) {
    for (size_t p = 0; p < (size_t)n * m; p++) out[p] = (uint8_t)(int)((double)(((float)r[p] + (float)g[p]) + (float)b[p]) / 3.0);
}

/* D1/Kernel.cu:697-720: only the pixel itself and its upper / left neighbours are visited (k < min(i+1, n)) */
void orc_discont(uint8_t *mat, const float *VMap, int n, int m, float thresh) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            int bi = i - 1 > 0 ? i - 1 : 0, ui = i + 1 < n ? i + 1 : n, bj = j - 1 > 0 ? j - 1 : 0, uj = j + 1 < m ? j + 1 : m;
            float depth = VMap[3 * ((size_t)i * m + j) + 2];
            int hit = 0;
            for (int k = bi; k < ui && !hit; k++)
                for (int h = bj; h < uj; h++) {
                    float z = VMap[3 * ((size_t)k * m + h) + 2];
                    if (z != 0.0f && fabsf(depth - z) > thresh) { hit = 1; break; }
                }
            if (hit) mat[(size_t)i * m + j] = 255;
        }
}

/* D1/Kernel.cu:768-778 */
void orc_thresh(uint8_t *out, const uint8_t *in, uint8_t thresh, const uint8_t *Label, int n, int m) {
    for (size_t p = 0; p < (size_t)n * m; p++) out[p] = (in[p] > thresh || Label[p] > 0) ? 0 : 255;
}

/* cv::erode(src, dst, getStructuringElement(MORPH_ELLIPSE, 5x5)) -- RG.cpp:320-326 */
void orc_erode_ellipse5(uint8_t *dst, const uint8_t *src, int n, int m) {
    static const int half[5] = {0, 2, 2, 2, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            int v = 255;
            for (int a = -2; a <= 2; a++)
                for (int b = -half[a + 2]; b <= half[a + 2]; b++) {
                    int ii = i + a, jj = j + b;
                    if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
                    int s = src[(size_t)ii * m + jj];
                    if (s < v) v = s;
                }
            dst[(size_t)i * m + j] = (uint8_t)v;
        }
}

/* D1/Utilities.cpp:1052-1092 with cv::floodFill(.., 4): blob[p] = raster index of the first pixel of
 * p's 4-connected component (-1 = background), sizes[first] = component size (0 elsewhere). */
void orc_find_blobs(const uint8_t *binary, int32_t *blob, int32_t *sizes, int n, int m) {
    size_t P = (size_t)n * m;
    for (size_t p = 0; p < P; p++) { blob[p] = -1; sizes[p] = 0; }
    size_t *stack = (size_t *)malloc(P * sizeof(size_t));
    for (size_t s = 0; s < P; s++) {
        if (!binary[s] || blob[s] != -1) continue;
        size_t top = 0;
        int cnt = 0;
        blob[s] = (int32_t)s; stack[top++] = s;
        while (top) {
            size_t p = stack[--top];
            cnt++;
            int i = (int)(p / m), j = (int)(p % m);
            const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
            for (int k = 0; k < 4; k++) {
                int ii = i + di[k], jj = j + dj[k];
                if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
                size_t q = (size_t)ii * m + jj;
                if (binary[q] && blob[q] == -1) { blob[q] = (int32_t)s; stack[top++] = q; }
            }
        }
        sizes[s] = cnt;
    }
    free(stack);
}

static int cmp_float(const void *a, const void *b) {
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* D1/RG.cpp:383-679: median plane fit of every blob larger than min_size (MIN_SIZE_PLAN) that holds at
 * least min_normals (1000) pixels with a normal.  Writes Label (value k+1 for the k-th kept plane, other
 * pixels untouched), equations (10 per plane: n, e1, e2, d) and centres (3 per plane).  Returns the
 * number of planes (<= max_planes). */
int orc_fit_planes(const int32_t *blob, const int32_t *sizes, const float *VMap, const float *NMap, const float *pose,
                   uint8_t *Label, float *equations, float *centers, int n, int m, int min_size, int min_normals,
                   int max_planes) {
    size_t P = (size_t)n * m;
    int nb = 0;
    for (size_t root = 0; root < P && nb < max_planes; root++) {
        int sz = sizes[root];
        if (sz <= min_size) continue;                                   /* RG.cpp:344 */
        size_t *px = (size_t *)malloc((size_t)sz * sizeof(size_t));
        int c = 0;
        for (size_t p = root; p < P && c < sz; p++)
            if (blob[p] == (int32_t)root) px[c++] = p;                  /* raster order, as FindBlobs lists them */
        float *nx = (float *)calloc((size_t)sz, sizeof(float)), *ny = (float *)calloc((size_t)sz, sizeof(float)),
              *nz = (float *)calloc((size_t)sz, sizeof(float)), *ds = (float *)calloc((size_t)sz, sizeof(float));
        int count = 0;
        for (int j = 0; j < sz; j++) {                                  /* RG.cpp:479-497: skipped pixels stay 0 in the lists */
            const float *N = NMap + 3 * px[j];
            if (N[0] == 0.0f && N[1] == 0.0f && N[2] == 0.0f) continue;
            count++;
            nx[j] = (N[0] * pose[0] + N[1] * pose[4]) + N[2] * pose[8];
            ny[j] = (N[0] * pose[1] + N[1] * pose[5]) + N[2] * pose[9];
            nz[j] = (N[0] * pose[2] + N[1] * pose[6]) + N[2] * pose[10];
        }
        if (count >= min_normals) {                                     /* RG.cpp:500-502 */
            qsort(nx, (size_t)sz, sizeof(float), cmp_float);
            qsort(ny, (size_t)sz, sizeof(float), cmp_float);
            qsort(nz, (size_t)sz, sizeof(float), cmp_float);
            float nm[3] = {nx[sz / 2], ny[sz / 2], nz[sz / 2]};         /* RG.cpp:525-533 */
            float norm = sqrtf((nm[0] * nm[0] + nm[1] * nm[1]) + nm[2] * nm[2]);
            nm[0] = nm[0] / norm; nm[1] = nm[1] / norm; nm[2] = nm[2] / norm;
            float cc[3] = {0.0f, 0.0f, 0.0f};
            for (int j = 0; j < sz; j++) {                              /* RG.cpp:557-578 */
                const float *V = VMap + 3 * px[j];
                float pt[3];
                pt[0] = ((V[0] * pose[0] + V[1] * pose[4]) + V[2] * pose[8]) + pose[12];
                pt[1] = ((V[0] * pose[1] + V[1] * pose[5]) + V[2] * pose[9]) + pose[13];
                pt[2] = ((V[0] * pose[2] + V[1] * pose[6]) + V[2] * pose[10]) + pose[14];
                cc[0] = cc[0] + pt[0]; cc[1] = cc[1] + pt[1]; cc[2] = cc[2] + pt[2];
                ds[j] = (pt[0] * nm[0] + pt[1] * nm[1]) + pt[2] * nm[2];
                Label[px[j]] = (uint8_t)(nb + 1);
            }
            centers[3 * nb] = cc[0] / (float)sz; centers[3 * nb + 1] = cc[1] / (float)sz; centers[3 * nb + 2] = cc[2] / (float)sz;
            qsort(ds, (size_t)sz, sizeof(float), cmp_float);
            float dist = ds[sz / 2];
            float e1[3], e2[3], ne;                                     /* RG.cpp:607-637 */
            if (nm[0] > nm[1] && nm[0] > nm[2]) { e1[0] = -nm[1]; e1[1] = nm[0]; e1[2] = 0.0f; }
            else if (nm[1] > nm[0] && nm[1] > nm[2]) { e1[0] = 0.0f; e1[1] = -nm[2]; e1[2] = nm[1]; }
            else { e1[0] = nm[2]; e1[1] = 0.0f; e1[2] = -nm[0]; }
            ne = sqrtf((e1[0] * e1[0] + e1[1] * e1[1]) + e1[2] * e1[2]);
            e1[0] /= ne; e1[1] /= ne; e1[2] /= ne;
            e2[0] = nm[1] * e1[2] - nm[2] * e1[1];
            e2[1] = nm[2] * e1[0] - nm[0] * e1[2];
            e2[2] = nm[0] * e1[1] - nm[1] * e1[0];
            float *E = equations + 10 * nb;
            E[0] = nm[0]; E[1] = nm[1]; E[2] = nm[2]; E[3] = e1[0]; E[4] = e1[1]; E[5] = e1[2];
            E[6] = e2[0]; E[7] = e2[1]; E[8] = e2[2]; E[9] = dist;
            nb++;
        }
        free(px); free(nx); free(ny); free(nz); free(ds);
    }
    return nb;
}
