// device_fns.hpp -- per-pixel device functions shared by the per-operator kernels (preprocess.hip,
// predict.hip) and the fused frame kernels (frame.hip).  One definition of each formula, so both
// launch shapes produce bit-identical results.
#pragma once
#include "common.hpp"

// The correctly rounded quotient a / b from y = RN(1 / b): q = RN(a y), r = a - b q in one FMA, RN(q + r y).  Where r is exact
// (q close enough to a / b for a - b q to fit the format) the last step rounds a / b (1 + e), |e| ~ 2^-2p, i.e. a / b itself
// unless a / b sits that close to a rounding boundary (Markstein).  That is NOT true of every pair of floats -- about one random
// f32 pair in 700 has an inexact residual, and divisors whose significand is nearly all ones always do -- so every use below
// is one of: (a) a constant divisor with a bounded numerator set, checked for EVERY numerator (u16 / 60000, byte / 255, the
// 256^3 colour sums / 3, all 2^23 significands / 100: tools/markstein_check.py, exact rational arithmetic); (b) an f64
// quotient whose operands come from floats or small constants (decode_cell), where b q needs at most ~80 bits, the residual
// always fits and no quotient of such operands can lie within 2^-100 of a boundary; (c) the two calibration quotients, whose
// 65 536 possible numerators are checked on the host against the division when the calibration is set (context.hip).
// Three instructions where the IEEE division sequence has eleven (f32) or fifteen (f64); the explicit FMAs are not
// contractions (-ffp-contract=off only governs a * b + c).  No overflow, underflow or non-finite value may occur on the way:
// the callers' range conditions.
__device__ __forceinline__ float quot_rcp(float a, float b, float y) { float q = a * y; return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q); }
__device__ __forceinline__ double quot_rcp(double a, double b, double y) { double q = a * y; return __builtin_fma(__builtin_fma(-b, q, a), y, q); }
// x / 255.0f for a byte and x / 3.0f for a sum of three such quotients (0 or in [1/255, 3]): constant denominators, bounded
// numerators -- the first checked for all 256 bytes, the second for all 16.7 M sums (tools/markstein_check.py)
__device__ __forceinline__ float div255(float x) { return quot_rcp(x, 255.0f, 1.0f / 255.0f); }
__device__ __forceinline__ float div3_unit(float x) { return quot_rcp(x, 3.0f, 1.0f / 3.0f); }
// (k - cx) / fx and (k - cy) / fy for a pixel index k < 65536, and the depth scale: from the host's reciprocals where the
// calibration allows it (CalibV::fast, uniform): |k - c| is 0 or in [2^-60, 2^21], the quotient and its residual are normal
__device__ __forceinline__ float col_over_fx(int j, const CalibV &c) {
    const float a = (float)j - c.v[2];
    if (c.fast) return quot_rcp(a, c.v[0], c.inv_fx);
    return a / c.v[0];
}
__device__ __forceinline__ float row_over_fy(int i, const CalibV &c) {
    const float a = (float)i - c.v[3];
    if (c.fast) return quot_rcp(a, c.v[1], c.inv_fy);
    return a / c.v[1];
}
__device__ __forceinline__ float depth_scale(const CalibV &c) { return c.fast ? c.s : c.v[9] / c.v[10]; }
// (float)((double)x / 100.0) == x / 100.0f for a float x (double rounding is innocuous for a quotient: 53 >= 2 * 24 + 2); for
// a finite x that is 0 or at least 2^-60 in magnitude -- a metric depth of the calibrated range -- from the reciprocal
__device__ __forceinline__ float depth_over_max(float x, const CalibV &c) {
    if (c.fast) return quot_rcp(x, (float)SLAM_MAX_DEPTH, 1.0f / (float)SLAM_MAX_DEPTH);
    return (float)((double)x / SLAM_MAX_DEPTH);
}

// ---- D1/Kernel.cu:33-52 VertexMapKinectKernel: vertex of pixel (i, j) from raw depth d -----------
__device__ __forceinline__ F3 vertex_from_depth(uint16_t d, int i, int j, const CalibV &c) {
    F3 r; r.x = r.y = r.z = 0.0f;
    if (d == 0) return r;
    float s = depth_scale(c);
    float fd = (float)d;
    float sd = s * fd;
    r.x = sd * col_over_fx(j, c);
    r.y = sd * row_over_fy(i, c);
    r.z = (-s) * fd;
    return r;
}

// ---- D1/Kernel.cu:123-307 NormalKernel ------------------------------------------------------------
// one of the four triangles around the pixel: normalised cross((a-p),(b-p)); returns 1 if counted
__device__ __forceinline__ int tri_normal(F3 p, F3 a, F3 b, F3 &o) {
    o.x = o.y = o.z = 0.0f;
    if (a.z != 0.0f && b.z != 0.0f) {
        float ax = a.x - p.x, ay = a.y - p.y, az = a.z - p.z;
        float bx = b.x - p.x, by = b.y - p.y, bz = b.z - p.z;
        float cx = ay * bz - az * by;
        float cy = az * bx - ax * bz;
        float cz = ax * by - ay * bx;
        float nn = (cx * cx + cy * cy + cz * cz);
        o.x = cx; o.y = cy; o.z = cz;
        if (nn != 0.0f) {
            float r = __builtin_sqrtf(nn);
            o.x = cx / r; o.y = cy / r; o.z = cz / r;
            return 1;
        }
    }
    return 0;
}
// normal of an interior pixel p (z != 0) from its 4-neighbourhood: dn = (i+1,j), up = (i-1,j), rt = (i,j+1), lf = (i,j-1)
__device__ __forceinline__ F3 normal_from_cross(F3 p, F3 dn, F3 up, F3 rt, F3 lf, int inverse) {
    F3 z; z.x = z.y = z.z = 0.0f;
    F3 n1, n2, n3, n4;
    int tot = tri_normal(p, dn, rt, n1);
    tot += tri_normal(p, rt, up, n2);
    tot += tri_normal(p, up, lf, n3);
    tot += tri_normal(p, lf, dn, n4);
    if (tot == 0) return z;
    float ft = (float)tot;
    float a0 = ((n1.x + n2.x) + n3.x) + n4.x, a1 = ((n1.y + n2.y) + n3.y) + n4.y, a2 = ((n1.z + n2.z) + n3.z) + n4.z;
    // x / 1, x / 2, x / 4 are x * 1, x * 0.5, x * 0.25 bit for bit (both are the correctly rounded value of the same real
    // number); only a count of 3 needs a division, and a pixel with exactly three valid triangles is rare (the rim of a
    // hole): the three IEEE divisions (~33 instructions of a ~330-instruction normal) run only in waves that hold one.
    const float inv = (tot == 4) ? 0.25f : ((tot == 2) ? 0.5f : 1.0f);
    if (__ballot(tot == 3) != 0ull) {
        a0 = (tot == 3) ? a0 / ft : a0 * inv; a1 = (tot == 3) ? a1 / ft : a1 * inv; a2 = (tot == 3) ? a2 / ft : a2 * inv;
    } else { a0 = a0 * inv; a1 = a1 * inv; a2 = a2 * inv; }
    float nn = __builtin_sqrtf((a0 * a0 + a1 * a1) + a2 * a2);
    if (nn == 0.0f) return z;
    F3 r;
    if (inverse) { r.x = (-a0) / nn; r.y = (-a1) / nn; r.z = (-a2) / nn; }
    else { r.x = a0 / nn; r.y = a1 / nn; r.z = a2 / nn; }
    return r;
}
__device__ __forceinline__ F3 normal_global(const float *__restrict__ VMap, int i, int j, int n, int m, int inverse) {
    size_t idx = (size_t)i * m + j;
    F3 p = ld3(VMap, idx);
    F3 z; z.x = z.y = z.z = 0.0f;
    if (i < 1 || i > n - 2 || j < 1 || j > m - 2 || p.z == 0.0f) return z;
    return normal_from_cross(p, ld3(VMap, idx + m), ld3(VMap, idx - m), ld3(VMap, idx + 1), ld3(VMap, idx - 1), inverse);
}

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
    return p;
}

// Pins loaded values in front of the branches that follow: passed through an empty asm they must all be present
// there, so the compiler cannot sink each load behind the branch that precedes its first use (which turns a batch
// of independent loads into a chain of dependent memory round trips).
__device__ __forceinline__ void pin_loaded(F3 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z)); }
__device__ __forceinline__ void pin_loaded(float4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }

// ---- D1/Kernel.cu:440-526 MergeDepthOffKernel: new (.w, mask) of one pixel -------------------------
// returns the new .w; newmask < 0 = mask unchanged
__device__ __forceinline__ float merge_pixel(float w_old, uint16_t d, uint8_t mask_old, const CalibV &c, int &newmask) {
    float dv = depth_scale(c) * (float)d;
    float dp = (float)((double)w_old * SLAM_MAX_DEPTH);
    float mv = (float)((double)(float)mask_old - 10.0);
    const float mt = 1.0f;
    float w = w_old;
    newmask = -1;
    if ((double)dv < 0.1) {
    } else if ((double)dp < 0.1 || __builtin_fabs((double)mv + 9.0) < 0.01) {
        w = depth_over_max(dv, c); newmask = (int)(uint8_t)((double)mt + 10.0);
    } else if ((double)dv < ((double)dp - SLAM_MAXTOLERANCE)) {
        w = depth_over_max(dv, c); newmask = (int)(uint8_t)((double)mt + 10.0);
    } else {
        double mx = 1.0 > 1.0 * (double)dv ? 1.0 : 1.0 * (double)dv;
        if ((double)dv > ((double)dp + mx * SLAM_MAXTOLERANCE)) {
            newmask = 10;
        } else {
            float am = __builtin_fabsf(mv);
            float avg = (dv * mt + am * dp) / (mt + am);
            w = depth_over_max(avg, c);                          // (both depths are at least 0.1 here, or avg is NaN and stays NaN)
            double mk = (double)(mt + am) + 10.0;
            if (mk > 255.0) mk = 255.0;
            newmask = (int)(uint8_t)mk;
        }
    }
    return w;
}

// ---- D1/Kernel.cu:54-72 VertexMapGLKernel -----------------------------------------------------------
__device__ __forceinline__ F3 vertex_from_w(float w, int i, int j, const CalibV &c) {
    double z = (double)w * SLAM_MAX_DEPTH;
    F3 r;
    r.x = (w == 0.0f) ? 0.0f : (float)(z * (double)col_over_fx(j, c));
    r.y = (w == 0.0f) ? 0.0f : (float)(z * (double)row_over_fy(i, c));
    r.z = (float)((double)(-w) * SLAM_MAX_DEPTH);
    return r;
}

// ---- D1/Kernel.cu:783-802 AddZBUFFOffKernel: .w of the seeded model ---------------------------------
__device__ __forceinline__ float zbuff_w(uint16_t d, const CalibV &c) {
    float z = depth_scale(c) * (float)d;
    return ((double)z > SLAM_MAX_DEPTH || (double)z < 0.0) ? 0.0f : depth_over_max(z, c);
}

// ---- D1/Kernel.cu:345-379 TransformVertexMapKernel: .w written next to a transformed vertex ---------
__device__ __forceinline__ float depth_attr(float z) {
    return ((double)(-z) > SLAM_MAX_DEPTH || (double)(-z) < 0.0) ? 0.0f : (float)((double)(-z) / SLAM_MAX_DEPTH);
}

// ---- D1/Kernel.cu:381-438 QuadTrimFrameKernel --------------------------------------------------------
__device__ __forceinline__ float dist3(F3 a, F3 b) {
    float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
    return __builtin_sqrtf((x * x + y * y) + z * z);
}
__device__ __forceinline__ float dist3_sq(F3 a, F3 b) {
    float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
    return (x * x + y * y) + z * z;
}
// max of the six edge lengths > thresh.  sqrtf is correctly rounded, hence monotonic:
// max_i sqrt(s_i) == sqrt(max_i s_i) bit for bit, so one square root decides.
__device__ __forceinline__ bool quad_trimmed(F3 a, F3 b, F3 c, F3 d, float thresh) {
    float mx = fmaxf(dist3_sq(a, b), dist3_sq(a, c));
    mx = fmaxf(dist3_sq(a, d), mx);
    mx = fmaxf(dist3_sq(b, c), mx);
    mx = fmaxf(dist3_sq(b, d), mx);
    mx = fmaxf(dist3_sq(c, d), mx);
    return __builtin_sqrtf(mx) > thresh;
}

// ---- compute rasteriser ---------------------------------------------------------------
// Vertices are snapped to 1/256 pixel; coverage uses exact 64-bit integer edge functions with a
// top-left rule (watertight, each pixel of a shared edge drawn once); depth and attributes are
// interpolated perspective-correctly in double; the z-buffer is a u64 atomicMin on
// (f32 depth bits << 32 | ~triangle id): nearest wins, the later primitive wins ties (GL_LEQUAL).
#define RASTER_MIN_DEPTH 1e-3f
#define RASTER_GUARD 16384.0f
#define ZBUF_EMPTY 0xFFFFFFFFFFFFFFFFull

struct RasTri { int X[3], Y[3]; double inv_d[3]; long long area; int order[3]; };

// projection of one vertex: shared by the two triangles of a quad (k_predict_draw projects four vertices, not six)
struct RasVert { int X, Y; double inv_d; bool ok; };
// Branch-free: the three rejection tests are evaluated together and applied with selects at the end (an early exit per test
// made the callers' three projections a cascade of nested branches with their results re-zeroed at every level).  A rejected
// vertex has all fields zero as before; the arithmetic on its garbage (division by a zero depth, ...) has no side effects.
__device__ __forceinline__ RasVert ras_project(F3 v, const CalibV &c) {
    RasVert P;
    const float dk = -v.z;
    const float u = (v.x / dk) * c.v[0] + c.v[2];
    const float w = (v.y / dk) * c.v[1] + c.v[3];
    const bool ok = (dk >= RASTER_MIN_DEPTH) & (__builtin_fabsf(u) <= RASTER_GUARD) & (__builtin_fabsf(w) <= RASTER_GUARD);
    const int X = f2i_rn(u * 256.0f), Y = f2i_rn(w * 256.0f);
    const double inv_d = 1.0 / (double)dk;
    P.X = ok ? X : 0;
    P.Y = ok ? Y : 0;
    P.inv_d = ok ? inv_d : 0.0;
    P.ok = ok;
    return P;
}
__device__ __forceinline__ bool ras_setup_projected(const RasVert &a, const RasVert &b, const RasVert &d, RasTri &T) {
    if (!a.ok || !b.ok || !d.ok) return false;
    T.X[0] = a.X; T.Y[0] = a.Y; T.inv_d[0] = a.inv_d; T.order[0] = 0;
    T.X[1] = b.X; T.Y[1] = b.Y; T.inv_d[1] = b.inv_d; T.order[1] = 1;
    T.X[2] = d.X; T.Y[2] = d.Y; T.inv_d[2] = d.inv_d; T.order[2] = 2;
    long long area = (long long)(T.X[1] - T.X[0]) * (T.Y[2] - T.Y[0]) - (long long)(T.Y[1] - T.Y[0]) * (T.X[2] - T.X[0]);
    if (area == 0) return false;
    if (area < 0) {
        int t;
        t = T.X[1]; T.X[1] = T.X[2]; T.X[2] = t;
        t = T.Y[1]; T.Y[1] = T.Y[2]; T.Y[2] = t;
        double dd = T.inv_d[1]; T.inv_d[1] = T.inv_d[2]; T.inv_d[2] = dd;
        T.order[1] = 2; T.order[2] = 1;
        area = -area;
    }
    T.area = area;
    return true;
}
__device__ __forceinline__ bool ras_setup(const F3 *vtx, const CalibV &c, RasTri &T) {
    return ras_setup_projected(ras_project(vtx[0], c), ras_project(vtx[1], c), ras_project(vtx[2], c), T);
}
__device__ __forceinline__ long long ras_edge(int ax, int ay, int bx, int by, int px, int py) {
    return (long long)(bx - ax) * (py - ay) - (long long)(by - ay) * (px - ax);
}
__device__ __forceinline__ bool ras_topleft(int ax, int ay, int bx, int by) {
    int dx = bx - ax, dy = by - ay;
    return (dy < 0) || (dy == 0 && dx < 0);
}
__device__ __forceinline__ bool ras_cover(const RasTri &T, int px, int py, long long *w) {
    int PX = px * 256, PY = py * 256;
    w[0] = ras_edge(T.X[1], T.Y[1], T.X[2], T.Y[2], PX, PY);
    w[1] = ras_edge(T.X[2], T.Y[2], T.X[0], T.Y[0], PX, PY);
    w[2] = ras_edge(T.X[0], T.Y[0], T.X[1], T.Y[1], PX, PY);
    if (w[0] < 0 || w[1] < 0 || w[2] < 0) return false;
    if (w[0] == 0 && !ras_topleft(T.X[1], T.Y[1], T.X[2], T.Y[2])) return false;
    if (w[1] == 0 && !ras_topleft(T.X[2], T.Y[2], T.X[0], T.Y[0])) return false;
    if (w[2] == 0 && !ras_topleft(T.X[0], T.Y[0], T.X[1], T.Y[1])) return false;
    return true;
}
// z-buffer pass for one triangle (already set up).  The three edge functions are affine in the pixel: they are evaluated
// once at the corner of the bounding box and then stepped (exact 64-bit integers, so every pixel sees the value ras_cover
// computes from scratch -- two 64-bit multiplies per edge and pixel there, one 64-bit add here).
__device__ __forceinline__ void ras_draw_tri(unsigned long long *__restrict__ zbuf, const RasTri &T, uint32_t tri_id, float znear,
                                             float zfar, int n, int m) {
    int xmn = min(T.X[0], min(T.X[1], T.X[2])), xmx = max(T.X[0], max(T.X[1], T.X[2]));
    int ymn = min(T.Y[0], min(T.Y[1], T.Y[2])), ymx = max(T.Y[0], max(T.Y[1], T.Y[2]));
    int x0 = max(0, (xmn + 255) >> 8), x1 = min(m - 1, xmx >> 8);
    int y0 = max(0, (ymn + 255) >> 8), y1 = min(n - 1, ymx >> 8);
    if (x0 > x1 || y0 > y1) return;
    // Fast path (round 3): a triangle whose bounding box spans at most 64 pixels either way -- every triangle of a depth
    // image's mesh -- has edge functions below 2^31 at every pixel centre of that box (|dx|, |dy| <= 2^14 and the pixel
    // within 2^14 + 256 of the edge's start: |w| < 2 * 2^14 * (2^14 + 256) < 2^30), so the SAME integers are formed and stepped
    // in 32 bits: one add instead of two per edge and pixel, one conversion to double instead of a 64-bit one.  The area
    // (sum of the three) fits likewise.  Same values, same doubles, same depth bits; larger triangles take the 64-bit loop.
    if (xmx - xmn <= 16384 && ymx - ymn <= 16384) {
        int row[3], sx[3], sy[3];
        bool tl[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int a = (k + 1) % 3, b = (k + 2) % 3;
            row[k] = (T.X[b] - T.X[a]) * (y0 * 256 - T.Y[a]) - (T.Y[b] - T.Y[a]) * (x0 * 256 - T.X[a]);
            sx[k] = -(T.Y[b] - T.Y[a]) * 256;
            sy[k] = (T.X[b] - T.X[a]) * 256;
            tl[k] = ras_topleft(T.X[a], T.Y[a], T.X[b], T.Y[b]);
        }
        const double area = (double)(int)T.area;
        for (int py = y0; py <= y1; py++) {
            int w[3] = {row[0], row[1], row[2]};
            for (int px = x0; px <= x1; px++) {
                const bool in = !((w[0] | w[1] | w[2]) < 0) && (w[0] != 0 || tl[0]) && (w[1] != 0 || tl[1]) && (w[2] != 0 || tl[2]);
                if (in) {
                    double W = ((double)w[0] * T.inv_d[0] + (double)w[1] * T.inv_d[1]) + (double)w[2] * T.inv_d[2];
                    float dpx = (float)(area / W);
                    if (dpx >= znear && dpx <= zfar) {
                        unsigned long long key = ((unsigned long long)__float_as_uint(dpx) << 32) | (unsigned long long)(0xFFFFFFFFu - tri_id);
                        atomicMin(&zbuf[(size_t)py * m + px], key);
                    }
                }
                w[0] += sx[0]; w[1] += sx[1]; w[2] += sx[2];
            }
            row[0] += sy[0]; row[1] += sy[1]; row[2] += sy[2];
        }
        return;
    }
    // edge k runs from vertex (k+1)%3 to vertex (k+2)%3 (as in ras_cover)
    long long row[3], sx[3], sy[3];
    bool tl[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        row[k] = ras_edge(T.X[a], T.Y[a], T.X[b], T.Y[b], x0 * 256, y0 * 256);
        sx[k] = -(long long)(T.Y[b] - T.Y[a]) * 256;          // one pixel to the right
        sy[k] = (long long)(T.X[b] - T.X[a]) * 256;           // one pixel down
        tl[k] = ras_topleft(T.X[a], T.Y[a], T.X[b], T.Y[b]);
    }
    for (int py = y0; py <= y1; py++) {
        long long w[3] = {row[0], row[1], row[2]};
        for (int px = x0; px <= x1; px++) {
            const bool in = !(w[0] < 0 || w[1] < 0 || w[2] < 0) && (w[0] != 0 || tl[0]) && (w[1] != 0 || tl[1]) && (w[2] != 0 || tl[2]);
            if (in) {
                double W = ((double)w[0] * T.inv_d[0] + (double)w[1] * T.inv_d[1]) + (double)w[2] * T.inv_d[2];
                float dpx = (float)((double)T.area / W);
                if (dpx >= znear && dpx <= zfar) {
                    unsigned long long key = ((unsigned long long)__float_as_uint(dpx) << 32) | (unsigned long long)(0xFFFFFFFFu - tri_id);
                    atomicMin(&zbuf[(size_t)py * m + px], key);
                }
            }
            w[0] += sx[0]; w[1] += sx[1]; w[2] += sx[2];
        }
        row[0] += sy[0]; row[1] += sy[1]; row[2] += sy[2];
    }
}
__device__ __forceinline__ void ras_draw(unsigned long long *__restrict__ zbuf, const F3 *vtx, uint32_t tri_id,
                                         const CalibV &c, float znear, float zfar, int n, int m) {
    RasTri T;
    if (!ras_setup(vtx, c, T)) return;
    ras_draw_tri(zbuf, T, tri_id, znear, zfar, n, m);
}
// resolve pass for the pixel (px, py) covered by the triangle of the three PROJECTED vertices with attributes att
// (r, g, b, depth/100)
__device__ __forceinline__ float4 ras_shade_projected(const RasVert &pa, const RasVert &pb, const RasVert &pd, const float4 *att,
                                                      int px, int py) {
    RasTri T; long long w[3];
    ras_setup_projected(pa, pb, pd, T);
    ras_cover(T, px, py, w);
    double b0 = (double)w[0] * T.inv_d[0], b1 = (double)w[1] * T.inv_d[1], b2 = (double)w[2] * T.inv_d[2];
    double W = (b0 + b1) + b2;
    // T.order is a permutation of {0,1,2}: select without dynamic indexing
    float4 A0 = att[0], A1 = (T.order[1] == 1) ? att[1] : att[2], A2 = (T.order[1] == 1) ? att[2] : att[1];
    float4 o;
    o.x = (float)(((b0 * (double)A0.x + b1 * (double)A1.x) + b2 * (double)A2.x) / W);
    o.y = (float)(((b0 * (double)A0.y + b1 * (double)A1.y) + b2 * (double)A2.y) / W);
    o.z = (float)(((b0 * (double)A0.z + b1 * (double)A1.z) + b2 * (double)A2.z) / W);
    o.w = (float)(((b0 * (double)A0.w + b1 * (double)A1.w) + b2 * (double)A2.w) / W);
    return o;
}
__device__ __forceinline__ float4 ras_shade(const F3 *vtx, const float4 *att, const CalibV &c, int px, int py) {
    return ras_shade_projected(ras_project(vtx[0], c), ras_project(vtx[1], c), ras_project(vtx[2], c), att, px, py);
}
__device__ __forceinline__ void quad_vertex_ids(uint32_t q, const uint32_t *__restrict__ index, int m, uint32_t *id) {
    if (index) {
        uint4 v = *(const uint4 *)(index + 4 * (size_t)q);
        id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
    } else {
        uint32_t i = q / (uint32_t)(m - 1), j = q % (uint32_t)(m - 1);
        id[0] = i * m + j; id[1] = id[0] + 1; id[2] = id[0] + m + 1; id[3] = id[0] + m;
    }
}

// ---- D1/Kernel.cu:309-343 ReadKernel: texel -> vertex ----------------------------------------------
__device__ __forceinline__ F3 vertex_from_texel(float w, int i, int j, const CalibV &c) {
    float z = (float)((double)w * SLAM_MAX_DEPTH);
    F3 r;
    r.x = z * col_over_fx(j, c);
    r.y = z * row_over_fy(i, c);
    r.z = -z;
    return r;
}

// ---- D1/Kernel.cu:634-667 ProjectMaskKernel: source pixel of the gather, -1 = out of view -----------
__device__ __forceinline__ long project_source(const float *P, const CalibV &c, F3 v, int n, int m) {
    if (is_zero3(v)) return -1;
    F3 r = xf_point(P, v);
    float az = __builtin_fabsf(r.z);
    int pu = f2i_rn((r.x / az) * c.v[0] + c.v[2]);
    int pv = f2i_rn((r.y / az) * c.v[1] + c.v[3]);
    // the reference only tests the upper bounds; negative projections are defined as out of view
    if (pv > n - 1 || pu > m - 1 || pv < 0 || pu < 0) return -1;
    return (long)pv * m + pu;
}

// ---- D1/RG.cu:252-320 SegmentKernel: label adopted by an unlabeled pixel (0 = none) ------------------
// LT = the label image's element type: uint8_t (the reference's, Frame.h:197) or uint16_t (models beyond 255 planes)
template <typename LT>
__device__ __forceinline__ LT segment_pixel(const LT *__restrict__ in, const float *__restrict__ VMap, F3 nl,
                                            const float *P, const float *__restrict__ Eq, float epsilon, float alpha,
                                            int i, int j, int n, int m) {
    size_t idx = (size_t)i * m + j;
    LT res = 0;
    F3 q = xf_dir(P, nl);
    if (!is_zero3(q)) {
        F3 pl = ld3(VMap, idx);
        F3 p = xf_point(P, pl);
        float depth = pl.z;
        const float thresh_depth = 0.003f;
        // The reference scans the 3x3 neighbourhood row by row and stops at the first neighbour that qualifies
        // (RG.cu:285-311).  All nine labels and depths are fetched first (independent loads, one round trip); the
        // scan then runs on registers in the same order, so the same neighbour wins.
        LT lab[9];
        float dz[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            int ki = i - 1 + k / 3, kj = j - 1 + k % 3;
            bool ok = ki >= 0 && ki < n && kj >= 0 && kj < m;
            size_t nb = ok ? (size_t)ki * m + kj : idx;
            lab[k] = ok ? in[nb] : (LT)0;
            dz[k] = VMap[3 * nb + 2];
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            LT l = lab[k];
            if (res == 0 && l > 0 && __builtin_fabsf(dz[k] - depth) < thresh_depth) {
                const float *E = Eq + 10 * ((int)l - 1);
                float ed = ((p.x * E[0] + p.y * E[1]) + p.z * E[2]) - E[9];
                float ea = (q.x * E[0] + q.y * E[1]) + q.z * E[2];
                if (!(__builtin_fabsf(ed) > epsilon || ea < alpha)) res = l;
            }
        }
    }
    return res;
}

// ---- D1/Primitive.cu:31-93 VertexFromBumpKernel: one Bump cell -> vertex, colour, packed mask ----------
struct BumpCell { float3 v; float4 c, k; };
// inv4 / inv5 = the correctly rounded reciprocals of p[4], p[5] as doubles, fast = every quotient of decode_cell may be formed
// from a reciprocal (below); both are functions of the plane, computed once per thread by load_param
struct BumpParam { float p[14]; double inv4, inv5; bool fast; };
// u8 / 255.0 (Primitive.cu:80-82,90): (float)((double)x / C) == x / (float)C for float x -- double
// rounding is innocuous for division (53 >= 2*24 + 2)
__device__ __forceinline__ float unorm8(uint8_t x) { return div255((float)x); }
// UNORM = functor returning x/255 for a byte (direct division, or a 256-entry LDS table)
template <typename UNORM>
__device__ __forceinline__ BumpCell decode_cell(uint16_t b0, uint16_t b1, uint16_t b2, uint8_t r, uint8_t g, uint8_t b, uint8_t mk,
                                                const BumpParam &P, float fi, float fj, UNORM unorm) {
    const float *param = P.p;
    BumpCell o;
    float x, y, d;
    if (P.fast) {                                               // (uniform: a property of the plane)
        // u16 / 60000.0f and u16 / 2000.0: bounded numerators, constant denominators.  (fi + s0) * 2 is 0 or in [2^-15, 2^18],
        // p[4], p[5] are finite and in [2^-40, 2^60] in magnitude: every quotient and residual is finite and normal.
        const float s0 = quot_rcp((float)b0, 60000.0f, 1.0f / 60000.0f);
        const float s1 = quot_rcp((float)b1, 60000.0f, 1.0f / 60000.0f);
        x = (float)(quot_rcp((double)(fi + s0) * 2.0, (double)param[4], P.inv4) + (double)param[6]);
        y = (float)(quot_rcp((double)(fj + s1) * 2.0, (double)param[5], P.inv5) + (double)param[7]);
        d = (float)((double)param[3] + (quot_rcp((double)(float)b2, 2000.0, 1.0 / 2000.0) - 15.0));
    } else {
        const float s0 = (float)b0 / 60000.0f;
        const float s1 = (float)b1 / 60000.0f;
        x = (float)((double)(fi + s0) * 2.0 / (double)param[4] + (double)param[6]);
        y = (float)((double)(fj + s1) * 2.0 / (double)param[5] + (double)param[7]);
        d = (float)((double)param[3] + (((double)(float)b2 / 2000.0) - 15.0));
    }
    o.v.x = (x * param[8] + y * param[11]) + d * param[0];
    o.v.y = (x * param[9] + y * param[12]) + d * param[1];
    o.v.z = (x * param[10] + y * param[13]) + d * param[2];
    o.c.x = unorm(r);
    o.c.y = unorm(g);
    o.c.z = unorm(b);
    o.c.w = 1.0f;
    if (P.fast) {
        // |v| < 2^123 under the bounds of load_param: 10 + v is 0 or at least 2^-21 in magnitude, and finite
        o.k.x = (float)quot_rcp(10.0 + (double)o.v.x, 20.0, 1.0 / 20.0);
        o.k.y = (float)quot_rcp(10.0 + (double)o.v.y, 20.0, 1.0 / 20.0);
        o.k.z = (float)quot_rcp(10.0 + (double)o.v.z, 20.0, 1.0 / 20.0);
    } else {
        o.k.x = (float)((10.0 + (double)o.v.x) / 20.0);
        o.k.y = (float)((10.0 + (double)o.v.y) / 20.0);
        o.k.z = (float)((10.0 + (double)o.v.z) / 20.0);
    }
    o.k.w = unorm(mk);
    if (mk < 11 && mk != 1) {                                   // Primitive.cu:47: empty cell -> zeros (branch-free select)
        o.v.x = o.v.y = o.v.z = 0.0f;
        o.c.x = o.c.y = o.c.z = o.c.w = 0.0f;
        o.k = o.c;
    }
    return o;
}
__device__ __forceinline__ BumpParam load_param(const float *__restrict__ param) {
    BumpParam P;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        P.p[k] = param[k];
        ok = ok && __builtin_fabsf(P.p[k]) <= 0x1p60f;          // finite and bounded (NaN fails the comparison)
    }
    ok = ok && __builtin_fabsf(P.p[4]) >= 0x1p-40f && __builtin_fabsf(P.p[5]) >= 0x1p-40f;
    P.fast = ok;
    P.inv4 = 1.0 / (double)P.p[4];
    P.inv5 = 1.0 / (double)P.p[5];
    return P;
}

// ---- D1/Primitive.cu:267-357 UpdateKernel: claim key and cell geometry (shared by k_update and the merge+claim fusion)
// Winner rule replacing the data race of UpdateKernel (D1/Primitive.cu:309-349), see DESIGN.md:
// all threads decide against the SNAPSHOT of the cell mask taken before any write; a gate-passing
// writer beats an eroder; among writers the highest mask_curr wins, ties go to the lowest frame
// pixel index; with only eroders on a cell a single erode step is applied.
// key = stamp(32) | writer(1) | mask_curr(8) | ~idx(23)
__device__ __forceinline__ unsigned long long claim_key(uint32_t stamp, bool writer, uint8_t mc, uint32_t idx) {
    uint32_t lo = ((uint32_t)(writer ? 1u : 0u) << 31) | ((uint32_t)mc << 23) | (0x7FFFFFu - idx);
    return ((unsigned long long)stamp << 32) | lo;
}

struct UpdCell { int L, cell; float sa, sb, err; };

// geometry of D1/Primitive.cu:278-307 for a pixel whose plane parameters Q are known
__device__ __forceinline__ bool update_cell(const F3 &p, const float *__restrict__ Q, int N_prim, int M_prim, UpdCell &o) {
    o.err = ((p.x * Q[0] + p.y * Q[1]) + p.z * Q[2]) - Q[3];
    float a = (p.x * Q[8] + p.y * Q[9]) + p.z * Q[10];
    float b = (p.x * Q[11] + p.y * Q[12]) + p.z * Q[13];
    a = (float)((double)((a - Q[6]) * Q[4]) / 2.0);
    b = (float)((double)((b - Q[7]) * Q[5]) / 2.0);
    int ii = f2i_rd(a), jj = f2i_rd(b);
    if (ii > N_prim - 1 || jj > M_prim - 1 || ii < 0 || jj < 0) return false;
    o.cell = ii * M_prim + jj; o.sa = a - (float)ii; o.sb = b - (float)jj;
    return true;
}

#define HALO_W (TILE_W + 2)
#define HALO_H (TILE_H + 2)
#define HALO_N (HALO_W * HALO_H)

// ---- OffLineFrame::LoadFrame (D1/Frame.cpp:43-77) + gpu_cpy + 2 x cv::gpu::Sobel (D1/RG.cpp:807-810) --
// one 64x4 tile (bx, by); shared by k_frame_preprocess and by the launch that appends the next frame's
// preprocessing to the Bump update (k_update_apply_preprocess)
struct PreprocessArgs {
    const uint16_t *depth; size_t dpitch;
    const uint8_t *bgr; size_t bpitch;
    float *VMap, *NMap, *RGB, *gx; size_t gx_pitch; float *gy; size_t gy_pitch;
    CalibV c;
    int n, m;
    float *aux;                                       // optional: the frame's auxiliary ICP records (rgbdslam_set_frame_aux), f32x4 per pixel
};
__device__ __forceinline__ void frame_preprocess_tile(const PreprocessArgs &a, const int bx, const int by) {
    const uint16_t *__restrict__ depth = a.depth; const size_t dpitch = a.dpitch;
    const uint8_t *__restrict__ bgr = a.bgr; const size_t bpitch = a.bpitch;
    float *__restrict__ VMap = a.VMap, *__restrict__ NMap = a.NMap, *__restrict__ RGB = a.RGB;
    float *__restrict__ gx = a.gx, *__restrict__ gy = a.gy;
    const size_t gx_pitch = a.gx_pitch, gy_pitch = a.gy_pitch;
    const CalibV &c = a.c;
    const int n = a.n, m = a.m;
    float *__restrict__ aux = a.aux;
    __shared__ float sv[HALO_H][HALO_W][3];   // vertices of the tile + halo
    __shared__ float si[HALO_H][HALO_W];      // intensities, halo by reflect-101
    __shared__ float sic[TILE_H][TILE_W];     // own pixels: the intensity in the ICP's order of additions (ICP.cu:167)
    const int j0 = bx * TILE_W, i0 = by * TILE_H;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int e = tid; e < HALO_N; e += TILE_W * TILE_H) {
        int r = e / HALO_W, cc = e % HALO_W;
        int ii = i0 + r - 1, jj = j0 + cc - 1;
        bool in_img = ii >= 0 && ii < n && jj >= 0 && jj < m;
        F3 v; v.x = v.y = v.z = 0.0f;
        if (in_img) v = vertex_from_depth(((const uint16_t *)((const char *)depth + (size_t)(n - ii - 1) * dpitch))[jj], ii, jj, c);
        sv[r][cc][0] = v.x; sv[r][cc][1] = v.y; sv[r][cc][2] = v.z;
        float inten = 0.0f;
        if (ii <= n && jj <= m) {   // one past the edge is still needed (reflected)
            int si_ = reflect101(ii, n), sj = reflect101(jj, m);
            const uint8_t *px = bgr + (size_t)(n - si_ - 1) * bpitch + 3 * (size_t)sj;
            // (float)((double)x / 255.0) == x / 255.0f: double rounding is innocuous for division (53 >= 2*24+2)
            float fr = div255((float)px[2]), fg = div255((float)px[1]), fb = div255((float)px[0]);       // Kernel.cu:749-751
            inten = div3_unit((fb + fg) + fr);                                                         // Kernel.cu:678
            if (in_img && r >= 1 && r <= TILE_H && cc >= 1 && cc <= TILE_W) {
                size_t idx = (size_t)ii * m + jj;
                st3(RGB, idx, fr, fg, fb);
                st3(VMap, idx, v.x, v.y, v.z);
                if (aux) sic[r - 1][cc - 1] = div3_unit((fr + fg) + fb);   // (r + g) + b over 3: the same sums, another order
            }
        }
        si[r][cc] = inten;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x, i = i0 + threadIdx.y;
    if (i >= n || j >= m) return;
    const int r = threadIdx.y + 1, cc = threadIdx.x + 1;
    size_t idx = (size_t)i * m + j;
    // normals (Frame.cpp:71, inverse = true)
    F3 nr; nr.x = nr.y = nr.z = 0.0f;
    F3 p; p.x = sv[r][cc][0]; p.y = sv[r][cc][1]; p.z = sv[r][cc][2];
    if (!(i < 1 || i > n - 2 || j < 1 || j > m - 2 || p.z == 0.0f)) {
        F3 dn, up, rt, lf;
        dn.x = sv[r + 1][cc][0]; dn.y = sv[r + 1][cc][1]; dn.z = sv[r + 1][cc][2];
        up.x = sv[r - 1][cc][0]; up.y = sv[r - 1][cc][1]; up.z = sv[r - 1][cc][2];
        rt.x = sv[r][cc + 1][0]; rt.y = sv[r][cc + 1][1]; rt.z = sv[r][cc + 1][2];
        lf.x = sv[r][cc - 1][0]; lf.y = sv[r][cc - 1][1]; lf.z = sv[r][cc - 1][2];
        nr = normal_from_cross(p, dn, up, rt, lf, 1);
    }
    st3(NMap, idx, nr.x, nr.y, nr.z);
    // 3x3 Sobel, separable, the accumulation order of k_intensity_sobel
    float rowd[3], rows[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = si[r - 1 + k][cc - 1], b = si[r - 1 + k][cc], d = si[r - 1 + k][cc + 1];
        float s = 0.0f;
        s = s + a * -1.0f; s = s + b * 0.0f; s = s + d * 1.0f;
        rowd[k] = s;
        s = 0.0f;
        s = s + a * 1.0f; s = s + b * 2.0f; s = s + d * 1.0f;
        rows[k] = s;
    }
    float s = 0.0f;
    s = s + rowd[0] * 1.0f; s = s + rowd[1] * 2.0f; s = s + rowd[2] * 1.0f;
    *(float *)((char *)gx + (size_t)i * gx_pitch + 4 * (size_t)j) = s;
    const float sgx = s;
    s = 0.0f;
    s = s + rows[0] * -1.0f; s = s + rows[1] * 0.0f; s = s + rows[2] * 1.0f;
    *(float *)((char *)gy + (size_t)i * gy_pitch + 4 * (size_t)j) = s;
    if (aux) {
        // what MatchKernelGaussNewton takes from this pixel when a model point projects onto it (ICP.cu:167-173), formed once
        // here from the values just stored: gradients / 8 (in double, as there), intensity, depth weight
        float4 x;
        x.x = (float)((double)sgx / 8.0);
        x.y = (float)((double)s / 8.0);
        x.z = sic[threadIdx.y][threadIdx.x];
        x.w = icp_depth_weight(p.z);
        *(float4 *)(aux + 4 * idx) = x;
    }
}

// ---- D1/Primitive.cu:267-357 UpdateKernel under the winner rule: one tile ---------------------------
// The per-pixel work is a chain of dependent gathers (label -> plane tables -> cell mask -> claim word -> stores).
// A wavefront is 64 consecutive pixels of one image row and almost always sees ONE label: then the plane's table
// entries and its 14 parameters are fetched once per wave through a wave-uniform address (scalar loads) instead of
// once per lane, which takes two vector-memory round trips out of the chain.
struct UpdateArgs {
    uint16_t *const *BumpTab; uint8_t *const *RGBTab; uint8_t *const *MaskTab; float *const *ParamTab;
    unsigned long long *const *ClaimTab;
    uint32_t stamp;
    const float *VMap, *RGBD; const uint8_t *Mask, *Label;
    const uint16_t *Label16;                          // non-null: 16-bit labels (models of more than 255 planes), Label unused
    PoseV P;
    const rgbdslam_pose_block *pb;
    const float *hist;
    int N_prim, M_prim, n, m, plane_lo, plane_hi;
    // "thin" frame source (plane-partitioned multi-GPU): W non-null -> the vertex is VertexMapGL's function of the
    // model depth W (Kernel.cu:54-72: what Merge left in VMap) and the colour the frame's byte / 255 (Kernel.cu:463-465, 739-753:
    // what Merge left in RGBD.rgb); VMap / RGBD are then unused.  Same operations on the same operands: the same bits.
    const float *W; const uint8_t *RGB8; CalibV c;
};
// one 64x4 tile (bx, by) of UpdateKernel; APPLY = second pass of the winner rule
template <bool APPLY, bool THIN = false>
__device__ __forceinline__ void update_tile(const UpdateArgs &a, const int bx, const int by) {
    uint16_t *const *__restrict__ BumpTab = a.BumpTab; uint8_t *const *__restrict__ RGBTab = a.RGBTab;
    uint8_t *const *__restrict__ MaskTab = a.MaskTab; float *const *__restrict__ ParamTab = a.ParamTab;
    unsigned long long *const *__restrict__ ClaimTab = a.ClaimTab;
    const uint32_t stamp = a.stamp;
    const float *__restrict__ VMap = a.VMap, *__restrict__ RGBD = a.RGBD;
    const uint8_t *__restrict__ Mask = a.Mask, *__restrict__ Label = a.Label;
    const uint16_t *__restrict__ Label16 = a.Label16;
    PoseV P = a.P;
    const rgbdslam_pose_block *__restrict__ pb = a.pb;
    const float *__restrict__ hist = a.hist;
    const int N_prim = a.N_prim, M_prim = a.M_prim, n = a.n, m = a.m, plane_lo = a.plane_lo, plane_hi = a.plane_hi;
    if (pb) {                                         // fused pipeline: pose from the device block, gated on it
        if (pb->status[0] != 1) return;
#pragma unroll
        for (int k = 0; k < 16; k++) P.v[k] = pb->P_glob[k];
    } else if (hist) {                                // the frame's immutable history entry: poseR(9) poset(3) tracked ...
        if (hist[12] != 1.0f) return;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            P.v[4 * c] = hist[3 * c]; P.v[4 * c + 1] = hist[3 * c + 1]; P.v[4 * c + 2] = hist[3 * c + 2]; P.v[4 * c + 3] = 0.0f;
        }
        P.v[12] = hist[9]; P.v[13] = hist[10]; P.v[14] = hist[11]; P.v[15] = 1.0f;
    }
    int j = bx * TILE_W + threadIdx.x, i = by * TILE_H + threadIdx.y;
    const bool in = i < n && j < m;
    const size_t idx = in ? (size_t)i * m + j : 0;
    int L = in ? (Label16 ? (int)Label16[idx] : (int)Label[idx]) - 1 : -1;
    const bool valid = L >= 0 && L >= plane_lo && L < plane_hi;
    const unsigned long long vmask = __ballot(valid);
    if (!vmask) return;
    const int L0 = __builtin_amdgcn_readfirstlane(__shfl(L, (int)__builtin_ctzll(vmask)));
    const bool one_plane = __ballot(valid && L != L0) == 0ull;
    if (!valid) return;
    const uint8_t mc = Mask[idx];
    const F3 p = xf_point(P.v, THIN ? vertex_from_w(a.W[idx], i, j, a.c) : ld3(VMap, idx));
    UpdCell u;
    uint8_t *maskb;
    unsigned long long *claimb;
    uint16_t *bumpb;
    uint8_t *rgbb;
    if (one_plane) {                                  // wave-uniform plane: table entries and parameters via uniform addresses
        const float *Q = ParamTab[L0];
        float q[14];
#pragma unroll
        for (int k = 0; k < 14; k++) q[k] = Q[k];
        maskb = MaskTab[L0]; claimb = ClaimTab[L0]; bumpb = BumpTab[L0]; rgbb = RGBTab[L0];
        if (!update_cell(p, q, N_prim, M_prim, u)) return;
        u.L = L0;
    } else {
        if (!update_cell(p, ParamTab[L], N_prim, M_prim, u)) return;
        u.L = L;
        maskb = MaskTab[L]; claimb = ClaimTab[L]; bumpb = BumpTab[L]; rgbb = RGBTab[L];
    }
    uint8_t *cellmask = maskb + u.cell;
    unsigned long long *slot = claimb + u.cell;
    const uint8_t mref = *cellmask;
    unsigned long long owner = 0ull;
    if (APPLY) owner = *slot;                         // independent of the mask test: both gathers are in flight together
    bool erode = (mc < 10 && mref != 1);                                      // Primitive.cu:313
    if (!erode && !(mc > mref)) return;                                       // Primitive.cu:341
    unsigned long long key = claim_key(stamp, !erode, erode ? (uint8_t)0 : mc, (uint32_t)idx);
    if (!APPLY) { atomicMax(slot, key); return; }
    if (owner != key) return;
    uint16_t *B = bumpb + 3 * (size_t)u.cell;
    uint8_t *C = rgbb + 3 * (size_t)u.cell;
    if (erode) {
        if (mref > 11) { *cellmask = (uint8_t)(mref - 1); }
        else { *cellmask = 10; B[0] = 0; B[1] = 0; B[2] = 0; C[0] = 0; C[1] = 0; C[2] = 0; }
    } else {
        float4 col;
        if (THIN) { col.x = unorm8(a.RGB8[3 * idx]); col.y = unorm8(a.RGB8[3 * idx + 1]); col.z = unorm8(a.RGB8[3 * idx + 2]); col.w = 0.0f; }
        else col = *(const float4 *)(RGBD + 4 * idx);
        B[0] = (uint16_t)f2i_rn((float)((double)u.sa * 60000.0));             // Primitive.cu:342-344
        B[1] = (uint16_t)f2i_rn((float)((double)u.sb * 60000.0));
        B[2] = (uint16_t)f2i_rn((float)(((double)u.err + 15.0) * 2000.0));
        C[0] = (uint8_t)f2i_rn((float)((double)col.x * 255.0));
        C[1] = (uint8_t)f2i_rn((float)((double)col.y * 255.0));
        C[2] = (uint8_t)f2i_rn((float)((double)col.z * 255.0));
        *cellmask = mc;
    }
}
