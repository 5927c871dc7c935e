// Minimal solvers for the RANSAC drivers, one sample per lane, float64:
//   * P3P (Grunert's quartic, as reviewed by Haralick et al.) on three 2D-3D pairs,
//     disambiguated by the reprojection of a fourth -- the model generator behind
//     cv.solvePnPRansac(flags=SOLVEPNP_P3P), vo_pipeline/poseEstimation.py:61-69;
//   * the 7-point algorithm (null space of the 7x9 epipolar system, cubic in the
//     pencil parameter, 1..3 real solutions) -- the model generator behind
//     cv2.findFundamentalMat(FM_RANSAC), vo_pipeline/bootstrap.py:246-252.
// These are independent implementations of the same geometry: candidate models
// agree with OpenCV's to solver precision, not bit for bit.
#include "vo_internal.h"
#include <math.h>

#define SOLVER_THREADS 64
#define P3P_LANES 4           // lanes per sample in p3p_kernel, one per root of the quartic

// ---- polynomial helpers ----------------------------------------------------------
// real roots of x^3 + b x^2 + c x + d; returns their number (1 or 3), polished
__device__ static int cubic_roots(double b, double c, double d, double *x)
{
    const double b3 = b / 3.0;
    const double p = c - b * b3, q = 2.0 * b3 * b3 * b3 - b3 * c + d;
    const double disc = 0.25 * q * q + p * p * p / 27.0;
    int n;
    if (disc > 0) {
        const double s = sqrt(disc);
        x[0] = cbrt(-0.5 * q + s) + cbrt(-0.5 * q - s) - b3;
        n = 1;
    } else if (p == 0) {
        x[0] = -b3;
        n = 1;
    } else {
        const double m = 2.0 * sqrt(-p / 3.0);
        double arg = 3.0 * q / (p * m);
        arg = fmin(1.0, fmax(-1.0, arg));
        const double th = acos(arg) / 3.0;
        const double two_pi_3 = 2.0943951023931954923;
        x[0] = m * cos(th) - b3;
        x[1] = m * cos(th - two_pi_3) - b3;
        x[2] = m * cos(th - 2.0 * two_pi_3) - b3;
        n = 3;
    }
    for (int i = 0; i < n; ++i)
        for (int it = 0; it < 2; ++it) {
            const double f = ((x[i] + b) * x[i] + c) * x[i] + d;
            const double df = (3.0 * x[i] + 2.0 * b) * x[i] + c;
            if (df != 0) x[i] -= f / df;
        }
    return n;
}

// real roots of a4 x^4 + ... + a0 (a4 != 0) by Ferrari's factorisation; returns their number.  The roots leave
// unpolished and as roots of the depressed quartic: quartic_polish(x[i]) finishes root i.
__device__ static int quartic_roots(double a4, double a3, double a2, double a1, double a0, double *x)
{
    const double a = a3 / a4, b = a2 / a4, c = a1 / a4, d = a0 / a4;
    // depressed quartic y^4 + p y^2 + q y + r, x = y - a/4
    const double a_4 = 0.25 * a;
    const double p = b - 6.0 * a_4 * a_4;
    const double q = c - 2.0 * b * a_4 + 8.0 * a_4 * a_4 * a_4;
    const double r = d - c * a_4 + b * a_4 * a_4 - 3.0 * a_4 * a_4 * a_4 * a_4;
    int n = 0;
    const double scale = fabs(p) + fabs(r) + 1.0;
    if (fabs(q) < 1e-14 * scale) {
        // biquadratic
        const double disc = p * p - 4.0 * r;
        if (disc >= 0) {
            const double s = sqrt(disc);
            const double y2a = 0.5 * (-p + s), y2b = 0.5 * (-p - s);
            if (y2a >= 0) { x[n++] = sqrt(y2a); x[n++] = -sqrt(y2a); }
            if (y2b >= 0) { x[n++] = sqrt(y2b); x[n++] = -sqrt(y2b); }
        }
    } else {
        // resolvent z^3 + 2p z^2 + (p^2 - 4r) z - q^2 = 0 has a positive real root
        double z[3];
        const int nz = cubic_roots(2.0 * p, p * p - 4.0 * r, -q * q, z);
        double zz = z[0];
        for (int i = 1; i < nz; ++i) zz = fmax(zz, z[i]);
        if (zz > 0) {
            const double beta = sqrt(zz), alpha = 0.5 * (p + zz), gamma = -q / (2.0 * beta);
            const double d1 = zz - 4.0 * (alpha + gamma), d2 = zz - 4.0 * (alpha - gamma);
            if (d1 >= 0) { const double s = sqrt(d1); x[n++] = 0.5 * (-beta + s); x[n++] = 0.5 * (-beta - s); }
            if (d2 >= 0) { const double s = sqrt(d2); x[n++] = 0.5 * (beta + s); x[n++] = 0.5 * (beta - s); }
        }
    }
    return n;
}

// A root y of the depressed quartic as a root of the quartic itself, x = y - a3 / (4 a4), after three Newton steps.
__device__ static double quartic_polish(double a4, double a3, double a2, double a1, double a0, double y)
{
    double v = y - 0.25 * (a3 / a4);
    for (int it = 0; it < 3; ++it) {
        const double f = (((a4 * v + a3) * v + a2) * v + a1) * v + a0;
        const double df = ((4.0 * a4 * v + 3.0 * a3) * v + 2.0 * a2) * v + a1;
        if (df != 0) v -= f / df;
    }
    return v;
}

__device__ static inline void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ static inline bool normalize3(double *a)
{
    const double n = sqrt(dot3(a, a));
    if (!(n > 0)) return false;
    a[0] /= n; a[1] /= n; a[2] /= n;
    return true;
}

// orthonormal triad (columns e1, e2, e3) spanned by three points
__device__ static bool triad(const double *A0, const double *A1, const double *A2, double T[3][3])
{
    double e1[3] = {A1[0] - A0[0], A1[1] - A0[1], A1[2] - A0[2]};
    double w[3] = {A2[0] - A0[0], A2[1] - A0[1], A2[2] - A0[2]};
    double e3[3], e2[3];
    if (!normalize3(e1)) return false;
    cross3(e1, w, e3);
    if (!normalize3(e3)) return false;
    cross3(e3, e1, e2);
    for (int i = 0; i < 3; ++i) { T[i][0] = e1[i]; T[i][1] = e2[i]; T[i][2] = e3[i]; }
    return true;
}

struct K9 { double k[9]; };

// subsets: (S,4) int32 indices into X/x.  Rt: (S,12) [R|t] of the best of the <= 4
// P3P solutions of points 0..2 as judged by point 3; valid: (S) 1 if one exists.
// blockIdx.y: the problem (a stream of a batched frame), whose arrays start bs.work / bs.out bytes after the problem
// before.  Samples below samp_lo are another launch's; with skip_bound, samples beyond *skip_bound (the RANSAC loop's
// iteration bound after an earlier stage, which only shrinks) are marked invalid unsolved.
__global__ __launch_bounds__(SOLVER_THREADS) void p3p_kernel(const float *__restrict__ X, const float *__restrict__ x, int n,
                                                            K9 K, K9 Kinv, const int *__restrict__ subsets, int S,
                                                            double *__restrict__ Rt, uint8_t *__restrict__ valid,
                                                            vo_batch_strides bs, int samp_lo, const int *__restrict__ skip_bound)
{
    // A kernel of the per-frame pose chain: a few short wavefronts that, with many streams in flight, share their SIMDs
    // with another launch group's tracker (five wavefronts issuing back to back).  Raised issue priority lets them through.
    __builtin_amdgcn_s_setprio(3);
    // Four lanes to a sample, one per root of the quartic: every lane forms the quartic and its roots (the same
    // instructions whether one lane or four execute them), then builds and judges the pose of its own root only, where
    // a lane per sample walked up to four roots one after the other, each behind its own tests; the four compare
    // their errors through shuffles.
    const int s = samp_lo + (blockIdx.x * SOLVER_THREADS + threadIdx.x) / P3P_LANES, slot = threadIdx.x % P3P_LANES;
    if (s >= S) return;                      // the lanes of a sample leave together
    {
        const size_t b = blockIdx.y;
        X = (const float *)((const char *)X + b * bs.work);
        x = (const float *)((const char *)x + b * bs.work);
        subsets = (const int *)((const char *)subsets + b * bs.work);
        Rt = (double *)((char *)Rt + b * bs.work);
        valid += b * bs.out;
        if (skip_bound) {
            skip_bound = (const int *)((const char *)skip_bound + b * bs.out);
            if (s > *skip_bound) { if (slot == 0) valid[s] = 0; return; }
        }
    }
    double P[4][3], j[3][3], obs3[2];
    bool ok = true;
    for (int k = 0; k < 4; ++k) {
        const int id = subsets[4 * s + k];
        ok = ok && id >= 0 && id < n;
        const int idc = min(max(id, 0), n - 1);
        for (int c = 0; c < 3; ++c) P[k][c] = X[3 * (int64_t)idc + c];
        const double u = x[2 * (int64_t)idc], v = x[2 * (int64_t)idc + 1];
        if (k < 3) {
            for (int c = 0; c < 3; ++c) j[k][c] = Kinv.k[3 * c] * u + Kinv.k[3 * c + 1] * v + Kinv.k[3 * c + 2];
            ok = normalize3(j[k]) && ok;
        } else {
            obs3[0] = u; obs3[1] = v;
        }
    }
    double best_e = 1e300, bestRt[12];
    bool have = false;
    double d23[3], d13[3], d12[3];
    for (int c = 0; c < 3; ++c) { d23[c] = P[1][c] - P[2][c]; d13[c] = P[0][c] - P[2][c]; d12[c] = P[0][c] - P[1][c]; }
    const double a2 = dot3(d23, d23), b2 = dot3(d13, d13), c2 = dot3(d12, d12);
    if (ok && fmin(a2, fmin(b2, c2)) > 1e-24) {
        const double ca = dot3(j[1], j[2]), cb = dot3(j[0], j[2]), cg = dot3(j[0], j[1]);
        const double q = (a2 - c2) / b2, p = (a2 + c2) / b2;
        const double A4 = (q - 1) * (q - 1) - 4 * c2 / b2 * ca * ca;
        const double A3 = 4 * (q * (1 - q) * cb - (1 - p) * ca * cg + 2 * c2 / b2 * ca * ca * cb);
        const double A2 = 2 * (q * q - 1 + 2 * q * q * cb * cb + 2 * (b2 - c2) / b2 * ca * ca - 4 * p * ca * cb * cg +
                               2 * (b2 - a2) / b2 * cg * cg);
        const double A1 = 4 * (-q * (1 + q) * cb + 2 * a2 / b2 * cg * cg * cb - (1 - p) * ca * cg);
        const double A0 = (1 + q) * (1 + q) - 4 * a2 / b2 * cg * cg;
        double roots[4] = {0.0, 0.0, 0.0, 0.0};
        int nr = 0;
        if (isfinite(A4) && isfinite(A3) && isfinite(A2) && isfinite(A1) && isfinite(A0) && fabs(A4) > 1e-300)
            nr = quartic_roots(A4, A3, A2, A1, A0, roots);
        double Tp[3][3];
        const bool tp_ok = triad(P[0], P[1], P[2], Tp);
        // this lane's root (a quartic has at most four, P3P_LANES of them)
        const double raw = slot == 0 ? roots[0] : slot == 1 ? roots[1] : slot == 2 ? roots[2] : roots[3];
        do {
            if (!(slot < nr && tp_ok)) break;
            const double v = quartic_polish(A4, A3, A2, A1, A0, raw);
            if (!(v > 0)) break;
            const double den = 2 * (cg - v * ca);
            if (fabs(den) < 1e-14) break;
            const double u = ((q - 1) * v * v - 2 * q * cb * v + 1 + q) / den;
            const double dd = 1 + v * v - 2 * v * cb;
            if (!(u > 0) || !(dd > 0)) break;
            const double s1 = sqrt(b2 / dd), s2 = u * s1, s3 = v * s1;
            double Q[3][3];
            for (int c = 0; c < 3; ++c) { Q[0][c] = s1 * j[0][c]; Q[1][c] = s2 * j[1][c]; Q[2][c] = s3 * j[2][c]; }
            double Tq[3][3];
            if (!triad(Q[0], Q[1], Q[2], Tq)) break;
            double R[3][3], t[3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) R[r][c] = Tq[r][0] * Tp[c][0] + Tq[r][1] * Tp[c][1] + Tq[r][2] * Tp[c][2];
            for (int r = 0; r < 3; ++r) t[r] = Q[0][r] - (R[r][0] * P[0][0] + R[r][1] * P[0][1] + R[r][2] * P[0][2]);
            double pc[3];
            for (int r = 0; r < 3; ++r) pc[r] = R[r][0] * P[3][0] + R[r][1] * P[3][1] + R[r][2] * P[3][2] + t[r];
            if (!(pc[2] > 0)) break;
            const double w0 = K.k[0] * pc[0] + K.k[1] * pc[1] + K.k[2] * pc[2];
            const double w1 = K.k[3] * pc[0] + K.k[4] * pc[1] + K.k[5] * pc[2];
            const double w2 = K.k[6] * pc[0] + K.k[7] * pc[1] + K.k[8] * pc[2];
            const double eu = w0 / w2 - obs3[0], ev = w1 / w2 - obs3[1];
            const double e = eu * eu + ev * ev;
            if (isfinite(e) && e < best_e) {
                best_e = e;
                have = true;
                for (int r = 0; r < 3; ++r) { bestRt[4 * r] = R[r][0]; bestRt[4 * r + 1] = R[r][1]; bestRt[4 * r + 2] = R[r][2]; bestRt[4 * r + 3] = t[r]; }
            }
        } while (false);
    }
    // the smallest error of the four wins, the earlier root on equal errors: what walking the roots in order and
    // keeping a strictly smaller error gives
    double win_e = best_e;
    int win = have ? slot : P3P_LANES;
#pragma unroll
    for (int d = 1; d < P3P_LANES; d <<= 1) {
        const double other_e = __shfl_xor(win_e, d);
        const int other = __shfl_xor(win, d);
        if (other < P3P_LANES && (win == P3P_LANES || other_e < win_e || (other_e == win_e && other < win))) { win_e = other_e; win = other; }
    }
    if (win == P3P_LANES) {
        if (slot == 0) {
            for (int k = 0; k < 12; ++k) Rt[12 * (int64_t)s + k] = 0.0;
            valid[s] = 0;
        }
    } else if (win == slot) {
        for (int k = 0; k < 12; ++k) Rt[12 * (int64_t)s + k] = bestRt[k];
        valid[s] = 1;
    }
}

static inline double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ static inline double det3d(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// subsets: (S,7) int32.  F: (S,3,9) up to three fundamental matrices (row-major, scaled to
// F[8] = 1 where possible, zero-filled otherwise); nsol: (S) number of real solutions.
__global__ __launch_bounds__(SOLVER_THREADS) void seven_point_kernel(const float *__restrict__ m1, const float *__restrict__ m2,
                                                                    int n, const int *__restrict__ subsets, int S,
                                                                    double *__restrict__ F, int *__restrict__ nsol)
{
    // The elimination swaps rows and columns chosen at run time, so the 7x9 system cannot live in
    // registers; it is kept in LDS, one 63-double column per lane (odd stride: lanes touching the
    // same element hit different banks), not in scratch memory, whose latency dominated the kernel.
    // The same holds for the column permutation and the two null-space vectors (scattered through it).
    __shared__ double A_lds[SOLVER_THREADS][63 + 18];
    __shared__ int perm_lds[SOLVER_THREADS][9];
    const int s = blockIdx.x * SOLVER_THREADS + threadIdx.x;
    if (s >= S) return;
    double(*A)[9] = reinterpret_cast<double(*)[9]>(A_lds[threadIdx.x]);
    double *f1 = A_lds[threadIdx.x] + 63, *f2 = f1 + 9;
    int *perm = perm_lds[threadIdx.x];
    bool ok = true;
    for (int k = 0; k < 7; ++k) {
        const int id = subsets[7 * s + k];
        ok = ok && id >= 0 && id < n;
        const int idc = min(max(id, 0), n - 1);
        const double x1 = m1[2 * (int64_t)idc], y1 = m1[2 * (int64_t)idc + 1];
        const double x2 = m2[2 * (int64_t)idc], y2 = m2[2 * (int64_t)idc + 1];
        A[k][0] = x2 * x1; A[k][1] = x2 * y1; A[k][2] = x2;
        A[k][3] = y2 * x1; A[k][4] = y2 * y1; A[k][5] = y2;
        A[k][6] = x1; A[k][7] = y1; A[k][8] = 1.0;
    }
    // Gauss-Jordan with complete pivoting; perm[c] = original column at position c
    for (int c = 0; c < 9; ++c) perm[c] = c;
    for (int r = 0; r < 7 && ok; ++r) {
        int pr = r, pc = r;
        double best = -1;
        for (int i = r; i < 7; ++i)
            for (int c = r; c < 9; ++c)
                if (fabs(A[i][c]) > best) { best = fabs(A[i][c]); pr = i; pc = c; }
        if (!(best > 1e-300)) { ok = false; break; }
        for (int c = 0; c < 9; ++c) { const double t = A[r][c]; A[r][c] = A[pr][c]; A[pr][c] = t; }
        for (int i = 0; i < 7; ++i) { const double t = A[i][r]; A[i][r] = A[i][pc]; A[i][pc] = t; }
        { const int t = perm[r]; perm[r] = perm[pc]; perm[pc] = t; }
        const double inv = 1.0 / A[r][r];
        for (int c = r; c < 9; ++c) A[r][c] *= inv;
        for (int i = 0; i < 7; ++i) {
            if (i == r) continue;
            const double f = A[i][r];
            if (f != 0) for (int c = r; c < 9; ++c) A[i][c] -= f * A[r][c];
        }
    }
    int count = 0;
    double *out = F + 27 * (int64_t)s;
    if (ok) {
        // null-space basis: free columns 7 and 8 of the permuted system
        for (int c = 0; c < 9; ++c) { f1[c] = 0; f2[c] = 0; }
        for (int r = 0; r < 7; ++r) { f1[perm[r]] = -A[r][7]; f2[perm[r]] = -A[r][8]; }
        f1[perm[7]] = 1.0; f2[perm[8]] = 1.0;
        // det(l f1 + (1-l) f2) = c3 l^3 + c2 l^2 + c1 l + c0 through four evaluations
        double M[9];
        double dv[4];
        const double ls[4] = {0.0, 1.0, -1.0, 2.0};
        for (int e = 0; e < 4; ++e) {
            for (int c = 0; c < 9; ++c) M[c] = ls[e] * f1[c] + (1.0 - ls[e]) * f2[c];
            dv[e] = det3d(M);
        }
        const double c0 = dv[0];
        const double c2 = 0.5 * (dv[1] + dv[2]) - c0;
        const double sm = 0.5 * (dv[1] - dv[2]);
        const double c3 = (dv[3] - c0 - 4.0 * c2 - 2.0 * sm) / 6.0;
        const double c1 = sm - c3;
        double roots[3];
        int nr = 0;
        const double cs = fabs(c0) + fabs(c1) + fabs(c2) + fabs(c3);
        if (isfinite(cs) && cs > 0) {
            if (fabs(c3) > 1e-14 * cs) {
                nr = cubic_roots(c2 / c3, c1 / c3, c0 / c3, roots);
            } else if (fabs(c2) > 1e-14 * cs) {
                const double disc = c1 * c1 - 4.0 * c2 * c0;
                if (disc >= 0) { const double sq = sqrt(disc); roots[0] = (-c1 + sq) / (2 * c2); roots[1] = (-c1 - sq) / (2 * c2); nr = 2; }
            } else if (fabs(c1) > 0) {
                roots[0] = -c0 / c1; nr = 1;
            }
        }
        for (int i = 0; i < nr; ++i) {
            const double l = roots[i];
            if (!isfinite(l)) continue;
            double Fm[9];
            for (int c = 0; c < 9; ++c) Fm[c] = l * f1[c] + (1.0 - l) * f2[c];
            if (fabs(Fm[8]) > 2.220446049250313e-16) { const double inv = 1.0 / Fm[8]; for (int c = 0; c < 9; ++c) Fm[c] *= inv; }
            for (int c = 0; c < 9; ++c) out[9 * count + c] = Fm[c];
            ++count;
        }
    }
    for (int k = 9 * count; k < 27; ++k) out[k] = 0.0;      // unused solution slots
    nsol[s] = count;
}

static bool invert3(const double *m, double *o)
{
    const double d = det3(m);
    if (!(fabs(d) > 0)) return false;
    o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
    return true;
}

extern "C" int vo_p3p_solve(vo_ctx *ctx, const float *X, const float *x, int n, const double *K, const int32_t *subsets,
                            int S, double *Rt, uint8_t *valid, int mem)
{
    VO_REQUIRE(ctx && K, "vo_p3p_solve: NULL argument");
    VO_REQUIRE(n >= 0 && S >= 0, "vo_p3p_solve: n=%d S=%d", n, S);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_p3p_solve: bad mem flag %d", mem);
    if (S == 0) return VO_OK;
    VO_REQUIRE(n >= 4 && X && x && subsets && Rt && valid, "vo_p3p_solve: NULL buffer or fewer than 4 points");
    K9 Kv, Ki;
    memcpy(Kv.k, K, sizeof(Kv.k));
    VO_REQUIRE(invert3(K, Ki.k), "vo_p3p_solve: K is singular");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const float *d_X = X, *d_x = x;
    const int32_t *d_sub = subsets;
    double *d_Rt = Rt;
    uint8_t *d_valid = valid;
    int rc;
    if (mem == VO_MEM_HOST) {
        void *p;
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * 12, &p))) return rc; d_X = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 8, &p))) return rc; d_x = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)S * 16, &p))) return rc; d_sub = (const int32_t *)p;
        if ((rc = vo_stage_reserve(ctx, 3, (size_t)S * 96, &p))) return rc; d_Rt = (double *)p;
        if ((rc = vo_stage_reserve(ctx, 4, (size_t)S, &p))) return rc; d_valid = (uint8_t *)p;
        VO_UPLOAD(ctx, (void *)d_X, X, (size_t)n * 12);
        VO_UPLOAD(ctx, (void *)d_x, x, (size_t)n * 8);
        VO_UPLOAD(ctx, (void *)d_sub, subsets, (size_t)S * 16);
    }
    hipLaunchKernelGGL(p3p_kernel, dim3((P3P_LANES * S + SOLVER_THREADS - 1) / SOLVER_THREADS), dim3(SOLVER_THREADS), 0, ctx->stream, d_X,
                       d_x, n, Kv, Ki, d_sub, S, d_Rt, d_valid, vo_batch_strides(), 0, (const int *)nullptr);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, Rt, d_Rt, (size_t)S * 96);
        VO_RESULT(ctx, valid, d_valid, (size_t)S);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

int vo_p3p_solve_dev(vo_ctx *ctx, const float *dX, const float *dx, int n, const double *K, const int32_t *dsub, int S,
                     double *dRt, uint8_t *dvalid, const vo_batch_strides &bs, int samp_lo, const int *skip_bound_dev)
{
    if (S <= samp_lo || bs.count <= 0) return VO_OK;
    K9 Kv, Ki;
    memcpy(Kv.k, K, sizeof(Kv.k));
    VO_REQUIRE(invert3(K, Ki.k), "vo_p3p_solve: K is singular");
    VO_REQUIRE(bs.count <= 65535, "vo_p3p_solve: %d problems in one launch", bs.count);
    hipLaunchKernelGGL(p3p_kernel, dim3((P3P_LANES * (S - samp_lo) + SOLVER_THREADS - 1) / SOLVER_THREADS, bs.count), dim3(SOLVER_THREADS), 0,
                       ctx->stream, dX, dx, n, Kv, Ki, dsub, S, dRt, dvalid, bs, samp_lo, skip_bound_dev);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

extern "C" int vo_seven_point_solve(vo_ctx *ctx, const float *m1, const float *m2, int n, const int32_t *subsets, int S,
                                    double *F, int32_t *nsol, int mem)
{
    VO_REQUIRE(ctx != nullptr, "vo_seven_point_solve: ctx is NULL");
    VO_REQUIRE(n >= 0 && S >= 0, "vo_seven_point_solve: n=%d S=%d", n, S);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_seven_point_solve: bad mem flag %d", mem);
    if (S == 0) return VO_OK;
    VO_REQUIRE(n >= 7 && m1 && m2 && subsets && F && nsol, "vo_seven_point_solve: NULL buffer or fewer than 7 points");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const float *d_m1 = m1, *d_m2 = m2;
    const int32_t *d_sub = subsets;
    double *d_F = F;
    int32_t *d_n = nsol;
    int rc;
    if (mem == VO_MEM_HOST) {
        void *p;
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * 8, &p))) return rc; d_m1 = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 8, &p))) return rc; d_m2 = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)S * 28, &p))) return rc; d_sub = (const int32_t *)p;
        if ((rc = vo_stage_reserve(ctx, 3, (size_t)S * 216, &p))) return rc; d_F = (double *)p;
        if ((rc = vo_stage_reserve(ctx, 4, (size_t)S * 4, &p))) return rc; d_n = (int32_t *)p;
        VO_UPLOAD(ctx, (void *)d_m1, m1, (size_t)n * 8);
        VO_UPLOAD(ctx, (void *)d_m2, m2, (size_t)n * 8);
        VO_UPLOAD(ctx, (void *)d_sub, subsets, (size_t)S * 28);
    }
    hipLaunchKernelGGL(seven_point_kernel, dim3((S + SOLVER_THREADS - 1) / SOLVER_THREADS), dim3(SOLVER_THREADS), 0, ctx->stream,
                       d_m1, d_m2, n, d_sub, S, d_F, d_n);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, F, d_F, (size_t)S * 216);
        VO_RESULT(ctx, nsol, d_n, (size_t)S * 4);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_solvers(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)p3p_kernel));
    return VO_OK;
}
