// Linear (DLT) triangulation of point pairs, one point per lane.
//
// Replaces the per-point Python loop of BootstrapInitializer._linear_triangulation
// (vo_pipeline/bootstrap.py:141-156), which _transform_matrix calls four times per
// bootstrap (bootstrap.py:110-124): A = [ [p1]x M1 ; [p2]x M2 ] (6x4), the point is
// the right singular vector of the smallest singular value, stored as float32 and
// divided by its last component.  [p1]x M1 is formed in float32 when M1 is float32
// (as in the reference, where M1 = K @ eye(3,4) is float32) and in float64
// otherwise; the null vector comes from a cyclic Jacobi eigen-decomposition of
// A^T A in float64.
#include "vo_internal.h"

struct M34 { double m[12]; };

__device__ static void jacobi4_min_eigvec(double S[4][4], double *v)
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) off += S[p][q] * S[p][q];
        const double diag = S[0][0] * S[0][0] + S[1][1] * S[1][1] + S[2][2] * S[2][2] + S[3][3] * S[3][3];
        if (!(off > 1e-40 * diag)) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                const double apq = S[p][q];
                if (apq == 0) continue;
                const double theta = (S[q][q] - S[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double skp = S[k][p], skq = S[k][q];
                    S[k][p] = c * skp - s * skq;
                    S[k][q] = s * skp + c * skq;
                }
                for (int k = 0; k < 4; ++k) {
                    const double spk = S[p][k], sqk = S[q][k];
                    S[p][k] = c * spk - s * sqk;
                    S[q][k] = s * spk + c * sqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int best = 0;
    for (int k = 1; k < 4; ++k)
        if (S[k][k] < S[best][best]) best = k;
    for (int k = 0; k < 4; ++k) v[k] = V[k][best];
}

// rows of [p]x M for a homogeneous pixel p (float32 skew entries as in the reference helper)
template <typename T>
__device__ static void cross_rows(const float *p, const double *M, double A[3][4])
{
    const T px = (T)p[0], py = (T)p[1], pz = (T)p[2];
    for (int c = 0; c < 4; ++c) {
        const T m0 = (T)M[c], m1 = (T)M[4 + c], m2 = (T)M[8 + c];
        // [p]x = [[0,-pz,py],[pz,0,-px],[-py,px,0]]; sums in matrix-product order
        A[0][c] = (double)((T)0 * m0 + (-pz) * m1 + py * m2);
        A[1][c] = (double)(pz * m0 + (T)0 * m1 + (-px) * m2);
        A[2][c] = (double)((-py) * m0 + px * m1 + (T)0 * m2);
    }
}

// One pair -> its point in camera 1 (float32 null vector divided by its last component).
__device__ static void triangulate_point(const float *__restrict__ p1, const float *__restrict__ p2, int64_t i, const M34 &M1,
                                         const M34 &M2, int m1_f32, int m2_f32, float P[4])
{
    float a[3], b[3];
    for (int c = 0; c < 3; ++c) { a[c] = p1[3 * i + c]; b[c] = p2[3 * i + c]; }
    double A[6][4];
    if (m1_f32) cross_rows<float>(a, M1.m, A); else cross_rows<double>(a, M1.m, A);
    if (m2_f32) cross_rows<float>(b, M2.m, A + 3); else cross_rows<double>(b, M2.m, A + 3);
    double S[4][4];
    for (int r = 0; r < 4; ++r)
        for (int c = r; c < 4; ++c) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += A[k][r] * A[k][c];
            S[r][c] = s;
            S[c][r] = s;
        }
    double v[4];
    jacobi4_min_eigvec(S, v);
    const float f0 = (float)v[0], f1 = (float)v[1], f2 = (float)v[2], f3 = (float)v[3];
    P[0] = f0 / f3;
    P[1] = f1 / f3;
    P[2] = f2 / f3;
    P[3] = f3 / f3;
}

__global__ __launch_bounds__(64) void triangulate_kernel(const float *__restrict__ p1, const float *__restrict__ p2, int n,
                                                        M34 M1, M34 M2, int m1_f32, int m2_f32, float *__restrict__ out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float P[4];
    triangulate_point(p1, p2, i, M1, M2, m1_f32, m2_f32, P);
    for (int c = 0; c < 4; ++c) out[4 * (int64_t)i + c] = P[c];
}

// Cheirality vote: blockIdx.y = motion hypothesis.  A lane triangulates its pair under that
// hypothesis, stores the point and tests its depth in both cameras; a wavefront adds its two
// ballots' population counts to the hypothesis' counter with one atomic.
#define VO_MAX_MOTIONS 8
struct MotionSet { M34 M2[VO_MAX_MOTIONS]; double row2[VO_MAX_MOTIONS][4]; };

__global__ __launch_bounds__(64) void cheirality_kernel(const float *__restrict__ p1, const float *__restrict__ p2, int n,
                                                       M34 M1, MotionSet set, int m1_f32, int m2_f32,
                                                       float *__restrict__ clouds, int *__restrict__ in_front)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int h = blockIdx.y;
    bool z1 = false, z2 = false;
    if (i < n) {
        const M34 &M2 = set.M2[h];
        float P[4];
        triangulate_point(p1, p2, i, M1, M2, m1_f32, m2_f32, P);
        float *o = clouds + 4 * ((int64_t)h * n + i);
        for (int c = 0; c < 4; ++c) o[c] = P[c];
        // depth in camera 2: third row of [R|t] times the float32 point, in float64
        double d = 0;
        for (int c = 0; c < 4; ++c) d += set.row2[h][c] * (double)P[c];
        z1 = P[2] > 0.0f;
        z2 = d > 0.0;
    }
    const int votes = __popcll(__ballot(z1)) + __popcll(__ballot(z2));
    if (threadIdx.x == 0 && votes) atomicAdd(in_front + h, votes);
}

extern "C" int vo_triangulate(vo_ctx *ctx, const float *pts1, const float *pts2, int n, const double *M1, const double *M2,
                              int m1_is_f32, int m2_is_f32, float *out, int mem)
{
    VO_REQUIRE(ctx && M1 && M2, "vo_triangulate: NULL argument");
    VO_REQUIRE(n >= 0, "vo_triangulate: n=%d", n);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_triangulate: bad mem flag %d", mem);
    if (n == 0) return VO_OK;
    VO_REQUIRE(pts1 && pts2 && out, "vo_triangulate: NULL point buffer");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    M34 a, b;
    memcpy(a.m, M1, sizeof(a.m));
    memcpy(b.m, M2, sizeof(b.m));
    const float *d1 = pts1, *d2 = pts2;
    float *d_out = out;
    int rc;
    if (mem == VO_MEM_HOST) {
        void *p;
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * 12, &p))) return rc; d1 = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 12, &p))) return rc; d2 = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)n * 16, &p))) return rc; d_out = (float *)p;
        VO_UPLOAD(ctx, (void *)d1, pts1, (size_t)n * 12);
        VO_UPLOAD(ctx, (void *)d2, pts2, (size_t)n * 12);
    }
    hipLaunchKernelGGL(triangulate_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d1, d2, n, a, b, m1_is_f32 ? 1 : 0,
                       m2_is_f32 ? 1 : 0, d_out);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, out, d_out, (size_t)n * 16);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

extern "C" int vo_cheirality_vote(vo_ctx *ctx, const float *pts1, const float *pts2, int n, const double *M1,
                                  const double *M2, const double *T_hat, int H, int m1_is_f32, int m2_is_f32,
                                  float *clouds, int32_t *in_front, int mem)
{
    VO_REQUIRE(ctx && M1 && M2 && T_hat && in_front, "vo_cheirality_vote: NULL argument");
    VO_REQUIRE(n >= 0 && H >= 1 && H <= VO_MAX_MOTIONS, "vo_cheirality_vote: n=%d, %d motion hypotheses (1..%d)", n, H,
               VO_MAX_MOTIONS);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_cheirality_vote: bad mem flag %d", mem);
    VO_REQUIRE(n == 0 || (pts1 && pts2 && clouds), "vo_cheirality_vote: NULL point buffer");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    int rc;
    void *p;
    // the hypotheses ride in the kernel-argument segment (1 KB)
    MotionSet set;
    memset(&set, 0, sizeof(set));
    for (int h = 0; h < H; ++h) {
        memcpy(set.M2[h].m, M2 + 12 * h, sizeof(double) * 12);
        memcpy(set.row2[h], T_hat + 12 * h + 8, sizeof(double) * 4);
    }
    int32_t *d_cnt = in_front;
    const float *d1 = pts1, *d2 = pts2;
    float *d_out = clouds;
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * 12, &p))) return rc; d1 = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 12, &p))) return rc; d2 = (const float *)p;
        // counters sit in front of the clouds so that one copy brings both back
        if ((rc = vo_stage_reserve(ctx, 2, 64 + (size_t)H * n * 16, &p))) return rc;
        d_cnt = (int32_t *)p;
        d_out = (float *)((char *)p + 64);
        if (n) {
            VO_UPLOAD(ctx, (void *)d1, pts1, (size_t)n * 12);
            VO_UPLOAD(ctx, (void *)d2, pts2, (size_t)n * 12);
        }
    }
    VO_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * H, ctx->stream));
    M34 a;
    memcpy(a.m, M1, sizeof(a.m));
    if (n) {
        hipLaunchKernelGGL(cheirality_kernel, dim3((n + 63) / 64, H), dim3(64), 0, ctx->stream, d1, d2, n, a, set,
                           m1_is_f32 ? 1 : 0, m2_is_f32 ? 1 : 0, d_out, d_cnt);
        VO_HIP_CHECK(hipGetLastError());
    }
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, in_front, d_cnt, sizeof(int32_t) * H);
        if (n) VO_RESULT(ctx, clouds, d_out, (size_t)H * n * 16);
        VO_RESULTS_FINISH(ctx);      // the call's one synchronisation
    }
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_twoview(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)triangulate_kernel));
    return VO_OK;
}
