// Bundle-adjustment residual, analytic Jacobian blocks and normal equations,
// one observation per lane, float64.
//
// Residual: BundleAdjustment._rotate/_reproject/_fun of the reference
// (vo_pipeline/bundleAdjustment.py:14-45), including the theta == 0 branch
// (v = 0, point unchanged).  The reference hands scipy a finite-difference
// Jacobian through the sparsity of bundleAdjustment.py:47-67 and never forms
// J^T J; here the 2x6 camera and 2x3 point blocks are analytic and the normal
// equations are accumulated on the fly:
//   Hcc (F,6,6) += Jc^T Jc   Hpp (P,3,3) += Jp^T Jp   Hcp (O,6,3) = Jc^T Jp
//   g (6F+3P) = J^T r        cost = 0.5 sum r^2 (all observations)
// Observations i < obs_start (those of the frozen first cameras) keep their
// residual but contribute no Jacobian rows, as in the reference's sparsity.
//
// HBM-bound by construction (~200 B per observation, dominated by the Hcp
// store); camera blocks are reduced across the wave with DPP before one atomic
// per wave and entry, point blocks use per-lane float64 atomics.
#include "vo_internal.h"

#define BA_THREADS 256

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v)
{
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes; valid in lane 63.
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<0xB1, 0xf>(v);
    v += dpp_d<0x4E, 0xf>(v);
    v += dpp_d<0x141, 0xf>(v);
    v += dpp_d<0x140, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    return v;
}

struct BaK { double k[9]; };

template <bool WITH_J>
__device__ __forceinline__ void ba_obs(const double *cam, const double *X, const BaK &K, const double *ob,
                                       double *r, double *Jc, double *Jp)
{
    const double rx = cam[0], ry = cam[1], rz = cam[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    double vx = 0, vy = 0, vz = 0;
    if (theta != 0) { vx = rx / theta; vy = ry / theta; vz = rz / theta; }
    const double dot = X[0] * vx + X[1] * vy + X[2] * vz;
    double s, c;
    sincos(theta, &s, &c);
    const double cxx = vy * X[2] - vz * X[1], cxy = vz * X[0] - vx * X[2], cxz = vx * X[1] - vy * X[0];
    const double fx_ = c * X[0] + s * cxx + dot * (1 - c) * vx;
    const double fy_ = c * X[1] + s * cxy + dot * (1 - c) * vy;
    const double fz_ = c * X[2] + s * cxz + dot * (1 - c) * vz;
    const double Xc = fx_ + cam[3], Yc = fy_ + cam[4], Zc = fz_ + cam[5];
    const double p0 = K.k[0] * Xc + K.k[1] * Yc + K.k[2] * Zc;
    const double p1 = K.k[3] * Xc + K.k[4] * Yc + K.k[5] * Zc;
    const double p2 = K.k[6] * Xc + K.k[7] * Yc + K.k[8] * Zc;
    r[0] = p0 / p2 - ob[0];
    r[1] = p1 / p2 - ob[1];
    if (!WITH_J) return;

    const double ip2 = 1.0 / p2;
    const double a0 = ip2, a2 = -p0 * ip2 * ip2, b1 = ip2, b2 = -p1 * ip2 * ip2;
    double G[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        G[0][k] = a0 * K.k[k] + a2 * K.k[6 + k];
        G[1][k] = b1 * K.k[3 + k] + b2 * K.k[6 + k];
    }
    const double v[3] = {vx, vy, vz};
    double R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = (1 - c) * v[i] * v[j] + (i == j ? c : 0);
    R[0][1] -= s * vz; R[0][2] += s * vy;
    R[1][0] += s * vz; R[1][2] -= s * vx;
    R[2][0] -= s * vy; R[2][1] += s * vx;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) Jp[a * 3 + j] = G[a][0] * R[0][j] + G[a][1] * R[1][j] + G[a][2] * R[2][j];

    double D[3][3];
    if (theta == 0) {
        D[0][0] = 0; D[0][1] = X[2]; D[0][2] = -X[1];
        D[1][0] = -X[2]; D[1][1] = 0; D[1][2] = X[0];
        D[2][0] = X[1]; D[2][1] = -X[0]; D[2][2] = 0;
    } else {
        const double cr[3] = {cxx, cxy, cxz};
        double dth[3], B[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) dth[i] = -s * X[i] + c * cr[i] + s * dot * v[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) B[i][j] = (1 - c) * (v[i] * X[j] + (i == j ? dot : 0));
        B[0][1] += s * X[2]; B[0][2] -= s * X[1];
        B[1][0] -= s * X[2]; B[1][2] += s * X[0];
        B[2][0] += s * X[1]; B[2][1] -= s * X[0];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double Bv = B[i][0] * v[0] + B[i][1] * v[1] + B[i][2] * v[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) D[i][j] = dth[i] * v[j] + (B[i][j] - Bv * v[j]) / theta;
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jc[a * 6 + j] = G[a][0] * D[0][j] + G[a][1] * D[1][j] + G[a][2] * D[2][j];
            Jc[a * 6 + 3 + j] = G[a][j];
        }
}

template <bool WITH_J>
__global__ __launch_bounds__(BA_THREADS) void ba_kernel(const double *__restrict__ params, int F, int P,
                                                        const int64_t *__restrict__ ci, const int64_t *__restrict__ pi,
                                                        const double *__restrict__ obs, int O, BaK K, int obs_start,
                                                        double *__restrict__ r, double *__restrict__ cost,
                                                        double *__restrict__ Hcc, double *__restrict__ Hpp,
                                                        double *__restrict__ Hcp, double *__restrict__ g)
{
    const int i = blockIdx.x * BA_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < O;
    int64_t c = 0, p = 0;
    bool ok = false;
    if (valid) {
        c = ci[i]; p = pi[i];
        ok = c >= 0 && c < F && p >= 0 && p < P;   // out-of-range indices never touch memory
    }
    double rr[2] = {0, 0}, Jc[12], Jp[6];
    const bool active = WITH_J && ok && i >= obs_start;
    if (ok) {
        double cam[6], X[3], ob[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) cam[k] = params[6 * c + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = params[6 * (int64_t)F + 3 * p + k];
        ob[0] = obs[2 * (int64_t)i]; ob[1] = obs[2 * (int64_t)i + 1];
        ba_obs<WITH_J>(cam, X, K, ob, rr, Jc, Jp);
        r[2 * (int64_t)i] = rr[0]; r[2 * (int64_t)i + 1] = rr[1];
    } else if (valid) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        r[2 * (int64_t)i] = qnan; r[2 * (int64_t)i + 1] = qnan;
        rr[0] = qnan;
    }
    // cost: wave sum, one atomic per wave
    double cs = wave_sum_d(rr[0] * rr[0] + rr[1] * rr[1]);
    if (lane == 63) unsafeAtomicAdd(cost, 0.5 * cs);
    if (!WITH_J) return;

    if (!active) {
#pragma unroll
        for (int k = 0; k < 12; ++k) Jc[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) Jp[k] = 0;
    }
    if (valid) {
        double *hx = Hcp + 18 * (int64_t)i;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) hx[a * 3 + b] = Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b];
    }
    // point blocks: per-lane atomics
    if (active) {
        double *hp = Hpp + 9 * p, *gp = g + 6 * (int64_t)F + 3 * p;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) unsafeAtomicAdd(hp + a * 3 + b, Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b]);
            unsafeAtomicAdd(gp + a, Jp[a] * rr[0] + Jp[3 + a] * rr[1]);
        }
    }
    // camera blocks: when every active lane of the wave sees one camera, reduce
    // over the wave first (the caller's observations are sorted by camera)
    const unsigned long long act = __ballot(active);
    if (act == 0) return;
    const int first = __ffsll((long long)act) - 1;
    const long long c0 = __shfl((long long)c, first);
    const bool uniform_cam = __ballot(active && c != c0) == 0;
    if (uniform_cam) {
        double *hc = Hcc + 36 * c0, *gc = g + 6 * c0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                double s = wave_sum_d(Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
                if (lane == 63) unsafeAtomicAdd(hc + a * 6 + b, s);
            }
            double s = wave_sum_d(Jc[a] * rr[0] + Jc[6 + a] * rr[1]);
            if (lane == 63) unsafeAtomicAdd(gc + a, s);
        }
    } else if (active) {
        double *hc = Hcc + 36 * c, *gc = g + 6 * c;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 6; ++b) unsafeAtomicAdd(hc + a * 6 + b, Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
            unsafeAtomicAdd(gc + a, Jc[a] * rr[0] + Jc[6 + a] * rr[1]);
        }
    }
}

template <bool WITH_J>
static int ba_common(vo_ctx *ctx, const double *params, int F, int P, const int64_t *ci, const int64_t *pi,
                     const double *obs, int O, const double *K, int obs_start, double *r, double *cost, double *Hcc,
                     double *Hpp, double *Hcp, double *g, int mem, const char *who)
{
    VO_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    VO_REQUIRE(F >= 0 && P >= 0 && O >= 0, "%s: F=%d P=%d O=%d", who, F, P, O);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "%s: bad mem flag %d", who, mem);
    VO_REQUIRE(K != nullptr && cost != nullptr, "%s: NULL K/cost", who);
    VO_REQUIRE(O == 0 || (params && ci && pi && obs && r), "%s: NULL input buffer", who);
    VO_REQUIRE(!WITH_J || (Hcc && Hpp && g && (O == 0 || Hcp)), "%s: NULL output buffer", who);
    VO_REQUIRE(obs_start >= 0, "%s: obs_start=%d", who, obs_start);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t np_ = 6 * (size_t)F + 3 * (size_t)P;
    int rc;
    BaK Kv;
    memcpy(Kv.k, K, sizeof(Kv.k));
    const double *d_params = params, *d_obs = obs;
    const int64_t *d_ci = ci, *d_pi = pi;
    double *d_r = r, *d_cost = cost, *d_Hcc = Hcc, *d_Hpp = Hpp, *d_Hcp = Hcp, *d_g = g;
    if (mem == VO_MEM_HOST) {
        for (int i = 0; i < O; ++i)
            VO_REQUIRE(ci[i] >= 0 && ci[i] < F && pi[i] >= 0 && pi[i] < P, "%s: observation %d has index out of range (camera %lld, point %lld)",
                       who, i, (long long)ci[i], (long long)pi[i]);
        // one staging block: params | obs | ci | pi | r | cost | Hcc | Hpp | g | Hcp
        size_t off[11], tot = 0;
        const size_t sz[10] = {np_ * 8, (size_t)O * 16, (size_t)O * 8, (size_t)O * 8, (size_t)O * 16, 8,
                               WITH_J ? (size_t)F * 288 : 0, WITH_J ? (size_t)P * 72 : 0, WITH_J ? np_ * 8 : 0,
                               WITH_J ? (size_t)O * 144 : 0};
        for (int k = 0; k < 10; ++k) { off[k] = tot; tot += (sz[k] + 255) / 256 * 256; }
        void *base;
        if ((rc = vo_stage_reserve(ctx, 0, tot, &base))) return rc;
        char *b = (char *)base;
        d_params = (double *)(b + off[0]); d_obs = (double *)(b + off[1]);
        d_ci = (int64_t *)(b + off[2]); d_pi = (int64_t *)(b + off[3]);
        d_r = (double *)(b + off[4]); d_cost = (double *)(b + off[5]);
        d_Hcc = (double *)(b + off[6]); d_Hpp = (double *)(b + off[7]); d_g = (double *)(b + off[8]);
        d_Hcp = (double *)(b + off[9]);
        if (np_) VO_HIP_CHECK(hipMemcpyAsync((void *)d_params, params, np_ * 8, hipMemcpyHostToDevice, ctx->stream));
        if (O) {
            VO_HIP_CHECK(hipMemcpyAsync((void *)d_obs, obs, (size_t)O * 16, hipMemcpyHostToDevice, ctx->stream));
            VO_HIP_CHECK(hipMemcpyAsync((void *)d_ci, ci, (size_t)O * 8, hipMemcpyHostToDevice, ctx->stream));
            VO_HIP_CHECK(hipMemcpyAsync((void *)d_pi, pi, (size_t)O * 8, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    VO_HIP_CHECK(hipMemsetAsync(d_cost, 0, 8, ctx->stream));
    if (WITH_J) {
        if (F) VO_HIP_CHECK(hipMemsetAsync(d_Hcc, 0, (size_t)F * 288, ctx->stream));
        if (P) VO_HIP_CHECK(hipMemsetAsync(d_Hpp, 0, (size_t)P * 72, ctx->stream));
        if (np_) VO_HIP_CHECK(hipMemsetAsync(d_g, 0, np_ * 8, ctx->stream));
    }
    if (O) {
        const unsigned blocks = (unsigned)((O + BA_THREADS - 1) / BA_THREADS);
        hipLaunchKernelGGL(ba_kernel<WITH_J>, dim3(blocks), dim3(BA_THREADS), 0, ctx->stream, d_params, F, P, d_ci, d_pi,
                           d_obs, O, Kv, obs_start, d_r, d_cost, d_Hcc, d_Hpp, d_Hcp, d_g);
        VO_HIP_CHECK(hipGetLastError());
    }
    if (mem == VO_MEM_HOST) {
        if (O) VO_HIP_CHECK(hipMemcpyAsync(r, d_r, (size_t)O * 16, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipMemcpyAsync(cost, d_cost, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (WITH_J) {
            if (F) VO_HIP_CHECK(hipMemcpyAsync(Hcc, d_Hcc, (size_t)F * 288, hipMemcpyDeviceToHost, ctx->stream));
            if (P) VO_HIP_CHECK(hipMemcpyAsync(Hpp, d_Hpp, (size_t)P * 72, hipMemcpyDeviceToHost, ctx->stream));
            if (np_) VO_HIP_CHECK(hipMemcpyAsync(g, d_g, np_ * 8, hipMemcpyDeviceToHost, ctx->stream));
            if (O) VO_HIP_CHECK(hipMemcpyAsync(Hcp, d_Hcp, (size_t)O * 144, hipMemcpyDeviceToHost, ctx->stream));
        }
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return VO_OK;
}

extern "C" int vo_ba_residual(vo_ctx *ctx, const double *params, int F, int P, const int64_t *ci, const int64_t *pi,
                              const double *obs, int O, const double *K, double *r, double *cost, int mem)
{
    return ba_common<false>(ctx, params, F, P, ci, pi, obs, O, K, 0, r, cost, nullptr, nullptr, nullptr, nullptr, mem,
                            "vo_ba_residual");
}

extern "C" int vo_ba_normal_eq(vo_ctx *ctx, const double *params, int F, int P, const int64_t *ci, const int64_t *pi,
                               const double *obs, int O, const double *K, int obs_start, double *r, double *cost,
                               double *Hcc, double *Hpp, double *Hcp, double *g, int mem)
{
    return ba_common<true>(ctx, params, F, P, ci, pi, obs, O, K, obs_start, r, cost, Hcc, Hpp, Hcp, g, mem,
                           "vo_ba_normal_eq");
}
