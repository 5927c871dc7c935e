// Batched RANSAC hypothesis scoring: every (hypothesis, correspondence) pair is
// independent.  One wavefront scores one hypothesis against a chunk of points:
// the model sits in scalar registers, lanes stride over the points with
// coalesced loads, inliers are counted with a 64-bit ballot + popcount and one
// integer atomic per wave (integer sums are order-independent, so counts and
// masks are bit-reproducible).  float64 arithmetic in source order with
// contraction off, as the CPU routines this replaces.
//
// Replaces the per-candidate scoring loops inside
//   cv2.findFundamentalMat(..., FM_RANSAC, ...)  vo_pipeline/bootstrap.py:246-252
//   cv.solvePnPRansac(..., flags=P3P, ...)       vo_pipeline/poseEstimation.py:61-69
#include "vo_internal.h"

#define SC_WAVES 4
#define SC_CHUNK 1024  // points per wave at most
// Points per wave for a launch that would otherwise put `waves` wavefronts (hypotheses x problems) on the chip: few
// hypotheses of one frame leave most of it idle while each wave walks a thousand points, sixteen dependent rounds of
// loads; their points are then split further, down to two rounds per wave, and the shares added up with atomics.
static int score_chunk(int n, int64_t waves)
{
    int chunk = SC_CHUNK;
    while (chunk > 128 && waves * ((n + chunk - 1) / chunk) < 2048) chunk >>= 1;
    return chunk;
}

__device__ __forceinline__ double std_max(double x, double y) { return (x < y) ? y : x; }

template <bool PNP>
__global__ __launch_bounds__(64 * SC_WAVES) void score_kernel(const float *__restrict__ pa, const float *__restrict__ pb,
                                                              int n, const double *__restrict__ models, int H,
                                                              double fx, double fy, double cx, double cy, float t,
                                                              int *__restrict__ counts, uint8_t *__restrict__ masks,
                                                              const int *__restrict__ n_dev, const int *__restrict__ first_dev,
                                                              vo_batch_strides bs, int hyp_lo, const int *__restrict__ skip_bound,
                                                              int chunk)
{
    // A kernel of the per-frame pose chain: a few short wavefronts that, with many streams in flight, share their SIMDs
    // with another launch group's tracker (five wavefronts issuing back to back).  Raised issue priority lets them through.
    __builtin_amdgcn_s_setprio(3);
    // n_dev / first_dev (optional): the point count and the index of the first model live on the device, left there
    // by earlier kernels of the same stream (a negative first model: nothing to score).
    // blockIdx.z: the problem (a stream of a batched frame), whose arrays start bs.work / bs.out bytes after the
    // problem before.  Candidates below hyp_lo are another launch's; with skip_bound, candidates beyond *skip_bound
    // (the RANSAC loop's iteration bound after an earlier stage, which only shrinks) keep their cleared count.
    const int lane = threadIdx.x & 63;
    const int h = hyp_lo + blockIdx.x * SC_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (h >= H) return;
    if (bs.count > 1 || skip_bound) {
        const size_t b = blockIdx.z;
        pa = (const float *)((const char *)pa + b * bs.work);
        pb = (const float *)((const char *)pb + b * bs.work);
        models = (const double *)((const char *)models + b * bs.work);
        counts = (int *)((char *)counts + b * bs.out);
        if (n_dev) n_dev = (const int *)((const char *)n_dev + b * bs.out);
        if (skip_bound && h > *(const int *)((const char *)skip_bound + b * bs.out)) return;
    }
    if (n_dev) n = *n_dev;
    int64_t first = 0;
    if (first_dev) {
        first = *first_dev;
        if (first < 0) return;
    }
    constexpr int MS = PNP ? 12 : 9;
    double m[MS];
#pragma unroll
    for (int k = 0; k < MS; ++k) m[k] = models[(first + h) * MS + k];
    const int i0 = blockIdx.y * chunk;
    const int i1 = min(n, i0 + chunk);
    if (i0 >= n && gridDim.y > 1) return;       // a chunk beyond a device-side point count adds nothing
    int good = 0;
    for (int i = i0 + lane; i < i1; i += 64) {
        bool in;
        if (PNP) {
            in = vo_pnp_inlier(m, pa, pb, i, fx, fy, cx, cy, t);
        } else {
            double x1 = pa[2 * i], y1 = pa[2 * i + 1], x2 = pb[2 * i], y2 = pb[2 * i + 1];
            double a, b, c, d1, d2, s1, s2;
            a = m[0] * x1 + m[1] * y1 + m[2];
            b = m[3] * x1 + m[4] * y1 + m[5];
            c = m[6] * x1 + m[7] * y1 + m[8];
            s2 = 1. / (a * a + b * b);
            d2 = x2 * a + y2 * b + c;
            a = m[0] * x2 + m[3] * y2 + m[6];
            b = m[1] * x2 + m[4] * y2 + m[7];
            c = m[2] * x2 + m[5] * y2 + m[8];
            s1 = 1. / (a * a + b * b);
            d1 = x1 * a + y1 * b + c;
            float err = (float)std_max(d1 * d1 * s1, d2 * d2 * s2);
            in = err <= t;
        }
        good += __popcll(__ballot(in));
        if (masks) masks[(int64_t)h * n + i] = in ? 1 : 0;
    }
    if (lane == 0) {
        if (gridDim.y == 1) counts[h] = good;
        else atomicAdd(&counts[h], good);
    }
}

template <bool PNP>
static int score_common(vo_ctx *ctx, const float *pa, const float *pb, int n, const double *models, int H,
                        const double *K, double thr, int *counts, uint8_t *masks, int mem, const char *who)
{
    VO_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    VO_REQUIRE(n >= 0 && H >= 0, "%s: n=%d H=%d", who, n, H);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "%s: bad mem flag %d", who, mem);
    if (H == 0) return VO_OK;
    VO_REQUIRE(models && counts, "%s: NULL model/count buffer", who);
    VO_REQUIRE(n == 0 || (pa && pb), "%s: NULL point buffer", who);
    VO_REQUIRE(!PNP || K, "%s: K is NULL", who);
    VO_REQUIRE((int64_t)H * (int64_t)(n > 0 ? n : 1) < (1ll << 40), "%s: problem too large", who);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const float t = (float)(thr * thr);
    constexpr int MS = PNP ? 12 : 9, PA = PNP ? 3 : 2;
    const float *d_pa = pa, *d_pb = pb;
    const double *d_m = models;
    int *d_c = counts;
    uint8_t *d_k = masks;
    double Kh[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int rc;
    if (mem == VO_MEM_HOST) {
        void *p;
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * PA * 4, &p))) return rc; d_pa = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 2 * 4, &p))) return rc; d_pb = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)H * MS * 8, &p))) return rc; d_m = (const double *)p;
        if ((rc = vo_stage_reserve(ctx, 3, (size_t)H * 4, &p))) return rc; d_c = (int *)p;
        if (masks) { if ((rc = vo_stage_reserve(ctx, 4, (size_t)H * n, &p))) return rc; d_k = (uint8_t *)p; }
        if (n) {
            VO_UPLOAD(ctx, (void *)d_pa, pa, (size_t)n * PA * 4);
            VO_UPLOAD(ctx, (void *)d_pb, pb, (size_t)n * 2 * 4);
        }
        VO_UPLOAD(ctx, (void *)d_m, models, (size_t)H * MS * 8);
    }
    if (PNP) memcpy(Kh, K, sizeof(Kh));  // K is always a host pointer: it travels as kernel arguments
    const int chunk = score_chunk(n, H);
    const int chunks = n > 0 ? (n + chunk - 1) / chunk : 1;
    VO_REQUIRE(chunks <= 65535, "%s: n=%d exceeds %d points", who, n, 65535 * SC_CHUNK);
    if (chunks > 1) VO_HIP_CHECK(hipMemsetAsync(d_c, 0, (size_t)H * 4, ctx->stream));
    dim3 grid((H + SC_WAVES - 1) / SC_WAVES, chunks);
    hipLaunchKernelGGL(score_kernel<PNP>, grid, dim3(64 * SC_WAVES), 0, ctx->stream, d_pa, d_pb, n, d_m, H, Kh[0], Kh[4],
                       Kh[2], Kh[5], t, d_c, d_k, (const int *)nullptr, (const int *)nullptr, vo_batch_strides(), 0,
                       (const int *)nullptr, chunk);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, counts, d_c, (size_t)H * 4);
        if (masks && n) VO_RESULT(ctx, masks, d_k, (size_t)H * n);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

extern "C" int vo_score_fundamental(vo_ctx *ctx, const float *m1, const float *m2, int n, const double *F, int H,
                                    double thr, int32_t *counts, uint8_t *masks, int mem)
{
    return score_common<false>(ctx, m1, m2, n, F, H, nullptr, thr, counts, masks, mem, "vo_score_fundamental");
}

extern "C" int vo_score_pnp(vo_ctx *ctx, const float *X, const float *x, int n, const double *K, const double *Rt,
                            int H, double thr, int32_t *counts, uint8_t *masks, int mem)
{
    return score_common<true>(ctx, X, x, n, Rt, H, K, thr, counts, masks, mem, "vo_score_pnp");
}

// vo_score_pnp on device pointers whose point count (and, optionally, first model) is only known on the device:
// n_cap bounds the count for the launch geometry, *n_dev is the count, models + 12 * *first_dev the first model;
// counts_cleared: an earlier kernel has zeroed dcounts (needed when the points span several workgroups, which
// add up their shares).  bs: several problems of this shape in one launch.  Candidates hyp_lo .. H-1 are scored,
// those beyond *skip_bound_dev left alone.  Asynchronous on the context's stream.
int vo_score_pnp_dev(vo_ctx *ctx, const float *dX, const float *dx, int n_cap, const int *n_dev, const double *K,
                     const double *dRt, int H, const int *first_dev, double thr, int32_t *dcounts, uint8_t *dmasks,
                     bool counts_cleared, const vo_batch_strides &bs, int hyp_lo, const int *skip_bound_dev)
{
    if (H <= hyp_lo || bs.count <= 0) return VO_OK;
    // without cleared counts a split costs a memset launch: more than it saves on the chains that call this
    const int chunk = counts_cleared ? score_chunk(n_cap, (int64_t)(H - hyp_lo) * bs.count) : SC_CHUNK;
    const int chunks = n_cap > 0 ? (n_cap + chunk - 1) / chunk : 1;
    VO_REQUIRE(chunks <= 65535 && bs.count <= 65535, "vo_score_pnp: n=%d exceeds %d points (or %d problems)", n_cap, 65535 * SC_CHUNK, bs.count);
    VO_REQUIRE(bs.count == 1 || (counts_cleared && !dmasks && !first_dev), "vo_score_pnp: batched scoring wants cleared counts and no masks");
    if (chunks > 1 && !counts_cleared) VO_HIP_CHECK(hipMemsetAsync(dcounts, 0, (size_t)H * 4, ctx->stream));
    const float t = (float)(thr * thr);
    dim3 grid((H - hyp_lo + SC_WAVES - 1) / SC_WAVES, chunks, bs.count);
    hipLaunchKernelGGL(score_kernel<true>, grid, dim3(64 * SC_WAVES), 0, ctx->stream, dX, dx, n_cap, dRt, H, K[0], K[4], K[2],
                       K[5], t, dcounts, dmasks, n_dev, first_dev, bs, hyp_lo, skip_bound_dev, chunk);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_score(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)score_kernel<true>));
    return VO_OK;
}
