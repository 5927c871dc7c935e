// Host-side RANSAC bookkeeping of the two cv2 calls on the hot path
// (cv2.findFundamentalMat(FM_RANSAC) at vo_pipeline/bootstrap.py:246-252 and
// cv.solvePnPRansac at vo_pipeline/poseEstimation.py:61-69): the sample drawing
// OpenCV 4.5.4 performs inside RANSACPointSetRegistrator::run -- cv::RNG
// (multiply-with-carry, initial state 2^64-1), getSubset's distinct-index rule and,
// for the fundamental matrix, the collinearity re-draw of
// FMEstimatorCallback::checkSubset / haveCollinearPoints.  Drawing every sample up
// front lets the GPU solve and score all of them in two launches.
#include "vo_internal.h"
#include <math.h>
#include <float.h>
#include <memory>
#include <new>

static inline uint32_t cv_rng_next(uint64_t &s)
{
    s = (uint64_t)(uint32_t)s * 4164903690u + (s >> 32);
    return (uint32_t)s;
}

// is the last of the m points (nearly) on a line through two earlier ones
static bool collinear_with_last(const float *pts, const int *idx, int m)
{
    const int i = m - 1;
    const float xi = pts[2 * idx[i]], yi = pts[2 * idx[i] + 1];
    for (int j = 0; j < i; ++j) {
        const double dx1 = pts[2 * idx[j]] - xi, dy1 = pts[2 * idx[j] + 1] - yi;
        for (int k = 0; k < j; ++k) {
            const double dx2 = pts[2 * idx[k]] - xi, dy2 = pts[2 * idx[k] + 1] - yi;
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}

extern "C" int vo_ransac_draw_subsets(uint64_t *rng_state, int count, int model_points, int n_subsets, const float *m1,
                                      const float *m2, int max_attempts, int32_t *subsets)
{
    VO_REQUIRE(rng_state && subsets, "vo_ransac_draw_subsets: NULL argument");
    VO_REQUIRE(model_points >= 1 && model_points <= 16, "vo_ransac_draw_subsets: model_points=%d out of [1,16]", model_points);
    VO_REQUIRE(count >= model_points, "vo_ransac_draw_subsets: %d points cannot fill subsets of %d", count, model_points);
    VO_REQUIRE(n_subsets >= 0 && max_attempts >= 1, "vo_ransac_draw_subsets: n_subsets=%d max_attempts=%d", n_subsets, max_attempts);
    VO_REQUIRE((m1 == nullptr) == (m2 == nullptr), "vo_ransac_draw_subsets: give both point sets or neither");
    uint64_t s = *rng_state;
    int idx[16];
    for (int q = 0; q < n_subsets; ++q) {
        for (int attempt = 0; attempt < max_attempts; ++attempt) {
            int have = 0;
            while (have < model_points) {
                const int j = (int)(cv_rng_next(s) % (uint32_t)count);   // RNG::uniform(0, count)
                bool dup = false;
                for (int k = 0; k < have; ++k) dup = dup || idx[k] == j;
                if (!dup) idx[have++] = j;
            }
            if (m1 && (collinear_with_last(m1, idx, model_points) || collinear_with_last(m2, idx, model_points))) continue;
            break;
        }
        for (int k = 0; k < model_points; ++k) subsets[(size_t)q * model_points + k] = idx[k];
    }
    *rng_state = s;
    return VO_OK;
}

// ---------------------------------------------------------------------------
// Whole RANSAC calls.  One entry point stands in for one cv2 call: all samples are
// drawn as OpenCV draws them, in chunks (256, then up to the loop's current bound); per chunk the
// GPU solves every minimal problem and scores every candidate against every
// correspondence (two launches, candidates and points stay on the device), the host
// reads back one inlier count per candidate and replays OpenCV's sequential loop over
// them -- best-so-far update, adaptive iteration count, early stop -- so the outcome
// equals that loop run on the same candidate sequence, and no chunk is solved past the
// iteration at which the loop ends; a last launch produces the winner's inlier mask.
// ---------------------------------------------------------------------------

// RANSACUpdateNumIters: iterations needed for confidence p at outlier ratio ep
static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmin(fmax(p, 0.), 1.);
    ep = fmin(fmax(ep, 0.), 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)lrint(num / denom);
}

// OpenCV's sequential loop, replayed over candidates that were solved and scored chunk by chunk.
struct RansacReplay {
    int niters, best = -1, best_c = -1, done = 0;    // done: iterations replayed so far
    explicit RansacReplay(int max_iters) : niters(max_iters) {}
    bool finished() const { return done >= niters; }
    // counts / used: the candidates of iterations [first, first + count), per_sample slots each (used == 0: no such
    // solution); candidate ids are global (iteration * per_sample + slot)
    void feed(const int32_t *counts, const uint8_t *used, int per_sample, int first, int count, int n, int model_points,
              double conf)
    {
        for (int it = first; it < first + count && it < niters; ++it) {
            for (int s = 0; s < per_sample; ++s) {
                const int c = (it - first) * per_sample + s;
                if (!used[c]) continue;
                const int good = counts[c];
                if (good > (best > model_points - 1 ? best : model_points - 1)) {
                    best = good;
                    best_c = it * per_sample + s;
                    niters = ransac_update_num_iters(conf, (double)(n - good) / n, model_points, niters);
                }
            }
            done = it + 1;
        }
    }
};

// Samples are solved and scored in chunks: 256 first (clean data stop the adaptive loop after a handful of
// iterations; the reference's parking parameters allow 10 000), then as many as the loop's current bound still asks
// for -- the bound only shrinks, so the second chunk normally ends the loop -- at most 4096 per launch.
static int ransac_chunk(int round, int first, int bound)
{
    if (round == 0) return 256;
    const int want = bound - first < 64 ? 64 : bound - first;
    return want < 4096 ? want : 4096;
}

void vo_ransac_replay_first_chunk(const int32_t *counts, const uint8_t *valid, int c, int n, int model_points, double conf,
                                  int max_iters, int *best, int *best_c, int *niters)
{
    RansacReplay loop(max_iters);
    loop.feed(counts, valid, 1, 0, c, n, model_points, conf);
    *best = loop.best;
    *best_c = loop.best_c;
    *niters = loop.niters;
}

int vo_ransac_advance_state(uint64_t *rng_state, int count, int model_points, int iterations)
{
    if (iterations <= 0) return VO_OK;
    std::unique_ptr<int32_t[]> scratch(new (std::nothrow) int32_t[(size_t)iterations * model_points]);
    if (!scratch) { vo_set_error("RANSAC: out of host memory"); return VO_ENOMEM; }
    return vo_ransac_draw_subsets(rng_state, count, model_points, iterations, nullptr, nullptr, 1, scratch.get());
}

static void matrix_from_rotvec(const double *rv, double *R)
{
    const double th = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (th == 0) { for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0); return; }
    const double kx = rv[0] / th, ky = rv[1] / th, kz = rv[2] / th, c = cos(th), s = sin(th), v = 1. - c;
    R[0] = c + v * kx * kx;      R[1] = v * kx * ky - s * kz; R[2] = v * kx * kz + s * ky;
    R[3] = v * kx * ky + s * kz; R[4] = c + v * ky * ky;      R[5] = v * ky * kz - s * kx;
    R[6] = v * kx * kz - s * ky; R[7] = v * ky * kz + s * kx; R[8] = c + v * kz * kz;
}

extern "C" int vo_solve_pnp_ransac(vo_ctx *ctx, const float *X, const float *x, int n, const double *K, double thr,
                                   double conf, int max_iters, int refine_iters, uint64_t *rng_state, double *Rt_out,
                                   uint8_t *inlier_mask, int32_t *n_inliers, int mem)
{
    VO_REQUIRE(ctx && K && Rt_out && inlier_mask && n_inliers, "vo_solve_pnp_ransac: NULL argument");
    VO_REQUIRE(n >= 0 && (n == 0 || (X && x)), "vo_solve_pnp_ransac: bad point buffers (n=%d)", n);
    VO_REQUIRE(max_iters >= 1 && max_iters <= (1 << 22) && refine_iters >= 0, "vo_solve_pnp_ransac: max_iters=%d refine_iters=%d", max_iters, refine_iters);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_solve_pnp_ransac: bad mem flag %d", mem);
    *n_inliers = 0;
    if (n < 4) return VO_OK;
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const int S = max_iters;
    int rc = VO_OK;
    void *p;
    const uint64_t state0 = rng_state ? *rng_state : 0xFFFFFFFFFFFFFFFFull;
    uint64_t state = state0;
    std::unique_ptr<int32_t[]> subsets_buf;          // all S subsets; only a loop that outlives its first chunk needs them
    // page-locked area.  Read back: a chunk's counts (4096 ints), validity flags (4096 bytes in 4096 ints), candidate
    // poses (4096 x 12 doubles), then the inlier mask (n bytes) and the refit's 8 doubles.  Sent: the first chunk's
    // subsets and, for host arrays, the points (copies out of page-locked memory do not hold the caller up).
    const size_t back_bytes = (((size_t)4096 * 8 + (size_t)4096 * 96 + 64 + (size_t)n + 64) + 255) & ~(size_t)255;
    const int c_first = ransac_chunk(0, 0, max_iters) < S ? ransac_chunk(0, 0, max_iters) : S;
    void *pin;
    if ((rc = vo_pinned_reserve(ctx, back_bytes + (size_t)c_first * 16 + (size_t)n * 20 + 64, &pin))) return rc;
    int32_t *counts = (int32_t *)pin;
    double *poses = (double *)((char *)pin + (size_t)4096 * 8);
    double *refined = poses + (size_t)4096 * 12;
    uint8_t *pin_mask = (uint8_t *)(refined + 8);
    int32_t *pin_sub = (int32_t *)((char *)pin + back_bytes);
    float *pin_X = (float *)(pin_sub + (size_t)c_first * 4), *pin_x = pin_X + (size_t)n * 3;
    double Rt[12] = {0};
    int32_t *subsets = nullptr;
    const float *dX = X, *dx = x;
    uint8_t *dmask = inlier_mask;
    // staging slots 0, 1 (points), 2 (subsets), 4 (candidates), 5 (counts + flags), 6 (mask), 7 (first chunk's results);
    // slot 3 belongs to vo_refine_pose
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * 12, &p))) return rc; dX = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 8, &p))) return rc; dx = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 6, (size_t)n, &p))) return rc; dmask = (uint8_t *)p;
        memcpy(pin_X, X, (size_t)n * 12);
        memcpy(pin_x, x, (size_t)n * 8);
    }
    if ((rc = vo_stage_reserve(ctx, 2, (size_t)S * 16, &p))) return rc;
    int32_t *dsub = (int32_t *)p;
    if ((rc = vo_stage_reserve(ctx, 4, (size_t)S * 96, &p))) return rc;
    double *dRt = (double *)p;
    if ((rc = vo_stage_reserve(ctx, 5, (size_t)S * 5 + 16, &p))) return rc;
    int32_t *dcounts = (int32_t *)p;
    uint8_t *dvalid = (uint8_t *)(dcounts + S);
    RansacReplay loop(max_iters);
    int first = 0, round = 0;
    {
        // First chunk with one read-back: the loop over its candidates is replayed on the device, which lets the
        // winner's mask and the refit follow without the host in between; the host then replays the loop itself on
        // the returned counts and keeps the device's result only when both agree and the loop has ended.
        const int c = ransac_chunk(0, 0, max_iters) < S ? ransac_chunk(0, 0, max_iters) : S;
        const size_t o_ref = sizeof(vo_frame_meta), o_counts = o_ref + 64, o_valid = o_counts + (((size_t)c * 4 + 63) & ~(size_t)63),
                     o_mask = o_valid + (((size_t)c + 63) & ~(size_t)63), out_bytes = o_mask + (mem == VO_MEM_HOST ? (size_t)n : 0);
        if ((rc = vo_stage_reserve(ctx, 7, out_bytes, &p))) return rc;
        char *dout = (char *)p;
        uint8_t *dmask0 = mem == VO_MEM_HOST ? (uint8_t *)(dout + o_mask) : inlier_mask;
        if ((rc = vo_ransac_draw_subsets(&state, n, 4, c, nullptr, nullptr, 1, pin_sub))) return rc;
        // subsets and (host arrays) points to the device in one launch
        const vo_copy_seg segs[3] = {{dsub, pin_sub, (size_t)c * 16}, {(void *)dX, pin_X, (size_t)n * 12}, {(void *)dx, pin_x, (size_t)n * 8}};
        if ((rc = vo_copy_pinned_async(ctx, segs, mem == VO_MEM_HOST ? 3 : 1))) return rc;
        if ((rc = vo_p3p_solve(ctx, dX, dx, n, K, dsub, c, dRt, dvalid, VO_MEM_DEVICE))) return rc;
        if ((rc = vo_score_pnp(ctx, dX, dx, n, K, dRt, c, thr, dcounts, nullptr, VO_MEM_DEVICE))) return rc;
        if ((rc = vo_pnp_first_chunk_tail(ctx, dX, dx, n, n, nullptr, dcounts, dvalid, c, max_iters, conf, K, thr, dRt, (vo_frame_meta *)dout,
                                          (int32_t *)(dout + o_counts), (uint8_t *)(dout + o_valid), dmask0, refine_iters,
                                          (double *)(dout + o_ref))))
            return rc;
        char *hout = (char *)pin;           // the read-back area is larger than this block
        if ((rc = vo_download_async(ctx, hout, dout, out_bytes))) return rc;
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const vo_frame_meta meta = *(const vo_frame_meta *)hout;
        loop.feed((const int32_t *)(hout + o_counts), (const uint8_t *)(hout + o_valid), 1, 0, c, n, 4, conf);
        const bool agree = loop.best_c == meta.win && loop.niters == meta.niters && (loop.best_c < 0 || loop.best == meta.best);
        first = c;
        round = 1;
        if (agree && (loop.finished() || first >= S)) {
            if (rng_state) {
                uint64_t st = state0;
                if ((rc = vo_ransac_advance_state(&st, n, 4, loop.niters < S ? loop.niters : S))) return rc;
                *rng_state = st;
            }
            if (loop.best_c < 0) return VO_OK;
            memcpy(Rt, meta.Rt, sizeof(Rt));
            const double *ref = (const double *)(hout + o_ref);
            bool finite = refine_iters > 0;
            for (int k = 0; k < 6 && finite; ++k) finite = isfinite(ref[k]);
            if (finite) {
                double R[9];
                matrix_from_rotvec(ref, R);
                for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) Rt[4 * r + cc] = R[3 * r + cc]; Rt[4 * r + 3] = ref[3 + r]; }
            }
            if (mem == VO_MEM_HOST) memcpy(inlier_mask, hout + o_mask, (size_t)n);
            memcpy(Rt_out, Rt, sizeof(Rt));
            *n_inliers = loop.best;
            return VO_OK;
        }
        if (loop.best_c >= 0) {
            if (agree) memcpy(Rt, meta.Rt, sizeof(Rt));
            else VO_HIP_CHECK(hipMemcpy(Rt, dRt + (size_t)loop.best_c * 12, sizeof(Rt), hipMemcpyDeviceToHost));
        }
    }
    if (first < S && !loop.finished()) {
        subsets_buf.reset(new (std::nothrow) int32_t[(size_t)S * 4]);
        if (!subsets_buf) { vo_set_error("vo_solve_pnp_ransac: out of host memory"); return VO_ENOMEM; }
        subsets = subsets_buf.get();
    }
    for (; first < S && !loop.finished(); ++round) {
        const int c = ransac_chunk(round, first, loop.niters) < S - first ? ransac_chunk(round, first, loop.niters) : S - first;
        if ((rc = vo_ransac_draw_subsets(&state, n, 4, c, nullptr, nullptr, 1, subsets + (size_t)first * 4))) return rc;
        VO_UPLOAD(ctx, dsub + (size_t)first * 4, subsets + (size_t)first * 4, (size_t)c * 16);
        if ((rc = vo_p3p_solve(ctx, dX, dx, n, K, dsub + (size_t)first * 4, c, dRt + (size_t)first * 12, dvalid + first, VO_MEM_DEVICE))) return rc;
        if ((rc = vo_score_pnp(ctx, dX, dx, n, K, dRt + (size_t)first * 12, c, thr, dcounts + first, nullptr, VO_MEM_DEVICE))) return rc;
        VO_HIP_CHECK(hipMemcpyAsync(counts, dcounts + first, (size_t)c * 4, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipMemcpyAsync(counts + 4096, dvalid + first, (size_t)c, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipMemcpyAsync(poses, dRt + (size_t)first * 12, (size_t)c * 96, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const int before = loop.best_c;
        loop.feed(counts, (const uint8_t *)(counts + 4096), 1, first, c, n, 4, conf);
        if (loop.best_c != before) memcpy(Rt, poses + (size_t)(loop.best_c - first) * 12, sizeof(Rt));
        first += c;
    }
    if (rng_state) {     // the generator after the iterations the loop ran, as OpenCV's own loop leaves it
        uint64_t st = state0;
        if ((rc = vo_ransac_advance_state(&st, n, 4, loop.niters < S ? loop.niters : S))) return rc;
        *rng_state = st;
    }
    const int win = loop.best_c;
    if (win < 0) return VO_OK;
    // winner's mask and the refit on its inliers back to back: the inlier count is already known from the loop
    const int32_t good = loop.best;
    if ((rc = vo_score_pnp(ctx, dX, dx, n, K, dRt + (size_t)win * 12, 1, thr, dcounts, dmask, VO_MEM_DEVICE))) return rc;
    const bool refit = good >= 4 && refine_iters > 0;
    // OpenCV refits with EPnP there; here Gauss-Newton on the same error, started from the winner where it lies
    if (refit && (rc = vo_refine_pose_from_Rt_async(ctx, dX, dx, n, nullptr, dmask, nullptr, K, dRt, win, nullptr, refine_iters, refined, nullptr)))
        return rc;
    if (mem == VO_MEM_HOST) VO_HIP_CHECK(hipMemcpyAsync(pin_mask, dmask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (mem == VO_MEM_HOST) memcpy(inlier_mask, pin_mask, (size_t)n);
    if (refit) {
        bool finite = true;
        for (int k = 0; k < 6; ++k) finite = finite && isfinite(refined[k]);
        if (finite) {
            double R[9];
            matrix_from_rotvec(refined, R);
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rt[4 * r + c] = R[3 * r + c]; Rt[4 * r + 3] = refined[3 + r]; }
        }
    }
    memcpy(Rt_out, Rt, sizeof(Rt));
    *n_inliers = good;
    return VO_OK;
}

extern "C" int vo_find_fundamental_ransac(vo_ctx *ctx, const float *m1, const float *m2, int n, double thr, double conf,
                                          int max_iters, uint64_t *rng_state, double *F_out, uint8_t *inlier_mask,
                                          int32_t *n_inliers)
{
    VO_REQUIRE(ctx && F_out && inlier_mask && n_inliers, "vo_find_fundamental_ransac: NULL argument");
    VO_REQUIRE(n >= 0 && (n == 0 || (m1 && m2)), "vo_find_fundamental_ransac: bad point buffers (n=%d)", n);
    VO_REQUIRE(max_iters >= 1 && max_iters <= (1 << 22), "vo_find_fundamental_ransac: max_iters=%d", max_iters);
    *n_inliers = 0;
    if (n < 8) return VO_OK;   // OpenCV needs >= 7 and switches estimators below 15; the RANSAC form is kept for n >= 8
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const int S = max_iters;
    int rc = VO_OK;
    void *p;
    const uint64_t state0 = rng_state ? *rng_state : 0xFFFFFFFFFFFFFFFFull;
    uint64_t state = state0;
    std::unique_ptr<int32_t[]> subsets_buf(new (std::nothrow) int32_t[(size_t)S * 7]);
    if (!subsets_buf) { vo_set_error("vo_find_fundamental_ransac: out of host memory"); return VO_ENOMEM; }
    // page-locked read-back area: a chunk's 3c counts and c solution counts (contiguous on the device too), room for the
    // 3c validity flags formed here, then the winner: F (9 doubles), its count, the inlier mask (n bytes)
    // behind it, sent: the first chunk's subsets and the points (one launch copies them out of page-locked memory)
    const size_t back_bytes = (((size_t)4096 * 20 + 128 + (size_t)n + 64) + 255) & ~(size_t)255;
    const int c_first = ransac_chunk(0, 0, max_iters) < S ? ransac_chunk(0, 0, max_iters) : S;
    void *pin;
    if ((rc = vo_pinned_reserve(ctx, back_bytes + (size_t)c_first * 28 + (size_t)n * 16 + 64, &pin))) return rc;
    int32_t *pin_sub = (int32_t *)((char *)pin + back_bytes);
    float *pin_m1 = (float *)(pin_sub + (size_t)c_first * 7), *pin_m2 = pin_m1 + (size_t)n * 2;
    int32_t *subsets = subsets_buf.get(), *counts = (int32_t *)pin;
    double *pin_F = (double *)((char *)pin + (size_t)4096 * 20);
    int32_t *pin_good = (int32_t *)(pin_F + 9);
    uint8_t *pin_mask = (uint8_t *)(pin_F + 12);
    if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * 8, &p))) return rc;
    float *d1 = (float *)p;
    if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 8, &p))) return rc;
    float *d2 = (float *)p;
    if ((rc = vo_stage_reserve(ctx, 2, (size_t)S * 28, &p))) return rc;
    int32_t *dsub = (int32_t *)p;
    if ((rc = vo_stage_reserve(ctx, 4, (size_t)S * 216, &p))) return rc;
    double *dF = (double *)p;
    if ((rc = vo_stage_reserve(ctx, 5, (size_t)S * 16 + 16, &p))) return rc;
    int32_t *dcounts = (int32_t *)p, *dnsol = dcounts + 3 * (size_t)S;
    if ((rc = vo_stage_reserve(ctx, 6, (size_t)n + 16, &p))) return rc;     // read back 16 bytes at a time
    uint8_t *dmask = (uint8_t *)p;
    memcpy(pin_m1, m1, (size_t)n * 8);
    memcpy(pin_m2, m2, (size_t)n * 8);
    RansacReplay loop(max_iters);
    for (int first = 0, round = 0; first < S && !loop.finished(); ++round) {
        const int c = ransac_chunk(round, first, loop.niters) < S - first ? ransac_chunk(round, first, loop.niters) : S - first;
        // the collinearity test of the sampler reads the points on the host
        if ((rc = vo_ransac_draw_subsets(&state, n, 7, c, m1, m2, 10000, subsets + (size_t)first * 7))) return rc;
        if (round == 0) {
            memcpy(pin_sub, subsets, (size_t)c * 28);
            const vo_copy_seg segs[3] = {{dsub, pin_sub, (size_t)c * 28}, {d1, pin_m1, (size_t)n * 8}, {d2, pin_m2, (size_t)n * 8}};
            if ((rc = vo_copy_pinned_async(ctx, segs, 3))) return rc;
        } else {
            VO_UPLOAD(ctx, dsub + (size_t)first * 7, subsets + (size_t)first * 7, (size_t)c * 28);
        }
        if ((rc = vo_seven_point_solve(ctx, d1, d2, n, dsub + (size_t)first * 7, c, dF + (size_t)first * 27, dnsol + first, VO_MEM_DEVICE))) return rc;
        if ((rc = vo_score_fundamental(ctx, d1, d2, n, dF + (size_t)first * 27, 3 * c, thr, dcounts + 3 * (size_t)first, nullptr, VO_MEM_DEVICE))) return rc;
        // read back by kernels that write the page-locked area (a copy engine's hand-over from the compute queue costs more
        // than these few KB take)
        if ((rc = vo_download_async(ctx, counts, dcounts + 3 * (size_t)first, (size_t)c * 12))) return rc;
        if ((rc = vo_download_async(ctx, counts + 3 * 4096, dnsol + first, (size_t)c * 4))) return rc;
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const int32_t *nsol = counts + 3 * 4096;
        uint8_t *used = (uint8_t *)(counts + 4 * 4096);
        for (int q = 0; q < c; ++q)
            for (int k = 0; k < 3; ++k) used[3 * q + k] = k < nsol[q];
        loop.feed(counts, used, 3, first, c, n, 7, conf);
        first += c;
    }
    if (rng_state) {     // the generator after the iterations the loop ran, as OpenCV's own loop leaves it
        const int ran = loop.niters < S ? loop.niters : S;
        uint64_t st = state0;
        std::unique_ptr<int32_t[]> scratch(new (std::nothrow) int32_t[(size_t)(ran > 0 ? ran : 1) * 7]);
        if (!scratch) { vo_set_error("vo_find_fundamental_ransac: out of host memory"); return VO_ENOMEM; }
        if (ran > 0 && (rc = vo_ransac_draw_subsets(&st, n, 7, ran, m1, m2, 10000, scratch.get()))) return rc;
        *rng_state = st;
    }
    const int win = loop.best_c;
    if (win < 0) return VO_OK;
    if ((rc = vo_score_fundamental(ctx, d1, d2, n, dF + (size_t)win * 9, 1, thr, dcounts, dmask, VO_MEM_DEVICE))) return rc;
    VO_HIP_CHECK(hipMemcpyAsync(pin_F, dF + (size_t)win * 9, 72, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(hipMemcpyAsync(pin_good, dcounts, 4, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = vo_download_async(ctx, pin_mask, dmask, (size_t)n))) return rc;
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(F_out, pin_F, 72);
    memcpy(inlier_mask, pin_mask, (size_t)n);
    *n_inliers = *pin_good;
    return VO_OK;
}
