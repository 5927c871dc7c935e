// One tracked frame of the pipeline in one call with one synchronisation: the body of
// ContinuousVO._process_frame up to the pose (vo_pipeline/continuousVO.py:160-180) --
//   cv.calcOpticalFlowPyrLK (PoseEstimation.KLT, vo_pipeline/poseEstimation.py:116-133),
//   the status == 1 filter of keypoints and landmarks (continuousVO.py:168-172),
//   cv.solvePnPRansac with its refit (PoseEstimation.PnP, poseEstimation.py:51-77).
// Called one after the other, the three steps cost three round trips to the host: the tracker's results come
// back to be filtered, the sampler needs the number of survivors, the RANSAC loop reads the inlier counts.  Here
// the survivors are compacted on the device, the first 256 samples are drawn on the device from the generator's
// raw output (which does not depend on the point count; only the modulus and the rejections do), the loop over
// their inlier counts is replayed by the kernel that writes the winner's mask, and the refit is launched on an index
// that still lives on the device.  Seven launches after the pyramid (tracker, preparation, P3P, scoring, mask, refit)
// with one copy in and one copy out.  The host reads everything back once, replays the loop itself on the returned counts
// (its own libm decides the adaptive bound) and, when the loop needs more than the first chunk, when the device's
// replay disagrees, or when the raw draws ran out, finishes with vo_solve_pnp_ransac on the compacted points.
// Either way the outcome is that of vo_klt_track followed by vo_solve_pnp_ransac on the rows with status 1.
#include "vo_internal.h"
#include <math.h>
#include <float.h>
#include <new>
#include <vector>

#define FS_THREADS 256                      // threads of a batched frame's compaction (a single frame takes FS_THREADS_ONE)
#define FS_THREADS_ONE 1024
#define FS_SAMPLES 256                      // samples of the first chunk
#define FS_DRAWS (4 * FS_SAMPLES + 512)     // raw generator outputs kept on the device: room for 512 rejected draws
#define FS_STAGE1 32                        // batched frames: samples solved and scored before the loop's bound is looked at

// One subset of four distinct indices drawn from position p of the generator's output (draw p = the low half of
// states[p + 1]), as RANSACPointSetRegistrator::getSubset does: index = draw % count, a repeated index costs another
// draw.  Returns the position after the subset; FS_DRAWS when the draws ran out (a subset whose last draw is the
// buffer's last entry counts as ran out too, which keeps the two cases apart at no cost but one subset).
__device__ static inline int frame_draw_subset(const uint64_t *__restrict__ states, int p, uint32_t m, int *idx)
{
    int have = 0;
    while (have < 4 && p < FS_DRAWS) {
        const int j = (int)((uint32_t)states[++p] % m);
        bool dup = false;
        for (int k = 0; k < have; ++k) dup = dup || idx[k] == j;
        if (!dup) idx[have++] = j;
    }
    return have == 4 && p < FS_DRAWS ? p : FS_DRAWS;
}

// Everything between the tracker and the minimal solver, one workgroup per stream (blockIdx.x; a single frame is
// one stream):
//  * rows with status 1 are compacted in order (landmarks and tracked points side by side; orig[j] = input row of
//    survivor j), the full-length inlier mask and the candidates' inlier counts are cleared;
//  * subsets 0 .. c0-1 are drawn by the first wavefront.  Where subset i starts depends on the rejected draws
//    before it, and those are rare (a draw repeats one of up to three earlier indices out of m): lane l assumes that
//    the next 64 subsets start four draws apart and tests its own four draws; the subsets before the first lane
//    that met a repeated index stand, that lane's subset is drawn the long way, and the wavefront goes on behind it.
//    With thousands of survivors a round settles 64 subsets; with a handful it settles one or two.
// Batched frames take the instance with four wavefronts (and no LDS to speak of), whose workgroups find room on
// compute units that a tracker launch of another stream group keeps busy; a single frame has the chip to itself and
// takes sixteen, a quarter of the compaction rounds.
// states: the generator's state after 0, 1, ..., FS_DRAWS draws (a draw is the low half of the state after it);
// stream b reads its own run of them, states_stride entries after stream b-1's (0: all streams draw alike).
// starts (optional, c0 + 1 entries): how many draws precede subset i.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void frame_prep_kernel(const uint8_t *__restrict__ status, const float *__restrict__ X,
                                                                const float *__restrict__ pts, int n,
                                                                const uint64_t *__restrict__ states, size_t states_stride, int c0,
                                                                float *__restrict__ Xc, float *__restrict__ xc, int *__restrict__ orig,
                                                                vo_frame_meta *__restrict__ meta, int *__restrict__ counts,
                                                                uint8_t *__restrict__ full, int *__restrict__ subsets,
                                                                unsigned short *__restrict__ starts, vo_batch_strides bs)
{
    // A kernel of the per-frame pose chain: a few short wavefronts that, with many streams in flight, share their SIMDs
    // with another launch group's tracker (five wavefronts issuing back to back).  Raised issue priority lets them through.
    __builtin_amdgcn_s_setprio(3);
    __shared__ int wcount[2][THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const size_t b = blockIdx.x;
        status += b * (size_t)n;
        X += 3 * b * (size_t)n;
        pts += 2 * b * (size_t)n;
        full += b * bs.mask;
        states += b * states_stride;
        Xc = (float *)((char *)Xc + b * bs.work);
        xc = (float *)((char *)xc + b * bs.work);
        orig = (int *)((char *)orig + b * bs.work);
        subsets = (int *)((char *)subsets + b * bs.work);
        if (starts) starts = (unsigned short *)((char *)starts + b * bs.work);
        meta = (vo_frame_meta *)((char *)meta + b * bs.out);
        counts = (int *)((char *)counts + b * bs.out);
    }
    for (int c = tid; c <= c0; c += THREADS) counts[c] = 0;
    // The landmarks of a single frame lie in page-locked host memory (they are read once, here): a read is a trip over
    // the bus, and read inside the rounds below -- a barrier apart -- the trips of six rounds came one after the other,
    // most of this kernel's time.  The first FS_AHEAD rounds' rows are therefore asked for up front, kept or not.
    // (status and tracked points with them: device memory, but a miss in every round all the same)
    constexpr int FS_AHEAD = 8;
    float ahead[FS_AHEAD][5];
    uint8_t ahead_status[FS_AHEAD];
#pragma unroll
    for (int r = 0; r < FS_AHEAD; ++r) {
        const int i = r * THREADS + tid;
        ahead_status[r] = 0;
        if (i < n) {
            ahead_status[r] = status[i];
            ahead[r][0] = X[3 * (int64_t)i]; ahead[r][1] = X[3 * (int64_t)i + 1]; ahead[r][2] = X[3 * (int64_t)i + 2];
            ahead[r][3] = pts[2 * (int64_t)i]; ahead[r][4] = pts[2 * (int64_t)i + 1];
        }
    }
    int base = 0;
    auto compact_round = [&](int i0, int round, const float *mine, uint8_t mine_status) {
        const int i = i0 + tid;
        const bool keep = i < n && (mine ? mine_status : status[i]) == 1;
        if (i < n) full[i] = 0;
        const unsigned long long b = __ballot(keep);
        if (lane == 0) wcount[round & 1][wv] = __popcll(b);
        __syncthreads();          // one barrier per round: the rounds alternate between two sets of counters
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) {
            const int c = wcount[round & 1][w];
            before += w < wv ? c : 0;
            total += c;
        }
        if (keep) {
            const int64_t j = base + before + __popcll(b & ((1ull << lane) - 1));
            if (mine) {
                Xc[3 * j] = mine[0]; Xc[3 * j + 1] = mine[1]; Xc[3 * j + 2] = mine[2];
                xc[2 * j] = mine[3]; xc[2 * j + 1] = mine[4];
            } else {
                Xc[3 * j] = X[3 * (int64_t)i]; Xc[3 * j + 1] = X[3 * (int64_t)i + 1]; Xc[3 * j + 2] = X[3 * (int64_t)i + 2];
                xc[2 * j] = pts[2 * (int64_t)i]; xc[2 * j + 1] = pts[2 * (int64_t)i + 1];
            }
            orig[j] = i;
        }
        base += total;
    };
#pragma unroll
    for (int r = 0; r < FS_AHEAD; ++r)
        if (r * THREADS < n) compact_round(r * THREADS, r, ahead[r], ahead_status[r]);
    for (int i0 = FS_AHEAD * THREADS, round = FS_AHEAD; i0 < n; i0 += THREADS, ++round) compact_round(i0, round, nullptr, 0);
    if (wv != 0) return;
    const int m = base;
    if (lane == 0) {
        meta->m = m;
        meta->win = -1;
        meta->best = 0;
        meta->niters = 0;
        meta->overflow = 0;
        meta->bound0 = 0x7fffffff;
        meta->state = states[0];
    }
    if (m < 4) {
        for (int c = lane; c < c0; c += 64) { subsets[4 * c] = 0; subsets[4 * c + 1] = 0; subsets[4 * c + 2] = 0; subsets[4 * c + 3] = 0; }
        if (starts) for (int c = lane; c <= c0; c += 64) starts[c] = 0;
        return;
    }
    int done = 0, pos = 0;          // subsets drawn, draws used (both uniform over the wavefront)
    bool overflow = false;
    while (done < c0) {
        const int p = pos + 4 * lane;
        int idx[4] = {0, 0, 0, 0};
        bool ok = false;
        if (done + lane < c0 && p + 4 < FS_DRAWS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) idx[k] = (int)((uint32_t)states[p + k + 1] % (uint32_t)m);
            ok = idx[0] != idx[1] && idx[0] != idx[2] && idx[0] != idx[3] && idx[1] != idx[2] && idx[1] != idx[3] && idx[2] != idx[3];
        }
        const unsigned long long bad = __ballot(!ok);
        const int good = bad ? __ffsll((long long)bad) - 1 : 64;       // lanes below: their subsets stand
        if (lane < good) {
            const int c = done + lane;
            subsets[4 * c] = idx[0]; subsets[4 * c + 1] = idx[1]; subsets[4 * c + 2] = idx[2]; subsets[4 * c + 3] = idx[3];
            if (starts) starts[c] = (unsigned short)p;
        }
        done += good;
        pos += 4 * good;
        if (done >= c0) break;
        if (good < 64) {
            // subset `done` met a repeated index (or the end of the draws): the long way, every lane alike
            const int end = frame_draw_subset(states, pos, (uint32_t)m, idx);
            if (end >= FS_DRAWS) { overflow = true; break; }
            if (lane == 0) {
                subsets[4 * done] = idx[0]; subsets[4 * done + 1] = idx[1]; subsets[4 * done + 2] = idx[2]; subsets[4 * done + 3] = idx[3];
                if (starts) starts[done] = (unsigned short)pos;
            }
            ++done;
            pos = end;
        }
    }
    if (overflow) {
        if (lane == 0) meta->overflow = 1;
        for (int c = done + lane; c < c0; c += 64) {
            subsets[4 * c] = 0; subsets[4 * c + 1] = 0; subsets[4 * c + 2] = 0; subsets[4 * c + 3] = 0;
            if (starts) starts[c] = FS_DRAWS;
        }
        pos = FS_DRAWS;
    }
    if (starts && lane == 0) starts[c0] = (unsigned short)pos;
}

// RANSACUpdateNumIters
__device__ static int frame_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmin(fmax(p, 0.), 1.);
    ep = fmin(fmax(ep, 0.), 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)lrint(num / denom);
}

// OpenCV's sequential loop over candidates 0 .. c0-1 (one per iteration, valid[c] == 0: none), replayed by one
// wavefront: candidate c improves the best when its count exceeds max(best, 3) and c is still below the iteration
// bound, which every improvement recomputes.  Only improvements change the state, so per 64 candidates the
// wavefront jumps from one improvement to the next.
__device__ static void frame_replay(const int *__restrict__ counts, const uint8_t *__restrict__ valid, int c0, int m, double conf,
                                    int lane, int &niters, int &best, int &best_c)
{
    bool stop = m < 4;
    for (int r = 0; r * 64 < c0 && !stop; ++r) {
        const int c = r * 64 + lane;
        const int v = (c < c0 && valid[c]) ? counts[c] : -1;
        while (true) {
            const int bar = best > 3 ? best : 3;
            const unsigned long long b = __ballot(v > bar);
            if (!b) break;
            const int first = __ffsll((long long)b) - 1;
            if (r * 64 + first >= niters) { stop = true; break; }
            best = __shfl(v, first);
            best_c = r * 64 + first;
            niters = frame_update_num_iters(conf, (double)(m - best) / m, 4, niters);
        }
    }
}

// Batched frames solve and score their first chunk in two stages: after the first c1 candidates this kernel (one
// wavefront per stream) leaves the loop's iteration bound in meta->bound0.  The bound only shrinks, so no candidate
// from bound0 on can matter any more, and the second stage's launches skip those beyond it (one more is kept: the
// host re-checks the loop with its own libm, whose bound may round the other way).
__global__ __launch_bounds__(64) void frame_bound_kernel(const int *__restrict__ counts, const uint8_t *__restrict__ valid, int c1,
                                                         int max_iters, double conf, vo_frame_meta *__restrict__ meta,
                                                         vo_batch_strides bs)
{
    // A kernel of the per-frame pose chain: a few short wavefronts that, with many streams in flight, share their SIMDs
    // with another launch group's tracker (five wavefronts issuing back to back).  Raised issue priority lets them through.
    __builtin_amdgcn_s_setprio(3);
    const size_t b = blockIdx.x;
    counts = (const int *)((const char *)counts + b * bs.out);
    valid += b * bs.out;
    meta = (vo_frame_meta *)((char *)meta + b * bs.out);
    int niters = max_iters, best = -1, best_c = -1;
    frame_replay(counts, valid, c1, meta->m, conf, threadIdx.x, niters, best, best_c);
    if (threadIdx.x == 0) meta->bound0 = niters;
}

// The winner of the first chunk and its inlier mask.  Every wavefront replays the loop over the chunk's candidates
// for itself, then scores its share of the survivors against the winner with the expression of the scoring kernel
// and writes the mask at the survivors' input rows.  blockIdx.y: the stream.  starts / states (optional): the draws
// that precede each subset and the generator's states, from which the state after the loop's iterations follows.
#define FM_WAVES 4
#define FM_POINTS 1024      // survivors per workgroup
__global__ __launch_bounds__(64 * FM_WAVES) void frame_mask_kernel(const int *__restrict__ counts, const uint8_t *__restrict__ valid,
                                                                   int c0, int max_iters, double conf, const double *__restrict__ Rt,
                                                                   const float *__restrict__ Xc, const float *__restrict__ xc,
                                                                   const int *__restrict__ orig, double fx, double fy, double cx,
                                                                   double cy, float t, vo_frame_meta *__restrict__ meta,
                                                                   uint8_t *__restrict__ full, int m_host,
                                                                   int *__restrict__ out_counts, uint8_t *__restrict__ out_valid,
                                                                   const unsigned short *__restrict__ starts,
                                                                   const uint64_t *__restrict__ states, size_t states_stride,
                                                                   vo_batch_strides bs)
{
    // A kernel of the per-frame pose chain: a few short wavefronts that, with many streams in flight, share their SIMDs
    // with another launch group's tracker (five wavefronts issuing back to back).  Raised issue priority lets them through.
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (bs.count > 1) {
        const size_t b = blockIdx.y;
        counts = (const int *)((const char *)counts + b * bs.out);
        valid += b * bs.out;
        Rt = (const double *)((const char *)Rt + b * bs.work);
        Xc = (const float *)((const char *)Xc + b * bs.work);
        xc = (const float *)((const char *)xc + b * bs.work);
        if (orig) orig = (const int *)((const char *)orig + b * bs.work);
        meta = (vo_frame_meta *)((char *)meta + b * bs.out);
        full += b * bs.mask;
        if (starts) starts = (const unsigned short *)((const char *)starts + b * bs.work);
        if (states) states += b * states_stride;
    }
    const int m = m_host >= 0 ? m_host : meta->m;
    if (blockIdx.x * FM_POINTS >= m && blockIdx.x > 0) return;        // no survivors for this workgroup
    if (out_counts && blockIdx.x == 0)
        for (int c = threadIdx.x; c < c0; c += 64 * FM_WAVES) { out_counts[c] = counts[c]; out_valid[c] = valid[c]; }
    int niters = max_iters, best = -1, best_c = -1;
    frame_replay(counts, valid, c0, m, conf, lane, niters, best, best_c);
    if (blockIdx.x == 0 && wv == 0) {
        if (lane == 0) {
            meta->m = m;
            meta->win = best_c;
            meta->best = best_c >= 0 ? best : 0;
            meta->niters = niters;
            if (m_host >= 0) meta->overflow = 0;
            if (starts && m >= 4 && niters <= c0) meta->state = states[starts[niters]];
        }
        if (best_c >= 0 && lane < 12) meta->Rt[lane] = Rt[12 * (int64_t)best_c + lane];
    }
    if (best_c < 0) return;
    double mdl[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) mdl[k] = Rt[12 * (int64_t)best_c + k];
    const int i0 = blockIdx.x * FM_POINTS + wv * (FM_POINTS / FM_WAVES);
    const int i1 = min(m, i0 + FM_POINTS / FM_WAVES);
    for (int i = i0 + lane; i < i1; i += 64) full[orig ? orig[i] : i] = vo_pnp_inlier(mdl, Xc, xc, i, fx, fy, cx, cy, t) ? 1 : 0;
}

int vo_pnp_first_chunk_tail(vo_ctx *ctx, const float *dX, const float *dx, int n_cap, int m_host, const int *row,
                            const int32_t *dcounts, const uint8_t *dvalid, int c0, int max_iters, double conf, const double *K,
                            double thr, const double *dRt, vo_frame_meta *d_meta, int32_t *out_counts, uint8_t *out_valid,
                            uint8_t *d_mask, int refine_iters, double *d_ref, const vo_batch_strides &bs,
                            const unsigned short *d_starts, const uint64_t *d_states, size_t states_stride)
{
    hipLaunchKernelGGL(frame_mask_kernel, dim3((n_cap + FM_POINTS - 1) / FM_POINTS, bs.count), dim3(64 * FM_WAVES), 0, ctx->stream,
                       dcounts, dvalid, c0, max_iters, conf, dRt, dX, dx, row, K[0], K[4], K[2], K[5], (float)(thr * thr), d_meta,
                       d_mask, m_host, out_counts, out_valid, d_starts, d_states, states_stride, bs);
    VO_HIP_CHECK(hipGetLastError());
    if (refine_iters > 0)
        return vo_refine_pose_from_Rt_async(ctx, dX, dx, n_cap, m_host >= 0 ? nullptr : &d_meta->m, d_mask, row, K, dRt, 0, &d_meta->win,
                                            refine_iters, nullptr, d_ref, bs);
    return VO_OK;
}

// mask[orig[j]] = mask_c[j] for the m survivors (the long path's mask comes in survivor order)
__global__ void frame_scatter_kernel(const int *__restrict__ orig, const uint8_t *__restrict__ mask_c, int m, uint8_t *__restrict__ full)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) full[orig[j]] = mask_c[j];
}

// cv::RNG (multiply-with-carry) states from `state` on: entry p = the state after p draws, FS_DRAWS + 1 of them,
// kept on the device per context
static void frame_states_from(uint64_t state, uint64_t *out)
{
    uint64_t s = state;
    out[0] = s;
    for (int k = 1; k <= FS_DRAWS; ++k) {
        s = (uint64_t)(uint32_t)s * 4164903690u + (s >> 32);
        out[k] = s;
    }
}

static int frame_raw_draws(vo_ctx *ctx, uint64_t state, const uint64_t **out)
{
    void *p;
    int rc;
    if ((rc = vo_stage_reserve(ctx, 9, sizeof(uint64_t) * (FS_DRAWS + 1), &p))) return rc;
    if (!ctx->rng_raw_ok || ctx->rng_raw_seed != state || ctx->rng_raw_dev != p) {
        frame_states_from(state, ctx->rng_raw_host);
        VO_HIP_CHECK(hipMemcpyAsync(p, ctx->rng_raw_host, sizeof(uint64_t) * (FS_DRAWS + 1), hipMemcpyHostToDevice, ctx->stream));
        ctx->rng_raw_seed = state;
        ctx->rng_raw_dev = p;
        ctx->rng_raw_ok = true;
    }
    *out = (const uint64_t *)p;
    return VO_OK;
}

static void frame_matrix_from_rotvec(const double *rv, double *R)
{
    const double th = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (th == 0) { for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0); return; }
    const double kx = rv[0] / th, ky = rv[1] / th, kz = rv[2] / th, c = cos(th), s = sin(th), v = 1. - c;
    R[0] = c + v * kx * kx;      R[1] = v * kx * ky - s * kz; R[2] = v * kx * kz + s * ky;
    R[3] = v * kx * ky + s * kz; R[4] = c + v * ky * ky;      R[5] = v * ky * kz - s * kx;
    R[6] = v * kx * kz - s * ky; R[7] = v * ky * kz + s * kx; R[8] = c + v * kz * kz;
}

extern "C" int vo_track_pnp(vo_ctx *ctx, const vo_pyr *prev, vo_pyr *next, const uint8_t *next_image, int64_t next_row_stride,
                            const float *prev_pts, const float *landmarks, int n, const vo_klt_params *klt, const double *K, double thr, double conf, int max_iters,
                            int refine_iters, uint64_t *rng_state, float *next_pts, uint8_t *status, double *Rt_out,
                            uint8_t *inlier_mask, int32_t *n_tracked, int32_t *n_inliers, int mem)
{
    VO_REQUIRE(ctx && prev && next && klt && K && Rt_out && n_tracked && n_inliers, "vo_track_pnp: NULL argument");
    VO_REQUIRE(n >= 0 && (n == 0 || (prev_pts && landmarks && next_pts && status && inlier_mask)), "vo_track_pnp: bad buffers (n=%d)", n);
    VO_REQUIRE(max_iters >= 1 && max_iters <= (1 << 22) && refine_iters >= 0, "vo_track_pnp: max_iters=%d refine_iters=%d", max_iters, refine_iters);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_track_pnp: bad mem flag %d", mem);
    VO_REQUIRE(prev->batch == 1 && next->batch == 1, "vo_track_pnp: pyramids of one image each (batch %d, %d)", prev->batch, next->batch);
    *n_tracked = 0;
    *n_inliers = 0;
    ctx->frame_path = 0;
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) {
        if (next_image) return vo_pyr_build(ctx, next, next_image, next_row_stride, 0, mem);
        return VO_OK;
    }
    const int c0 = max_iters < FS_SAMPLES ? max_iters : FS_SAMPLES;
    const uint64_t state0 = rng_state ? *rng_state : 0xFFFFFFFFFFFFFFFFull;
    int rc;
    const uint64_t *d_raw;
    if ((rc = frame_raw_draws(ctx, state0, &d_raw))) return rc;
    const bool host = mem == VO_MEM_HOST;
    // One device block: [work | outputs], the outputs contiguous so that one copy takes everything the host reads back
    // out (into page-locked memory); the host's input arrays are staged in the page-locked buffer as well.
    const size_t o_pts = 0, o_X = host ? (((size_t)n * 8 + 255) & ~(size_t)255) : 0,
                 in_bytes = host ? o_X + (((size_t)n * 12 + 255) & ~(size_t)255) : 0;
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_Xc = take((size_t)n * 12), o_xc = take((size_t)n * 8), o_orig = take((size_t)n * 4), o_maskc = take((size_t)n),
                 o_sub = take((size_t)c0 * 16), o_Rt = take((size_t)c0 * 96);
    const size_t out0 = off;
    const size_t o_meta = take(sizeof(vo_frame_meta)), o_ref = take(64), o_counts = take((size_t)c0 * 4 + 4), o_valid = take((size_t)c0);
    const size_t head_bytes = off - out0;
    const size_t o_next = take(host ? (size_t)n * 8 : 0), o_status = take(host ? (size_t)n : 0), o_full = take(host ? (size_t)n : 0);
    const size_t out_bytes = off - out0;
    void *blk;
    if ((rc = vo_stage_reserve(ctx, 8, off, &blk))) return rc;
    char *base = (char *)blk;
    const float *d_pts = prev_pts, *d_X = landmarks;       // host arrays: read in place from page-locked memory, see below
    float *d_next = host ? (float *)(base + o_next) : next_pts;
    uint8_t *d_status = host ? (uint8_t *)(base + o_status) : status;
    uint8_t *d_full = host ? (uint8_t *)(base + o_full) : inlier_mask;
    float *d_Xc = (float *)(base + o_Xc), *d_xc = (float *)(base + o_xc);
    int *d_orig = (int *)(base + o_orig), *d_sub = (int *)(base + o_sub), *d_counts = (int *)(base + o_counts);
    uint8_t *d_maskc = (uint8_t *)(base + o_maskc), *d_valid = (uint8_t *)(base + o_valid);
    vo_frame_meta *d_meta = (vo_frame_meta *)(base + o_meta);
    double *d_Rt = (double *)(base + o_Rt), *d_ref = (double *)(base + o_ref);
    const size_t img_bytes = host && next_image ? (((size_t)next->w[0] * next->h[0] + 255) & ~(size_t)255) : 0;
    void *pin;
    if ((rc = vo_pinned_reserve(ctx, img_bytes + in_bytes + out_bytes, &pin))) return rc;
    char *hin = (char *)pin + img_bytes, *hout = hin + in_bytes;       // hout mirrors base + out0
    const vo_frame_meta *h_meta = (const vo_frame_meta *)(hout + (o_meta - out0));
    const double *h_ref = (const double *)(hout + (o_ref - out0));

    // the new frame's pyramid first: its upload and the points' share the page-locked buffer, nothing waits
    if (next_image) {
        if (host) rc = vo_pyr_build_pinned(ctx, next, next_image, next_row_stride, (uint8_t *)pin);
        else rc = vo_pyr_build(ctx, next, next_image, next_row_stride, 0, VO_MEM_DEVICE);
        if (rc) return rc;
    }
    if (host) {
        // Keypoints and landmarks are read once each (by the tracker and by the compaction): the kernels take them
        // straight from the page-locked buffer, which the device addresses, instead of waiting for a copy engine to
        // hand over to the compute queue (~20 us of idle queue per hand-over on this platform).
        memcpy(hin + o_pts, prev_pts, (size_t)n * 8);
        memcpy(hin + o_X, landmarks, (size_t)n * 12);
        d_pts = (const float *)(hin + o_pts);
        d_X = (const float *)(hin + o_X);
    }
    if ((rc = vo_klt_track(ctx, prev, next, d_pts, nullptr, n, klt, d_next, d_status, nullptr, VO_MEM_DEVICE))) return rc;
    hipLaunchKernelGGL(frame_prep_kernel<FS_THREADS_ONE>, dim3(1), dim3(FS_THREADS_ONE), 0, ctx->stream, d_status, d_X, d_next, n, d_raw, (size_t)0, c0, d_Xc,
                       d_xc, d_orig, d_meta, d_counts, d_full, d_sub, (unsigned short *)nullptr, vo_batch_strides());
    VO_HIP_CHECK(hipGetLastError());
    if (n >= 4) {
        if ((rc = vo_p3p_solve(ctx, d_Xc, d_xc, n, K, d_sub, c0, d_Rt, d_valid, VO_MEM_DEVICE))) return rc;
        if ((rc = vo_score_pnp_dev(ctx, d_Xc, d_xc, n, &d_meta->m, K, d_Rt, c0, nullptr, thr, d_counts, nullptr, true))) return rc;
        if ((rc = vo_pnp_first_chunk_tail(ctx, d_Xc, d_xc, n, -1, d_orig, d_counts, d_valid, c0, max_iters, conf, K, thr, d_Rt, d_meta,
                                          nullptr, nullptr, d_full, refine_iters, d_ref)))
            return rc;
    }
    if ((rc = vo_download_async(ctx, hout, base + out0, host ? out_bytes : head_bytes))) return rc;
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const vo_frame_meta meta = *h_meta;
    const int m = meta.m;
    *n_tracked = m;
    if (host) {
        memcpy(next_pts, hout + (o_next - out0), (size_t)n * 8);
        memcpy(status, hout + (o_status - out0), (size_t)n);
    }
    if (m < 4 || n < 4) {
        if (host) memset(inlier_mask, 0, (size_t)n);      // a device-side mask was cleared by the preparation kernel
        return VO_OK;
    }
    // the host's own replay of the loop decides (its libm computes the adaptive bound OpenCV's would)
    int hb, hbc, hn;
    vo_ransac_replay_first_chunk((const int32_t *)(hout + (o_counts - out0)), (const uint8_t *)(hout + (o_valid - out0)), c0, m, 4, conf,
                                 max_iters, &hb, &hbc, &hn);
    const bool finished = hn <= c0 || c0 >= max_iters;
    const bool agree = !meta.overflow && hbc == meta.win && hn == meta.niters && (hbc < 0 || hb == meta.best);
    if (finished && agree) {
        ctx->frame_path = 1;
        if (rng_state) {     // the generator after the iterations the loop ran, as OpenCV's own loop leaves it
            const int ran = hn < max_iters ? hn : max_iters;
            if ((rc = vo_ransac_advance_state(rng_state, m, 4, ran))) return rc;
        }
        if (hbc < 0) {
            if (host) memset(inlier_mask, 0, (size_t)n);
            return VO_OK;
        }
        double Rt[12];
        memcpy(Rt, meta.Rt, sizeof(Rt));
        if (refine_iters > 0) {
            bool finite = true;
            for (int k = 0; k < 6; ++k) finite = finite && isfinite(h_ref[k]);
            if (finite) {
                double R[9];
                frame_matrix_from_rotvec(h_ref, R);
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rt[4 * r + c] = R[3 * r + c]; Rt[4 * r + 3] = h_ref[3 + r]; }
            }
        }
        memcpy(Rt_out, Rt, sizeof(Rt));
        if (host) memcpy(inlier_mask, hout + (o_full - out0), (size_t)n);
        *n_inliers = hb;
        return VO_OK;
    }
    // the loop wants more samples than the first chunk (or the fast path cannot vouch for its result): the whole
    // RANSAC call on the survivors, which are already on the device
    ctx->frame_path = !agree ? 3 : 2;
    int32_t good = 0;
    if ((rc = vo_solve_pnp_ransac(ctx, d_Xc, d_xc, m, K, thr, conf, max_iters, refine_iters, rng_state, Rt_out, d_maskc, &good, VO_MEM_DEVICE)))
        return rc;
    *n_inliers = good;
    VO_HIP_CHECK(hipMemsetAsync(d_full, 0, (size_t)n, ctx->stream));
    if (good > 0) {
        hipLaunchKernelGGL(frame_scatter_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, d_orig, d_maskc, m, d_full);
        VO_HIP_CHECK(hipGetLastError());
    }
    if (host) {
        if ((rc = vo_pinned_reserve(ctx, (size_t)n, &pin))) return rc;
        VO_HIP_CHECK(hipMemcpyAsync(pin, d_full, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        memcpy(inlier_mask, pin, (size_t)n);
    } else {
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return VO_OK;
}

extern "C" int vo_track_pnp_path(const vo_ctx *ctx) { return ctx ? ctx->frame_path : 0; }

// ---------------------------------------------------------------------------
// The tracked frame of many streams at once: vo_track_pnp for every image pair of two batched pyramids, with one
// launch chain and one synchronisation for all of them.  Tracker (already batched), then per stream -- one
// workgroup, or one grid row, each -- the compaction and the sampler, the P3P solves, the scoring, the loop's replay
// with the winner's mask, and the refit; one copy brings every stream's bookkeeping record, refit and inlier counts
// to the host, which re-checks each stream's loop as vo_track_pnp does and sends the streams whose loop outlives the
// first chunk (or whose device-side bookkeeping it cannot vouch for) through vo_solve_pnp_ransac one by one.
// The first chunk is solved and scored in two stages (FS_STAGE1 samples, then the rest up to the loop's bound after
// those): clean streams end their loop within a handful of iterations, as OpenCV's does.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void frame_states_kernel(const uint64_t *__restrict__ seeds, int B, uint64_t *__restrict__ states)
{
    // one lane per stream: the generator is a serial recurrence (1 536 dependent multiply-adds, ~15 us)
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    uint64_t s = seeds[b];
    uint64_t *o = states + (size_t)b * (FS_DRAWS + 1);
    o[0] = s;
    for (int k = 1; k <= FS_DRAWS; ++k) {
        s = (uint64_t)(uint32_t)s * 4164903690u + (s >> 32);
        o[k] = s;
    }
}

struct vo_frame_batch {
    vo_devbuf work, out, states, in_pts, in_X, o_next, o_status, o_full;   // device
    void *pin = nullptr;            // page-locked: [mirror of `out` | generator seeds]
    size_t pin_cap = 0;
    hipEvent_t done = nullptr;
    bool default_states_ok = false; // `states` holds the one run from OpenCV's initial state
    const void *default_states_dev = nullptr;
    // the call between begin and finish
    bool pending = false;
    int B = 0, n = 0, c0 = 0, max_iters = 0, refine_iters = 0, mem = 0;
    double K[9], thr = 0, conf = 0;
    size_t ws = 0, os = 0, o_Xc = 0, o_xc = 0, o_orig = 0, o_maskc = 0, o_meta = 0, o_ref = 0, o_counts = 0, o_valid = 0;
    bool has_seeds = false;
    std::vector<uint64_t> seeds;
    float *h_next = nullptr;        // the caller's host arrays (VO_MEM_HOST)
    uint8_t *h_status = nullptr, *h_mask = nullptr;
    uint8_t *d_full = nullptr;      // inlier masks on the device (the caller's, or o_full)
};

static int fb_reserve(vo_ctx *ctx, vo_devbuf &b, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    if (b.cap < bytes) {
        if (b.ptr) {
            VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            VO_HIP_CHECK(hipFree(b.ptr));
            b.ptr = nullptr;
            b.cap = 0;
        }
        const size_t cap = bytes + bytes / 4;
        hipError_t e = hipMalloc(&b.ptr, cap);
        if (e != hipSuccess) { vo_set_error("device allocation of %zu bytes failed: %s", cap, hipGetErrorString(e)); return VO_ENOMEM; }
        b.cap = cap;
    }
    *out = b.ptr;
    return VO_OK;
}

void vo_frame_batch_free(vo_ctx *ctx)
{
    vo_frame_batch *fb = ctx->fb;
    if (!fb) return;
    for (vo_devbuf *b : {&fb->work, &fb->out, &fb->states, &fb->in_pts, &fb->in_X, &fb->o_next, &fb->o_status, &fb->o_full})
        if (b->ptr) (void)hipFree(b->ptr);
    if (fb->pin) (void)hipHostFree(fb->pin);
    if (fb->done) (void)hipEventDestroy(fb->done);
    delete fb;
    ctx->fb = nullptr;
}

extern "C" int vo_track_pnp_batch_begin(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts,
                                        const float *landmarks, int n, const vo_klt_params *klt, const double *K, double thr,
                                        double conf, int max_iters, int refine_iters, const uint64_t *rng_state, float *next_pts,
                                        uint8_t *status, uint8_t *inlier_mask, int mem)
{
    VO_REQUIRE(ctx && prev && next && klt && K, "vo_track_pnp_batch: NULL argument");
    VO_REQUIRE(n >= 1 && prev_pts && landmarks && next_pts && status && inlier_mask, "vo_track_pnp_batch: bad buffers (n=%d)", n);
    VO_REQUIRE(max_iters >= 1 && max_iters <= (1 << 22) && refine_iters >= 0, "vo_track_pnp_batch: max_iters=%d refine_iters=%d", max_iters, refine_iters);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_track_pnp_batch: bad mem flag %d", mem);
    VO_REQUIRE(prev->batch == next->batch && prev->batch >= 1 && prev->batch <= 65535, "vo_track_pnp_batch: pyramids of %d and %d images", prev->batch, next->batch);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->fb) {
        ctx->fb = new (std::nothrow) vo_frame_batch();
        if (!ctx->fb) { vo_set_error("vo_track_pnp_batch: out of host memory"); return VO_ENOMEM; }
        VO_HIP_CHECK(hipEventCreateWithFlags(&ctx->fb->done, hipEventDisableTiming));
    }
    vo_frame_batch *fb = ctx->fb;
    VO_REQUIRE(!fb->pending, "vo_track_pnp_batch_begin: the context's previous batch has not been finished");
    const int B = prev->batch, c0 = max_iters < FS_SAMPLES ? max_iters : FS_SAMPLES;
    const size_t cnt = (size_t)B * n;
    int rc;
    // one stream's work block and read-back block
    size_t off = 0;
    auto take = [&off](size_t bytes, size_t align) { off = (off + align - 1) & ~(align - 1); const size_t o = off; off += bytes; return o; };
    const size_t o_Xc = take((size_t)n * 12, 256), o_xc = take((size_t)n * 8, 256), o_orig = take((size_t)n * 4, 256),
                 o_maskc = take((size_t)n, 256), o_sub = take((size_t)c0 * 16, 256), o_Rt = take((size_t)c0 * 96, 256),
                 o_starts = take((size_t)(c0 + 1) * 2, 256);
    const size_t ws = (off + 255) & ~(size_t)255;
    off = 0;
    const size_t o_meta = take(sizeof(vo_frame_meta), 16), o_ref = take(64, 16), o_counts = take((size_t)c0 * 4 + 4, 16),
                 o_valid = take((size_t)c0, 16);
    const size_t os = (off + 255) & ~(size_t)255;
    void *p;
    if ((rc = fb_reserve(ctx, fb->work, ws * B, &p))) return rc;
    char *work = (char *)p;
    if ((rc = fb_reserve(ctx, fb->out, os * B, &p))) return rc;
    char *out = (char *)p;
    const size_t pin_need = os * B + (((size_t)B * 8 + 255) & ~(size_t)255);
    if (fb->pin_cap < pin_need) {
        if (fb->pin) {
            VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            VO_HIP_CHECK(hipHostFree(fb->pin));
            fb->pin = nullptr;
            fb->pin_cap = 0;
        }
        VO_HIP_CHECK(hipHostMalloc(&fb->pin, pin_need + pin_need / 4, hipHostMallocDefault));
        fb->pin_cap = pin_need + pin_need / 4;
    }
    // generator states: one run for all streams from OpenCV's initial state, or one per stream from the caller's seeds
    const uint64_t *d_states;
    size_t states_stride = 0;
    if (!rng_state) {
        if ((rc = fb_reserve(ctx, fb->states, sizeof(uint64_t) * (FS_DRAWS + 1), &p))) return rc;
        if (!fb->default_states_ok || fb->default_states_dev != p) {
            frame_states_from(0xFFFFFFFFFFFFFFFFull, ctx->rng_raw_host);
            ctx->rng_raw_ok = false;      // the single-frame call's cache shares the host array
            VO_HIP_CHECK(hipMemcpyAsync(p, ctx->rng_raw_host, sizeof(uint64_t) * (FS_DRAWS + 1), hipMemcpyHostToDevice, ctx->stream));
            VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            fb->default_states_ok = true;
            fb->default_states_dev = p;
        }
        d_states = (const uint64_t *)p;
        fb->has_seeds = false;
    } else {
        if ((rc = fb_reserve(ctx, fb->states, sizeof(uint64_t) * (FS_DRAWS + 1) * B, &p))) return rc;
        fb->default_states_ok = false;
        uint64_t *pin_seeds = (uint64_t *)((char *)fb->pin + os * B);
        memcpy(pin_seeds, rng_state, (size_t)B * 8);
        fb->seeds.assign(rng_state, rng_state + B);
        fb->has_seeds = true;
        hipLaunchKernelGGL(frame_states_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, (const uint64_t *)pin_seeds, B, (uint64_t *)p);
        VO_HIP_CHECK(hipGetLastError());
        d_states = (const uint64_t *)p;
        states_stride = FS_DRAWS + 1;
    }
    const float *d_pts = prev_pts, *d_X = landmarks;
    float *d_next = next_pts;
    uint8_t *d_status = status, *d_full = inlier_mask;
    if (mem == VO_MEM_HOST) {
        if ((rc = fb_reserve(ctx, fb->in_pts, cnt * 8, &p))) return rc; d_pts = (const float *)p;
        if ((rc = fb_reserve(ctx, fb->in_X, cnt * 12, &p))) return rc; d_X = (const float *)p;
        if ((rc = fb_reserve(ctx, fb->o_next, cnt * 8, &p))) return rc; d_next = (float *)p;
        if ((rc = fb_reserve(ctx, fb->o_status, cnt, &p))) return rc; d_status = (uint8_t *)p;
        if ((rc = fb_reserve(ctx, fb->o_full, cnt, &p))) return rc; d_full = (uint8_t *)p;
        VO_UPLOAD(ctx, d_pts, prev_pts, cnt * 8);
        VO_UPLOAD(ctx, d_X, landmarks, cnt * 12);
    }
    vo_batch_strides bs;
    bs.count = B; bs.work = ws; bs.out = os; bs.mask = (size_t)n;
    float *d_Xc = (float *)(work + o_Xc), *d_xc = (float *)(work + o_xc);
    int *d_orig = (int *)(work + o_orig), *d_sub = (int *)(work + o_sub), *d_counts = (int *)(out + o_counts);
    uint8_t *d_valid = (uint8_t *)(out + o_valid);
    unsigned short *d_starts = (unsigned short *)(work + o_starts);
    vo_frame_meta *d_meta = (vo_frame_meta *)(out + o_meta);
    double *d_Rt = (double *)(work + o_Rt), *d_ref = (double *)(out + o_ref);
    if (ctx->prof_tracker) VO_HIP_CHECK(hipEventRecord(ctx->prof0, ctx->stream));
    if ((rc = vo_klt_track(ctx, prev, next, d_pts, nullptr, n, klt, d_next, d_status, nullptr, VO_MEM_DEVICE))) return rc;
    if (ctx->prof_tracker) {
        VO_HIP_CHECK(hipEventRecord(ctx->prof1, ctx->stream));
        ctx->prof_recorded = true;
    }
    hipLaunchKernelGGL(frame_prep_kernel<FS_THREADS>, dim3(B), dim3(FS_THREADS), 0, ctx->stream, (const uint8_t *)d_status, d_X, (const float *)d_next, n,
                       d_states, states_stride, c0, d_Xc, d_xc, d_orig, d_meta, d_counts, d_full, d_sub, d_starts, bs);
    VO_HIP_CHECK(hipGetLastError());
    // VO_FRAME_STAGE1 in the environment moves the first stage's size (0 or >= the chunk: one stage; comparison runs)
    const char *stage1_env = getenv("VO_FRAME_STAGE1");
    const int stage1 = stage1_env ? atoi(stage1_env) : FS_STAGE1;
    const int c1 = stage1 > 0 && stage1 < c0 ? stage1 : c0;
    if ((rc = vo_p3p_solve_dev(ctx, d_Xc, d_xc, n, K, d_sub, c1, d_Rt, d_valid, bs))) return rc;
    if ((rc = vo_score_pnp_dev(ctx, d_Xc, d_xc, n, &d_meta->m, K, d_Rt, c1, nullptr, thr, d_counts, nullptr, true, bs))) return rc;
    if (c1 < c0) {
        hipLaunchKernelGGL(frame_bound_kernel, dim3(B), dim3(64), 0, ctx->stream, (const int *)d_counts, (const uint8_t *)d_valid, c1, max_iters,
                           conf, d_meta, bs);
        VO_HIP_CHECK(hipGetLastError());
        if ((rc = vo_p3p_solve_dev(ctx, d_Xc, d_xc, n, K, d_sub, c0, d_Rt, d_valid, bs, c1, &d_meta->bound0))) return rc;
        if ((rc = vo_score_pnp_dev(ctx, d_Xc, d_xc, n, &d_meta->m, K, d_Rt, c0, nullptr, thr, d_counts, nullptr, true, bs, c1, &d_meta->bound0)))
            return rc;
    }
    if ((rc = vo_pnp_first_chunk_tail(ctx, d_Xc, d_xc, n, -1, d_orig, d_counts, d_valid, c0, max_iters, conf, K, thr, d_Rt, d_meta, nullptr,
                                      nullptr, d_full, refine_iters, d_ref, bs, d_starts, d_states, states_stride)))
        return rc;
    if ((rc = vo_download_async(ctx, fb->pin, out, os * B))) return rc;
    VO_HIP_CHECK(hipEventRecord(fb->done, ctx->stream));
    fb->pending = true;
    fb->B = B; fb->n = n; fb->c0 = c0; fb->max_iters = max_iters; fb->refine_iters = refine_iters; fb->mem = mem;
    memcpy(fb->K, K, sizeof(fb->K));
    fb->thr = thr; fb->conf = conf;
    fb->ws = ws; fb->os = os; fb->o_Xc = o_Xc; fb->o_xc = o_xc; fb->o_orig = o_orig; fb->o_maskc = o_maskc;
    fb->o_meta = o_meta; fb->o_ref = o_ref; fb->o_counts = o_counts; fb->o_valid = o_valid;
    fb->h_next = next_pts; fb->h_status = status; fb->h_mask = inlier_mask;
    fb->d_full = d_full;
    return VO_OK;
}

extern "C" int vo_track_pnp_batch_finish(vo_ctx *ctx, uint64_t *rng_state, double *Rt_out, int32_t *n_tracked, int32_t *n_inliers,
                                         int32_t *path)
{
    VO_REQUIRE(ctx && Rt_out && n_tracked && n_inliers, "vo_track_pnp_batch_finish: NULL argument");
    vo_frame_batch *fb = ctx->fb;
    VO_REQUIRE(fb && fb->pending, "vo_track_pnp_batch_finish: no batch was begun on this context");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    fb->pending = false;
    VO_HIP_CHECK(hipEventSynchronize(fb->done));
    const int B = fb->B, n = fb->n, c0 = fb->c0, max_iters = fb->max_iters;
    const char *hout = (const char *)fb->pin;
    std::vector<int> slow;
    for (int b = 0; b < B; ++b) {
        const char *hb_ = hout + (size_t)b * fb->os;
        const vo_frame_meta meta = *(const vo_frame_meta *)(hb_ + fb->o_meta);
        const double *h_ref = (const double *)(hb_ + fb->o_ref);
        const int m = meta.m;
        n_tracked[b] = m;
        n_inliers[b] = 0;
        if (path) path[b] = 0;
        if (rng_state && fb->has_seeds) rng_state[b] = fb->seeds[b];
        else if (rng_state) rng_state[b] = 0xFFFFFFFFFFFFFFFFull;
        if (m < 4 || n < 4) continue;        // the masks were cleared by the preparation kernel
        int hb, hbc, hn;
        vo_ransac_replay_first_chunk((const int32_t *)(hb_ + fb->o_counts), (const uint8_t *)(hb_ + fb->o_valid), c0, m, 4, fb->conf, max_iters,
                                     &hb, &hbc, &hn);
        const bool finished = hn <= c0 || c0 >= max_iters;
        const bool agree = !meta.overflow && hbc == meta.win && hn == meta.niters && (hbc < 0 || hb == meta.best);
        if (!(finished && agree)) {
            if (path) path[b] = !agree ? 3 : 2;
            slow.push_back(b);
            continue;
        }
        if (path) path[b] = 1;
        if (rng_state) rng_state[b] = meta.state;
        if (hbc < 0) continue;
        double *Rt = Rt_out + 12 * (size_t)b;
        memcpy(Rt, meta.Rt, sizeof(meta.Rt));
        if (fb->refine_iters > 0) {
            bool finite = true;
            for (int k = 0; k < 6; ++k) finite = finite && isfinite(h_ref[k]);
            if (finite) {
                double R[9];
                frame_matrix_from_rotvec(h_ref, R);
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rt[4 * r + c] = R[3 * r + c]; Rt[4 * r + 3] = h_ref[3 + r]; }
            }
        }
        n_inliers[b] = hb;
    }
    // streams whose loop wants more samples than the first chunk (or whose device-side bookkeeping the host cannot
    // vouch for): the whole RANSAC call on the stream's survivors, which are already on the device
    int rc;
    for (int b : slow) {
        char *wb = (char *)fb->work.ptr + (size_t)b * fb->ws;
        const int m = n_tracked[b];
        uint64_t st = fb->has_seeds ? fb->seeds[b] : 0xFFFFFFFFFFFFFFFFull;
        int32_t good = 0;
        uint8_t *d_maskc = (uint8_t *)(wb + fb->o_maskc), *d_full = fb->d_full + (size_t)b * n;
        if ((rc = vo_solve_pnp_ransac(ctx, (const float *)(wb + fb->o_Xc), (const float *)(wb + fb->o_xc), m, fb->K, fb->thr, fb->conf, max_iters,
                                      fb->refine_iters, &st, Rt_out + 12 * (size_t)b, d_maskc, &good, VO_MEM_DEVICE)))
            return rc;
        n_inliers[b] = good;
        if (rng_state) rng_state[b] = st;
        VO_HIP_CHECK(hipMemsetAsync(d_full, 0, (size_t)n, ctx->stream));
        if (good > 0) {
            hipLaunchKernelGGL(frame_scatter_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, (const int *)(wb + fb->o_orig),
                               (const uint8_t *)d_maskc, m, d_full);
            VO_HIP_CHECK(hipGetLastError());
        }
    }
    if (fb->mem == VO_MEM_HOST) {
        const size_t cnt = (size_t)B * n;
        VO_HIP_CHECK(hipMemcpyAsync(fb->h_next, fb->o_next.ptr, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipMemcpyAsync(fb->h_status, fb->o_status.ptr, cnt, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipMemcpyAsync(fb->h_mask, fb->o_full.ptr, cnt, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } else if (!slow.empty()) {
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return VO_OK;
}

extern "C" int vo_track_pnp_batch(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts, const float *landmarks,
                                  int n, const vo_klt_params *klt, const double *K, double thr, double conf, int max_iters,
                                  int refine_iters, uint64_t *rng_state, float *next_pts, uint8_t *status, double *Rt_out,
                                  uint8_t *inlier_mask, int32_t *n_tracked, int32_t *n_inliers, int32_t *path, int mem)
{
    VO_REQUIRE(Rt_out && n_tracked && n_inliers, "vo_track_pnp_batch: NULL argument");
    int rc = vo_track_pnp_batch_begin(ctx, prev, next, prev_pts, landmarks, n, klt, K, thr, conf, max_iters, refine_iters, rng_state, next_pts,
                                      status, inlier_mask, mem);
    if (rc) return rc;
    return vo_track_pnp_batch_finish(ctx, rng_state, Rt_out, n_tracked, n_inliers, path);
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_frame(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)frame_prep_kernel<FS_THREADS>));
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)frame_prep_kernel<FS_THREADS_ONE>));
    return VO_OK;
}
