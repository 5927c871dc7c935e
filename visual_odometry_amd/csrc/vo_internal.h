// Internal declarations shared by the HIP translation units of libvo_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/vo_hip.h"

#define VO_MAX_LEVELS 10

void vo_set_error(const char *fmt, ...);

#define VO_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            vo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VO_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define VO_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            vo_set_error(__VA_ARGS__);                                                  \
            return VO_EINVAL;                                                           \
        }                                                                               \
    } while (0)

// Growable device scratch owned by a context (staging of host-side arguments).
struct vo_devbuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct vo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;    // created by vo_ctx_create_own_stream, destroyed with the context
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    vo_devbuf stage[10];         // 0-7: staging of host-side arguments; 8, 9: vo_track_pnp's block and generator states
    void *pinned = nullptr;      // page-locked host buffer for small read-backs (a copy into pageable memory blocks the caller)
    size_t pinned_cap = 0;
    // cv::RNG states that stage[9] holds (vo_track_pnp draws its first samples on the device): entry p is the state
    // after p draws, a draw is the low half of the state after it
    uint64_t rng_raw_host[1537];
    uint64_t rng_raw_seed = 0;
    const void *rng_raw_dev = nullptr;
    bool rng_raw_ok = false;
    int frame_path = 0;          // how the last vo_track_pnp ended (vo_track_pnp_path)
    // page-locked ring that host arrays pass through on their way to the device (vo_upload)
    char *up_ring = nullptr;
    size_t up_cap = 0, up_off = 0;
    // page-locked area that results pass through on their way to the caller's arrays (vo_result / vo_results_finish)
    char *res_area = nullptr;
    size_t res_off = 0;
    struct { void *dst; const char *src; size_t bytes; } res_pending[16];
    int res_count = 0;
    // device pool of the last destroyed bundle-adjustment solver, handed to the next one that fits (vo_ba_create)
    char *ba_pool = nullptr;
    size_t ba_pool_bytes = 0;
    // device and page-locked state of vo_track_pnp_batch (vo_frame.hip), kept from call to call
    struct vo_frame_batch *fb = nullptr;
    // vo_sift_detect_compute: room for scale-space extrema and raw keypoints (grown when an image overflows it)
    int sift_cand_cap = 0, sift_kp_cap = 0;
    // vo_profile_tracker: event pair around the tracker launch of the next vo_track_pnp_batch_begin
    bool prof_tracker = false, prof_recorded = false;
    hipEvent_t prof0 = nullptr, prof1 = nullptr;
};

// vo_frame.hip: frees what ctx->fb holds (vo_ctx_destroy)
void vo_frame_batch_free(vo_ctx *ctx);

bool vo_ctx_is_live(const vo_ctx *ctx);

// One per translation unit with kernels: asks the runtime for the attributes of one of the file's kernels, which
// loads the file's device code on the current device.  vo_ctx_create calls them all the first time a device gets a
// context, so that the tens of milliseconds of code loading are spent there and not inside the first frame.
int vo_load_code_ba(void);
int vo_load_code_cc(void);
int vo_load_code_frame(void);
int vo_load_code_harris(void);
int vo_load_code_klt(void);
int vo_load_code_match(void);
int vo_load_code_pyramid(void);
int vo_load_code_score(void);
int vo_load_code_sift(void);
int vo_load_code_solvers(void);
int vo_load_code_subpix(void);
int vo_load_code_twoview(void);

// Reprojection test of solvePnPRansac's scoring: point i of pa (n,3) under the 3x4 model m against pb (n,2);
// float64 in this order, the squared error compared in float32 (every kernel that scores PnP candidates uses it,
// so their masks agree bit for bit).
__device__ __forceinline__ bool vo_pnp_inlier(const double *m, const float *__restrict__ pa, const float *__restrict__ pb, int64_t i,
                                              double fx, double fy, double cx, double cy, float t)
{
    double Xw = pa[3 * i], Yw = pa[3 * i + 1], Zw = pa[3 * i + 2];
    double xc = m[0] * Xw + m[1] * Yw + m[2] * Zw + m[3];
    double yc = m[4] * Xw + m[5] * Yw + m[6] * Zw + m[7];
    double zc = m[8] * Xw + m[9] * Yw + m[10] * Zw + m[11];
    zc = zc ? 1. / zc : 1;
    xc *= zc;
    yc *= zc;
    float u = (float)(xc * fx + cx), v = (float)(yc * fy + cy);
    float du = pb[2 * i] - u, dv = pb[2 * i + 1] - v;
    float err = du * du + dv * dv;
    return err <= t;
}

// Looks up the current device's copy of the fixed scratch array of vo_klt_track_fb_noalloc (vo_klt.hip).
int vo_klt_noalloc_setup(int device);

// Page-locked host scratch of at least `bytes` (contents survive until the next call that asks for more).
int vo_pinned_reserve(vo_ctx *ctx, size_t bytes, void **out);

// Up to three small host arrays, already in page-locked memory, into device buffers by one kernel that reads them over
// the bus (byte counts multiples of 4).  A copy engine would do the same, but handing over from the engine to the
// compute queue leaves the queue idle for ~10-20 us, more than these few KB take.  Asynchronous.
struct vo_copy_seg { void *dst; const void *src; size_t bytes; };
int vo_copy_pinned_async(vo_ctx *ctx, const vo_copy_seg *segs, int count);
// One larger block (a frame) the same way, 16 bytes per lane: both buffers 16-byte aligned with room for the byte count
// rounded up to 16.
int vo_copy_pinned_wide_async(vo_ctx *ctx, void *dst, const void *src, size_t bytes);

// Host array -> device buffer, asynchronous on the context's stream, the caller's memory free on return.  Arrays up to
// 1 MiB are copied into a page-locked ring and brought over by a kernel that reads the ring across the bus (no copy
// engine, so no hand-over to the compute queue in front of the kernel that consumes them -- worth ~100 us per frame-sized
// upload on this platform); larger ones take hipMemcpyAsync.  A region of the ring is reused only after a wait for
// the stream.  vo_upload_2d packs `height` rows of `width` bytes (source pitch `src_stride`) tightly.
int vo_upload(vo_ctx *ctx, void *dst, const void *src, size_t bytes);
int vo_upload_2d(vo_ctx *ctx, void *dst, const void *src, size_t src_stride, size_t width, size_t height);
#define VO_UPLOAD(ctx, dst, src, bytes)                                                  \
    do {                                                                                \
        int _u = vo_upload((ctx), (void *)(dst), (src), (bytes));                       \
        if (_u) return _u;                                                              \
    } while (0)
#define VO_UPLOAD_2D(ctx, dst, src, src_stride, width, height)                           \
    do {                                                                                \
        int _u = vo_upload_2d((ctx), (void *)(dst), (src), (src_stride), (width), (height)); \
        if (_u) return _u;                                                              \
    } while (0)

// Device buffer -> page-locked host memory by a kernel that writes across the bus (both 16-byte aligned, room for the
// byte count rounded up to 16 on both sides): the read-back then follows the producing kernel on the compute queue
// instead of waiting for a copy engine to take over.  Asynchronous; the host reads after synchronising the stream.
int vo_download_async(vo_ctx *ctx, void *pinned_dst, const void *dev_src, size_t bytes);

// Device buffer -> the caller's (pageable) host array: vo_result queues a kernel that writes a page-locked region and
// notes where the bytes belong; vo_results_finish waits for the stream once and copies every noted region home.  (A copy
// straight into pageable memory blocks the caller per array, and a copy engine's hand-over costs more than these few KB.)
// Arrays beyond 1 MiB, or more than 16 per call, take hipMemcpyAsync.
int vo_result(vo_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
int vo_results_finish(vo_ctx *ctx);
void vo_results_abandon(vo_ctx *ctx);      // forget the noted regions (a call that fails between noting and finishing)
#define VO_RESULT(ctx, dst, src, bytes)                                                  \
    do {                                                                                \
        int _r = vo_result((ctx), (void *)(dst), (src), (bytes));                       \
        if (_r) { vo_results_abandon(ctx); return _r; }                                 \
    } while (0)
#define VO_RESULTS_FINISH(ctx)                                                           \
    do {                                                                                \
        int _r = vo_results_finish(ctx);                                                \
        if (_r) return _r;                                                              \
    } while (0)

// Make stage[slot] at least `bytes` large; returns its device pointer in *out.
int vo_stage_reserve(vo_ctx *ctx, int slot, size_t bytes, void **out);

// Several independent problems of one shape in one launch (the streams of vo_track_pnp_batch): problem b's arrays
// start `work` bytes (points, subsets, candidate models, row maps), `out` bytes (counts, validity flags, the
// bookkeeping record and the refit's result) or `mask` bytes (the caller's inlier mask) after problem b-1's.
// count = 1 with zero strides is the single problem every other caller has.
struct vo_batch_strides {
    int count = 1;
    size_t work = 0, out = 0, mask = 0;
};

// vo_score.hip: vo_score_pnp on device pointers whose point count (*n_dev <= n_cap) and, optionally, first model
// (*first_dev; negative: nothing to do) are known only on the device; counts_cleared: dcounts is already zero.
// hyp_lo_dev (optional): candidates below *hyp_lo_dev are left alone -- and, with skip_bound_dev, candidates from
// *skip_bound_dev + 1 on as well (the RANSAC loop cannot reach them any more).  Asynchronous.
int vo_score_pnp_dev(vo_ctx *ctx, const float *dX, const float *dx, int n_cap, const int *n_dev, const double *K,
                     const double *dRt, int H, const int *first_dev, double thr, int32_t *dcounts, uint8_t *dmasks,
                     bool counts_cleared, const vo_batch_strides &bs = vo_batch_strides(), int hyp_lo = 0,
                     const int *skip_bound_dev = nullptr);
// vo_solvers.hip: vo_p3p_solve on device pointers, for bs.count problems; samples below samp_lo are left alone, and
// with skip_bound_dev those from *skip_bound_dev + 1 on as well.  Asynchronous.
int vo_p3p_solve_dev(vo_ctx *ctx, const float *dX, const float *dx, int n, const double *K, const int32_t *dsub, int S,
                     double *dRt, uint8_t *dvalid, const vo_batch_strides &bs, int samp_lo = 0,
                     const int *skip_bound_dev = nullptr);

// vo_ba.hip: pose refits on device pointers; both enqueue the copy of {pose (6), cost, iterations} into host_out
// and do not synchronise.  The second starts from [R|t] number sel (or *sel_dev) of dRt, with the number of
// correspondences optionally on the device (*n_dev <= n), selecting correspondence i by dmask[mask_row[i]] when
// mask_row is given, and leaving the result in d_out (device) instead of host_out when d_out is given.
int vo_refine_pose_async(vo_ctx *ctx, const float *dX, const float *dx, int n, const uint8_t *dmask, const double *K,
                         const double *pose_in, int max_iters, double *host_out);
int vo_refine_pose_from_Rt_async(vo_ctx *ctx, const float *dX, const float *dx, int n, const int *n_dev, const uint8_t *dmask,
                                 const int *mask_row, const double *K, const double *dRt, int sel, const int *sel_dev,
                                 int max_iters, double *host_out, double *d_out,
                                 const vo_batch_strides &bs = vo_batch_strides());

// What the device-side RANSAC bookkeeping of a first chunk of PnP samples leaves behind (vo_frame.hip).
struct vo_frame_meta {
    int m;            // correspondences (vo_track_pnp: those with status 1)
    int win;          // winning sample of the first chunk, -1: none
    int best;         // its inlier count
    int niters;       // the loop's iteration bound after the first chunk
    int overflow;     // vo_track_pnp: the raw draws did not suffice for the chunk's subsets
    int bound0;       // staged first chunk: the loop's iteration bound after the first stage's candidates
    uint64_t state;   // the generator after the iterations the loop ran (when it ended inside the chunk)
    double Rt[12];    // the winner
};
// vo_frame.hip: after the c0 candidates dRt (validity dvalid) have been scored into dcounts -- replay of OpenCV's loop
// over them, the winner's inlier mask into d_mask[row[i]] (row = NULL: d_mask[i]) and the refit started from the
// winner, all on device-resident state: the winner goes to *d_meta, the refit's {pose (6), cost, iterations} to d_ref.
// m_host >= 0: the number of correspondences; < 0: it is d_meta->m, left there by an earlier kernel.  out_counts /
// out_valid (optional): copies of dcounts / dvalid next to d_meta, so that one copy takes everything to the host.
// Asynchronous.
int vo_pnp_first_chunk_tail(vo_ctx *ctx, const float *dX, const float *dx, int n_cap, int m_host, const int *row,
                            const int32_t *dcounts, const uint8_t *dvalid, int c0, int max_iters, double conf, const double *K,
                            double thr, const double *dRt, vo_frame_meta *d_meta, int32_t *out_counts, uint8_t *out_valid,
                            uint8_t *d_mask, int refine_iters, double *d_ref, const vo_batch_strides &bs = vo_batch_strides(),
                            const unsigned short *d_starts = nullptr, const uint64_t *d_states = nullptr, size_t states_stride = 0);

// vo_pyramid.hip: one host image into a one-image pyramid through page-locked memory the caller provides (w * h
// bytes, untouched until the stream has run the copy); does not wait for the device.
int vo_pyr_build_pinned(vo_ctx *ctx, vo_pyr *pyr, const uint8_t *img, int64_t row_stride, uint8_t *pin);

// vo_ransac.hip: OpenCV's RANSAC loop replayed over the candidates of the first c iterations (one candidate per
// iteration, valid[i] == 0: none) -> best count, its iteration (-1: none), the loop's iteration bound afterwards.
void vo_ransac_replay_first_chunk(const int32_t *counts, const uint8_t *valid, int c, int n, int model_points, double conf,
                                  int max_iters, int *best, int *best_c, int *niters);
// vo_ransac.hip: advance a cv::RNG state past `iterations` subsets of model_points distinct indices below count.
int vo_ransac_advance_state(uint64_t *rng_state, int count, int model_points, int iterations);

struct vo_pyr {
    vo_ctx *ctx = nullptr;
    int batch = 0, win = 0, max_level = 0;  // max_level: effective last level
    int h[VO_MAX_LEVELS] = {0}, w[VO_MAX_LEVELS] = {0};
    int64_t pitch[VO_MAX_LEVELS] = {0};       // bytes between rows
    int64_t img_stride[VO_MAX_LEVELS] = {0};  // bytes between images
    uint8_t *lvl[VO_MAX_LEVELS] = {nullptr};  // device pointers; lvl[0] may alias caller memory
    uint8_t *pool = nullptr;                  // levels >= 1
    uint8_t *own0 = nullptr;                  // library-owned level 0 (host-memory builds)
    size_t own0_bytes = 0;
    bool built = false;
};
