/*
 * libvo_hip -- C-ABI of the MI355X (gfx950) data-parallel front end of the
 * visual-odometry pipeline.  Every entry point is `extern "C"`, takes plain
 * pointers and sizes, returns 0 on success or a negative VO_E* code, and never
 * throws; vo_last_error() holds the message of the calling thread's last failure.
 *
 * The reference (pure Python) has no FFI; the boundary these functions replace is
 * the cv2/scipy call made inside each vo_pipeline method.  Each declaration cites
 * the reference call site it stands in for (paths relative to the reference root).
 *
 * Memory: the caller owns every input and output buffer.  `mem` says where they
 * live: VO_MEM_HOST (the library stages through its own device buffers and
 * synchronises before returning) or VO_MEM_DEVICE (raw device pointers, e.g.
 * torch tensor.data_ptr(); the call is asynchronous on the context's stream).
 * A context is bound to one GPU and one HIP stream and is not re-entrant.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_OK 0
#define VO_EINVAL (-1)   /* bad argument (shape, null pointer, range) */
#define VO_EHIP (-2)     /* a HIP runtime call failed */
#define VO_ENOMEM (-3)   /* allocation failed */
#define VO_ESTATE (-4)   /* object used in the wrong state */

#define VO_MEM_HOST 0
#define VO_MEM_DEVICE 1

typedef struct vo_ctx vo_ctx;
typedef struct vo_pyr vo_pyr;

/* ---- context ------------------------------------------------------------ */
const char *vo_last_error(void);
int vo_version(void);
/* stream: a hipStream_t (may be NULL = the device's default stream). */
int vo_ctx_create(int device, void *stream, vo_ctx **out);
/* A context on a stream of its own (non-blocking: not ordered with the device's default stream), destroyed with it:
 * for work that runs beside another context's, e.g. the two-view bootstrap beside a bundle-adjustment window. */
int vo_ctx_create_own_stream(int device, vo_ctx **out);
int vo_ctx_destroy(vo_ctx *ctx);
int vo_sync(vo_ctx *ctx);
/* Timing helpers (HIP events on the context's stream). */
int vo_timer_start(vo_ctx *ctx);
int vo_timer_stop_ms(vo_ctx *ctx, float *ms);  /* records stop, synchronises, returns elapsed */

/* Plain device buffers on the context's GPU, for callers that bring no tensor library (the reference keeps every
 * array in numpy; the pipeline's bootstrap keeps SIFT descriptors on the GPU between extraction and matching --
 * vo_pipeline/bootstrap.py:179-191 -- without crossing the bus twice).  vo_dev_upload is asynchronous on the
 * context's stream and the host array is free on return; vo_dev_free and vo_dev_download synchronise the stream. */
int vo_dev_alloc(vo_ctx *ctx, size_t bytes, void **out);
int vo_dev_free(vo_ctx *ctx, void *ptr);
int vo_dev_upload(vo_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int vo_dev_download(vo_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);

/* ---- image pyramid ------------------------------------------------------
 * Replaces the pyramid that cv.calcOpticalFlowPyrLK builds internally for both
 * images on every call (vo_pipeline/poseEstimation.py:122-132): Gaussian 5-tap
 * pyrDown, REFLECT_101, level count clamped so every level is larger than `win`.
 * A vo_pyr holds `batch` images of one size; level 0 aliases the caller's device
 * buffer (VO_MEM_DEVICE: it must stay valid while the pyramid is in use) or a
 * library-owned copy (VO_MEM_HOST).
 */
int vo_pyr_create(vo_ctx *ctx, int batch, int height, int width, int max_level, int win, vo_pyr **out);
int vo_pyr_destroy(vo_pyr *pyr);
int vo_pyr_max_level(const vo_pyr *pyr);  /* effective last level (0-based) */
int vo_pyr_level_shape(const vo_pyr *pyr, int level, int *height, int *width, int64_t *pitch);
/* img: batch images, row stride `row_stride` bytes, image stride `img_stride` bytes.  VO_MEM_HOST: the caller's
 * buffer is free again when the call returns; the pyramid itself is built on the context's stream, ahead of
 * whatever is queued on it next (images up to 1 MiB are staged in page-locked memory and nothing waits for the
 * device; larger ones are uploaded and built before the call returns). */
int vo_pyr_build(vo_ctx *ctx, vo_pyr *pyr, const uint8_t *img, int64_t row_stride, int64_t img_stride, int mem);
/* Copy one level of one image to a tight host buffer (tests / debugging). */
int vo_pyr_download(vo_ctx *ctx, const vo_pyr *pyr, int image, int level, uint8_t *host_out);

/* ---- pyramidal Lucas-Kanade ---------------------------------------------
 * Replaces cv.calcOpticalFlowPyrLK(img0, img1, float32(prev_kpts), None,
 * winSize, maxLevel, criteria=(COUNT|EPS, iters, eps), flags=
 * OPTFLOW_LK_GET_MIN_EIGENVALS, minEigThreshold) -- PoseEstimation.KLT,
 * vo_pipeline/poseEstimation.py:116-133; caller vo_pipeline/continuousVO.py:167.
 * prev/next: pyramids of the same geometry.  Points are (batch, n, 2) float32
 * (x, y); image b of `prev` is tracked into image b of `next`.
 * init_pts may be NULL (start at prev_pts, as the reference does).
 * status: (batch, n) uint8, 1 = tracked.  err: (batch, n) float32 min-eigenvalue,
 * may be NULL.
 * Accuracy: positions agree with a scalar evaluation of the same algorithm to 0.05 px
 * (observed <= 0.003 px) at the reference's criteria (10 iterations, eps 1e-3).  The
 * kernels add the 2x2 systems exactly in int32 where the scalar code adds float32 terms,
 * so a step whose squared length lies within rounding of eps^2 can stop one iteration
 * earlier or later; with a coarse eps (>= ~0.03 px) that last step is itself larger
 * than 0.05 px and a few keypoints in 10^5 land up to ~0.2 px apart.
 */
typedef struct vo_klt_params {
    int max_iters;     /* criteria count (clamped to [0,100]) */
    float eps;         /* criteria epsilon (clamped to [0,10], squared internally) */
    float min_eig;     /* minEigThreshold */
} vo_klt_params;

int vo_klt_track(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next,
                 const float *prev_pts, const float *init_pts, int n,
                 const vo_klt_params *params,
                 float *next_pts, uint8_t *status, float *err, int mem);

/* Forward + backward track and the bidirectional test the reference's
 * TrackPoints.trackKLT intends (vo_pipeline/trackPoints.py:19-64, 154-161):
 * accept when |p - back(fwd(p))| < lambda in both axes.  mask: (batch, n) uint8.
 * next_pts holds the forward result for accepted points and (0,0) otherwise. */
int vo_klt_track_fb(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next,
                    const float *prev_pts, int n, const vo_klt_params *params,
                    float lambda, float *next_pts, uint8_t *mask, int mem);

/* vo_klt_track_fb_noalloc has no counterpart in the reference.  It is vo_klt_track_fb for callers
 * that cannot allocate device memory while they run -- tight serving loops, harnesses that forbid
 * hipMalloc: same kernels, same results, device pointers only
 * (VO_MEM_DEVICE semantics, asynchronous on the context's stream), and the intermediate results
 * (back-tracked points, the two status arrays: 10 bytes per keypoint) live in a fixed array in
 * device memory of VO_KLT_FB_NOALLOC_MAX_KEYPOINTS entries (2.5 MiB per GPU, held until the process
 * ends) that vo_ctx_create locates once per device.  The call allocates and frees nothing.
 * batch * n beyond the maximum -> VO_EINVAL.  All calls on one GPU share the array: calls on one
 * stream are ordered; a call from another stream (another context) while the previous call's
 * kernels are still running is refused with VO_ESTATE -- wait for that stream first.  Host threads
 * that call at the same moment must serialise themselves.
 */
#define VO_KLT_FB_NOALLOC_MAX_KEYPOINTS 262144
int vo_klt_track_fb_noalloc(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts, int n,
                            const vo_klt_params *params, float lambda, float *next_pts, uint8_t *mask);

/* ---- RANSAC hypothesis scoring --------------------------------------------
 * vo_score_fundamental replaces the per-candidate error/inlier loop inside
 * cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC, thr, conf, maxIters)
 * (vo_pipeline/bootstrap.py:246-252): err = (float)max(d1^2/|l1|^2, d2^2/|l2|^2)
 * in float64, inlier iff err <= (float)(thr*thr).
 * m1, m2: (n,2) float32; F: (H,9) float64 row-major; counts: (H) int32;
 * masks: (H,n) uint8 or NULL.
 */
int vo_score_fundamental(vo_ctx *ctx, const float *m1, const float *m2, int n, const double *F, int H,
                         double thr, int32_t *counts, uint8_t *masks, int mem);
/* vo_score_pnp replaces the per-candidate reprojection/inlier loop inside
 * cv.solvePnPRansac(pointcloud, img_key_points, K, None, flags=P3P, ...)
 * (vo_pipeline/poseEstimation.py:61-69): projection in float64 without
 * distortion, err = (float)|x - proj|^2, inlier iff err <= (float)(thr*thr).
 * X: (n,3) float32; x: (n,2) float32; K: (3,3) float64 row-major, ALWAYS a host
 * pointer (passed to the kernel by value);
 * Rt: (H,12) float64 row-major [R|t] world->camera.
 */
int vo_score_pnp(vo_ctx *ctx, const float *X, const float *x, int n, const double *K, const double *Rt, int H,
                 double thr, int32_t *counts, uint8_t *masks, int mem);

/* ---- bundle-adjustment residual and normal equations ------------------------
 * vo_ba_residual replaces BundleAdjustment._fun (with _reproject/_rotate),
 * vo_pipeline/bundleAdjustment.py:14-45: params = [F x (rvec, t) | P x (X,Y,Z)]
 * float64, camera_indices / point_indices (O) int64, observations (O,2) float64,
 * K (3,3) float64 row-major (ALWAYS a host pointer).  r: (2O) interleaved
 * residuals; *cost = 0.5 * sum r^2 (the quantity scipy's TRF minimises for the
 * call at bundleAdjustment.py:96-97).  cost lives in the same memory space as r.
 * An observation whose indices are out of range yields NaN (VO_MEM_DEVICE) or
 * VO_EINVAL (VO_MEM_HOST); it never touches memory out of bounds.
 */
int vo_ba_residual(vo_ctx *ctx, const double *params, int F, int P, const int64_t *camera_indices,
                   const int64_t *point_indices, const double *obs, int O, const double *K,
                   double *r, double *cost, int mem);
/* vo_ba_normal_eq additionally builds the normal equations from analytic
 * Jacobian blocks (the reference lets scipy difference _fun through the sparsity
 * of bundleAdjustment.py:47-67 instead): Hcc (F,6,6), Hpp (P,3,3), Hcp (O,6,3)
 * per observation, g (6F+3P) = J^T r.  Observations i < obs_start contribute no
 * Jacobian rows: obs_start = number of observations of the first int(F/4)+1
 * cameras, exactly the rows bundleAdjustment.py:54-65 leaves empty.
 */
int vo_ba_normal_eq(vo_ctx *ctx, const double *params, int F, int P, const int64_t *camera_indices,
                    const int64_t *point_indices, const double *obs, int O, const double *K, int obs_start,
                    double *r, double *cost, double *Hcc, double *Hpp, double *Hcp, double *g, int mem);

/* ---- Harris / Shi-Tomasi response, non-maximum suppression, top-K ------------
 * vo_harris_response replaces cv.cornerHarris(float32(img), 2, 3, 0.1) in
 * FeatureExtractor.harris (vo_pipeline/featureExtraction.py:37-40).  img: batch
 * uint8 images; R: (batch, height, width) float32, tight.  mode 0 = Harris
 * (a*c - b*b - k*(a+c)^2), mode 1 = min eigenvalue (Shi-Tomasi), the quantity the
 * tracker thresholds (minEigThreshold, vo_pipeline/poseEstimation.py:129).
 */
int vo_harris_response(vo_ctx *ctx, const uint8_t *img, int batch, int height, int width, int64_t row_stride,
                       int64_t img_stride, int block_size, int ksize, float k, int mode, float *R, int mem);
/* vo_nms_topk stands in for the dilate / threshold(0.01*max) / connected-component
 * centroid steps of FeatureExtractor.harris (vo_pipeline/featureExtraction.py:41-50):
 * it keeps local maxima of R above rel_thresh*max(R) within a (2*radius+1)^2 window
 * (ties: earlier raster index) and returns the K strongest per image, strongest
 * first.  xy: (batch, K, 2) float32 (x, y), zero-filled past count[b]; scores:
 * (batch, K) float32 or NULL; count: (batch) int32.  K <= 4096.
 */
int vo_nms_topk(vo_ctx *ctx, const float *R, int batch, int height, int width, float rel_thresh, int radius,
                int K, float *xy, float *scores, int32_t *count, int mem);

/* vo_detect_corners = vo_harris_response followed by vo_nms_topk on the same images,
 * with identical results, in one call: the response map stays in a library-owned
 * device buffer (no caller allocation, no host round trip) and its per-image maximum
 * is reduced inside the response pass.  It stands in for the whole of
 * FeatureExtractor.harris (vo_pipeline/featureExtraction.py:37-51) and supplies the
 * Shi-Tomasi corners the bootstrap tracks (mode 1).  Arguments as in the two calls above.
 */
int vo_detect_corners(vo_ctx *ctx, const uint8_t *img, int batch, int height, int width, int64_t row_stride,
                      int64_t img_stride, float k, int mode, float rel_thresh, int radius, int K, float *xy,
                      float *scores, int32_t *count, int mem);


/* ---- minimal solvers (model generators of the two RANSAC calls) ---------------
 * vo_p3p_solve: per sample of four 2D-3D pairs, the P3P pose of pairs 0..2 that
 * best reprojects pair 3 -- the candidate generator inside
 * cv.solvePnPRansac(flags=SOLVEPNP_P3P) (vo_pipeline/poseEstimation.py:61-69).
 * X: (n,3) float32; x: (n,2) float32; K: (3,3) float64 row-major (host pointer);
 * subsets: (S,4) int32; Rt: (S,12) float64 [R|t] (zeros when invalid);
 * valid: (S) uint8.
 */
int vo_p3p_solve(vo_ctx *ctx, const float *X, const float *x, int n, const double *K, const int32_t *subsets,
                 int S, double *Rt, uint8_t *valid, int mem);
/* vo_seven_point_solve: per sample of seven correspondences, the 1..3 fundamental
 * matrices of the 7-point algorithm -- the candidate generator inside
 * cv2.findFundamentalMat(FM_RANSAC) (vo_pipeline/bootstrap.py:246-252).
 * m1, m2: (n,2) float32; subsets: (S,7) int32; F: (S,3,9) float64 (unused slots
 * zero); nsol: (S) int32.
 */
int vo_seven_point_solve(vo_ctx *ctx, const float *m1, const float *m2, int n, const int32_t *subsets, int S,
                         double *F, int32_t *nsol, int mem);

/* ---- two-view geometry ---------------------------------------------------------
 * vo_triangulate replaces BootstrapInitializer._linear_triangulation
 * (vo_pipeline/bootstrap.py:141-156): per pair the null vector of
 * [ [p1]x M1 ; [p2]x M2 ], stored as float32 and divided by its last component.
 * pts1, pts2: (n,3) float32 homogeneous pixels; M1, M2: (3,4) float64 row-major,
 * host pointers; m*_is_f32 != 0 forms [p]x M in float32 (the reference's M1 is
 * float32); out: (n,4) float32.
 */
int vo_triangulate(vo_ctx *ctx, const float *pts1, const float *pts2, int n, const double *M1, const double *M2,
                   int m1_is_f32, int m2_is_f32, float *out, int mem);

/* vo_cheirality_vote replaces the four-way motion disambiguation of
 * BootstrapInitializer._transform_matrix (vo_pipeline/bootstrap.py:108-124): for each of the
 * H candidate motions [R|t] (the reference tries (R1,-u), (R1,u), (R2,-u), (R2,u)) every pair is
 * triangulated as in vo_triangulate and the points in front of both cameras are counted, in one
 * launch.  M1 (3,4) and M2 (H,3,4) = K @ [R|t] are the projection matrices, T_hat (H,3,4) the
 * motions [R|t] themselves; all three float64 row-major host pointers.  clouds (H,n,4) float32:
 * the triangulated points of every candidate (camera-1 frame, last column 1); in_front (H) int32 =
 * #(z > 0 in camera 1) + #(z > 0 in camera 2), the depth in camera 2 being the third row of
 * [R|t] times the float32 point, evaluated in float64.  clouds and in_front follow `mem`
 * (VO_MEM_HOST: one synchronisation for the whole call).  H <= 8.
 */
int vo_cheirality_vote(vo_ctx *ctx, const float *pts1, const float *pts2, int n, const double *M1, const double *M2,
                       const double *T_hat, int H, int m1_is_f32, int m2_is_f32, float *clouds, int32_t *in_front,
                       int mem);

/* ---- bundle-adjustment solve ------------------------------------------------------
 * Device-resident Levenberg-Marquardt state for one window; replaces the
 * least_squares(..., method='trf') call of BundleAdjustment.bundle_adjustment
 * (vo_pipeline/bundleAdjustment.py:96-97) on the problem defined there: parameters
 * [F x (rvec, t) | P x XYZ], observations before obs_start without Jacobian rows
 * (bundleAdjustment.py:54-65), cost 0.5*|r|^2.
 *   vo_ba_create     uploads indices/observations once (ci, pi, obs follow `mem`; K is a host pointer)
 *   vo_ba_linearize  sets the current parameters and builds residual + normal equations; cost -> host
 *   vo_ba_try_step   candidate = current + LM step for damping lambda (Schur complement on the
 *                    landmarks, Cholesky of the 6F x 6F camera system, back-substitution), linearised
 *                    as well; cost_new -> host, *solved = 0 when the reduced system was not positive definite
 *   vo_ba_accept     the candidate becomes the current point (buffer swap, nothing recomputed)
 *   vo_ba_params     copies the current (candidate = 0) or candidate (1) parameter vector out
 * vo_ba_linearize and vo_ba_try_step synchronise the stream (they return costs to the host).
 */
typedef struct vo_ba vo_ba;
int vo_ba_create(vo_ctx *ctx, int F, int P, int O, const int64_t *ci, const int64_t *pi, const double *obs,
                 const double *K, int obs_start, int mem, vo_ba **out);
/* vo_ba_window_assemble (host-side, no GPU work): the observation arrays of a bundle-adjustment window from the
 * per-frame keypoint / trajectory lists, i.e. the loop of ContinuousVO._bundle_adjustment that fills camera_indices,
 * point_indices and keypoints and numbers the window's landmarks in order of first appearance
 * (vo_pipeline/continuousVO.py:286-306), in one pass.  Frame f of the F window frames has counts[f] observations:
 * keypoints kps[f] (counts[f] x 2 float32) of trajectories ids[f] (counts[f] int32, non-negative).  Fills ci
 * (frame of every observation, ascending), pi (landmark number of its trajectory), obs (the keypoints, concatenated)
 * -- O = sum(counts) entries each -- and traj_of_point (the trajectory of landmark 0, 1, ...; room for O entries),
 * *n_points = landmarks in the window.  Returns 1, with nothing written, when the ids are too sparse for a table (their
 * span exceeds 8 O + 1024: the caller numbers them by sorting instead); VO_EINVAL for a negative id. */
int vo_ba_window_assemble(int F, const int32_t *counts, const float *const *kps, const int32_t *const *ids, int64_t *ci,
                          int64_t *pi, float *obs, int32_t *traj_of_point, int32_t *n_points);
/* vo_ba_camera_params (host-side, no GPU work): the F window poses M (F x 3 x 4 float64, rows of [R|t]) as the
 * solver's camera parameters (F x 6: rotation vector, translation), i.e. the per-frame cv.Rodrigues loop of
 * BundleAdjustment.bundle_adjustment (vo_pipeline/bundleAdjustment.py:86-91).  As cv.Rodrigues does, the rotation
 * block is first replaced by the nearest rotation (the poses may have been through float32); the rotation vector has
 * length in [0, pi].  VO_EINVAL for a block that is singular, a reflection or not finite. */
int vo_ba_camera_params(int F, const double *M, double *params);
/* vo_ba_create_filtered: vo_ba_create preceded by the outlier pre-filter of BundleAdjustment._preprocess_data
 * (vo_pipeline/bundleAdjustment.py:111-134) on the device: obs is (O,2) float64, or float32 when obs_f32 is set (the
 * pipeline's keypoints, widened on the device exactly as the host conversion would); the residual is evaluated at x0
 * (6F + 3P float64, follows `mem`), observations whose squared reprojection error exceeds max_sq_error are dropped (BA_DISTANCE_TH,
 * params.py:44; the reference's test is "difference > threshold"), the kept ones are compacted in order into the
 * solver, and the observations of cameras below n_frozen_cams (int(F/4)+1, bundleAdjustment.py:54-65; counted, the
 * observations being sorted by camera) get no Jacobian rows.  One upload of the observations, one synchronisation.
 * n_kept / obs_start (host): observations kept, and how many of them lead without Jacobian rows.  The solver equals
 * the one vo_ba_create builds from the kept observations. */
int vo_ba_create_filtered(vo_ctx *ctx, int F, int P, int O, const int64_t *ci, const int64_t *pi, const void *obs,
                          int obs_f32, const double *K, const double *x0, double max_sq_error, int n_frozen_cams, int mem,
                          vo_ba **out, int32_t *n_kept, int32_t *obs_start);
void vo_ba_destroy(vo_ba *ba);
/* vo_ba_solve: the whole least_squares(...) call of BundleAdjustment.bundle_adjustment
 * (vo_pipeline/bundleAdjustment.py:96-97) on a solver object, with the loop's decisions on the device: linearise at x0,
 * then Levenberg-Marquardt trial steps (as vo_ba_try_step) queued several at a time -- after each, one device thread
 * compares the two costs, accepts (damping / 3) or rejects (damping x 10) and applies the stop rules: an accepted step that
 * gained less than ftol of the cost (stop_reason 1), a rejected step that moved the cost by less than ftol (2; only with
 * stop_on_flat_reject), ten rejected steps in a row (3), max_iters accepted steps (4).  The host reads the bookkeeping
 * record back once per round of queued steps instead of one cost per step.  x0 and x_out (6F+3P float64) follow `mem`;
 * stats (host, may be NULL).  The decisions are those of a host loop over vo_ba_linearize / vo_ba_try_step / vo_ba_accept
 * with the same rules.  Synchronises the stream. */
typedef struct vo_ba_lm_stats {
    double cost_start, cost_final, lambda;
    int32_t trial_steps, accepted, iterations, stop_reason;
} vo_ba_lm_stats;
int vo_ba_solve(vo_ctx *ctx, vo_ba *ba, const double *x0, int max_iters, double ftol, double lambda0,
                int stop_on_flat_reject, double *x_out, int mem, vo_ba_lm_stats *stats);
/* The two halves of vo_ba_solve: _begin uploads x0, linearises and queues the first round of trial steps without
 * waiting for anything; _finish reads the loop's record (queuing further rounds while the loop has not ended) and
 * returns the result.  Between the two the caller is free to use other contexts -- ContinuousVO builds the next
 * two-view bootstrap (vo_pipeline/continuousVO.py:223-236: the adjustment, then the bootstrap) on a second context
 * while the window's trial steps run.  first_round: trial steps queued by _begin (0: five, what vo_ba_solve queues; a
 * caller that overlaps other work asks for more, a surplus step costing only its empty launches).  One solve at a time
 * per solver: a second _begin before _finish, and vo_ba_linearize / _try_step / _accept / _params between the two
 * (the queued loop owns the solver's two slots) -> VO_ESTATE. */
int vo_ba_solve_begin(vo_ctx *ctx, vo_ba *ba, const double *x0, int max_iters, double ftol, double lambda0,
                      int stop_on_flat_reject, int first_round, int mem);
int vo_ba_solve_finish(vo_ctx *ctx, vo_ba *ba, double *x_out, int mem, vo_ba_lm_stats *stats);
int vo_ba_linearize(vo_ctx *ctx, vo_ba *ba, const double *x, int mem, double *cost);
int vo_ba_try_step(vo_ctx *ctx, vo_ba *ba, double lambda, double *cost_new, int *solved);
int vo_ba_accept(vo_ctx *ctx, vo_ba *ba);
int vo_ba_params(vo_ctx *ctx, vo_ba *ba, int candidate, double *x, int mem);

/* ---- RANSAC bookkeeping and refit ----------------------------------------------------
 * vo_ransac_draw_subsets (host-side, no GPU work): the sample drawing of OpenCV's
 * RANSAC loop behind cv2.findFundamentalMat (vo_pipeline/bootstrap.py:246-252) and
 * cv.solvePnPRansac (vo_pipeline/poseEstimation.py:61-69): cv::RNG multiply-with-carry
 * generator (*rng_state in/out, OpenCV starts from 2^64-1), distinct indices per subset;
 * with m1/m2 ((count,2) float32, host) subsets whose last point is collinear with an earlier
 * pair in either set are redrawn (up to max_attempts).  subsets: (n_subsets, model_points) int32.
 *
 * vo_refine_pose: Gauss-Newton refit of one pose on the correspondences selected by
 * mask (NULL = all), the step cv.solvePnPRansac performs on the inliers after its loop.
 * X (n,3) float32, x (n,2) float32, mask (n) uint8 follow `mem`; K (3,3), pose_in and
 * pose_out (rotation vector, translation: 6 float64) and cost_out are host pointers.
 * Synchronises the stream.
 */
int vo_ransac_draw_subsets(uint64_t *rng_state, int count, int model_points, int n_subsets, const float *m1,
                           const float *m2, int max_attempts, int32_t *subsets);
int vo_refine_pose(vo_ctx *ctx, const float *X, const float *x, int n, const uint8_t *mask, const double *K,
                   const double *pose_in, int max_iters, double *pose_out, double *cost_out, int mem);

/* ---- whole RANSAC calls ---------------------------------------------------------------
 * vo_solve_pnp_ransac replaces cv.solvePnPRansac(pointcloud, img_key_points, K, None,
 * flags=SOLVEPNP_P3P, reprojectionError=thr, confidence=conf, iterationsCount=max_iters)
 * at vo_pipeline/poseEstimation.py:61-69 in one call: max_iters four-point samples drawn as
 * OpenCV draws them, every P3P candidate solved and scored on the GPU (candidates never
 * leave the device), OpenCV's sequential loop -- best-so-far update, adaptive iteration
 * count -- replayed on the host over the inlier counts, and a Gauss-Newton refit of the
 * winner on its inliers (refine_iters; OpenCV refits with EPnP there).  X (n,3) float32,
 * x (n,2) float32 and inlier_mask (n) uint8 follow `mem`; K (3,3) float64, rng_state
 * (in/out cv::RNG state, NULL = OpenCV's initial 2^64-1), Rt_out (3,4) float64 [R|t]
 * world->camera and n_inliers are host pointers.  n_inliers = 0: no pose found (the
 * condition under which cv2 returns success = False).  Synchronises the stream.
 *
 * vo_find_fundamental_ransac replaces cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC, thr,
 * conf, max_iters) at vo_pipeline/bootstrap.py:246-252 the same way (7-point candidates, up
 * to three per sample; no refit, as in OpenCV).  m1, m2 (n,2) float32, F_out (3,3) float64,
 * inlier_mask (n) uint8 and n_inliers are host pointers (the sampler's collinearity test
 * reads the points on the host).
 */
int vo_solve_pnp_ransac(vo_ctx *ctx, const float *X, const float *x, int n, const double *K, double thr,
                        double conf, int max_iters, int refine_iters, uint64_t *rng_state, double *Rt_out,
                        uint8_t *inlier_mask, int32_t *n_inliers, int mem);
int vo_find_fundamental_ransac(vo_ctx *ctx, const float *m1, const float *m2, int n, double thr, double conf,
                               int max_iters, uint64_t *rng_state, double *F_out, uint8_t *inlier_mask,
                               int32_t *n_inliers);

/* ---- one tracked frame: tracker, status filter and PnP-RANSAC in one call ---------------------------
 * vo_track_pnp replaces the body of ContinuousVO._process_frame up to the pose
 * (vo_pipeline/continuousVO.py:160-180): cv.calcOpticalFlowPyrLK from `prev` to `next`
 * (vo_pipeline/poseEstimation.py:116-133), the status == 1 filter of keypoints and landmarks
 * (continuousVO.py:168-172) and cv.solvePnPRansac with its refit on the survivors
 * (poseEstimation.py:51-77) -- with one synchronisation instead of the three round trips of the separate
 * calls.  The outcome is that of vo_klt_track (init_pts = NULL) followed by vo_solve_pnp_ransac on the rows
 * with status 1, in their order.  prev / next: pyramids of one image each; prev built, next built or -- when
 * next_image is given (one grayscale uint8 image with row pitch next_row_stride, following `mem`) -- built from
 * it here first, as vo_pyr_build would, with the upload of a host image sharing the page-locked copy of the
 * points instead of waiting for the device on its own.  prev_pts (n,2) float32 and
 * landmarks (n,3) float32, row i of one belonging to row i of the other; next_pts (n,2) float32, status (n)
 * uint8 as vo_klt_track returns them; inlier_mask (n) uint8 over the INPUT rows (0 for rows that were not
 * tracked); all five follow `mem`.  *n_tracked (host) = rows with status 1, *n_inliers (host) = 0 when no pose
 * was found (fewer than four survivors, no valid sample), and Rt_out (12 doubles, host, row-major [R|t]) is
 * then left untouched.  K, thr, conf, max_iters, refine_iters, rng_state as in vo_solve_pnp_ransac.  Always
 * synchronises the stream (the pose returns to the host).
 */
int vo_track_pnp(vo_ctx *ctx, const vo_pyr *prev, vo_pyr *next, const uint8_t *next_image, int64_t next_row_stride,
                 const float *prev_pts, const float *landmarks, int n, const vo_klt_params *params, const double *K, double thr, double conf, int max_iters,
                 int refine_iters, uint64_t *rng_state, float *next_pts, uint8_t *status, double *Rt_out,
                 uint8_t *inlier_mask, int32_t *n_tracked, int32_t *n_inliers, int mem);
/* How the context's last vo_track_pnp call ended (no counterpart in the reference; for tests and profiling):
 * 0 no pose was sought (fewer than four survivors), 1 within the first 256 samples, on the single
 * synchronisation, 2 the loop asked for more samples and vo_solve_pnp_ransac finished on the device-resident
 * survivors, 3 the same because the device-side sampler or replay could not vouch for its result. */
int vo_track_pnp_path(const vo_ctx *ctx);

/* ---- the tracked frame of many independent streams in one launch chain ---------------------------------
 * vo_track_pnp_batch is vo_track_pnp (above; vo_pipeline/continuousVO.py:160-180 with
 * vo_pipeline/poseEstimation.py:51-77 and 116-133) for every image pair of two batched pyramids at once:
 * stream b tracks image b of `prev` into image b of `next` (both built), filters its rows by status and runs
 * cv.solvePnPRansac's loop and refit on its survivors.  The reference runs one camera stream per process; a server
 * that runs many of them batches their frames here, with one launch chain and ONE synchronisation for all streams.
 * The outcome per stream equals vo_track_pnp on that stream alone (given the same tracker kernel, which the library
 * picks by the number of keypoints in the call): tracked points, status, pose, inlier mask, counts, generator state.
 * prev_pts (batch,n,2) float32, landmarks (batch,n,3) float32, next_pts (batch,n,2) float32, status (batch,n)
 * uint8, inlier_mask (batch,n) uint8 follow `mem`; a stream with fewer than n keypoints pads with points outside
 * the image (they lose their status and drop out).  K (3,3) float64 is shared by the streams (host pointer).
 * Host arrays of `batch` entries: rng_state (in/out cv::RNG states; NULL = every stream starts from OpenCV's
 * initial 2^64-1, as each cv.solvePnPRansac call does), Rt_out (batch,12) float64 (a stream without pose keeps its
 * rows untouched), n_tracked, n_inliers (0: no pose), path (may be NULL; per stream as vo_track_pnp_path).
 * Streams whose loop outlives the first 256 samples are finished one by one through vo_solve_pnp_ransac after the
 * synchronisation.
 *
 * vo_track_pnp_batch_begin / _finish are the two halves: begin enqueues everything on the context's stream and
 * returns without waiting (rng_state is read there; with VO_MEM_HOST the three output arrays must stay valid
 * until finish); finish waits for the chain, re-checks each stream's loop on the host and writes the host outputs.
 * A caller with several contexts begins a batch on each before finishing the first, so that their chains overlap
 * on the device.  One batch per context may be in flight.
 */
int vo_track_pnp_batch(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts, const float *landmarks,
                       int n, const vo_klt_params *params, const double *K, double thr, double conf, int max_iters,
                       int refine_iters, uint64_t *rng_state, float *next_pts, uint8_t *status, double *Rt_out,
                       uint8_t *inlier_mask, int32_t *n_tracked, int32_t *n_inliers, int32_t *path, int mem);
int vo_track_pnp_batch_begin(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts,
                             const float *landmarks, int n, const vo_klt_params *params, const double *K, double thr,
                             double conf, int max_iters, int refine_iters, const uint64_t *rng_state, float *next_pts,
                             uint8_t *status, uint8_t *inlier_mask, int mem);
int vo_track_pnp_batch_finish(vo_ctx *ctx, uint64_t *rng_state, double *Rt_out, int32_t *n_tracked, int32_t *n_inliers,
                              int32_t *path);

/* Measurement aid (no counterpart in the reference): while enabled, vo_track_pnp_batch_begin brackets its tracker
 * launch with a HIP event pair on the context's stream; vo_profile_tracker_ms waits for the pair of the last such
 * begin and returns the launch's duration.  An event pair costs a few microseconds of idle queue: benchmarks enable
 * it for a fraction of their steps. */
int vo_profile_tracker(vo_ctx *ctx, int enable);
int vo_profile_tracker_ms(vo_ctx *ctx, float *ms);

/* ---- corner mask and sub-pixel refinement of FeatureExtractor.harris -------------------------
 * vo_harris_mask replaces cv.cornerHarris + cv.dilate + cv.threshold + np.uint8
 * (vo_pipeline/featureExtraction.py:40-43): mask = 255 where the 3x3 maximum of the response
 * (pixels outside the image ignored) exceeds (float)(rel_thresh * max response), else 0.
 * img as in vo_harris_response; mask (batch, height, width) uint8, tight.
 * vo_corner_subpix replaces cv.cornerSubPix(img, corners, (win_w, win_h), (-1,-1),
 * (EPS + MAX_ITER, max_iters, eps)) (featureExtraction.py:47-50) on one grayscale uint8 image
 * (the reference converts it to float32, which holds the same values): corners (count,2)
 * float32 (x, y) are refined in place.  Windows that leave the image sample with coordinates
 * clamped to the border.  Half-sizes up to 6.
 */
int vo_harris_mask(vo_ctx *ctx, const uint8_t *img, int batch, int height, int width, int64_t row_stride,
                   int64_t img_stride, float k, int mode, double rel_thresh, uint8_t *mask, int mem);
int vo_corner_subpix(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, float *corners,
                     int count, int win_w, int win_h, int max_iters, double eps, int mem);

/* ---- connected components; the whole of FeatureExtractor.harris ------------------------------------
 * vo_connected_components replaces cv.connectedComponentsWithStats(dst)
 * (vo_pipeline/featureExtraction.py:45) on one uint8 mask (nonzero = foreground): 8-connectivity,
 * label 0 = background.  order 0 numbers the components in raster order of their first pixel,
 * order 1 in raster order of their first 2x2 block (the numbering of the block-based labeller of
 * OpenCV >= 4.5.2).  labels (height, width) int32, tight, may be NULL; centroids (max_labels, 2)
 * float64 (x, y) = coordinate sums / area (an empty background sits at (0, 0)); areas (max_labels)
 * int32, may be NULL; all three follow `mem`.  *n_labels (host) = number of components + 1; rows
 * beyond max_labels are not written (at most ceil(h/2)*ceil(w/2) + 1 labels exist).  Synchronises
 * the stream (the count returns to the host).
 *
 * vo_harris_corners replaces the body of FeatureExtractor.harris (featureExtraction.py:37-51) in one
 * call with one synchronisation: cv.cornerHarris(float32(img), 2, 3, k), 3x3 dilation, threshold at
 * rel_thresh * max, connected components, float32 centroids (background first), cv.cornerSubPix with
 * half-window (win_w, win_h), max_iters, eps.  corners (max_corners, 2) float32 (x, y) follows `mem`;
 * *n_corners (host) = number of labels; rows beyond max_corners are dropped.
 */
int vo_connected_components(vo_ctx *ctx, const uint8_t *mask, int height, int width, int64_t row_stride, int order,
                            int32_t *labels, double *centroids, int32_t *areas, int max_labels, int32_t *n_labels, int mem);
int vo_harris_corners(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, float k, double rel_thresh,
                      int order, int win_w, int win_h, int max_iters, double eps, float *corners, int max_corners,
                      int32_t *n_corners, int mem);

/* ---- SIFT keypoints and descriptors ------------------------------------------------------------------
 * vo_sift_detect_compute replaces cv.SIFT_create(nfeatures=0, nOctaveLayers, contrastThreshold,
 * edgeThreshold, sigma).detectAndCompute(img, None) -- FeatureExtractor.get_kp,
 * vo_pipeline/featureExtraction.py:20-35, called for both bootstrap images at
 * vo_pipeline/bootstrap.py:179-181 (the reference's arguments: 4, 0.03, 10, 1.6).  One grayscale uint8
 * image (follows `mem`).  keypoints: HOST array of max_keypoints records in OpenCV's KeyPoint fields
 * (pt.x, pt.y, size, angle in degrees, response, octave packed as octave | layer << 8 | offset << 16),
 * sorted and de-duplicated as cv2 returns them; descriptors (max_keypoints, 128) float32 with byte values
 * (follows `mem`; may be NULL: detect only).  *n_keypoints (host) = keypoints found; only the first
 * max_keypoints are written (a caller that wants them all calls again with that much room).  The library's own room
 * for extrema and raw keypoints grows with the image that needs it: as cv2's nfeatures = 0, there is no bound but
 * memory.  Synchronises the stream (keypoints are ordered on the host).
 *
 * vo_sift_scale_space_image (tests / debugging): builds the scale space of the image and copies one Gaussian
 * (which = 0, index 0 .. layers + 2) or difference (which = 1, index 0 .. layers + 1) image of `octave` to the
 * tight host buffer host_out (may be NULL to query the size).
 */
typedef struct vo_sift_keypoint {
    float x, y, size, angle, response;
    int32_t octave;
} vo_sift_keypoint;
int vo_sift_detect_compute(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, int n_octave_layers,
                           double contrast_threshold, double edge_threshold, double sigma, vo_sift_keypoint *keypoints,
                           float *descriptors, int max_keypoints, int32_t *n_keypoints, int mem);
int vo_sift_scale_space_image(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, int n_octave_layers,
                              double sigma, int which, int octave, int index, float *host_out, int *out_width, int *out_height,
                              int mem);

/* ---- descriptor matching -----------------------------------------------------------------
 * vo_knn_match2 replaces matcher.knnMatch(des1, des2, k=2) inside
 * FeatureMatcher.match_descriptors (vo_pipeline/featureMatching.py:52-63) and the ratio test
 * of FeatureMatcher.match_filter (featureMatching.py:65-69).  The search is exhaustive: the
 * result of cv.BFMatcher() (featureMatching.py:45), which the FLANN kd-tree matcher of
 * featureMatching.py:46-50 approximates.  des1 (n1,dim) and des2 (n2,dim) float32, row-major,
 * dim = 128 (device pointers 16-byte aligned).  Per query row q: idx[q] = the two train rows with the smallest
 * float32 L2 distance sqrtf(sum_k (a_k-b_k)^2), nearest first, equal distances to the lower
 * row; dist[q] = those distances; keep[q] (may be NULL) = dist[q][0] < ratio * dist[q][1],
 * evaluated in double as the reference's Python expression is.  A missing neighbour (n2 < 2)
 * is index -1 at distance +inf and fails the ratio test.  idx (n1,2) int32, dist (n1,2)
 * float32, keep (n1) uint8 follow `mem`.
 */
int vo_knn_match2(vo_ctx *ctx, const float *des1, int n1, const float *des2, int n2, int dim, double ratio,
                  int32_t *idx, float *dist, uint8_t *keep, int mem);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
