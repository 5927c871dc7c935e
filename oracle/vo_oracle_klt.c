/*
 * CPU ORACLE (test infrastructure, not product code) -- image pyramid + pyramidal
 * Lucas-Kanade tracker.
 *
 * Plain-C restatement of the algorithm the reference reaches through
 *   cv.calcOpticalFlowPyrLK(img0, img1, np.float32(prev_kpts), None,
 *                           winSize=(21,21), maxLevel=8, minEigThreshold=1e-2,
 *                           criteria=(COUNT|EPS, 10, 0.001),
 *                           flags=OPTFLOW_LK_GET_MIN_EIGENVALS)
 * at /root/reference vo_pipeline/poseEstimation.py:116-133 (caller
 * vo_pipeline/continuousVO.py:167).  The arithmetic lives in the third-party
 * dependency opencv-python-headless==4.5.4.60 (environment.yml:93), whose sources
 * and binaries are absent from this build environment, and the reference holds no
 * test vectors for this call.  PARITY UNPINNED: this file restates the published
 * OpenCV 4.5.4 algorithm (modules/video/src/lkpyramid.cpp scalar path,
 * modules/imgproc/src/pyramids.cpp) -- it has not been checked against a cv2 run.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define W_BITS 14

static inline int reflect101(int i, int n)
{
    /* BORDER_REFLECT_101: ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...  (n == 1 -> 0) */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

static inline int cv_floor(float v)
{
    int i = (int)v;
    return i - (i > v);
}

static inline int cv_round(float v)
{
    /* OpenCV cvRound: SSE2 cvtsd2si on the double value -> round half to even. */
    return (int)lrint((double)v);
}

/* Number of the last pyramid level that buildOpticalFlowPyramid keeps:
 * after building level l the next size is ((w+1)/2,(h+1)/2); when that is
 * <= win in either dimension the build stops at l. */
int vo_or_pyr_max_level(int w, int h, int max_level, int win)
{
    int level;
    for (level = 0; level <= max_level; ++level) {
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) return level;
    }
    return max_level;
}

/* pyrDown for 8-bit single channel: separable [1 4 6 4 1], REFLECT_101,
 * dst = (sum + 128) >> 8, dst(x,y) centred on src(2x,2y). */
void vo_or_pyr_down(const uint8_t *src, int w, int h, int sstride,
                    uint8_t *dst, int dstride)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int *rows = (int *)malloc(sizeof(int) * 5 * (size_t)dw);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y - 2 + k, h);
            const uint8_t *s = src + (size_t)sy * sstride;
            int *r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; ++x) {
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w);
                int x2 = 2 * x, x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                r[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        for (int x = 0; x < dw; ++x) {
            int v = rows[x] + rows[4 * dw + x] + 4 * (rows[dw + x] + rows[3 * dw + x]) +
                    6 * rows[2 * dw + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* Scharr derivative as calcSharrDeriv: smooth [3 10 3], diff [-1 0 1], no
 * normalisation, REFLECT_101 inside the image.  dst is h*w*2 int16, (dx,dy)
 * interleaved. */
void vo_or_scharr(const uint8_t *src, int w, int h, int stride, int16_t *dst)
{
    int *t0 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    int *t1 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = src + (size_t)reflect101(y - 1, h) * stride;
        const uint8_t *r1 = src + (size_t)y * stride;
        const uint8_t *r2 = src + (size_t)reflect101(y + 1, h) * stride;
        for (int x = 0; x < w; ++x) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        int xl = reflect101(-1, w), xr = reflect101(w, w);
        t0[0] = t0[xl + 1]; t1[0] = t1[xl + 1];
        t0[w + 1] = t0[xr + 1]; t1[w + 1] = t1[xr + 1];
        int16_t *d = dst + (size_t)y * w * 2;
        for (int x = 0; x < w; ++x) {
            d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0);
    free(t1);
}

/* diagnostic counters of the last vo_or_klt call: Gauss-Newton iterations and
 * (keypoint, level) pairs that reached the iteration loop */
static long g_iters, g_levels;
void vo_or_klt_stats(long *iters, long *levels) { *iters = g_iters; *levels = g_levels; }
/* optional per-keypoint log of Gauss-Newton iterations, log[p * 16 + level]; NULL switches it off */
static int *g_iter_log;
void vo_or_klt_set_iter_log(int *log) { g_iter_log = log; }

typedef struct {
    int w, h;
    uint8_t *img;   /* h*w */
    int16_t *der;   /* h*w*2, only for the previous image */
} or_level;

/* image sample with the REFLECT_101 border that buildOpticalFlowPyramid adds */
static inline int pix(const or_level *L, int x, int y)
{
    return L->img[(size_t)reflect101(y, L->h) * L->w + reflect101(x, L->w)];
}
/* derivative sample; the derivative image is padded with constant 0 */
static inline int der(const or_level *L, int x, int y, int c)
{
    if (x < 0 || y < 0 || x >= L->w || y >= L->h) return 0;
    return L->der[((size_t)y * L->w + x) * 2 + c];
}

static inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

static or_level *build_levels(const uint8_t *img, int w, int h, int stride, int levels, int with_der)
{
    or_level *L = (or_level *)calloc((size_t)levels, sizeof(or_level));
    for (int l = 0; l < levels; ++l) {
        if (l == 0) {
            L[0].w = w; L[0].h = h;
            L[0].img = (uint8_t *)malloc((size_t)w * h);
            for (int y = 0; y < h; ++y) memcpy(L[0].img + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
        } else {
            L[l].w = (L[l - 1].w + 1) / 2; L[l].h = (L[l - 1].h + 1) / 2;
            L[l].img = (uint8_t *)malloc((size_t)L[l].w * L[l].h);
            vo_or_pyr_down(L[l - 1].img, L[l - 1].w, L[l - 1].h, L[l - 1].w, L[l].img, L[l].w);
        }
        if (with_der) {
            L[l].der = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)L[l].w * L[l].h);
            vo_or_scharr(L[l].img, L[l].w, L[l].h, L[l].w, L[l].der);
        }
    }
    return L;
}

static void free_levels(or_level *L, int levels)
{
    for (int l = 0; l < levels; ++l) { free(L[l].img); free(L[l].der); }
    free(L);
}

/* Copy pyramid level `level` of img (w x h) into out (tight, ((w+1)/2..) sized). */
int vo_or_pyr_level(const uint8_t *img, int w, int h, int stride, int level, uint8_t *out)
{
    or_level *L = build_levels(img, w, h, stride, level + 1, 0);
    memcpy(out, L[level].img, (size_t)L[level].w * L[level].h);
    free_levels(L, level + 1);
    return 0;
}

/*
 * Pyramidal LK, translation model.  pts/out_pts are (n,2) float32 (x,y).
 * init_pts (nullable) = OPTFLOW_USE_INITIAL_FLOW start values.
 * status: 1 = tracked.  err = min eigenvalue (OPTFLOW_LK_GET_MIN_EIGENVALS).
 * max_iters / eps are the raw criteria values (clamped and squared here as
 * calcOpticalFlowPyrLK does).  Returns the effective max level.
 */
int vo_or_klt(const uint8_t *img0, const uint8_t *img1, int w, int h, int stride,
              const float *pts, const float *init_pts, int n, int win, int max_level,
              int max_iters, double eps, float min_eig_thr,
              float *out_pts, uint8_t *status, float *err)
{
    if (max_iters < 0) max_iters = 0;
    if (max_iters > 100) max_iters = 100;
    if (eps < 0) eps = 0;
    if (eps > 10) eps = 10;
    const double eps2 = eps * eps;
    const int maxL = vo_or_pyr_max_level(w, h, max_level, win);
    const int levels = maxL + 1;
    or_level *P = build_levels(img0, w, h, stride, levels, 1);
    or_level *Q = build_levels(img1, w, h, stride, levels, 0);
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    int16_t *Iw = (int16_t *)malloc(sizeof(int16_t) * 3 * (size_t)win * win);
    int16_t *dIw = Iw + (size_t)win * win;

    for (int i = 0; i < n; ++i) { status[i] = 1; if (err) err[i] = 0.f; }
    g_iters = 0; g_levels = 0;

    for (int level = maxL; level >= 0; --level) {
        const or_level *I = &P[level], *J = &Q[level];
        for (int p = 0; p < n; ++p) {
            float sc = (float)(1. / (1 << level));
            float px = pts[2 * p] * sc, py = pts[2 * p + 1] * sc;
            float nx, ny;
            if (level == maxL) {
                if (init_pts) { nx = init_pts[2 * p] * sc; ny = init_pts[2 * p + 1] * sc; }
                else { nx = px; ny = py; }
            } else {
                nx = out_pts[2 * p] * 2.f; ny = out_pts[2 * p + 1] * 2.f;
            }
            out_pts[2 * p] = nx; out_pts[2 * p + 1] = ny;

            px -= half; py -= half;
            int ipx = cv_floor(px), ipy = cv_floor(py);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
                if (level == 0) { status[p] = 0; if (err) err[p] = 0.f; }
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < win; ++y)
                for (int x = 0; x < win; ++x) {
                    int X = ipx + x, Y = ipy + y;
                    int ival = descale(pix(I, X, Y) * iw00 + pix(I, X + 1, Y) * iw01 +
                                       pix(I, X, Y + 1) * iw10 + pix(I, X + 1, Y + 1) * iw11, W_BITS - 5);
                    int ixval = descale(der(I, X, Y, 0) * iw00 + der(I, X + 1, Y, 0) * iw01 +
                                        der(I, X, Y + 1, 0) * iw10 + der(I, X + 1, Y + 1, 0) * iw11, W_BITS);
                    int iyval = descale(der(I, X, Y, 1) * iw00 + der(I, X + 1, Y, 1) * iw01 +
                                        der(I, X, Y + 1, 1) * iw10 + der(I, X + 1, Y + 1, 1) * iw11, W_BITS);
                    Iw[y * win + x] = (int16_t)ival;
                    dIw[(y * win + x) * 2] = (int16_t)ixval;
                    dIw[(y * win + x) * 2 + 1] = (int16_t)iyval;
                    iA11 += (float)(ixval * ixval);
                    iA12 += (float)(ixval * iyval);
                    iA22 += (float)(iyval * iyval);
                }
            float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                           (float)(2 * win * win);
            if (err) err[p] = minEig;
            if (minEig < min_eig_thr || D < 1.1920928955078125e-07f) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0, pdy = 0;
            ++g_levels;
            for (int j = 0; j < max_iters; ++j) {
                ++g_iters;
                if (g_iter_log) ++g_iter_log[p * 16 + level];
                int inx = cv_floor(nx), iny = cv_floor(ny);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float ib1 = 0, ib2 = 0;
                for (int y = 0; y < win; ++y)
                    for (int x = 0; x < win; ++x) {
                        int X = inx + x, Y = iny + y;
                        int diff = descale(pix(J, X, Y) * iw00 + pix(J, X + 1, Y) * iw01 +
                                           pix(J, X, Y + 1) * iw10 + pix(J, X + 1, Y + 1) * iw11,
                                           W_BITS - 5) - Iw[y * win + x];
                        ib1 += (float)(diff * dIw[(y * win + x) * 2]);
                        ib2 += (float)(diff * dIw[(y * win + x) * 2 + 1]);
                    }
                float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
                float dx = (float)((A12 * b2 - A22 * b1) * D);
                float dy = (float)((A12 * b1 - A11 * b2) * D);
                nx += dx; ny += dy;
                out_pts[2 * p] = nx + half; out_pts[2 * p + 1] = ny + half;
                if ((double)dx * dx + (double)dy * dy <= eps2) break;
                if (j > 0 && fabs(dx + pdx) < 0.01 && fabs(dy + pdy) < 0.01) {
                    out_pts[2 * p] -= dx * 0.5f; out_pts[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
        }
    }
    free(Iw);
    free_levels(P, levels);
    free_levels(Q, levels);
    return maxL;
}
