/*
 * CPU restatement (test infrastructure, never on the product path) of
 *   cv.SIFT_create(nfeatures=0, nOctaveLayers=4, contrastThreshold=0.03, edgeThreshold=10, sigma=1.6)
 *       .detectAndCompute(img, None)
 * as the reference calls it (vo_pipeline/featureExtraction.py:20-35; callers vo_pipeline/bootstrap.py:179-181).
 *
 * The arithmetic lives in opencv-python-headless 4.5.4.60 (reference environment.yml:93), which is not
 * installed here and whose source is not under the reference tree: this file restates the published
 * algorithm of OpenCV 4.5.4's modules/features2d/src/sift.dispatch.cpp and sift.simd.hpp in their scalar
 * form.  PARITY UNPINNED against cv2 (the reference ships no SIFT vectors).  Known departures from a cv2
 * build: exp() of the Gaussian weights is evaluated by the polynomial sift_exp below where OpenCV calls its
 * own table-driven hal::exp32f (and 2^x of the keypoint size by the same polynomial instead of powf), SIMD builds of OpenCV add in another order, and the Gaussian kernels come
 * from libm's exp where OpenCV 4.5.4 uses a soft-float routine (all last-bit effects).
 *
 * Stages: float image doubled by bilinear resize and blurred to sigma (createInitialImage); Gaussian pyramid
 * of nOctaveLayers + 3 images per octave; differences; 26-neighbour extrema refined by up to five Newton
 * steps with the contrast and edge tests (adjustLocalExtrema); 36-bin orientation histograms, one keypoint
 * per peak >= 0.8 max; keypoints sorted and de-duplicated (KeyPointsFilter::removeDuplicatedSorted), scaled
 * back to the input image; 4x4x8 descriptors with trilinear interpolation, clamped at 0.2, scaled by 512 and
 * saturated to bytes (stored as float32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_ORI_SIG_FCTR 1.5f
#define SIFT_ORI_RADIUS (3 * SIFT_ORI_SIG_FCTR)
#define SIFT_ORI_PEAK_RATIO 0.8f
#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_DESCR_SCL_FCTR 3.f
#define SIFT_DESCR_MAG_THR 0.2f
#define SIFT_INT_DESCR_FCTR 512.f
#define SIFT_INIT_SIGMA 0.5f
#define SIFT_FLT_EPSILON 1.1920928955078125e-07f

typedef struct { float x, y, size, angle, response; int32_t octave; } sift_kp;

static int sift_round(double v) { return (int)lrint(v); }       /* cvRound: to nearest, ties to even */

/* cos and sin of a float angle t (radians, 0 <= t <= 2 pi) as OpenCV's descriptor rotation takes them (cosf / sinf of
 * the float product angle * (float)(pi / 180)): evaluated in float64 -- argument reduced by multiples of pi/2 given in
 * two parts, fdlibm's kernel polynomials on [-pi/4, pi/4] -- in this fixed operation order and rounded to float.  The
 * float64 result is within one unit in its last place, so the float is the correctly rounded one except when the exact
 * value lies within ~1e-16 (relative) of a float rounding boundary; written out (instead of calling libm) so that a
 * CPU and a GPU evaluate the very same expression. */
void vo_or_sift_sincos(float t, float *c, float *s)
{
    const double x = (double)t;
    const int k = (int)(x * 0.63661977236758138 + 0.5);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double r2 = r * r;
    const double sn = r + r * r2 * (-1.66666666666666324348e-01 + r2 * (8.33333333332248946124e-03 + r2 * (-1.98412698298579493134e-04 +
                      r2 * (2.75573137070700676789e-06 + r2 * (-2.50507602534068634195e-08 + r2 * 1.58969099521155010221e-10)))));
    const double cs = 1.0 - 0.5 * r2 + r2 * r2 * (4.16666666666666019037e-02 + r2 * (-1.38888888888741095749e-03 + r2 * (2.48015872894767294178e-05 +
                      r2 * (-2.75573143513906633035e-07 + r2 * (2.08757232129817482790e-09 + r2 * -1.13596475577881948265e-11)))));
    switch (k & 3) {
    case 0: *c = (float)cs; *s = (float)sn; break;
    case 1: *c = (float)-sn; *s = (float)cs; break;
    case 2: *c = (float)-cs; *s = (float)-sn; break;
    default: *c = (float)sn; *s = (float)-cs; break;
    }
}

static int sift_floor(float v) { int i = (int)v; return i - (i > v); }

/* 2^t: integer part by exponent arithmetic, fraction by a degree-6 polynomial (relative error ~2e-7).  Plain
 * float operations in a fixed order, so any IEEE machine gives the same bits (libm's expf / powf differ in the
 * last bit between implementations). */
static float sift_exp2(float t)
{
    if (t < -126.f) return 0.f;
    float n = floorf(t), f = t - n;
    float p = 1.530276e-4f;
    p = p * f + 1.339887e-3f;
    p = p * f + 9.618437e-3f;
    p = p * f + 5.550357e-2f;
    p = p * f + 2.402265e-1f;
    p = p * f + 6.931472e-1f;
    p = p * f + 1.0f;
    return ldexpf(p, (int)n);
}
static float sift_exp(float x) { return sift_exp2(x * 1.44269504088896341f); }

/* OpenCV's fastAtan2: degrees in [0, 360), polynomial of the smaller/larger ratio (max error ~0.3 deg) */
static float sift_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f,
                p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446049250313e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.220446049250313e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

/* cv::getGaussianKernel(ksize, sigma, CV_32F) with ksize = cvRound(sigma * 8 + 1) | 1 (float images) */
int vo_or_sift_gauss_kernel(double sigma, float *k)
{
    int n = sift_round(sigma * 8 + 1) | 1;
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        k[i] = (float)exp(s2 * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
    return n;
}

/* GaussianBlur on float32, BORDER_REFLECT_101: rows with the taps in order, columns with symmetric pairs */
static void gauss_blur(const float *src, float *dst, float *tmp, int w, int h, double sigma)
{
    float k[64];
    const int n = vo_or_sift_gauss_kernel(sigma, k), r = n / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = src[(size_t)y * w + reflect101(x - r, w)] * k[0];
            for (int j = 1; j < n; ++j) s += src[(size_t)y * w + reflect101(x - r + j, w)] * k[j];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[r] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= r; ++j)
                s += k[r + j] * (tmp[(size_t)reflect101(y + j, h) * w + x] + tmp[(size_t)reflect101(y - j, h) * w + x]);
            dst[(size_t)y * w + x] = s;
        }
}

/* cv::resize(float, 2x, INTER_LINEAR): pixel centres aligned, clamped at the border */
static void upsample2(const uint8_t *img, int w, int h, int64_t stride, float *dst)
{
    const int W = 2 * w, H = 2 * h;
    for (int y = 0; y < H; ++y) {
        float fy = (float)((y + 0.5) * 0.5 - 0.5);
        int sy = sift_floor(fy);
        fy -= sy;
        if (sy < 0) { sy = 0; fy = 0; }
        if (sy >= h - 1) { sy = h - 1; fy = 0; }
        const int sy1 = sy + 1 < h ? sy + 1 : h - 1;
        const float b0 = 1.f - fy, b1 = fy;
        for (int x = 0; x < W; ++x) {
            float fx = (float)((x + 0.5) * 0.5 - 0.5);
            int sx = sift_floor(fx);
            fx -= sx;
            if (sx < 0) { sx = 0; fx = 0; }
            if (sx >= w - 1) { sx = w - 1; fx = 0; }
            const int sx1 = sx + 1 < w ? sx + 1 : w - 1;
            const float a0 = 1.f - fx, a1 = fx;
            const float r0 = (float)img[sy * stride + sx] * a0 + (float)img[sy * stride + sx1] * a1;
            const float r1 = (float)img[sy1 * stride + sx] * a0 + (float)img[sy1 * stride + sx1] * a1;
            dst[(size_t)y * W + x] = r0 * b0 + r1 * b1;
        }
    }
}

typedef struct {
    int n_oct, layers, first_octave;
    int w[32], h[32];
    float **gauss; /* n_oct * (layers + 3) */
    float **dog;   /* n_oct * (layers + 2) */
} sift_pyr;

static const float *G(const sift_pyr *P, int o, int i) { return P->gauss[o * (P->layers + 3) + i]; }
static const float *D(const sift_pyr *P, int o, int i) { return P->dog[o * (P->layers + 2) + i]; }

static int adjust_extremum(const sift_pyr *P, int octv, int *layer, int *r, int *c, double contrast_thr, double edge_thr,
                           double sigma, sift_kp *kp)
{
    const int w = P->w[octv], h = P->h[octv], nl = P->layers;
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < SIFT_MAX_INTERP_STEPS; ++i) {
        const float *img = D(P, octv, *layer), *prev = D(P, octv, *layer - 1), *next = D(P, octv, *layer + 1);
        const size_t at = (size_t)*r * w + *c;
        const float dD0 = (img[at + 1] - img[at - 1]) * deriv_scale, dD1 = (img[at + w] - img[at - w]) * deriv_scale,
                    dD2 = (next[at] - prev[at]) * deriv_scale;
        const float v2 = img[at] * 2;
        const float dxx = (img[at + 1] + img[at - 1] - v2) * second, dyy = (img[at + w] + img[at - w] - v2) * second,
                    dss = (next[at] + prev[at] - v2) * second;
        const float dxy = (img[at + w + 1] - img[at + w - 1] - img[at - w + 1] + img[at - w - 1]) * cross;
        const float dxs = (next[at + 1] - next[at - 1] - prev[at + 1] + prev[at - 1]) * cross;
        const float dys = (next[at + w] - next[at - w] - prev[at + w] + prev[at - w]) * cross;
        /* X = H^-1 dD by the closed 3x3 form of cv::Matx::solve */
        const float a00 = dxx, a01 = dxy, a02 = dxs, a10 = dxy, a11 = dyy, a12 = dys, a20 = dxs, a21 = dys, a22 = dss;
        float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
        float X0 = 0, X1 = 0, X2 = 0;
        if (d != 0) {
            d = 1 / d;
            X0 = d * (dD0 * (a11 * a22 - a12 * a21) - a01 * (dD1 * a22 - a12 * dD2) + a02 * (dD1 * a21 - a11 * dD2));
            X1 = d * (a00 * (dD1 * a22 - a12 * dD2) - dD0 * (a10 * a22 - a12 * a20) + a02 * (a10 * dD2 - dD1 * a20));
            X2 = d * (a00 * (a11 * dD2 - dD1 * a21) - a01 * (a10 * dD2 - dD1 * a20) + dD0 * (a10 * a21 - a11 * a20));
        }
        xi = -X2; xr = -X1; xc = -X0;
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(2147483647 / 3) || fabsf(xr) > (float)(2147483647 / 3) || fabsf(xc) > (float)(2147483647 / 3)) return 0;
        *c += sift_round(xc); *r += sift_round(xr); *layer += sift_round(xi);
        if (*layer < 1 || *layer > nl || *c < SIFT_IMG_BORDER || *c >= w - SIFT_IMG_BORDER || *r < SIFT_IMG_BORDER ||
            *r >= h - SIFT_IMG_BORDER)
            return 0;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return 0;
    {
        const float *img = D(P, octv, *layer), *prev = D(P, octv, *layer - 1), *next = D(P, octv, *layer + 1);
        const size_t at = (size_t)*r * w + *c;
        const float dD0 = (img[at + 1] - img[at - 1]) * deriv_scale, dD1 = (img[at + w] - img[at - w]) * deriv_scale,
                    dD2 = (next[at] - prev[at]) * deriv_scale;
        const float t = dD0 * xc + dD1 * xr + dD2 * xi;
        const float contr = img[at] * img_scale + t * 0.5f;
        if (fabsf(contr) * nl < (float)contrast_thr) return 0;
        const float v2 = img[at] * 2.f;
        const float dxx = (img[at + 1] + img[at - 1] - v2) * second, dyy = (img[at + w] + img[at - w] - v2) * second;
        const float dxy = (img[at + w + 1] - img[at + w - 1] - img[at - w + 1] + img[at - w - 1]) * cross;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * (float)edge_thr >= ((float)edge_thr + 1) * ((float)edge_thr + 1) * det) return 0;
        kp->x = (*c + xc) * (1 << octv);
        kp->y = (*r + xr) * (1 << octv);
        kp->octave = octv + (*layer << 8) + (sift_round((xi + 0.5) * 255) << 16);
        kp->size = (float)sigma * sift_exp2((*layer + xi) / nl) * (1 << octv) * 2;
        kp->response = fabsf(contr);
    }
    return 1;
}

static float orientation_hist(const float *img, int w, int h, int px, int py, int radius, float sigma, float *hist, int n)
{
    const int len = (radius * 2 + 1) * (radius * 2 + 1);
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    float *tmp = (float *)calloc((size_t)n + 4, sizeof(float)), *th = tmp + 2;
    (void)len;
    for (int i = -radius; i <= radius; ++i) {
        const int y = py + i;
        if (y <= 0 || y >= h - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = px + j;
            if (x <= 0 || x >= w - 1) continue;
            const float dx = img[(size_t)y * w + x + 1] - img[(size_t)y * w + x - 1];
            const float dy = img[(size_t)(y - 1) * w + x] - img[(size_t)(y + 1) * w + x];
            const float wgt = sift_exp((i * i + j * j) * expf_scale);
            const float ori = sift_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
            int bin = sift_round((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            th[bin] += wgt * mag;
        }
    }
    th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
    float maxval = 0;
    for (int i = 0; i < n; ++i) {
        hist[i] = (th[i - 2] + th[i + 2]) * (1.f / 16.f) + (th[i - 1] + th[i + 1]) * (4.f / 16.f) + th[i] * (6.f / 16.f);
        if (i == 0 || hist[i] > maxval) maxval = hist[i];
    }
    free(tmp);
    return maxval;
}

static void descriptor(const float *img, int w, int h, float ptx, float pty, float ori, float scl, float *dst)
{
    const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    const int px = sift_round(ptx), py = sift_round(pty);
    float cos_t, sin_t;
    vo_or_sift_sincos(ori * (float)(3.14159265358979323846 / 180.), &cos_t, &sin_t);
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = sift_round(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt((double)w * w + (double)h * h);
    if (radius > rmax) radius = rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    float hist[(SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_HIST_BINS + 2)];
    memset(hist, 0, sizeof(hist));
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1)) continue;
            const float dx = img[(size_t)r * w + c + 1] - img[(size_t)r * w + c - 1];
            const float dy = img[(size_t)(r - 1) * w + c] - img[(size_t)(r + 1) * w + c];
            const float wgt = sift_exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (sift_atan2_deg(dy, dx) - ori) * bins_per_rad;
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            const int r0 = sift_floor(rbin), c0 = sift_floor(cbin);
            int o0 = sift_floor(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v111 = v_rc11 * obin, v110 = v_rc11 - v111, v101 = v_rc10 * obin, v100 = v_rc10 - v101;
            const float v011 = v_rc01 * obin, v010 = v_rc01 - v011, v001 = v_rc00 * obin, v000 = v_rc00 - v001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx] += v000; hist[idx + 1] += v001;
            hist[idx + (n + 2)] += v010; hist[idx + (n + 3)] += v011;
            hist[idx + (d + 2) * (n + 2)] += v100; hist[idx + (d + 2) * (n + 2) + 1] += v101;
            hist[idx + (d + 3) * (n + 2)] += v110; hist[idx + (d + 3) * (n + 2) + 1] += v111;
        }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; ++k) dst[(i * d + j) * n + k] = hist[idx + k];
        }
    const int len = d * d * n;
    float nrm2 = 0;
    for (int k = 0; k < len; ++k) nrm2 += dst[k] * dst[k];
    const float thr = sqrtf(nrm2) * SIFT_DESCR_MAG_THR;
    nrm2 = 0;
    for (int k = 0; k < len; ++k) {
        const float v = dst[k] < thr ? dst[k] : thr;
        dst[k] = v;
        nrm2 += v * v;
    }
    const float sq = sqrtf(nrm2);
    nrm2 = SIFT_INT_DESCR_FCTR / (sq > SIFT_FLT_EPSILON ? sq : SIFT_FLT_EPSILON);
    for (int k = 0; k < len; ++k) {
        int v = sift_round(dst[k] * nrm2);
        dst[k] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static int kp_less(const void *pa, const void *pb)
{
    const sift_kp *a = (const sift_kp *)pa, *b = (const sift_kp *)pb;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    if (a->angle != b->angle) return a->angle < b->angle ? -1 : 1;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    if (a->octave != b->octave) return a->octave > b->octave ? -1 : 1;
    return 0;
}

int vo_or_sift_num_octaves(int w, int h)
{
    const int m = 2 * (w < h ? w : h);
    return sift_round(log((double)m) / log(2.) - 2) + 1;
}

static sift_pyr *build_pyramid(const uint8_t *img, int w, int h, int64_t stride, int layers, double sigma)
{
    sift_pyr *P = (sift_pyr *)calloc(1, sizeof(sift_pyr));
    P->layers = layers; P->first_octave = -1;
    P->n_oct = vo_or_sift_num_octaves(w, h);
    P->gauss = (float **)calloc((size_t)P->n_oct * (layers + 3), sizeof(float *));
    P->dog = (float **)calloc((size_t)P->n_oct * (layers + 2), sizeof(float *));
    const int W = 2 * w, H = 2 * h;
    float *dbl = (float *)malloc(sizeof(float) * (size_t)W * H), *tmp = (float *)malloc(sizeof(float) * (size_t)W * H);
    upsample2(img, w, h, stride, dbl);
    double sig[16];
    sig[0] = sigma;
    const double k = pow(2., 1. / layers);
    for (int i = 1; i < layers + 3; ++i) {
        const double sp = pow(k, (double)(i - 1)) * sigma, st = sp * k;
        sig[i] = sqrt(st * st - sp * sp);
    }
    double sd = sigma * sigma - (double)(SIFT_INIT_SIGMA * 2) * (SIFT_INIT_SIGMA * 2);
    sd = sqrt(sd > 0.01 ? sd : 0.01);
    for (int o = 0; o < P->n_oct; ++o) {
        P->w[o] = o == 0 ? W : P->w[o - 1] / 2;
        P->h[o] = o == 0 ? H : P->h[o - 1] / 2;
        const int ow = P->w[o], oh = P->h[o];
        for (int i = 0; i < layers + 3; ++i) {
            float *dst = (float *)malloc(sizeof(float) * (size_t)(ow > 0 ? ow : 1) * (oh > 0 ? oh : 1));
            P->gauss[o * (layers + 3) + i] = dst;
            if (ow <= 0 || oh <= 0) continue;
            if (o == 0 && i == 0) gauss_blur(dbl, dst, tmp, ow, oh, sd);
            else if (i == 0) {
                const float *src = G(P, o - 1, layers);
                const int sw = P->w[o - 1];
                for (int y = 0; y < oh; ++y)
                    for (int x = 0; x < ow; ++x) dst[(size_t)y * ow + x] = src[(size_t)(2 * y) * sw + 2 * x];   /* INTER_NEAREST, exact half */
            } else gauss_blur(G(P, o, i - 1), dst, tmp, ow, oh, sig[i]);
        }
        for (int i = 0; i < layers + 2; ++i) {
            float *dst = (float *)malloc(sizeof(float) * (size_t)(ow > 0 ? ow : 1) * (oh > 0 ? oh : 1));
            P->dog[o * (layers + 2) + i] = dst;
            const float *a = G(P, o, i), *b = G(P, o, i + 1);
            for (size_t t = 0; t < (size_t)(ow > 0 ? ow : 0) * (oh > 0 ? oh : 0); ++t) dst[t] = b[t] - a[t];
        }
    }
    free(dbl); free(tmp);
    return P;
}

static void free_pyramid(sift_pyr *P)
{
    for (int i = 0; i < P->n_oct * (P->layers + 3); ++i) free(P->gauss[i]);
    for (int i = 0; i < P->n_oct * (P->layers + 2); ++i) free(P->dog[i]);
    free(P->gauss); free(P->dog); free(P);
}

/* Copy Gaussian (which = 0) or difference (which = 1) image (octave, index) out; returns w * h, sizes in *ow, *oh. */
int vo_or_sift_pyramid_image(const uint8_t *img, int h, int w, int64_t stride, int layers, double sigma, int which, int octave,
                             int index, float *out, int *ow, int *oh)
{
    sift_pyr *P = build_pyramid(img, w, h, stride, layers, sigma);
    int n = -1;
    if (octave >= 0 && octave < P->n_oct && index >= 0 && index < layers + (which ? 2 : 3)) {
        *ow = P->w[octave]; *oh = P->h[octave];
        n = *ow * *oh;
        if (out) memcpy(out, which ? D(P, octave, index) : G(P, octave, index), sizeof(float) * (size_t)n);
    }
    free_pyramid(P);
    return n;
}

/* keypoints: max_kp records {x, y, size, angle, response, octave}; descriptors: max_kp x 128 float32 (may be NULL).
 * Returns the number of keypoints found (only the first max_kp are written). */
int vo_or_sift_detect_compute(const uint8_t *img, int h, int w, int64_t stride, int layers, double contrast_thr, double edge_thr,
                              double sigma, sift_kp *keypoints, float *descriptors, int max_kp)
{
    sift_pyr *P = build_pyramid(img, w, h, stride, layers, sigma);
    const int threshold = (int)floor(0.5 * contrast_thr / layers * 255);
    size_t cap = 4096, cnt = 0;
    sift_kp *kps = (sift_kp *)malloc(sizeof(sift_kp) * cap);
    const int n = SIFT_ORI_HIST_BINS;
    for (int o = 0; o < P->n_oct; ++o) {
        const int ow = P->w[o], oh = P->h[o];
        for (int i = 1; i <= layers; ++i) {
            const float *img1 = D(P, o, i), *prev = D(P, o, i - 1), *next = D(P, o, i + 1);
            for (int r = SIFT_IMG_BORDER; r < oh - SIFT_IMG_BORDER; ++r)
                for (int c = SIFT_IMG_BORDER; c < ow - SIFT_IMG_BORDER; ++c) {
                    const size_t at = (size_t)r * ow + c;
                    const float val = img1[at];
                    if (!(fabsf(val) > threshold)) continue;
                    int is_max = val > 0, is_min = val < 0;
                    for (int dy = -1; dy <= 1 && (is_max || is_min); ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            const size_t q = at + (size_t)((long)dy * ow + dx);
                            const float a = prev[q], b = img1[q], cc = next[q];
                            if (is_max && !(val >= a && val >= b && val >= cc)) is_max = 0;
                            if (is_min && !(val <= a && val <= b && val <= cc)) is_min = 0;
                        }
                    if (!is_max && !is_min) continue;
                    int r1 = r, c1 = c, layer = i;
                    sift_kp kp;
                    if (!adjust_extremum(P, o, &layer, &r1, &c1, contrast_thr, edge_thr, sigma, &kp)) continue;
                    const float scl_octv = kp.size * 0.5f / (1 << o);
                    float hist[SIFT_ORI_HIST_BINS];
                    const float omax = orientation_hist(G(P, o, layer), ow, oh, c1, r1, sift_round(SIFT_ORI_RADIUS * scl_octv),
                                                        SIFT_ORI_SIG_FCTR * scl_octv, hist, n);
                    const float mag_thr = omax * SIFT_ORI_PEAK_RATIO;
                    for (int j = 0; j < n; ++j) {
                        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
                            kp.angle = 360.f - (360.f / n) * bin;
                            if (fabsf(kp.angle - 360.f) < SIFT_FLT_EPSILON) kp.angle = 0.f;
                            if (cnt == cap) { cap *= 2; kps = (sift_kp *)realloc(kps, sizeof(sift_kp) * cap); }
                            kps[cnt++] = kp;
                        }
                    }
                }
        }
    }
    /* KeyPointsFilter::removeDuplicatedSorted, then back to the coordinates of the input image */
    qsort(kps, cnt, sizeof(sift_kp), kp_less);
    size_t m = 0;
    for (size_t i = 0; i < cnt; ++i) {
        if (m > 0 && kps[i].x == kps[m - 1].x && kps[i].y == kps[m - 1].y && kps[i].size == kps[m - 1].size &&
            kps[i].angle == kps[m - 1].angle)
            continue;
        kps[m++] = kps[i];
    }
    for (size_t i = 0; i < m; ++i) {
        kps[i].octave = (kps[i].octave & ~255) | ((kps[i].octave + P->first_octave) & 255);
        kps[i].x *= 0.5f; kps[i].y *= 0.5f; kps[i].size *= 0.5f;
    }
    for (size_t i = 0; i < m && i < (size_t)max_kp; ++i) {
        keypoints[i] = kps[i];
        if (descriptors) {
            int octave = kps[i].octave & 255;
            const int layer = (kps[i].octave >> 8) & 255;
            octave = octave < 128 ? octave : (-128 | octave);
            const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
            const float size = kps[i].size * scale;
            const int o = octave - P->first_octave;
            float angle = 360.f - kps[i].angle;
            if (fabsf(angle - 360.f) < SIFT_FLT_EPSILON) angle = 0.f;
            descriptor(G(P, o, layer), P->w[o], P->h[o], kps[i].x * scale, kps[i].y * scale, angle, size * 0.5f,
                       descriptors + 128 * i);
        }
    }
    free(kps);
    free_pyramid(P);
    return (int)m;
}
