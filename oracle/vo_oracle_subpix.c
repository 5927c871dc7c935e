/*
 * CPU checker for the tail of FeatureExtractor.harris -- TEST INFRASTRUCTURE ONLY (imported by
 * tests/ and the CPU-baseline legs of the benchmarks; never by the product path).
 *
 * Restates, for vo_pipeline/featureExtraction.py:41-50:
 *   cv.dilate(dst, None)                       3x3 maximum, pixels outside the image ignored  (:41)
 *   cv.threshold(dst, 0.01*dst.max(), 255, 0)  dst > (float)threshold                          (:42)
 *   cv.cornerSubPix(img, centroids, (5,5), (-1,-1), (EPS+MAX_ITER, 50, 0.001))                 (:47-50)
 * from the published OpenCV 4.5.4 algorithm (modules/imgproc/src/cornersubpix.cpp, samplers.cpp).
 * Parity unpinned against cv2 (not installed; the reference holds no vectors).  One known
 * simplification: windows that leave the image are sampled with coordinates clamped to the border
 * and the same four-tap formula; OpenCV's border branch weights the replicated pixel with two taps,
 * which can differ in the last float bit.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>

void vo_or_harris_mask(const float *R, int h, int w, double rel_thresh, uint8_t *mask)
{
    float mx = R[0];
    for (int64_t i = 1; i < (int64_t)h * w; ++i)
        if (R[i] > mx) mx = R[i];
    const float thr = (float)(rel_thresh * (double)mx);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float m = R[(int64_t)y * w + x];
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    const float v = R[(int64_t)yy * w + xx];
                    if (v > m) m = v;
                }
            mask[(int64_t)y * w + x] = m > thr ? 255 : 0;
        }
}

/* exp(-y^2) exp(-x^2) window weights of cornerSubPix, (2*win_h+1) x (2*win_w+1) */
void vo_or_subpix_weights(int win_w, int win_h, float *mask)
{
    const int ww = 2 * win_w + 1, wh = 2 * win_h + 1;
    for (int i = 0; i < wh; ++i) {
        const float y = (float)(i - win_h) / win_h;
        const float vy = expf(-y * y);
        for (int j = 0; j < ww; ++j) {
            const float x = (float)(j - win_w) / win_w;
            mask[i * ww + j] = (float)(vy * expf(-x * x));
        }
    }
}

static int clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

void vo_or_corner_subpix(const uint8_t *img, int h, int w, int64_t stride, float *corners, int count, int win_w, int win_h,
                         int max_iters, double eps)
{
    const int ww = 2 * win_w + 1, wh = 2 * win_h + 1, bw = ww + 2, bh = wh + 2;
    float mask[33 * 33], buf[35 * 35];
    if (win_w < 1 || win_h < 1 || win_w > 16 || win_h > 16) return;
    vo_or_subpix_weights(win_w, win_h, mask);
    if (max_iters < 1) max_iters = 1;
    if (max_iters > 100) max_iters = 100;
    eps = eps > 0 ? eps : 0;
    eps *= eps;
    for (int p = 0; p < count; ++p) {
        const float ctx = corners[2 * p], cty = corners[2 * p + 1];
        float cx = ctx, cy = cty;
        int iter = 0;
        double err = 0;
        do {
            /* getRectSubPix: bw x bh samples centred on (cx, cy) */
            const float ox = cx - (bw - 1) * 0.5f, oy = cy - (bh - 1) * 0.5f;
            const int ix = (int)floorf(ox), iy = (int)floorf(oy);
            const float a = ox - ix, b = oy - iy;
            const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
            for (int i = 0; i < bh; ++i) {
                const uint8_t *r0 = img + (int64_t)clampi(iy + i, h) * stride, *r1 = img + (int64_t)clampi(iy + i + 1, h) * stride;
                for (int j = 0; j < bw; ++j) {
                    const int x0 = clampi(ix + j, w), x1 = clampi(ix + j + 1, w);
                    buf[i * bw + j] = (float)r0[x0] * a11 + (float)r0[x1] * a12 + (float)r1[x0] * a21 + (float)r1[x1] * a22;
                }
            }
            double A = 0, B = 0, C = 0, bb1 = 0, bb2 = 0;
            for (int i = 0; i < wh; ++i) {
                const float *s = buf + (i + 1) * bw + 1;
                const double py = i - win_h;
                for (int j = 0; j < ww; ++j) {
                    const double m = mask[i * ww + j];
                    const double tgx = s[j + 1] - s[j - 1];
                    const double tgy = s[j + bw] - s[j - bw];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    const double px = j - win_w;
                    A += gxx; B += gxy; C += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            const double det = A * C - B * B;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            const double scale = 1.0 / det;
            const float nx = (float)(cx + C * scale * bb1 - B * scale * bb2);
            const float ny = (float)(cy - B * scale * bb1 + A * scale * bb2);
            err = (double)((nx - cx) * (nx - cx) + (ny - cy) * (ny - cy));
            cx = nx; cy = ny;
            if (cx < 0 || cx >= w || cy < 0 || cy >= h) break;
        } while (++iter < max_iters && err > eps);
        if (fabsf(cx - ctx) > win_w || fabsf(cy - cty) > win_h) { cx = ctx; cy = cty; }
        corners[2 * p] = cx;
        corners[2 * p + 1] = cy;
    }
}
