/*
 * CPU ORACLE (test infrastructure, not product code) -- Harris / Shi-Tomasi corner
 * response and non-maximum suppression with top-K selection.
 *
 * Response: restates cv.cornerHarris(float32(img), blockSize=2, ksize=3, k=0.1)
 * as FeatureExtractor.harris calls it (/root/reference
 * vo_pipeline/featureExtraction.py:37-40): float32 Sobel 3x3 with scale
 * 1/((1<<(ksize-1))*blockSize) = 1/8 folded into the smoothing taps,
 * BORDER_REFLECT_101; products dx*dx, dx*dy, dy*dy; un-normalised 2x2 box sum
 * with anchor (1,1) (offsets {-1,0}) and REFLECT_101 on the product maps;
 * R = a*c - b*b - k*(a+c)*(a+c) in float32 (the vectorised form OpenCV uses for
 * the bulk of each row).  mode 1 gives the min-eigenvalue (Shi-Tomasi) response
 * of cv.cornerMinEigenVal on the same structure tensor.
 * The arithmetic lives in opencv-python-headless==4.5.4.60 (environment.yml:93),
 * absent here, and the reference holds no vectors for it: PARITY UNPINNED against
 * cv2 (restated from imgproc corner.cpp / filter.simd.hpp).
 *
 * NMS/top-K: the reference thresholds the dilated map at 0.01*max, labels
 * connected components and refines centroids (featureExtraction.py:41-50); the
 * product path instead keeps strict local maxima above rel_thresh*max within a
 * (2r+1)^2 window and returns the K strongest, ordered by (response desc, raster
 * index asc).  This file defines that rule for the parity tests.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline int refl(int i, int n)
{
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/* img: h x w uint8 (stride bytes); R: h x w float32. mode 0 = Harris, 1 = min eigenvalue. */
void vo_or_harris_response(const uint8_t *img, int w, int h, int stride, float k, int mode, float *R)
{
    float *dx = (float *)malloc(sizeof(float) * (size_t)w * h);
    float *dy = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = img + (size_t)refl(y - 1, h) * stride;
        const uint8_t *r1 = img + (size_t)y * stride;
        const uint8_t *r2 = img + (size_t)refl(y + 1, h) * stride;
        for (int x = 0; x < w; ++x) {
            int xl = refl(x - 1, w), xr = refl(x + 1, w);
            /* row pass (differences / smoothing along x), then column pass */
            float d0 = (float)r0[xr] - (float)r0[xl], d1 = (float)r1[xr] - (float)r1[xl], d2 = (float)r2[xr] - (float)r2[xl];
            dx[(size_t)y * w + x] = d1 * 0.25f + (d0 + d2) * 0.125f;
            float s0 = (float)r0[x] * 0.25f + ((float)r0[xl] + (float)r0[xr]) * 0.125f;
            float s2 = (float)r2[x] * 0.25f + ((float)r2[xl] + (float)r2[xr]) * 0.125f;
            dy[(size_t)y * w + x] = s2 - s0;
        }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float a = 0, b = 0, c = 0;
            /* row sums first (x-1, x), then the two rows (y-1, y) */
            for (int yy = -1; yy <= 0; ++yy) {
                int Y = refl(y + yy, h);
                int X0 = refl(x - 1, w), X1 = x;
                float ax = dx[(size_t)Y * w + X0], ay = dy[(size_t)Y * w + X0];
                float bx = dx[(size_t)Y * w + X1], by = dy[(size_t)Y * w + X1];
                float ra = ax * ax + bx * bx, rb = ax * ay + bx * by, rc = ay * ay + by * by;
                if (yy == -1) { a = ra; b = rb; c = rc; }
                else { a = a + ra; b = b + rb; c = c + rc; }
            }
            float out;
            if (mode == 0) {
                float ac_bb = a * c - b * b;
                float ac = a + c;
                out = ac_bb - k * ac * ac;
            } else {
                float ha = a * 0.5f, hc = c * 0.5f;
                out = (ha + hc) - sqrtf((ha - hc) * (ha - hc) + b * b);
            }
            R[(size_t)y * w + x] = out;
        }
    free(dx);
    free(dy);
}

typedef struct { float v; int idx; } cand;
static int cand_cmp(const void *pa, const void *pb)
{
    const cand *a = (const cand *)pa, *b = (const cand *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* Local maxima of R above rel_thresh * max(R) within radius r (ties: the earlier
 * raster index wins), strongest K first.  xy: (K,2) float32 (x,y); scores: (K)
 * or NULL.  Unused rows are zero.  Returns the number of corners written. */
int vo_or_nms_topk(const float *R, int w, int h, float rel_thresh, int radius, int K, float *xy, float *scores)
{
    float mx = R[0];
    for (size_t i = 1; i < (size_t)w * h; ++i) if (R[i] > mx) mx = R[i];
    const float thr = rel_thresh * mx;
    cand *cs = (cand *)malloc(sizeof(cand) * (size_t)w * h);
    int n = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float v = R[(size_t)y * w + x];
            if (!(v > thr)) continue;
            int ok = 1;
            for (int yy = y - radius; yy <= y + radius && ok; ++yy) {
                if (yy < 0 || yy >= h) continue;
                for (int xx = x - radius; xx <= x + radius; ++xx) {
                    if (xx < 0 || xx >= w || (xx == x && yy == y)) continue;
                    float q = R[(size_t)yy * w + xx];
                    int before = (yy < y) || (yy == y && xx < x);
                    if (before ? (q >= v) : (q > v)) { ok = 0; break; }
                }
            }
            if (ok) { cs[n].v = v; cs[n].idx = y * w + x; ++n; }
        }
    qsort(cs, (size_t)n, sizeof(cand), cand_cmp);
    int m = n < K ? n : K;
    for (int i = 0; i < K; ++i) {
        xy[2 * i] = i < m ? (float)(cs[i].idx % w) : 0.f;
        xy[2 * i + 1] = i < m ? (float)(cs[i].idx / w) : 0.f;
        if (scores) scores[i] = i < m ? cs[i].v : 0.f;
    }
    free(cs);
    return m;
}
