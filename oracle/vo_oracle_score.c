/*
 * CPU ORACLE (test infrastructure, not product code) -- RANSAC hypothesis scoring.
 *
 * (1) Fundamental-matrix hypotheses: the error and inlier rule that
 *     cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC, thr, conf, maxIters)
 *     applies to every candidate F (reference call:
 *     vo_pipeline/bootstrap.py:246-252).
 * (2) Pose hypotheses: the reprojection error and inlier rule that
 *     cv.solvePnPRansac(pointcloud, img_key_points, K, None, flags=P3P, ...)
 *     applies to every candidate (R|t) (reference call:
 *     vo_pipeline/poseEstimation.py:61-69).
 *
 * The arithmetic lives in opencv-python-headless==4.5.4.60 (environment.yml:93),
 * absent from this build environment; the reference holds no test vectors for
 * these calls.  PARITY UNPINNED against cv2: this restates the published OpenCV
 * 4.5.4 routines (calib3d fundam.cpp FMEstimatorCallback::computeError,
 * solvepnp.cpp PnPRansacCallback::computeError + projectPoints without
 * distortion, ptsetreg.cpp findInliers).  Double arithmetic in source order, no
 * contraction, so a GPU kernel doing the same operations matches bit for bit.
 */
#include <stdint.h>
#include <stddef.h>

/* std::max(x, y) == (x < y) ? y : x, NaN behaviour included */
static inline double std_max(double x, double y) { return (x < y) ? y : x; }

/* m1, m2: (n,2) float32 pixel coordinates; F: (H,9) float64 row-major;
 * counts: (H) int32; masks: (H,n) uint8 or NULL. inlier iff err <= (float)(thr*thr). */
void vo_or_score_fundamental(const float *m1, const float *m2, int n, const double *F, int H,
                             double thr, int32_t *counts, uint8_t *masks)
{
    const float t = (float)(thr * thr);
    for (int h = 0; h < H; ++h) {
        const double *f = F + (size_t)h * 9;
        int good = 0;
        for (int i = 0; i < n; ++i) {
            double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
            double a, b, c, d1, d2, s1, s2;
            a = f[0] * x1 + f[1] * y1 + f[2];
            b = f[3] * x1 + f[4] * y1 + f[5];
            c = f[6] * x1 + f[7] * y1 + f[8];
            s2 = 1. / (a * a + b * b);
            d2 = x2 * a + y2 * b + c;
            a = f[0] * x2 + f[3] * y2 + f[6];
            b = f[1] * x2 + f[4] * y2 + f[7];
            c = f[2] * x2 + f[5] * y2 + f[8];
            s1 = 1. / (a * a + b * b);
            d1 = x1 * a + y1 * b + c;
            float err = (float)std_max(d1 * d1 * s1, d2 * d2 * s2);
            int in = err <= t;
            good += in;
            if (masks) masks[(size_t)h * n + i] = (uint8_t)in;
        }
        counts[h] = good;
    }
}

/* X: (n,3) float32 world points; x: (n,2) float32 observations; K: (3,3) float64
 * row-major (fx, fy, cx, cy are used); Rt: (H,12) float64 row-major [R|t];
 * inlier iff (float)|x - proj|^2 <= (float)(thr*thr). */
void vo_or_score_pnp(const float *X, const float *x, int n, const double *K, const double *Rt, int H,
                     double thr, int32_t *counts, uint8_t *masks)
{
    const float t = (float)(thr * thr);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int h = 0; h < H; ++h) {
        const double *M = Rt + (size_t)h * 12;
        int good = 0;
        for (int i = 0; i < n; ++i) {
            double Xw = X[3 * i], Yw = X[3 * i + 1], Zw = X[3 * i + 2];
            double xc = M[0] * Xw + M[1] * Yw + M[2] * Zw + M[3];
            double yc = M[4] * Xw + M[5] * Yw + M[6] * Zw + M[7];
            double zc = M[8] * Xw + M[9] * Yw + M[10] * Zw + M[11];
            zc = zc ? 1. / zc : 1;
            xc *= zc;
            yc *= zc;
            float u = (float)(xc * fx + cx), v = (float)(yc * fy + cy);
            float du = x[2 * i] - u, dv = x[2 * i + 1] - v;
            float err = du * du + dv * dv;
            int in = err <= t;
            good += in;
            if (masks) masks[(size_t)h * n + i] = (uint8_t)in;
        }
        counts[h] = good;
    }
}
