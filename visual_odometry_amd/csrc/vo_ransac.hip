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
