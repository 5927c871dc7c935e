/*
 * CPU checker for descriptor matching -- TEST INFRASTRUCTURE ONLY (imported by tests/,
 * __graft_entry__.smoke() and the CPU-baseline legs of the benchmarks; never by the
 * product path).
 *
 * Restates matcher.knnMatch(des1, des2, k=2) of FeatureMatcher.match_descriptors
 * (vo_pipeline/featureMatching.py:58) for the exhaustive matcher cv.BFMatcher()
 * (featureMatching.py:45; default NORM_L2, no cross check): for every query row the two
 * train rows of smallest distance sqrtf(sum_k (a_k - b_k)^2), nearest first; and the ratio
 * test of match_filter (featureMatching.py:68-69), d0 < threshold * d1 in double.
 *
 * Parity unpinned against cv2 (opencv-python-headless 4.5.4.60 is not installed and the
 * reference holds no vectors for it): OpenCV adds the 128 squared differences in SIMD
 * partial sums whose order is not restated here.  For descriptors that hold integers in
 * [0, 255] -- what cv.SIFT emits -- every order gives the same float32 sum, so on such
 * inputs the distances equal OpenCV's whatever its order; for other inputs the sum below
 * (ascending k, one fused multiply-add per term) is this build's definition.  Equal
 * distances go to the lower train row.
 */
#include <math.h>
#include <stdint.h>

__attribute__((target("fma")))
void vo_or_knn_match2(const float *des1, int n1, const float *des2, int n2, int dim, double ratio,
                      int32_t *idx, float *dist, uint8_t *keep)
{
    for (int q = 0; q < n1; ++q) {
        const float *a = des1 + (int64_t)q * dim;
        float s0 = INFINITY, s1 = INFINITY;
        int32_t i0 = -1, i1 = -1;
        for (int t = 0; t < n2; ++t) {
            const float *b = des2 + (int64_t)t * dim;
            float s = 0.f;
            for (int k = 0; k < dim; ++k) {
                const float d = a[k] - b[k];
                s = __builtin_fmaf(d, d, s);
            }
            if (i0 < 0 || s < s0) {
                s1 = s0; i1 = i0;
                s0 = s; i0 = t;
            } else if (i1 < 0 || s < s1) {
                s1 = s; i1 = t;
            }
        }
        const float d0 = i0 < 0 ? INFINITY : sqrtf(s0), d1 = i1 < 0 ? INFINITY : sqrtf(s1);
        idx[2 * (int64_t)q] = i0;
        idx[2 * (int64_t)q + 1] = i1;
        dist[2 * (int64_t)q] = d0;
        dist[2 * (int64_t)q + 1] = d1;
        if (keep) keep[q] = (i1 >= 0 && (double)d0 < ratio * (double)d1) ? 1 : 0;
    }
}
