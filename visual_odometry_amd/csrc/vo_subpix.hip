// Sub-pixel corner refinement, one corner per lane.
//
// Replaces cv.cornerSubPix(img, centroids, (5,5), (-1,-1), (EPS + MAX_ITER, 50, 0.001)) of
// FeatureExtractor.harris (vo_pipeline/featureExtraction.py:47-50): per corner, iterate
//   sample a (2*win+3)^2 patch around the estimate (bilinear, float32),
//   central differences -> gradient, weighted by exp(-x^2) exp(-y^2) over the window,
//   solve the 2x2 system sum(g g^T) q = sum(g g^T p) in float64,
// until the step is below eps, the estimate leaves the image or max_iters is reached; a result
// farther than the window from its start is discarded.  Every lane runs the scalar algorithm in its
// own order (float64 sums in raster order of the window), so results equal the CPU checker's bit
// for bit; a lane's patch lives in LDS, sample-major, so that the 64 lanes of a wavefront hit 64 banks.
// Windows that leave the image sample with coordinates clamped to the border.
#include "vo_internal.h"
#include <math.h>
#include <float.h>

#define SP_MAX_WIN 6   // (2*6+3)^2 floats x 64 lanes = 57.6 KB of LDS
struct SubpixWeights { float m[(2 * SP_MAX_WIN + 1) * (2 * SP_MAX_WIN + 1)]; };

__global__ __launch_bounds__(64) void corner_subpix_kernel(const uint8_t *__restrict__ img, int h, int w, int64_t stride,
                                                          float *__restrict__ corners, int count, int win_w, int win_h,
                                                          int max_iters, double eps2, SubpixWeights wt)
{
    extern __shared__ float patch[];   // [(2*win_h+3) * (2*win_w+3)][64]
    const int lane = threadIdx.x, p = blockIdx.x * 64 + lane;
    if (p >= count) return;
    const int ww = 2 * win_w + 1, wh = 2 * win_h + 1, bw = ww + 2, bh = wh + 2;
    const float ctx = corners[2 * p], cty = corners[2 * p + 1];
    float cx = ctx, cy = cty;
    int iter = 0;
    double err = 0;
    do {
        const float ox = cx - (bw - 1) * 0.5f, oy = cy - (bh - 1) * 0.5f;
        const int ix = (int)floorf(ox), iy = (int)floorf(oy);
        const float a = ox - ix, b = oy - iy;
        const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
        for (int i = 0; i < bh; ++i) {
            const uint8_t *r0 = img + (int64_t)min(max(iy + i, 0), h - 1) * stride;
            const uint8_t *r1 = img + (int64_t)min(max(iy + i + 1, 0), h - 1) * stride;
            for (int j = 0; j < bw; ++j) {
                const int x0 = min(max(ix + j, 0), w - 1), x1 = min(max(ix + j + 1, 0), w - 1);
                patch[(i * bw + j) * 64 + lane] = (float)r0[x0] * a11 + (float)r0[x1] * a12 + (float)r1[x0] * a21 + (float)r1[x1] * a22;
            }
        }
        double A = 0, B = 0, C = 0, bb1 = 0, bb2 = 0;
        for (int i = 0; i < wh; ++i) {
            const double py = i - win_h;
            for (int j = 0; j < ww; ++j) {
                const int c = (i + 1) * bw + j + 1;
                const double m = wt.m[i * ww + j];
                const double tgx = patch[(c + 1) * 64 + lane] - patch[(c - 1) * 64 + lane];
                const double tgy = patch[(c + bw) * 64 + lane] - patch[(c - bw) * 64 + lane];
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
    } while (++iter < max_iters && err > eps2);
    if (fabsf(cx - ctx) > win_w || fabsf(cy - cty) > win_h) { cx = ctx; cy = cty; }
    corners[2 * p] = cx;
    corners[2 * p + 1] = cy;
}

extern "C" int vo_corner_subpix(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, float *corners,
                                int count, int win_w, int win_h, int max_iters, double eps, int mem)
{
    VO_REQUIRE(ctx && img, "vo_corner_subpix: NULL argument");
    VO_REQUIRE(height >= 1 && width >= 1 && row_stride >= width, "vo_corner_subpix: bad shape h=%d w=%d", height, width);
    VO_REQUIRE(win_w >= 1 && win_h >= 1 && win_w <= SP_MAX_WIN && win_h <= SP_MAX_WIN, "vo_corner_subpix: window half-size (%d,%d) out of [1,%d]", win_w,
               win_h, SP_MAX_WIN);
    VO_REQUIRE(count >= 0, "vo_corner_subpix: count=%d", count);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_corner_subpix: bad mem flag %d", mem);
    if (count == 0) return VO_OK;
    VO_REQUIRE(corners != nullptr, "vo_corner_subpix: corners is NULL");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    SubpixWeights wt;
    memset(&wt, 0, sizeof(wt));
    const int ww = 2 * win_w + 1, wh = 2 * win_h + 1;
    for (int i = 0; i < wh; ++i) {                      // the window weights, in float as OpenCV forms them
        const float y = (float)(i - win_h) / win_h;
        const float vy = expf(-y * y);
        for (int j = 0; j < ww; ++j) {
            const float x = (float)(j - win_w) / win_w;
            wt.m[i * ww + j] = (float)(vy * expf(-x * x));
        }
    }
    max_iters = max_iters < 1 ? 1 : (max_iters > 100 ? 100 : max_iters);
    eps = eps > 0 ? eps : 0;
    int rc;
    void *p;
    const uint8_t *d_img = img;
    float *d_c = corners;
    int64_t pitch = row_stride;
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)width * height, &p))) return rc; d_img = (const uint8_t *)p;
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)count * 8, &p))) return rc; d_c = (float *)p;
        VO_HIP_CHECK(hipMemcpy2DAsync((void *)d_img, width, img, row_stride, width, height, hipMemcpyHostToDevice, ctx->stream));
        VO_HIP_CHECK(hipMemcpyAsync(d_c, corners, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
        pitch = width;
    }
    const size_t lds = (size_t)(ww + 2) * (wh + 2) * 64 * sizeof(float);
    hipLaunchKernelGGL(corner_subpix_kernel, dim3((count + 63) / 64), dim3(64), lds, ctx->stream, d_img, height, width, pitch, d_c,
                       count, win_w, win_h, max_iters, eps * eps, wt);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_HIP_CHECK(hipMemcpyAsync(corners, d_c, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return VO_OK;
}
