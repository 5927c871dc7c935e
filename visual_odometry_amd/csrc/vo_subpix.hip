// Sub-pixel corner refinement, one corner per wavefront.
//
// Replaces cv.cornerSubPix(img, centroids, (5,5), (-1,-1), (EPS + MAX_ITER, 50, 0.001)) of
// FeatureExtractor.harris (vo_pipeline/featureExtraction.py:47-50): per corner, iterate
//   sample a (2*win+3)^2 patch around the estimate (bilinear, float32),
//   central differences -> gradient, weighted by exp(-x^2) exp(-y^2) over the window,
//   solve the 2x2 system sum(g g^T) q = sum(g g^T p) in float64,
// until the step is below eps, the estimate leaves the image or max_iters is reached; a result
// farther than the window from its start is discarded.  The iteration is a serial chain, so a
// corner gets a whole wavefront: the lanes share the patch samples and the per-element terms (LDS),
// and five lanes each add one of the five float64 sums in the raster order of the window -- the
// scalar algorithm's order, so results equal the CPU checker's bit for bit.  (289 corners that run
// all 50 rounds: 0.33 ms per call; with the summing lanes forming the terms themselves 1.5 ms.)
// Windows that leave the image sample with coordinates clamped to the border.
#include "vo_internal.h"
#include <math.h>
#include <float.h>

#define SP_MAX_WIN 6
struct SubpixWeights { float m[(2 * SP_MAX_WIN + 1) * (2 * SP_MAX_WIN + 1)]; };

__global__ __launch_bounds__(64) void corner_subpix_kernel(const uint8_t *__restrict__ img, int h, int w, int64_t stride,
                                                          float *__restrict__ corners, int count, int win_w, int win_h,
                                                          int max_iters, double eps2, SubpixWeights wt,
                                                          const int *__restrict__ last_dev)
{
    __shared__ float patch[(2 * SP_MAX_WIN + 3) * (2 * SP_MAX_WIN + 3)];
    __shared__ double terms[(2 * SP_MAX_WIN + 1) * (2 * SP_MAX_WIN + 1) * 5];
    const int lane = threadIdx.x, p = blockIdx.x;
    if (last_dev && p > *last_dev) return;      // count known on the device only: corners 0 .. *last_dev exist
    const int ww = 2 * win_w + 1, wh = 2 * win_h + 1, bw = ww + 2, bh = wh + 2;
    const float ctx = corners[2 * p], cty = corners[2 * p + 1];
    float cx = ctx, cy = cty;      // every lane carries the same estimate
    int iter = 0;
    double err = 0;
    do {
        const float ox = cx - (bw - 1) * 0.5f, oy = cy - (bh - 1) * 0.5f;
        const int ix = (int)floorf(ox), iy = (int)floorf(oy);
        const float a = ox - ix, b = oy - iy;
        const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
        for (int s = lane; s < bw * bh; s += 64) {
            const int i = s / bw, j = s - i * bw;
            const uint8_t *r0 = img + (int64_t)min(max(iy + i, 0), h - 1) * stride;
            const uint8_t *r1 = img + (int64_t)min(max(iy + i + 1, 0), h - 1) * stride;
            const int x0 = min(max(ix + j, 0), w - 1), x1 = min(max(ix + j + 1, 0), w - 1);
            patch[s] = (float)r0[x0] * a11 + (float)r0[x1] * a12 + (float)r1[x0] * a21 + (float)r1[x1] * a22;
        }
        __syncthreads();
        // the five terms of every window element, all lanes; then lanes 0..4 add one sum each in raster order
        for (int e = lane; e < ww * wh; e += 64) {
            const int i = e / ww, j = e - i * ww;
            const int c = (i + 1) * bw + j + 1;
            const double py = i - win_h, px = j - win_w;
            const double m = wt.m[e];
            const double tgx = patch[c + 1] - patch[c - 1];
            const double tgy = patch[c + bw] - patch[c - bw];
            const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
            terms[e * 5] = gxx;
            terms[e * 5 + 1] = gxy;
            terms[e * 5 + 2] = gyy;
            terms[e * 5 + 3] = gxx * px + gxy * py;
            terms[e * 5 + 4] = gxy * px + gyy * py;
        }
        __syncthreads();
        double sum = 0;
        if (lane < 5)
            for (int e = 0; e < ww * wh; ++e) sum += terms[e * 5 + lane];
        const double A = __shfl(sum, 0, 64), B = __shfl(sum, 1, 64), C = __shfl(sum, 2, 64), bb1 = __shfl(sum, 3, 64),
                     bb2 = __shfl(sum, 4, 64);
        __syncthreads();               // the patch is rewritten in the next round
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
    if (lane == 0) {
        corners[2 * p] = cx;
        corners[2 * p + 1] = cy;
    }
}

static void subpix_weights(SubpixWeights &wt, int win_w, int win_h)
{
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
}

// Refinement of corners 0 .. *d_last (a count that only the device knows yet; at most max_count corners),
// device pointers, asynchronous.
int vo_corner_subpix_device(vo_ctx *ctx, const uint8_t *d_img, int height, int width, int64_t pitch, float *d_corners,
                            int max_count, const int *d_last, int win_w, int win_h, int max_iters, double eps)
{
    VO_REQUIRE(win_w >= 1 && win_h >= 1 && win_w <= SP_MAX_WIN && win_h <= SP_MAX_WIN, "corner refinement: window half-size (%d,%d) out of [1,%d]",
               win_w, win_h, SP_MAX_WIN);
    if (max_count <= 0) return VO_OK;
    SubpixWeights wt;
    subpix_weights(wt, win_w, win_h);
    max_iters = max_iters < 1 ? 1 : (max_iters > 100 ? 100 : max_iters);
    eps = eps > 0 ? eps : 0;
    hipLaunchKernelGGL(corner_subpix_kernel, dim3(max_count), dim3(64), 0, ctx->stream, d_img, height, width, pitch, d_corners,
                       max_count, win_w, win_h, max_iters, eps * eps, wt, d_last);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
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
    subpix_weights(wt, win_w, win_h);
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
        if (row_stride == width)        // one linear copy: a 2D copy from pageable memory goes row by row
            VO_UPLOAD(ctx, (void *)d_img, img, (size_t)width * height);
        else
            VO_UPLOAD_2D(ctx, (void *)d_img, img, row_stride, width, height);
        VO_UPLOAD(ctx, d_c, corners, (size_t)count * 8);
        pitch = width;
    }
    hipLaunchKernelGGL(corner_subpix_kernel, dim3(count), dim3(64), 0, ctx->stream, d_img, height, width, pitch, d_c,
                       count, win_w, win_h, max_iters, eps * eps, wt, (const int *)nullptr);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, corners, d_c, (size_t)count * 8);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_subpix(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)corner_subpix_kernel));
    return VO_OK;
}
