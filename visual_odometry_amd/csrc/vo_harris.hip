// Harris / Shi-Tomasi corner response and non-maximum suppression + top-K.
//
// Response: cv.cornerHarris(float32(img), blockSize=2, ksize=3, k) as
// FeatureExtractor.harris calls it (vo_pipeline/featureExtraction.py:37-40):
// float32 Sobel 3x3 scaled by 1/8 (REFLECT_101), 2x2 un-normalised box sum of
// the derivative products with offsets {-1,0} (REFLECT_101 on the product maps),
// R = a*c - b*b - k*(a+c)^2; mode 1 = min-eigenvalue (Shi-Tomasi) response.
// One workgroup stages a (8+3)x(32+3) image tile in LDS, derives the 9x33
// derivative tile from it and emits 8x32 responses; image bytes are read once
// from HBM per tile (+halo), the response is written once.
//
// NMS/top-K: per image (1) max reduction, (2) strict local maxima above
// rel_thresh*max in a (2r+1)^2 window are appended as 64-bit keys
// {ordered(response), ~raster index}, (3) one workgroup radix-selects the K-th
// largest key, gathers the K winners into LDS, bitonic-sorts them and writes
// (x, y) strongest first.  Keys are unique, so the output is deterministic.
#include "vo_internal.h"
#include <limits.h>

#define HR_TW 32
#define HR_TH 8
#define HR_IW (HR_TW + 3)
#define HR_IH (HR_TH + 3)
#define HR_DW (HR_TW + 1)
#define HR_DH (HR_TH + 1)
#define TOPK_MAX 4096
#define TOPK_THREADS 1024

__device__ __forceinline__ int refl1(int i, int n)
{
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

__global__ __launch_bounds__(256) void harris_response_kernel(const uint8_t *__restrict__ img, int w, int h,
                                                              int64_t pitch, int64_t istride, float k, int mode,
                                                              float *__restrict__ R)
{
    __shared__ float tile[HR_IH][HR_IW + 1];
    __shared__ float dxs[HR_DH][HR_DW + 1];
    __shared__ float dys[HR_DH][HR_DW + 1];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * HR_TW, y0 = blockIdx.y * HR_TH;
    const uint8_t *src = img + (int64_t)blockIdx.z * istride;
    float *dst = R + (int64_t)blockIdx.z * w * h;

    for (int idx = t; idx < HR_IH * HR_IW; idx += 256) {
        int r = idx / HR_IW, c = idx - r * HR_IW;
        int y = refl1(y0 - 2 + r, h), x = refl1(x0 - 2 + c, w);
        tile[r][c] = (float)src[(int64_t)y * pitch + x];
    }
    __syncthreads();
    for (int idx = t; idx < HR_DH * HR_DW; idx += 256) {
        int r = idx / HR_DW, c = idx - r * HR_DW;   // derivative at image (x0-1+c, y0-1+r) = tile (c+1, r+1)
        const float *p0 = &tile[r][c], *p1 = &tile[r + 1][c], *p2 = &tile[r + 2][c];
        float d0 = p0[2] - p0[0], d1 = p1[2] - p1[0], d2 = p2[2] - p2[0];
        float dx = d1 * 0.25f + (d0 + d2) * 0.125f;
        float s0 = p0[1] * 0.25f + (p0[0] + p0[2]) * 0.125f;
        float s2 = p2[1] * 0.25f + (p2[0] + p2[2]) * 0.125f;
        float dy = s2 - s0;
        // positions left of / above the image stand for the product maps at the
        // mirrored coordinate: undo the sign the mirrored image gives the odd part
        if (x0 - 1 + c < 0) dx = -dx;
        if (y0 - 1 + r < 0) dy = -dy;
        dxs[r][c] = dx;
        dys[r][c] = dy;
    }
    __syncthreads();
    const int tx = t & (HR_TW - 1), ty = t >> 5;
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx >= w || gy >= h) return;
    float a, b, c;
    {
        float ax = dxs[ty][tx], ay = dys[ty][tx], bx = dxs[ty][tx + 1], by = dys[ty][tx + 1];
        a = ax * ax + bx * bx; b = ax * ay + bx * by; c = ay * ay + by * by;
        ax = dxs[ty + 1][tx]; ay = dys[ty + 1][tx]; bx = dxs[ty + 1][tx + 1]; by = dys[ty + 1][tx + 1];
        float ra = ax * ax + bx * bx, rb = ax * ay + bx * by, rc = ay * ay + by * by;
        a = a + ra; b = b + rb; c = c + rc;
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
    dst[(int64_t)gy * w + gx] = out;
}

__device__ __forceinline__ uint32_t ordered_key(float v)
{
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void max_key_kernel(const float *__restrict__ R, int64_t npix, uint32_t *__restrict__ maxkey)
{
    __shared__ uint32_t wave_max[4];
    const float *src = R + (int64_t)blockIdx.y * npix;
    uint32_t m = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (((npix & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4 *v4 = reinterpret_cast<const float4 *>(src);
        const int64_t n4 = npix >> 2;
        for (; i + stride < n4; i += 2 * stride) {   // two independent 16-byte loads in flight
            const float4 a = v4[i], b = v4[i + stride];
            m = max(m, max(max(ordered_key(a.x), ordered_key(a.y)), max(ordered_key(a.z), ordered_key(a.w))));
            m = max(m, max(max(ordered_key(b.x), ordered_key(b.y)), max(ordered_key(b.z), ordered_key(b.w))));
        }
        for (; i < n4; i += stride) {
            const float4 a = v4[i];
            m = max(m, max(max(ordered_key(a.x), ordered_key(a.y)), max(ordered_key(a.z), ordered_key(a.w))));
        }
    } else {
        for (; i < npix; i += stride) m = max(m, ordered_key(src[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(&maxkey[blockIdx.y], max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3])));
}

// before/after rule of the suppression: an earlier raster neighbour wins ties
__device__ __forceinline__ bool beaten(float q, float v, bool before) { return before ? (q >= v) : (q > v); }

// Stage 1: pixels above the threshold that are maxima of their 3x3 neighbourhood.
// radius == 1: they are final (written as 64-bit keys); else their raster indices
// go to a per-image list for the full-window test.
__global__ __launch_bounds__(256) void nms3x3_kernel(const float *__restrict__ R, int w, int h, float rel_thresh,
                                                     const uint32_t *__restrict__ maxkey, int final_stage,
                                                     unsigned long long *__restrict__ cand, int cap,
                                                     uint32_t *__restrict__ list, int list_cap, int *__restrict__ counter)
{
    __shared__ int s_n, s_base;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const float *src = R + (int64_t)b * w * h;
    const float thr = rel_thresh * key_value(maxkey[b]);
    float v = 0.f;
    bool ok = x < w && y < h;
    if (ok) {
        v = src[(int64_t)y * w + x];
        ok = v > thr;
    }
    if (ok) {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                if (dx == 0 && dy == 0) continue;
                const int xx = x + dx, yy = y + dy;
                if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
                ok = ok && !beaten(src[(int64_t)yy * w + xx], v, dy < 0 || (dy == 0 && dx < 0));
            }
    }
    // block-level compaction: one global atomic per workgroup instead of one per survivor
    const uint32_t idx = (uint32_t)(y * w + x);
    int slot = -1;
    if (ok) slot = atomicAdd(&s_n, 1);
    __syncthreads();
    if (threadIdx.x == 0 && s_n > 0) s_base = atomicAdd(&counter[b], s_n);
    __syncthreads();
    if (slot < 0) return;
    const int pos = s_base + slot;
    if (final_stage) {
        if (pos < cap) cand[(int64_t)b * cap + pos] = ((unsigned long long)ordered_key(v) << 32) | (0xffffffffu - idx);
    } else {
        if (pos < list_cap) list[(int64_t)b * list_cap + pos] = idx;
    }
}

// Stage 2: one wavefront per 3x3 maximum scans its (2r+1)^2 window, 64 neighbours
// per step; survivors are appended as 64-bit keys.
__global__ __launch_bounds__(256) void nms_window_kernel(const float *__restrict__ R, int w, int h, int radius,
                                                         const uint32_t *__restrict__ list, int list_cap,
                                                         const int *__restrict__ list_count,
                                                         unsigned long long *__restrict__ cand, int cap,
                                                         int *__restrict__ counter)
{
    __shared__ unsigned long long s_keys[4][64];   // survivors of one wave, flushed with one atomic
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = min(list_count[b], list_cap);
    const int side = 2 * radius + 1, area = side * side;
    const float *src = R + (int64_t)b * w * h;
    int kept = 0;
    for (int c = blockIdx.x * 4 + wv; c < n; c += gridDim.x * 4) {
        const uint32_t idx = list[(int64_t)b * list_cap + c];
        const int x = (int)(idx % (uint32_t)w), y = (int)(idx / (uint32_t)w);
        const float v = src[idx];
        bool lost = false;
        for (int k = lane; k < area; k += 64) {
            const int dy = k / side - radius, dx = k % side - radius;
            const int xx = x + dx, yy = y + dy;
            if ((dx | dy) != 0 && xx >= 0 && xx < w && yy >= 0 && yy < h)
                lost = lost || beaten(src[(int64_t)yy * w + xx], v, dy < 0 || (dy == 0 && dx < 0));
        }
        if (__ballot(lost) == 0) {
            if (lane == 0) s_keys[wv][kept] = ((unsigned long long)ordered_key(v) << 32) | (0xffffffffu - idx);
            ++kept;
        }
        if (kept == 64 || c + gridDim.x * 4 >= n) {
            // flush (wave-uniform): LDS accesses of one wave are ordered
            int base = 0;
            if (lane == 0 && kept > 0) base = atomicAdd(&counter[b], kept);
            base = __shfl(base, 0);
            if (lane < kept && base + lane < cap) cand[(int64_t)b * cap + base + lane] = s_keys[wv][lane];
            kept = 0;
        }
    }
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const unsigned long long *__restrict__ cand, int cap,
                                                            const int *__restrict__ counter, int w, int K,
                                                            float *__restrict__ xy, float *__restrict__ scores,
                                                            int *__restrict__ count)
{
    __shared__ unsigned long long sel[TOPK_MAX];
    __shared__ int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_need, s_nsel;
    const int b = blockIdx.x, t = threadIdx.x;
    const unsigned long long *c = cand + (int64_t)b * cap;
    const int n = min(counter[b], cap);
    const int m = min(n, K);
    unsigned long long T = 0;  // K-th largest key (everything >= T is selected)
    if (n > K) {
        if (t == 0) { s_prefix = 0; s_need = K; }
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (t < 256) hist[t] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            const unsigned long long himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
            for (int i = t; i < n; i += TOPK_THREADS) {
                const unsigned long long key = c[i];
                if ((key & himask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
            }
            __syncthreads();
            if (t == 0) {
                int need = s_need, bin = 255;
                for (; bin > 0; --bin) {
                    if (hist[bin] >= need) break;
                    need -= hist[bin];
                }
                s_need = need;
                s_prefix = prefix | ((unsigned long long)bin << shift);
            }
            __syncthreads();
        }
        T = s_prefix;
    }
    if (t == 0) s_nsel = 0;
    for (int i = t; i < TOPK_MAX; i += TOPK_THREADS) sel[i] = 0;
    __syncthreads();
    for (int i = t; i < n; i += TOPK_THREADS) {
        const unsigned long long key = c[i];
        if (key >= T) {
            const int pos = atomicAdd(&s_nsel, 1);
            if (pos < TOPK_MAX) sel[pos] = key;
        }
    }
    __syncthreads();
    // bitonic sort, descending, over the next power of two >= m
    int np2 = 1;
    while (np2 < m) np2 <<= 1;
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < np2 / 2; i += TOPK_THREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = sel[lo], bb = sel[hi];
                if (desc ? (a < bb) : (a > bb)) { sel[lo] = bb; sel[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = t; i < K; i += TOPK_THREADS) {
        float fx = 0.f, fy = 0.f, sc = 0.f;
        if (i < m) {
            const unsigned long long key = sel[i];
            const uint32_t idx = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
            fx = (float)(idx % (uint32_t)w);
            fy = (float)(idx / (uint32_t)w);
            sc = key_value((uint32_t)(key >> 32));
        }
        xy[((int64_t)b * K + i) * 2] = fx;
        xy[((int64_t)b * K + i) * 2 + 1] = fy;
        if (scores) scores[(int64_t)b * K + i] = sc;
    }
    if (t == 0) count[b] = m;
}

extern "C" int vo_harris_response(vo_ctx *ctx, const uint8_t *img, int batch, int height, int width, int64_t row_stride,
                                  int64_t img_stride, int block_size, int ksize, float k, int mode, float *R, int mem)
{
    VO_REQUIRE(ctx && img && R, "vo_harris_response: NULL argument");
    VO_REQUIRE(batch >= 1 && batch <= 65535 && height >= 3 && width >= 3, "vo_harris_response: bad shape batch=%d h=%d w=%d", batch, height, width);
    VO_REQUIRE(block_size == 2 && ksize == 3, "vo_harris_response: only blockSize=2, ksize=3 (the reference's values) are implemented");
    VO_REQUIRE(mode == 0 || mode == 1, "vo_harris_response: mode %d (0 = Harris, 1 = min eigenvalue)", mode);
    VO_REQUIRE(row_stride >= width && (batch == 1 || img_stride >= row_stride * height), "vo_harris_response: bad strides");
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_harris_response: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const uint8_t *d_img = img;
    float *d_R = R;
    int64_t pitch = row_stride, istride = img_stride;
    int rc;
    const size_t npix = (size_t)width * height;
    if (mem == VO_MEM_HOST) {
        void *p;
        if ((rc = vo_stage_reserve(ctx, 0, npix * batch, &p))) return rc; d_img = (const uint8_t *)p;
        if ((rc = vo_stage_reserve(ctx, 1, npix * batch * 4, &p))) return rc; d_R = (float *)p;
        if (row_stride == width && (batch == 1 || img_stride == (int64_t)npix)) {
            VO_HIP_CHECK(hipMemcpyAsync((void *)d_img, img, npix * batch, hipMemcpyHostToDevice, ctx->stream));
        } else {
            for (int b = 0; b < batch; ++b)
                VO_HIP_CHECK(hipMemcpy2DAsync((void *)(d_img + b * npix), width, img + (size_t)b * img_stride, row_stride,
                                              width, height, hipMemcpyHostToDevice, ctx->stream));
        }
        pitch = width; istride = (int64_t)npix;
    }
    dim3 grid((width + HR_TW - 1) / HR_TW, (height + HR_TH - 1) / HR_TH, batch);
    VO_REQUIRE(grid.y <= 65535u, "vo_harris_response: image too tall");
    hipLaunchKernelGGL(harris_response_kernel, grid, dim3(256), 0, ctx->stream, d_img, width, height, pitch, istride, k, mode, d_R);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_HIP_CHECK(hipMemcpyAsync(R, d_R, npix * batch * 4, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return VO_OK;
}

extern "C" int vo_nms_topk(vo_ctx *ctx, const float *R, int batch, int height, int width, float rel_thresh, int radius,
                           int K, float *xy, float *scores, int32_t *count, int mem)
{
    VO_REQUIRE(ctx && R && xy && count, "vo_nms_topk: NULL argument");
    VO_REQUIRE(batch >= 1 && batch <= 65535 && height >= 1 && width >= 1, "vo_nms_topk: bad shape");
    VO_REQUIRE((int64_t)height * width < (1ll << 31), "vo_nms_topk: image too large");
    VO_REQUIRE(radius >= 1 && radius <= 32, "vo_nms_topk: radius %d out of [1,32]", radius);
    VO_REQUIRE(K >= 1 && K <= TOPK_MAX, "vo_nms_topk: K=%d out of [1,%d]", K, TOPK_MAX);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_nms_topk: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npix = (size_t)width * height;
    // two local maxima are more than `radius` apart, so each owns a (radius+1)^2 cell
    const int cap = ((height + radius) / (radius + 1)) * ((width + radius) / (radius + 1));
    int rc;
    void *p;
    const float *d_R = R;
    float *d_xy = xy, *d_sc = scores;
    int *d_cnt = count;
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 1, npix * batch * 4, &p))) return rc; d_R = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)batch * K * 8, &p))) return rc; d_xy = (float *)p;
        if ((rc = vo_stage_reserve(ctx, 3, (size_t)batch * K * 4, &p))) return rc; d_sc = scores ? (float *)p : nullptr;
        if ((rc = vo_stage_reserve(ctx, 4, (size_t)batch * 4, &p))) return rc; d_cnt = (int *)p;
        VO_HIP_CHECK(hipMemcpyAsync((void *)d_R, R, npix * batch * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = vo_stage_reserve(ctx, 5, (size_t)batch * cap * 8, &p))) return rc;
    unsigned long long *d_cand = (unsigned long long *)p;
    // 3x3 maxima are at least 2 apart: at most ceil(h/2)*ceil(w/2) of them
    const int list_cap = ((height + 1) / 2) * ((width + 1) / 2);
    uint32_t *d_list = nullptr;
    if (radius > 1) {
        if ((rc = vo_stage_reserve(ctx, 7, (size_t)batch * list_cap * 4, &p))) return rc;
        d_list = (uint32_t *)p;
    }
    if ((rc = vo_stage_reserve(ctx, 6, (size_t)batch * 12, &p))) return rc;
    uint32_t *d_max = (uint32_t *)p;
    int *d_counter = (int *)p + batch;       // final candidates per image
    int *d_list_count = (int *)p + 2 * batch;  // 3x3 maxima per image
    VO_HIP_CHECK(hipMemsetAsync(p, 0, (size_t)batch * 12, ctx->stream));
    int mblocks = (int)((npix / 4 + 256 * 4 - 1) / (256 * 4));
    if (mblocks > 512) mblocks = 512;
    if (mblocks < 1) mblocks = 1;
    hipLaunchKernelGGL(max_key_kernel, dim3(mblocks, batch), dim3(256), 0, ctx->stream, d_R, (int64_t)npix, d_max);
    dim3 grid((width + 63) / 64, (height + 3) / 4, batch);
    VO_REQUIRE(grid.y <= 65535u, "vo_nms_topk: image too tall");
    hipLaunchKernelGGL(nms3x3_kernel, grid, dim3(256), 0, ctx->stream, d_R, width, height, rel_thresh,
                       (const uint32_t *)d_max, radius == 1 ? 1 : 0, d_cand, cap, d_list, list_cap,
                       radius == 1 ? d_counter : d_list_count);
    if (radius > 1)
        hipLaunchKernelGGL(nms_window_kernel, dim3(512, batch), dim3(256), 0, ctx->stream, d_R, width, height, radius,
                           (const uint32_t *)d_list, list_cap, (const int *)d_list_count, d_cand, cap, d_counter);
    hipLaunchKernelGGL(topk_kernel, dim3(batch), dim3(TOPK_THREADS), 0, ctx->stream, (const unsigned long long *)d_cand, cap,
                       (const int *)d_counter, width, K, d_xy, d_sc, d_cnt);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_HIP_CHECK(hipMemcpyAsync(xy, d_xy, (size_t)batch * K * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (scores) VO_HIP_CHECK(hipMemcpyAsync(scores, d_sc, (size_t)batch * K * 4, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipMemcpyAsync(count, d_cnt, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return VO_OK;
}
