// Gaussian image pyramid (pyrDown) for batches of 8-bit grayscale images.
//
// Semantics follow the pyramid cv.calcOpticalFlowPyrLK builds internally for the
// reference's KLT call (vo_pipeline/poseEstimation.py:122-132): separable
// [1 4 6 4 1] filter, BORDER_REFLECT_101, dst = (sum + 128) >> 8, destination
// size ((w+1)/2, (h+1)/2), and a level count clamped so that every level stays
// larger than the tracking window.
//
// HBM-bound byte work: one workgroup turns a (2*64+3) x (2*16+3) input tile,
// staged once in LDS, into a 64 x 16 output tile; rows are filtered horizontally
// into 16-bit sums in LDS, then vertically, and each thread stores four output
// pixels as one aligned dword (level pitches are multiples of 64 bytes).
#include "vo_internal.h"
#include <new>

#define PD_TW 64          // output tile width
#define PD_TH 16          // output tile height
#define PD_IW (2 * PD_TW + 3)
#define PD_IH (2 * PD_TH + 3)
#define PD_IS 132         // LDS row stride of the input tile (bytes)

__device__ __forceinline__ int reflect101(int i, int n)
{
    // valid for -n < i < 2n-1, which the window/halo sizes used here guarantee
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return i;
}

typedef unsigned short pd_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pd_u16x2 pd_pair(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_bit_cast(pd_u16x2, __builtin_amdgcn_perm(hi, lo, sel));
}

__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t *__restrict__ src, int sw, int sh,
                                                       int64_t spitch, int64_t sstride,
                                                       uint8_t *__restrict__ dst, int dw, int dh,
                                                       int64_t dpitch, int64_t dstride, int tiles_x, int tiles_y, int batch)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[PD_IH * PD_IS];
    __shared__ __attribute__((aligned(16))) uint16_t hsum[PD_IH * PD_TW];

    const int t = threadIdx.x;
    // workgroups b and b+8 run on the same XCD (round-robin dispatch): give every XCD whole
    // images, so the rows two neighbouring tiles share are fetched into one L2 only
    const int per_image = tiles_x * tiles_y;
    int bid = blockIdx.x, img, tl;
    if ((batch & 7) == 0) {
        const int xcd = bid & 7, j = bid >> 3;
        img = (j / per_image) * 8 + xcd;
        tl = j % per_image;
    } else {
        img = bid / per_image;
        tl = bid % per_image;
    }
    const int ox0 = (tl % tiles_x) * PD_TW, oy0 = (tl / tiles_x) * PD_TH;
    const uint8_t *simg = src + (int64_t)img * sstride;
    uint8_t *dimg = dst + (int64_t)img * dstride;
    const int ix0 = 2 * ox0 - 2, iy0 = 2 * oy0 - 2;

    // stage the input tile.  Interior tiles: unaligned 4-byte reads, consecutive lanes
    // on consecutive dwords of a row (33 dwords cover the 131 bytes needed), all five
    // reads of a lane in flight before the first LDS store; tiles touching the image
    // border read bytes with REFLECT_101 applied.
    const bool interior = ix0 >= 0 && iy0 >= 0 && ix0 + 4 * (PD_IS / 4) <= sw && iy0 + PD_IH <= sh;
    if (interior) {
        const uint8_t *base = simg + (int64_t)iy0 * spitch + ix0;
        uint32_t *tile32 = reinterpret_cast<uint32_t *>(tile);
        constexpr int ITEMS = PD_IH * (PD_IS / 4), ROUNDS = (ITEMS + 255) / 256;
        uint32_t q[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int idx = min(t + 256 * k, ITEMS - 1);
            const int r = (idx * 1986) >> 16, c = idx - r * (PD_IS / 4);   // idx / 33 for idx < 2560
            __builtin_memcpy(&q[k], base + (int64_t)r * spitch + 4 * c, 4);
        }
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) tile32[min(t + 256 * k, ITEMS - 1)] = q[k];
    } else {
        // tiles touching the image border: same dword items with the row reflected per item;
        // only the dwords that straddle the left/right border fall back to four byte reads
        // with REFLECT_101.  Positions needed only by outputs outside the image may fall
        // outside the reflect range; they are clamped (their results are never stored).
        uint32_t *tile32 = reinterpret_cast<uint32_t *>(tile);
        constexpr int ITEMS = PD_IH * (PD_IS / 4), ROUNDS = (ITEMS + 255) / 256;
        uint32_t q[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int idx = min(t + 256 * k, ITEMS - 1);
            const int r = (idx * 1986) >> 16, c = idx - r * (PD_IS / 4);
            int y = min(max(iy0 + r, -(sh - 1)), 2 * sh - 2);
            y = sh > 1 ? reflect101(y, sh) : 0;
            const uint8_t *row = simg + (int64_t)y * spitch;
            const int x = ix0 + 4 * c;
            if (x >= 0 && x + 3 < sw) {
                __builtin_memcpy(&q[k], row + x, 4);
            } else {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int xx = min(max(x + b, -(sw - 1)), 2 * sw - 2);
                    xx = sw > 1 ? reflect101(xx, sw) : 0;
                    v |= (uint32_t)row[xx] << (8 * b);
                }
                q[k] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) tile32[min(t + 256 * k, ITEMS - 1)] = q[k];
    }
    __syncthreads();

    // horizontal 5-tap on every staged row, two adjacent outputs per item: the eight bytes
    // b0..b7 at the pair's origin become 16-bit lane pairs {b0,b2} {b1,b3} {b2,b4} {b3,b5}
    // {b4,b6}, so packed 16-bit math yields {h(ox), h(ox+1)} for one dword store
    {
        const uint32_t *tile32 = reinterpret_cast<const uint32_t *>(tile);
        uint32_t *hsum32 = reinterpret_cast<uint32_t *>(hsum);
        for (int idx = t; idx < PD_IH * (PD_TW / 2); idx += 256) {
            const int r = idx >> 5, pr = idx & 31;
            const uint32_t lo = tile32[r * (PD_IS / 4) + pr], hi = tile32[r * (PD_IS / 4) + pr + 1];
            const pd_u16x2 p0 = pd_pair(hi, lo, 0x0c020c00u), p1 = pd_pair(hi, lo, 0x0c030c01u),
                           p2 = pd_pair(hi, lo, 0x0c040c02u), p3 = pd_pair(hi, lo, 0x0c050c03u),
                           p4 = pd_pair(hi, lo, 0x0c060c04u);
            const pd_u16x2 h = (p0 + p4) + (p1 + p3) * (unsigned short)4 + p2 * (unsigned short)6;
            hsum32[idx] = __builtin_bit_cast(uint32_t, h);
        }
    }
    __syncthreads();

    // vertical 5-tap; thread -> 4 consecutive outputs of one row, packed 16-bit math
    // (255 * 16 * 16 + 128 < 65536)
    const int oxg = (t & 15) * 4, gx = ox0 + oxg;
#pragma unroll
    for (int half = 0; half < PD_TH / 16; ++half) {
        const int oy = (t >> 4) + 16 * half, gy = oy0 + oy;
        if (gy < dh && gx < dw) {
            const uint2 *h0 = reinterpret_cast<const uint2 *>(hsum + (2 * oy) * PD_TW + oxg);
            constexpr int RS = PD_TW / 4;  // row stride in uint2
            const uint2 a0 = h0[0], a1 = h0[RS], a2 = h0[2 * RS], a3 = h0[3 * RS], a4 = h0[4 * RS];
            uint32_t out2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const pd_u16x2 v0 = __builtin_bit_cast(pd_u16x2, k ? a0.y : a0.x), v1 = __builtin_bit_cast(pd_u16x2, k ? a1.y : a1.x),
                               v2 = __builtin_bit_cast(pd_u16x2, k ? a2.y : a2.x), v3 = __builtin_bit_cast(pd_u16x2, k ? a3.y : a3.x),
                               v4 = __builtin_bit_cast(pd_u16x2, k ? a4.y : a4.x);
                const pd_u16x2 v = ((v0 + v4) + (v1 + v3) * (unsigned short)4 + v2 * (unsigned short)6 + (unsigned short)128) >> (unsigned short)8;
                out2[k] = __builtin_bit_cast(uint32_t, v);
            }
            // bytes past dw land in the row's pitch padding
            *reinterpret_cast<uint32_t *>(dimg + (int64_t)gy * dpitch + gx) = __builtin_amdgcn_perm(out2[1], out2[0], 0x06040200u);
        }
    }
}

static int eff_max_level(int w, int h, int max_level, int win)
{
    for (int level = 0; level <= max_level; ++level) {
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) return level;
    }
    return max_level;
}

extern "C" int vo_pyr_create(vo_ctx *ctx, int batch, int height, int width, int max_level, int win, vo_pyr **out)
{
    VO_REQUIRE(ctx && out, "vo_pyr_create: NULL argument");
    VO_REQUIRE(batch >= 1 && height >= 1 && width >= 1, "vo_pyr_create: bad shape batch=%d h=%d w=%d", batch, height, width);
    VO_REQUIRE(max_level >= 0 && win >= 1, "vo_pyr_create: bad max_level=%d win=%d", max_level, win);
    VO_REQUIRE(height > win && width > win, "vo_pyr_create: image %dx%d must be larger than the window %d", width, height, win);
    if (max_level > VO_MAX_LEVELS - 1) max_level = VO_MAX_LEVELS - 1;
    vo_pyr *p = new (std::nothrow) vo_pyr();
    if (!p) { vo_set_error("vo_pyr_create: out of host memory"); return VO_ENOMEM; }
    p->ctx = ctx;
    p->batch = batch;
    p->win = win;
    p->max_level = eff_max_level(width, height, max_level, win);
    p->w[0] = width;
    p->h[0] = height;
    size_t total = 0;
    size_t off[VO_MAX_LEVELS] = {0};
    for (int l = 1; l <= p->max_level; ++l) {
        p->w[l] = (p->w[l - 1] + 1) / 2;
        p->h[l] = (p->h[l - 1] + 1) / 2;
        p->pitch[l] = ((int64_t)p->w[l] + 63) / 64 * 64;
        p->img_stride[l] = p->pitch[l] * p->h[l];
        off[l] = total;
        total += (size_t)p->img_stride[l] * batch;
    }
    total += 256;  // slack: the tracker's 4-byte row reads may run 3 bytes past a row
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc((void **)&p->pool, total);
    if (e != hipSuccess) {
        delete p;
        vo_set_error("vo_pyr_create: device allocation of %zu bytes failed: %s", total, hipGetErrorString(e));
        return VO_ENOMEM;
    }
    for (int l = 1; l <= p->max_level; ++l) p->lvl[l] = p->pool + off[l];
    *out = p;
    return VO_OK;
}

extern "C" int vo_pyr_destroy(vo_pyr *pyr)
{
    if (!pyr) return VO_OK;
    if (vo_ctx_is_live(pyr->ctx)) {
        (void)hipSetDevice(pyr->ctx->device);
        (void)hipStreamSynchronize(pyr->ctx->stream);
    }
    if (pyr->pool) (void)hipFree(pyr->pool);
    if (pyr->own0) (void)hipFree(pyr->own0);
    delete pyr;
    return VO_OK;
}

extern "C" int vo_pyr_max_level(const vo_pyr *pyr) { return pyr ? pyr->max_level : VO_EINVAL; }

extern "C" int vo_pyr_level_shape(const vo_pyr *pyr, int level, int *height, int *width, int64_t *pitch)
{
    VO_REQUIRE(pyr != nullptr, "vo_pyr_level_shape: pyr is NULL");
    VO_REQUIRE(level >= 0 && level <= pyr->max_level, "vo_pyr_level_shape: level %d out of [0,%d]", level, pyr->max_level);
    if (height) *height = pyr->h[level];
    if (width) *width = pyr->w[level];
    if (pitch) *pitch = pyr->pitch[level];
    return VO_OK;
}

extern "C" int vo_pyr_build(vo_ctx *ctx, vo_pyr *pyr, const uint8_t *img, int64_t row_stride, int64_t img_stride, int mem)
{
    VO_REQUIRE(ctx && pyr && img, "vo_pyr_build: NULL argument");
    VO_REQUIRE(pyr->ctx == ctx, "vo_pyr_build: pyramid belongs to another context");
    VO_REQUIRE(row_stride >= pyr->w[0], "vo_pyr_build: row_stride %lld < width %d", (long long)row_stride, pyr->w[0]);
    VO_REQUIRE(pyr->batch == 1 || img_stride >= row_stride * pyr->h[0], "vo_pyr_build: img_stride %lld too small", (long long)img_stride);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_pyr_build: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    if (mem == VO_MEM_HOST) {
        // tight library-owned copy of level 0 (+ slack for 4-byte row reads)
        size_t need = (size_t)pyr->w[0] * pyr->h[0] * pyr->batch + 256;
        if (pyr->own0_bytes < need) {
            if (pyr->own0) { VO_HIP_CHECK(hipStreamSynchronize(ctx->stream)); VO_HIP_CHECK(hipFree(pyr->own0)); pyr->own0 = nullptr; }
            hipError_t e = hipMalloc((void **)&pyr->own0, need);
            if (e != hipSuccess) { vo_set_error("vo_pyr_build: device allocation of %zu bytes failed", need); return VO_ENOMEM; }
            pyr->own0_bytes = need;
        }
        const size_t img_bytes = (size_t)pyr->w[0] * pyr->h[0];
        if (row_stride == pyr->w[0] && (pyr->batch == 1 || img_stride == (int64_t)img_bytes)) {
            // tight input: one linear copy (2-D copies from pageable memory go row by row)
            VO_HIP_CHECK(hipMemcpyAsync(pyr->own0, img, img_bytes * pyr->batch, hipMemcpyHostToDevice, ctx->stream));
        } else {
            for (int b = 0; b < pyr->batch; ++b)
                VO_HIP_CHECK(hipMemcpy2DAsync(pyr->own0 + b * img_bytes, pyr->w[0], img + (size_t)b * img_stride, row_stride,
                                              pyr->w[0], pyr->h[0], hipMemcpyHostToDevice, ctx->stream));
        }
        pyr->lvl[0] = pyr->own0;
        pyr->pitch[0] = pyr->w[0];
        pyr->img_stride[0] = (int64_t)pyr->w[0] * pyr->h[0];
    } else {
        pyr->lvl[0] = const_cast<uint8_t *>(img);
        pyr->pitch[0] = row_stride;
        pyr->img_stride[0] = img_stride;
    }
    for (int l = 1; l <= pyr->max_level; ++l) {
        const int tiles_x = (pyr->w[l] + PD_TW - 1) / PD_TW, tiles_y = (pyr->h[l] + PD_TH - 1) / PD_TH;
        const int64_t blocks = (int64_t)tiles_x * tiles_y * pyr->batch;
        VO_REQUIRE(blocks <= 0x7fffffff, "vo_pyr_build: %lld tiles exceed the grid limit", (long long)blocks);
        hipLaunchKernelGGL(pyr_down_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, pyr->lvl[l - 1], pyr->w[l - 1],
                           pyr->h[l - 1], pyr->pitch[l - 1], pyr->img_stride[l - 1], pyr->lvl[l], pyr->w[l], pyr->h[l],
                           pyr->pitch[l], pyr->img_stride[l], tiles_x, tiles_y, pyr->batch);
    }
    VO_HIP_CHECK(hipGetLastError());
    pyr->built = true;
    if (mem == VO_MEM_HOST) VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_pyr_download(vo_ctx *ctx, const vo_pyr *pyr, int image, int level, uint8_t *host_out)
{
    VO_REQUIRE(ctx && pyr && host_out, "vo_pyr_download: NULL argument");
    VO_REQUIRE(pyr->built, "vo_pyr_download: pyramid not built");
    VO_REQUIRE(image >= 0 && image < pyr->batch && level >= 0 && level <= pyr->max_level, "vo_pyr_download: image/level out of range");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    VO_HIP_CHECK(hipMemcpy2DAsync(host_out, pyr->w[level], pyr->lvl[level] + (int64_t)image * pyr->img_stride[level],
                                  pyr->pitch[level], pyr->w[level], pyr->h[level], hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}
