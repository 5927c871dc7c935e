// Gaussian image pyramid (pyrDown) for batches of 8-bit grayscale images.
//
// Semantics follow the pyramid cv.calcOpticalFlowPyrLK builds internally for the
// reference's KLT call (vo_pipeline/poseEstimation.py:122-132): separable
// [1 4 6 4 1] filter, BORDER_REFLECT_101, dst = (sum + 128) >> 8, destination
// size ((w+1)/2, (h+1)/2), and a level count clamped so that every level stays
// larger than the tracking window.
//
// HBM-bound byte work with no reuse beyond a 5x5 stencil: the kernel keeps nothing in LDS;
// every lane streams the input rows of its four output columns through registers (see
// pyr_down_rows_kernel).  Level pitches are multiples of 64 bytes, so a lane's four output
// pixels leave as one aligned dword.
#include "vo_internal.h"
#include <new>

typedef unsigned short pd_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pd_u16x2 pd_pair(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_bit_cast(pd_u16x2, __builtin_amdgcn_perm(hi, lo, sel));
}

// No LDS, no barrier.  A lane owns four neighbouring output columns and
// PR_RO consecutive output rows (8 for large levels, 4 where that is needed to fill the chip); it issues the reads of all 2*PR_RO+3 input rows it needs (three
// unaligned dwords = 12 bytes per row) before the first use, so a wavefront keeps ~14 KB in flight,
// filters each row horizontally with packed 16-bit math (the bytes b0..b11 of a row become 16-bit lane
// pairs {b0,b2} {b1,b3} ... through v_perm_b32, so one packed instruction yields two neighbouring
// outputs; 255 * 16 * 16 + 128 < 65536) and runs the vertical
// filter over a rolling window of five filtered rows held in registers.  Consecutive lanes take
// consecutive column groups (a row of 4-byte stores per wavefront and output row).  Rows are
// reflected (REFLECT_101) by address; lanes whose 12-byte window would leave the row read the 12
// bytes that end at (or start at) the row border instead and gather their window from them with
// per-lane byte selectors; wavefronts without such a lane skip the gather.

__device__ __forceinline__ void pr_hrow(uint32_t n0, uint32_t n1, uint32_t n2, pd_u16x2 &h01, pd_u16x2 &h23)
{
    const pd_u16x2 a0 = pd_pair(n1, n0, 0x0c020c00u), a1 = pd_pair(n1, n0, 0x0c030c01u), a2 = pd_pair(n1, n0, 0x0c040c02u),
                   a3 = pd_pair(n1, n0, 0x0c050c03u), a4 = pd_pair(n1, n0, 0x0c060c04u);
    const pd_u16x2 b1 = pd_pair(n2, n1, 0x0c030c01u), b2 = pd_pair(n2, n1, 0x0c040c02u),
                   b3 = pd_pair(n2, n1, 0x0c050c03u), b4 = pd_pair(n2, n1, 0x0c060c04u);
    h01 = (a0 + a4) + (a1 + a3) * (unsigned short)4 + a2 * (unsigned short)6;
    h23 = (a4 + b4) + (b1 + b3) * (unsigned short)4 + b2 * (unsigned short)6;   // a4 = {b4, b6} is the pair's first tap
}

template <int PR_RO>
__global__ __launch_bounds__(256) void pyr_down_rows_kernel(const uint8_t *__restrict__ src, int sw, int sh, int64_t spitch,
                                                            int64_t sstride, uint8_t *__restrict__ dst, int dw, int dh,
                                                            int64_t dpitch, int64_t dstride, int groups, int strips,
                                                            int wgs_per_image, int batch)
{
    constexpr int PR_ROWS = 2 * PR_RO + 3;
    int bid = blockIdx.x, img, wg;
    if ((batch & 7) == 0) {     // workgroups b and b+8 share an XCD: whole images per XCD
        const int xcd = bid & 7, j = bid >> 3;
        img = (j / wgs_per_image) * 8 + xcd;
        wg = j % wgs_per_image;
    } else {
        img = bid / wgs_per_image;
        wg = bid % wgs_per_image;
    }
    const int tid = wg * 256 + threadIdx.x;
    if (tid >= groups * strips) return;
    const int strip = tid / groups, g = tid - strip * groups;
    const int ox = 4 * g, oy0 = strip * PR_RO;
    const int cx = 2 * ox - 2;                          // first input column of the lane's window
    const bool inrow = cx >= 0 && cx + 12 <= sw;
    const int colbase = min(max(cx, 0), sw - 12);       // the launch requires sw >= 12
    // Addresses as the image's base (wave-uniform: scalar registers) plus one unsigned 32-bit offset per lane -- the
    // build checks that a level's pitch fits 24 bits and an image 32 -- formed with the 24-bit multiply: written with
    // 64-bit products, the nineteen row addresses of a lane were 84 quarter-rate multiplies, a fifth of the kernel's
    // vector time.
    const uint8_t *simg = src + (int64_t)img * sstride;
    const unsigned sp = (unsigned)spitch;

    uint32_t w[PR_ROWS][3];
#pragma unroll
    for (int k = 0; k < PR_ROWS; ++k) {
        int y = 2 * oy0 - 2 + k;
        const int a = max(y, -y);                        // REFLECT_101 for one fold, clamped (rows needed only by
        y = max(min(a, 2 * sh - 2 - a), 0);              // outputs below the image)
        const uint8_t *row = simg + (__umul24((unsigned)y, sp) + (unsigned)colbase);
        __builtin_memcpy(&w[k][0], row, 4);
        __builtin_memcpy(&w[k][1], row + 4, 4);
        __builtin_memcpy(&w[k][2], row + 8, 4);
    }

    const bool gather = __ballot(!inrow) != 0;           // wave-uniform
    uint32_t selA[3] = {0x03020100u, 0x07060504u, 0u}, selB[3] = {0x03020100u, 0x03020100u, 0x07060504u};
    if (gather) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            uint32_t sa = 0, sb = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int col = cx + 4 * j + b;
                col = col < 0 ? -col : col;
                col = col >= sw ? 2 * sw - 2 - col : col;
                col = min(max(col, 0), sw - 1);
                const int pos = min(max(col - colbase, 0), 11);      // byte of the 12 that were read
                sa |= (uint32_t)(pos < 8 ? pos : 0) << (8 * b);       // from {w1, w0}
                sb |= (uint32_t)(pos < 8 ? b : pos - 4) << (8 * b);   // keep that byte, or take w2's byte pos - 8
            }
            selA[j] = sa; selB[j] = sb;
        }
    }

    pd_u16x2 h[5][2];
    uint8_t *dimg = dst + (int64_t)img * dstride;
    const unsigned dp = (unsigned)dpitch;
#pragma unroll
    for (int k = 0; k < PR_ROWS; ++k) {
        uint32_t n0 = w[k][0], n1 = w[k][1], n2 = w[k][2];
        if (gather) {
            const uint32_t t0 = __builtin_amdgcn_perm(n1, n0, selA[0]), t1 = __builtin_amdgcn_perm(n1, n0, selA[1]),
                           t2 = __builtin_amdgcn_perm(n1, n0, selA[2]);
            n0 = __builtin_amdgcn_perm(n2, t0, selB[0]);
            n1 = __builtin_amdgcn_perm(n2, t1, selB[1]);
            const uint32_t m2 = __builtin_amdgcn_perm(n2, t2, selB[2]);
            n2 = m2;
        }
        pr_hrow(n0, n1, n2, h[k % 5][0], h[k % 5][1]);
        if (k >= 4 && (k & 1) == 0) {
            const int o = (k - 4) / 2, gy = oy0 + o;     // rows k-4 .. k are the five taps
            uint32_t out2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const pd_u16x2 v0 = h[(k - 4) % 5][q], v1 = h[(k - 3) % 5][q], v2 = h[(k - 2) % 5][q], v3 = h[(k - 1) % 5][q],
                               v4 = h[k % 5][q];
                const pd_u16x2 v = ((v0 + v4) + (v1 + v3) * (unsigned short)4 + v2 * (unsigned short)6 + (unsigned short)128) >> (unsigned short)8;
                out2[q] = __builtin_bit_cast(uint32_t, v);
            }
            // bytes past dw land in the row's pitch padding
            if (gy < dh)
                *reinterpret_cast<uint32_t *>(dimg + (__umul24((unsigned)gy, dp) + (unsigned)ox)) = __builtin_amdgcn_perm(out2[1], out2[0], 0x06040200u);
        }
    }
}

#define PR_TALL_MIN_LANES (1024 * 64 * 2)   // 1024 SIMDs x 64 lanes x 2 wavefronts

static int eff_max_level(int w, int h, int max_level, int win)
{
    for (int level = 0; level <= max_level; ++level) {
        if (w < 12) return level;   // the kernel reads 12-byte windows of a source row
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

static int pyr_own_level0(vo_ctx *ctx, vo_pyr *pyr);
static int pyr_launch_levels(vo_ctx *ctx, vo_pyr *pyr);
#define PYR_STAGED_MAX_BYTES ((size_t)1 << 20)     // host images up to this size go through vo_upload's page-locked ring

// ---- the small levels of a few images in one launch ------------------------------------------------------------
// A level of a single 640x480 frame is a launch at its floor (3-4 us for 20-80 thousand pixels), and a level waits
// for the one before it: three such levels were 12 us of a tracked frame's 140.  Here the last PT_LEVELS_MAX or fewer
// levels are built by one launch without any exchange between workgroups: a workgroup owns a PT_TILE x PT_TILE tile of
// the LAST level, brings the part of the source level that tile depends on into LDS (2r + 3 pixels a side per level,
// r the side one level up: 19, 41, 85 for three levels), and computes level after level from LDS into LDS, borders
// reflected on the levels' own coordinates.  A pixel of an intermediate level is written to memory by the one
// workgroup whose tile it lies under (x >> s, y >> s with s the levels in between); what neighbours need of it they
// compute again themselves (1.6 times the pixels on the largest level).  Same arithmetic as pyr_down_rows_kernel.
#define PT_TILE 4
#define PT_LEVELS_MAX 4
#define PT_THREADS 256
struct PyrTailArgs {
    const uint8_t *src;             // the level the first built level is formed from
    int sw, sh;
    int64_t spitch, sstride;        // bytes between its rows / images
    int levels;                     // levels built
    int src_dwords;                 // the source's rows start on 4-byte boundaries and may be read up to their pitch
    uint8_t *dst[PT_LEVELS_MAX];
    int w[PT_LEVELS_MAX], h[PT_LEVELS_MAX];
    int64_t pitch[PT_LEVELS_MAX], stride[PT_LEVELS_MAX];
    int tiles_x, tiles_y;
};

__device__ __forceinline__ int pt_reflect(int i, int n)         // REFLECT_101, |overshoot| < n
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

__global__ __launch_bounds__(PT_THREADS) void pyr_tail_kernel(PyrTailArgs a)
{
    extern __shared__ uint8_t pt_lds[];
    const int tile = blockIdx.x % (a.tiles_x * a.tiles_y), image = blockIdx.x / (a.tiles_x * a.tiles_y);
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int last = a.levels - 1;
    // the rectangle of every level this tile depends on, clipped to the level: lo[k] .. hi[k] (k = -1: the source)
    int xlo[PT_LEVELS_MAX + 1], xhi[PT_LEVELS_MAX + 1], ylo[PT_LEVELS_MAX + 1], yhi[PT_LEVELS_MAX + 1];
    xlo[last + 1] = tx * PT_TILE; xhi[last + 1] = min(a.w[last], (tx + 1) * PT_TILE) - 1;
    ylo[last + 1] = ty * PT_TILE; yhi[last + 1] = min(a.h[last], (ty + 1) * PT_TILE) - 1;
#pragma unroll
    for (int k = PT_LEVELS_MAX - 1; k >= 0; --k) {
        if (k > last) continue;
        const int wk = k == 0 ? a.sw : a.w[k - 1], hk = k == 0 ? a.sh : a.h[k - 1];     // the level below level k's output
        xlo[k] = max(0, 2 * xlo[k + 1] - 2); xhi[k] = min(wk - 1, 2 * xhi[k + 1] + 2);
        ylo[k] = max(0, 2 * ylo[k + 1] - 2); yhi[k] = min(hk - 1, 2 * yhi[k + 1] + 2);
    }
    // two LDS areas used in turn; the source rectangle is the largest
    const int side0 = ((PT_TILE + 3) << PT_LEVELS_MAX) - 3;
    uint8_t *cur = pt_lds, *nxt = pt_lds + (side0 + 8) * side0;               // rows of the source rounded out to dwords
    // The source rectangle was written by the launch before this one, on other XCDs: every read is a trip to memory,
    // and a thread's reads must all be in flight together.  Rows of a library-owned level start on 64-byte
    // boundaries: the rectangle comes in as whole dwords (eight or so per thread instead of twenty-eight bytes), its
    // left edge moved down to a multiple of four.  A caller's level 0 (any pitch) is read byte by byte.
    const int x_origin = a.src_dwords ? (xlo[0] & ~3) : xlo[0];
    const int src_row = a.src_dwords ? ((xhi[0] - x_origin) / 4 + 1) * 4 : xhi[0] - xlo[0] + 1;      // bytes per row in LDS
    {
        const int rh = yhi[0] - ylo[0] + 1;
        const uint8_t *g = a.src + image * a.sstride;
        if (a.src_dwords) {
            const int rd = src_row / 4, n = rd * rh;
            for (int i0 = threadIdx.x; i0 < n; i0 += 16 * PT_THREADS) {
                uint32_t v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * PT_THREADS;
                    const int y = i / rd, x = i - y * rd;
                    v[u] = i < n ? *(const uint32_t *)(g + (int64_t)(ylo[0] + y) * a.spitch + x_origin + 4 * x) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * PT_THREADS;
                    if (i < n) ((uint32_t *)cur)[i] = v[u];
                }
            }
        } else {
            const int n = src_row * rh;
            for (int i0 = threadIdx.x; i0 < n; i0 += 16 * PT_THREADS) {
                uint8_t v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * PT_THREADS;
                    const int y = i / src_row, x = i - y * src_row;
                    v[u] = i < n ? g[(int64_t)(ylo[0] + y) * a.spitch + x_origin + x] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * PT_THREADS;
                    if (i < n) cur[i] = v[u];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PT_LEVELS_MAX; ++k) {
        if (k > last) break;
        const int sw = k == 0 ? a.sw : a.w[k - 1], sh = k == 0 ? a.sh : a.h[k - 1];
        const int srw = k == 0 ? src_row : xhi[k] - xlo[k] + 1;                 // row length of the rectangle in `cur`
        const int sx0 = k == 0 ? x_origin : xlo[k];                             // its first column
        const int rw = xhi[k + 1] - xlo[k + 1] + 1, rh = yhi[k + 1] - ylo[k + 1] + 1;
        const int shift = last - k;
        uint8_t *g = a.dst[k] + image * a.stride[k];
        for (int i = threadIdx.x; i < rw * rh; i += PT_THREADS) {
            const int ry = i / rw, rx = i - ry * rw;
            const int x = xlo[k + 1] + rx, y = ylo[k + 1] + ry;
            int cx[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) cx[t] = pt_reflect(2 * x - 2 + t, sw) - sx0;
            int v = 0;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const uint8_t *row = cur + (pt_reflect(2 * y - 2 + r, sh) - ylo[k]) * srw;
                const int hsum = row[cx[2]] * 6 + (row[cx[1]] + row[cx[3]]) * 4 + row[cx[0]] + row[cx[4]];
                v += hsum * (r == 2 ? 6 : (r == 1 || r == 3) ? 4 : 1);
            }
            const uint8_t out = (uint8_t)((v + 128) >> 8);
            nxt[i] = out;
            if ((x >> shift) / PT_TILE == tx && (y >> shift) / PT_TILE == ty) g[(int64_t)y * a.pitch[k] + x] = out;
        }
        __syncthreads();
        uint8_t *t = cur; cur = nxt; nxt = t;
    }
}

// Levels first .. max_level by pyr_tail_kernel, when that pays: a handful of images whose level `first` is small.
static bool pyr_tail_applies(const vo_pyr *pyr, int first)
{
    const int levels = pyr->max_level - first + 1;
    return levels >= 2 && levels <= PT_LEVELS_MAX && pyr->batch <= 4 && (int64_t)pyr->w[first] * pyr->h[first] <= 65536;
}

static int pyr_launch_tail(vo_ctx *ctx, vo_pyr *pyr, int first)
{
    PyrTailArgs a;
    a.src = pyr->lvl[first - 1]; a.sw = pyr->w[first - 1]; a.sh = pyr->h[first - 1];
    a.spitch = pyr->pitch[first - 1]; a.sstride = pyr->img_stride[first - 1];
    a.levels = pyr->max_level - first + 1;
    a.src_dwords = first >= 2;      // a level of the pyramid's own pool (pitch a multiple of 64); level 0 may be the caller's
    for (int k = 0; k < PT_LEVELS_MAX; ++k) {
        const int l = first + (k < a.levels ? k : a.levels - 1);
        a.dst[k] = pyr->lvl[l]; a.w[k] = pyr->w[l]; a.h[k] = pyr->h[l]; a.pitch[k] = pyr->pitch[l]; a.stride[k] = pyr->img_stride[l];
    }
    a.tiles_x = (a.w[a.levels - 1] + PT_TILE - 1) / PT_TILE;
    a.tiles_y = (a.h[a.levels - 1] + PT_TILE - 1) / PT_TILE;
    const int side0 = ((PT_TILE + 3) << PT_LEVELS_MAX) - 3, side1 = ((PT_TILE + 3) << (PT_LEVELS_MAX - 1)) - 3;
    const size_t lds = (size_t)(side0 + 8) * side0 + (size_t)side1 * side1;
    static_assert((size_t)(((PT_TILE + 3) << PT_LEVELS_MAX) - 3) * (((PT_TILE + 3) << PT_LEVELS_MAX) - 3) * 5 / 4 < 64 * 1024,
                  "the two areas fit the LDS a launch gets by default");
    hipLaunchKernelGGL(pyr_tail_kernel, dim3((unsigned)(a.tiles_x * a.tiles_y * pyr->batch)), dim3(PT_THREADS), lds, ctx->stream, a);
    return VO_OK;
}

extern "C" int vo_pyr_build(vo_ctx *ctx, vo_pyr *pyr, const uint8_t *img, int64_t row_stride, int64_t img_stride, int mem)
{
    VO_REQUIRE(ctx && pyr && img, "vo_pyr_build: NULL argument");
    VO_REQUIRE(pyr->ctx == ctx, "vo_pyr_build: pyramid belongs to another context");
    VO_REQUIRE(row_stride >= pyr->w[0], "vo_pyr_build: row_stride %lld < width %d", (long long)row_stride, pyr->w[0]);
    VO_REQUIRE(pyr->batch == 1 || img_stride >= row_stride * pyr->h[0], "vo_pyr_build: img_stride %lld too small", (long long)img_stride);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_pyr_build: bad mem flag %d", mem);
    // the tracker forms row offsets with a 24-bit multiply and 32-bit sums
    VO_REQUIRE(row_stride < (1 << 24) && (int64_t)pyr->h[0] * row_stride < (1ll << 32),
               "vo_pyr_build: row_stride %lld with %d rows exceeds the tracker's addressing (row_stride < 16 MiB, image < 4 GiB)",
               (long long)row_stride, pyr->h[0]);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    if (mem == VO_MEM_HOST) {
        int rc = pyr_own_level0(ctx, pyr);
        if (rc) return rc;
        const size_t img_bytes = (size_t)pyr->w[0] * pyr->h[0];
        if (img_bytes <= PYR_STAGED_MAX_BYTES) {
            // Frames of the drop-in calls pass through the context's page-locked ring (vo_upload): the caller's buffer is
            // free when this returns and nothing waits for the device -- the upload and the level kernels queue on the
            // stream ahead of whatever uses the pyramid.
            for (int b = 0; b < pyr->batch; ++b)
                VO_UPLOAD_2D(ctx, pyr->own0 + b * img_bytes, img + (size_t)b * img_stride, row_stride, pyr->w[0], pyr->h[0]);
            return pyr_launch_levels(ctx, pyr);
        }
        if (row_stride == pyr->w[0] && (pyr->batch == 1 || img_stride == (int64_t)img_bytes)) {
            // tight input: one linear copy (2-D copies from pageable memory go row by row)
            VO_HIP_CHECK(hipMemcpyAsync(pyr->own0, img, img_bytes * pyr->batch, hipMemcpyHostToDevice, ctx->stream));
        } else {
            for (int b = 0; b < pyr->batch; ++b)
                VO_HIP_CHECK(hipMemcpy2DAsync(pyr->own0 + b * img_bytes, pyr->w[0], img + (size_t)b * img_stride, row_stride,
                                              pyr->w[0], pyr->h[0], hipMemcpyHostToDevice, ctx->stream));
        }
    } else {
        pyr->lvl[0] = const_cast<uint8_t *>(img);
        pyr->pitch[0] = row_stride;
        pyr->img_stride[0] = img_stride;
    }
    int rc = pyr_launch_levels(ctx, pyr);
    if (rc) return rc;
    if (mem == VO_MEM_HOST) VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// Level 0 in a tight library-owned buffer (+ slack for 4-byte row reads): where host images are uploaded to.
static int pyr_own_level0(vo_ctx *ctx, vo_pyr *pyr)
{
    const size_t need = (size_t)pyr->w[0] * pyr->h[0] * pyr->batch + 256;
    if (pyr->own0_bytes < need) {
        if (pyr->own0) { VO_HIP_CHECK(hipStreamSynchronize(ctx->stream)); VO_HIP_CHECK(hipFree(pyr->own0)); pyr->own0 = nullptr; }
        hipError_t e = hipMalloc((void **)&pyr->own0, need);
        if (e != hipSuccess) { vo_set_error("vo_pyr_build: device allocation of %zu bytes failed", need); return VO_ENOMEM; }
        pyr->own0_bytes = need;
    }
    pyr->lvl[0] = pyr->own0;
    pyr->pitch[0] = pyr->w[0];
    pyr->img_stride[0] = (int64_t)pyr->w[0] * pyr->h[0];
    return VO_OK;
}

// One host image into a one-image pyramid through caller-provided page-locked memory (w * h bytes at `pin`, which
// must stay untouched until the stream has run the copy): nothing here waits for the device.
int vo_pyr_build_pinned(vo_ctx *ctx, vo_pyr *pyr, const uint8_t *img, int64_t row_stride, uint8_t *pin)
{
    VO_REQUIRE(pyr->ctx == ctx && pyr->batch == 1, "vo_pyr_build: a one-image pyramid of this context is needed");
    VO_REQUIRE(row_stride >= pyr->w[0], "vo_pyr_build: row_stride %lld < width %d", (long long)row_stride, pyr->w[0]);
    int rc = pyr_own_level0(ctx, pyr);
    if (rc) return rc;
    const size_t w = (size_t)pyr->w[0], h = (size_t)pyr->h[0];
    if ((size_t)row_stride == w) memcpy(pin, img, w * h);
    else for (size_t y = 0; y < h; ++y) memcpy(pin + y * w, img + y * (size_t)row_stride, w);
    // by a kernel that reads the page-locked memory: a copy engine's hand-over to the compute queue would cost more than
    // the frame takes over the bus (pin is 16-byte aligned with room for w * h rounded up to 16, as level 0 is)
    if ((rc = vo_copy_pinned_wide_async(ctx, pyr->own0, pin, w * h))) return rc;
    return pyr_launch_levels(ctx, pyr);
}

// Levels 1 .. max_level from level 0, asynchronous.
static int pyr_launch_levels(vo_ctx *ctx, vo_pyr *pyr)
{
    for (int l = 1; l <= pyr->max_level; ++l) {
        if (pyr_tail_applies(pyr, l)) {
            int rc = pyr_launch_tail(ctx, pyr, l);
            if (rc) return rc;
            break;
        }
        const int groups = (pyr->w[l] + 3) / 4;
        // eight output rows per lane unless that leaves fewer than ~2 wavefronts per SIMD
        const bool tall = (int64_t)groups * ((pyr->h[l] + 7) / 8) * pyr->batch >= PR_TALL_MIN_LANES;
        const int ro = tall ? 8 : 4, strips = (pyr->h[l] + ro - 1) / ro;
        const int wgs = (groups * strips + 255) / 256;
        const int64_t blocks = (int64_t)wgs * pyr->batch;
        VO_REQUIRE(blocks <= 0x7fffffff, "vo_pyr_build: %lld workgroups exceed the grid limit", (long long)blocks);
        if (tall)
            hipLaunchKernelGGL(pyr_down_rows_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, pyr->lvl[l - 1],
                               pyr->w[l - 1], pyr->h[l - 1], pyr->pitch[l - 1], pyr->img_stride[l - 1], pyr->lvl[l], pyr->w[l],
                               pyr->h[l], pyr->pitch[l], pyr->img_stride[l], groups, strips, wgs, pyr->batch);
        else
            hipLaunchKernelGGL(pyr_down_rows_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, pyr->lvl[l - 1],
                               pyr->w[l - 1], pyr->h[l - 1], pyr->pitch[l - 1], pyr->img_stride[l - 1], pyr->lvl[l], pyr->w[l],
                               pyr->h[l], pyr->pitch[l], pyr->img_stride[l], groups, strips, wgs, pyr->batch);
    }
    VO_HIP_CHECK(hipGetLastError());
    pyr->built = true;
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

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_pyramid(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)pyr_down_rows_kernel<8>));
    return VO_OK;
}
