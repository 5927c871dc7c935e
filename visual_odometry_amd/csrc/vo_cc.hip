// Connected-component labelling of a binary mask with per-component statistics, and the whole
// corner pipeline of FeatureExtractor.harris built on it.
//
// Replaces cv.connectedComponentsWithStats(dst) of FeatureExtractor.harris
// (vo_pipeline/featureExtraction.py:45): 8-connectivity, label 0 = background, labels 1..n in the order
// in which a sequential labeller first meets the components, and per label the area and the centroid
// (sum of pixel coordinates / area, float64).  Two numberings are offered:
//   order 0  raster order of a component's first pixel (two-pass labellers; scipy.ndimage.label),
//   order 1  raster order of a component's first 2x2 block (rows 2r, 2r+1; columns 2c, 2c+1): the order
//            of the block-based labeller OpenCV >= 4.5.2 uses for 8-connectivity.
//
// Method: union-find over pixel indices with atomicMin links (a parent is always a smaller index, so a
// tree's root is the component's smallest index).  The index of pixel (x, y) is y*w + x for order 0 and the
// block-major index 4*((y/2)*ceil(w/2) + x/2) + 2*(y%2) + x%2 for order 1, so that in both cases "smallest
// index" means "met first" and the label of a component is 1 + the number of roots with a smaller index --
// a prefix sum over root flags.  Passes (all HBM/L2-bound integer work on a few bytes per pixel):
//   init     a wavefront takes 64 pixels of a row; every foreground pixel links to the start of its
//            horizontal run inside the segment (one ballot)
//   merge    links across segment borders and to the row above, one union per adjacency of two runs
//   flatten  every pixel points at its root; roots are counted per 1024 indices
//   scan     exclusive prefix of the per-block root counts (one workgroup)
//   rank     label of each root
//   stats    labels out; area / coordinate sums with integer atomics, aggregated per run, per wavefront
//            and per workgroup before they leave the CU (a blob that covers most of the image would
//            otherwise queue a third of a million atomics on three addresses)
//   finish   background row from the totals, centroids = sums / area in float64
// Integer sums make areas and centroids independent of the order of the atomics: results are
// bit-identical to the sequential checker.
#include "vo_internal.h"

#define CC_BLK 1024

struct CcGeom {
    int w, h, wb, order;
    int n_idx;  // size of the index space
};

__device__ __forceinline__ int cc_index(const CcGeom &g, int x, int y)
{
    return g.order == 0 ? y * g.w + x : ((((y >> 1) * g.wb + (x >> 1)) << 2) | ((y & 1) << 1) | (x & 1));
}

__device__ __forceinline__ int cc_find(const int *L, int i)
{
    int p = L[i];
    while (p != i) { i = p; p = L[i]; }
    return i;
}

// Join the sets of a and b.  Reads may be stale (another wavefront may have linked a root meanwhile):
// the atomicMin on the supposed root tells, and the loop follows the link it finds.
__device__ void cc_union(int *L, int a, int b)
{
    for (;;) {
        a = cc_find(L, a);
        b = cc_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + a, b);
        if (old == a) return;   // a was a root and now hangs under b
        a = old;                // a had a parent already (smaller than a): join that with b
    }
}

__device__ __forceinline__ bool bit_at(unsigned long long m, int i) { return (m >> i) & 1ull; }

__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t *__restrict__ mask, int64_t stride, CcGeom g, int *__restrict__ L)
{
    const int lane = threadIdx.x & 63, y = blockIdx.y * 4 + (threadIdx.x >> 6), x0 = blockIdx.x * 64, x = x0 + lane;
    const bool fg = x < g.w && y < g.h && mask[(int64_t)y * stride + x] != 0;
    const unsigned long long bits = __ballot(fg);
    if (!fg) {
        if (x < g.w && y < g.h) L[cc_index(g, x, y)] = -1;
        return;
    }
    const unsigned long long below = ~bits & ((1ull << lane) - 1ull);
    const int s = below ? 64 - __clzll((long long)below) : 0;   // first lane of this pixel's run
    L[cc_index(g, x, y)] = cc_index(g, x0 + s, y);
}

__global__ __launch_bounds__(256) void cc_merge_kernel(const uint8_t *__restrict__ mask, int64_t stride, CcGeom g, int *__restrict__ L)
{
    const int lane = threadIdx.x & 63, y = blockIdx.y * 4 + (threadIdx.x >> 6), x0 = blockIdx.x * 64, x = x0 + lane;
    if (y >= g.h) return;                                   // wave-uniform
    const uint8_t *row = mask + (int64_t)y * stride;
    const bool fg = x < g.w && row[x] != 0;
    const bool upf = y > 0 && x < g.w && row[x - stride] != 0;
    const unsigned long long bits = __ballot(fg), ubits = __ballot(upf);
    if (bits == 0) return;
    // the pixels beside the segment, in this row and the row above
    const bool cl = x0 > 0 && row[x0 - 1] != 0, cr = x0 + 64 < g.w && row[x0 + 64] != 0;
    const bool ul = y > 0 && x0 > 0 && row[x0 - 1 - stride] != 0, ur = y > 0 && x0 + 64 < g.w && row[x0 + 64 - stride] != 0;
    if (!fg) return;
    const bool w_fg = lane > 0 ? bit_at(bits, lane - 1) : cl;
    const bool e_fg = lane < 63 ? bit_at(bits, lane + 1) : cr;
    const bool upl = lane > 0 ? bit_at(ubits, lane - 1) : ul;
    const bool upr = lane < 63 ? bit_at(ubits, lane + 1) : ur;
    const int p = cc_index(g, x, y);
    if (lane == 0 && w_fg) cc_union(L, p, cc_index(g, x - 1, y));     // the run continues from the segment to the left
    if (y == 0) return;
    // One union per adjacency of two runs: where the pixel above is set, the leftmost pixel of the overlap makes
    // the link (a pixel whose left neighbour and upper-left neighbour are both set leaves it to that neighbour);
    // a diagonal neighbour counts only when the pixel above is clear, and is left to the horizontal neighbour
    // below it when that one is set.
    if (upf) {
        if (!(w_fg && upl)) cc_union(L, p, cc_index(g, x, y - 1));
    } else {
        if (upl && !w_fg) cc_union(L, p, cc_index(g, x - 1, y - 1));
        if (upr && !e_fg) cc_union(L, p, cc_index(g, x + 1, y - 1));
    }
}

// in-block exclusive prefix of a flag, and the block's total in every thread
__device__ __forceinline__ int block_prefix(bool flag, int &total)
{
    __shared__ int wave_cnt[CC_BLK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();                    // the array may still be read from a previous call
    if (lane == 0) wave_cnt[wv] = __popcll(b);
    __syncthreads();
    int before = 0, all = 0;
    for (int k = 0; k < CC_BLK / 64; ++k) {
        const int c = wave_cnt[k];
        before += k < wv ? c : 0;
        all += c;
    }
    total = all;
    return before + within;
}

__global__ __launch_bounds__(CC_BLK) void cc_flatten_kernel(CcGeom g, int *__restrict__ L, int *__restrict__ blockcnt)
{
    const int i = blockIdx.x * CC_BLK + threadIdx.x;
    bool root = false;
    if (i < g.n_idx) {
        const int v = L[i];
        if (v >= 0) {
            const int r = cc_find(L, i);
            if (r != v) L[i] = r;
            root = r == i;
        }
    }
    int total;
    block_prefix(root, total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

// meta[0] = number of components
__global__ __launch_bounds__(CC_BLK) void cc_scan_kernel(const int *__restrict__ blockcnt, int nblk, int *__restrict__ blockoff,
                                                        int *__restrict__ meta)
{
    __shared__ int part[CC_BLK];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += CC_BLK) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? blockcnt[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < CC_BLK; d <<= 1) {             // inclusive scan in place
            const int add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const int carry = carry_s;
        if (i < nblk) blockoff[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == CC_BLK - 1) carry_s = carry + part[CC_BLK - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) meta[0] = carry_s;
}

__global__ __launch_bounds__(CC_BLK) void cc_rank_kernel(CcGeom g, const int *__restrict__ L, const int *__restrict__ blockoff,
                                                        int *__restrict__ rank)
{
    const int i = blockIdx.x * CC_BLK + threadIdx.x;
    const bool root = i < g.n_idx && L[i] == i;
    int total;
    const int before = block_prefix(root, total);
    if (root) rank[i] = blockoff[blockIdx.x] + before + 1;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    const int lo = __shfl_xor((int)(v & 0xffffffffull), m), hi = __shfl_xor((int)(v >> 32), m);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// Block = 16 wavefronts = 16 rows x 64 pixels.
__global__ __launch_bounds__(CC_BLK) void cc_stats_kernel(const uint8_t *__restrict__ mask, int64_t stride, CcGeom g,
                                                         const int *__restrict__ L, const int *__restrict__ rank,
                                                         int32_t *__restrict__ labels, unsigned *__restrict__ area,
                                                         unsigned long long *__restrict__ sx, unsigned long long *__restrict__ sy)
{
    __shared__ int w_comp[16];
    __shared__ unsigned w_area[16];
    __shared__ unsigned long long w_sx[16], w_sy[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.y * 16 + wv, x = blockIdx.x * 64 + lane;
    const bool in = x < g.w && y < g.h;
    const bool fg = in && mask[(int64_t)y * stride + x] != 0;
    int comp = 0;
    if (fg) comp = rank[L[cc_index(g, x, y)]];
    if (in && labels) labels[(int64_t)y * g.w + x] = comp;
    const unsigned long long bits = __ballot(fg);
    // a run's first lane carries the run's sums
    const bool start = fg && !(lane > 0 && bit_at(bits, lane - 1));
    unsigned a = 0;
    unsigned long long vx = 0, vy = 0;
    if (start) {
        const unsigned long long rest = ~(bits >> lane);
        const unsigned len = rest ? (unsigned)__builtin_ctzll(rest) : 64u;     // lane 0 of an all-set segment
        a = len;
        vx = (unsigned long long)len * (unsigned)x + (unsigned long long)len * (len - 1) / 2;
        vy = (unsigned long long)len * (unsigned)y;
    }
    const unsigned long long sm = __ballot(start);
    int my_comp = 0;
    unsigned my_a = 0;
    unsigned long long my_x = 0, my_y = 0;
    if (sm != 0) {
        const int c0 = __shfl(comp, __ffsll((long long)sm) - 1);
        if (__ballot(start && comp != c0) == 0) {
            // one component in this segment: add the runs up inside the wavefront
            for (int m = 32; m >= 1; m >>= 1) {
                a += (unsigned)__shfl_xor((int)a, m);
                vx += shfl_xor_u64(vx, m);
                vy += shfl_xor_u64(vy, m);
            }
            my_comp = c0; my_a = a; my_x = vx; my_y = vy;
        } else if (start) {
            atomicAdd(area + comp, a);
            atomicAdd(sx + comp, vx);
            atomicAdd(sy + comp, vy);
        }
    }
    if (lane == 0) { w_comp[wv] = my_comp; w_area[wv] = my_a; w_sx[wv] = my_x; w_sy[wv] = my_y; }
    __syncthreads();
    if (wv == 0) {
        const int c = lane < 16 ? w_comp[lane] : 0;
        unsigned ba = lane < 16 ? w_area[lane] : 0u;
        unsigned long long bx = lane < 16 ? w_sx[lane] : 0ull, by = lane < 16 ? w_sy[lane] : 0ull;
        const unsigned long long have = __ballot(c != 0);
        if (have != 0) {
            const int c0 = __shfl(c, __ffsll((long long)have) - 1);
            if (__ballot(c != 0 && c != c0) == 0) {
                for (int m = 8; m >= 1; m >>= 1) {
                    ba += (unsigned)__shfl_xor((int)ba, m);
                    bx += shfl_xor_u64(bx, m);
                    by += shfl_xor_u64(by, m);
                }
                if (lane == 0) {
                    atomicAdd(area + c0, ba);
                    atomicAdd(sx + c0, bx);
                    atomicAdd(sy + c0, by);
                }
            } else if (c != 0) {
                atomicAdd(area + c, ba);
                atomicAdd(sx + c, bx);
                atomicAdd(sy + c, by);
            }
        }
    }
}

// Row 0 (background) from the totals; centroids and areas of rows 0 .. min(n, max_labels - 1).
// corners (optional): the same centroids as float32 (x, y) pairs, the input of the sub-pixel refinement.
__global__ __launch_bounds__(CC_BLK) void cc_finish_kernel(CcGeom g, const int *__restrict__ meta, unsigned *__restrict__ area,
                                                          unsigned long long *__restrict__ sx, unsigned long long *__restrict__ sy,
                                                          int max_labels, double *__restrict__ centroids, int32_t *__restrict__ areas_out,
                                                          float *__restrict__ corners, int max_corners)
{
    __shared__ unsigned long long red[3][CC_BLK / 64];
    const int n = meta[0];
    unsigned long long ta = 0, tx = 0, ty = 0;
    for (int k = 1 + threadIdx.x; k <= n; k += CC_BLK) { ta += area[k]; tx += sx[k]; ty += sy[k]; }
    for (int m = 32; m >= 1; m >>= 1) { ta += shfl_xor_u64(ta, m); tx += shfl_xor_u64(tx, m); ty += shfl_xor_u64(ty, m); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ta; red[1][threadIdx.x >> 6] = tx; red[2][threadIdx.x >> 6] = ty; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long A = 0, X = 0, Y = 0;
        for (int k = 0; k < CC_BLK / 64; ++k) { A += red[0][k]; X += red[1][k]; Y += red[2][k]; }
        const unsigned long long W = (unsigned long long)g.w, H = (unsigned long long)g.h;
        area[0] = (unsigned)(W * H - A);
        sx[0] = H * (W * (W - 1) / 2) - X;
        sy[0] = W * (H * (H - 1) / 2) - Y;
    }
    __syncthreads();
    __threadfence_block();
    for (int k = threadIdx.x; k <= n; k += CC_BLK) {
        const unsigned ar = area[k];
        const double d = (double)(ar ? ar : 1u);          // an empty background (all-foreground mask) sits at (0, 0)
        const double cx = (double)sx[k] / d, cy = (double)sy[k] / d;
        if (k < max_labels) {
            if (centroids) { centroids[2 * (int64_t)k] = cx; centroids[2 * (int64_t)k + 1] = cy; }
            if (areas_out) areas_out[k] = (int32_t)ar;
        }
        if (corners && k < max_corners) { corners[2 * (int64_t)k] = (float)cx; corners[2 * (int64_t)k + 1] = (float)cy; }
    }
}

struct CcWork {
    CcGeom g;
    int nblk, cap;
    int *L, *rank, *meta, *blockcnt, *blockoff;
    unsigned *area;
    unsigned long long *sx, *sy;
};

// Reserve the work buffers (context slots 3, 4, 5) and fill the geometry.
static int cc_prepare(vo_ctx *ctx, int height, int width, int order, CcWork &wk)
{
    CcGeom &g = wk.g;
    g.w = width; g.h = height; g.wb = (width + 1) / 2; g.order = order;
    const int hb = (height + 1) / 2;
    const int64_t n_idx = order == 0 ? (int64_t)width * height : 4ll * g.wb * hb;
    VO_REQUIRE(n_idx < (1ll << 31), "connected components: image too large (%d x %d)", width, height);
    g.n_idx = (int)n_idx;
    wk.nblk = (g.n_idx + CC_BLK - 1) / CC_BLK;
    wk.cap = g.wb * hb + 1;                          // isolated pixels on a 2-pixel grid, plus the background
    int rc;
    void *p;
    if ((rc = vo_stage_reserve(ctx, 3, (size_t)g.n_idx * 4, &p))) return rc; wk.L = (int *)p;
    if ((rc = vo_stage_reserve(ctx, 4, (size_t)g.n_idx * 4, &p))) return rc; wk.rank = (int *)p;
    const size_t table = 256 + (size_t)wk.nblk * 8 + 256 + (size_t)wk.cap * 20 + 256;
    if ((rc = vo_stage_reserve(ctx, 5, table, &p))) return rc;
    char *b = (char *)p;
    wk.meta = (int *)b; b += 256;
    wk.sx = (unsigned long long *)b; b += (size_t)wk.cap * 8;
    wk.sy = (unsigned long long *)b; b += (size_t)wk.cap * 8;
    wk.area = (unsigned *)b; b += ((size_t)wk.cap * 4 + 255) / 256 * 256;
    wk.blockcnt = (int *)b; b += (size_t)wk.nblk * 4;
    wk.blockoff = (int *)b;
    return VO_OK;
}

// All passes on device pointers, asynchronous on the context's stream.  Leaves the number of components in wk.meta[0].
static int cc_run(vo_ctx *ctx, CcWork &wk, const uint8_t *d_mask, int64_t stride, int32_t *d_labels, double *d_centroids,
                  int32_t *d_areas, int max_labels, float *d_corners, int max_corners)
{
    const CcGeom &g = wk.g;
    if (g.order != 0) VO_HIP_CHECK(hipMemsetAsync(wk.L, 0xff, (size_t)g.n_idx * 4, ctx->stream));   // padding entries = background
    VO_HIP_CHECK(hipMemsetAsync(wk.meta, 0, 256 + (size_t)wk.cap * 20, ctx->stream));                   // meta, sx, sy, area
    const dim3 rows4((g.w + 63) / 64, (g.h + 3) / 4), rows16((g.w + 63) / 64, (g.h + 15) / 16);
    VO_REQUIRE(rows4.y <= 65535u, "connected components: image too tall");
    hipLaunchKernelGGL(cc_init_kernel, rows4, dim3(256), 0, ctx->stream, d_mask, stride, g, wk.L);
    hipLaunchKernelGGL(cc_merge_kernel, rows4, dim3(256), 0, ctx->stream, d_mask, stride, g, wk.L);
    hipLaunchKernelGGL(cc_flatten_kernel, dim3(wk.nblk), dim3(CC_BLK), 0, ctx->stream, g, wk.L, wk.blockcnt);
    hipLaunchKernelGGL(cc_scan_kernel, dim3(1), dim3(CC_BLK), 0, ctx->stream, (const int *)wk.blockcnt, wk.nblk, wk.blockoff, wk.meta);
    hipLaunchKernelGGL(cc_rank_kernel, dim3(wk.nblk), dim3(CC_BLK), 0, ctx->stream, g, (const int *)wk.L, (const int *)wk.blockoff,
                       wk.rank);
    hipLaunchKernelGGL(cc_stats_kernel, rows16, dim3(CC_BLK), 0, ctx->stream, d_mask, stride, g, (const int *)wk.L,
                       (const int *)wk.rank, d_labels, wk.area, wk.sx, wk.sy);
    hipLaunchKernelGGL(cc_finish_kernel, dim3(1), dim3(CC_BLK), 0, ctx->stream, g, (const int *)wk.meta, wk.area, wk.sx, wk.sy,
                       max_labels, d_centroids, d_areas, d_corners, max_corners);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

extern "C" int vo_connected_components(vo_ctx *ctx, const uint8_t *mask, int height, int width, int64_t row_stride, int order,
                                       int32_t *labels, double *centroids, int32_t *areas, int max_labels, int32_t *n_labels,
                                       int mem)
{
    VO_REQUIRE(ctx && mask && n_labels, "vo_connected_components: NULL argument");
    VO_REQUIRE(height >= 1 && width >= 1 && row_stride >= width, "vo_connected_components: bad shape h=%d w=%d stride=%lld", height,
               width, (long long)row_stride);
    VO_REQUIRE(order == 0 || order == 1, "vo_connected_components: order %d (0 = first pixel, 1 = first 2x2 block)", order);
    VO_REQUIRE(max_labels >= 0 && (max_labels == 0 || centroids || areas), "vo_connected_components: max_labels=%d without outputs", max_labels);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_connected_components: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    CcWork wk;
    int rc = cc_prepare(ctx, height, width, order, wk);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    const uint8_t *d_mask = mask;
    int32_t *d_labels = labels, *d_areas = areas;
    double *d_cent = centroids;
    int64_t pitch = row_stride;
    if (max_labels > wk.cap) max_labels = wk.cap;
    if (mem == VO_MEM_HOST) {
        void *p;
        if ((rc = vo_stage_reserve(ctx, 0, npix, &p))) return rc; d_mask = (const uint8_t *)p;
        if (labels) { if ((rc = vo_stage_reserve(ctx, 1, npix * 4, &p))) return rc; d_labels = (int32_t *)p; }
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)max_labels * 20 + 256, &p))) return rc;
        d_cent = centroids ? (double *)p : nullptr;
        d_areas = areas ? (int32_t *)((char *)p + (size_t)max_labels * 16) : nullptr;
        if (row_stride == width)
            VO_UPLOAD(ctx, (void *)d_mask, mask, npix);
        else
            VO_UPLOAD_2D(ctx, (void *)d_mask, mask, row_stride, width, height);
        pitch = width;
    }
    if ((rc = cc_run(ctx, wk, d_mask, pitch, d_labels, d_cent, d_areas, max_labels, nullptr, 0))) return rc;
    int n_comp = 0;
    VO_RESULT(ctx, &n_comp, wk.meta, 4);
    VO_RESULTS_FINISH(ctx);
    *n_labels = n_comp + 1;
    if (mem == VO_MEM_HOST) {
        const int rows = *n_labels < max_labels ? *n_labels : max_labels;
        if (labels) VO_RESULT(ctx, labels, d_labels, npix * 4);
        if (centroids && rows) VO_RESULT(ctx, centroids, d_cent, (size_t)rows * 16);
        if (areas && rows) VO_RESULT(ctx, areas, d_areas, (size_t)rows * 4);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

// Steps of the corner pipeline that live in other translation units (device pointers, asynchronous).
int vo_harris_mask_device(vo_ctx *ctx, const uint8_t *d_img, int height, int width, int64_t pitch, float k, int mode,
                          double rel_thresh, uint8_t *d_mask);
int vo_corner_subpix_device(vo_ctx *ctx, const uint8_t *d_img, int height, int width, int64_t pitch, float *d_corners,
                            int max_count, const int *d_count_minus_1, int win_w, int win_h, int max_iters, double eps);

extern "C" int vo_harris_corners(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, float k,
                                 double rel_thresh, int order, int win_w, int win_h, int max_iters, double eps, float *corners,
                                 int max_corners, int32_t *n_corners, int mem)
{
    VO_REQUIRE(ctx && img && corners && n_corners, "vo_harris_corners: NULL argument");
    VO_REQUIRE(height >= 3 && width >= 3 && row_stride >= width, "vo_harris_corners: bad shape h=%d w=%d", height, width);
    VO_REQUIRE(order == 0 || order == 1, "vo_harris_corners: order %d", order);
    VO_REQUIRE(max_corners >= 1, "vo_harris_corners: max_corners=%d", max_corners);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_harris_corners: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    CcWork wk;
    int rc = cc_prepare(ctx, height, width, order, wk);
    if (rc) return rc;
    if (max_corners > wk.cap) max_corners = wk.cap;
    const size_t npix = (size_t)width * height;
    void *p;
    const uint8_t *d_img = img;
    int64_t pitch = row_stride;
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 0, npix, &p))) return rc; d_img = (const uint8_t *)p;
        if (row_stride == width)
            VO_UPLOAD(ctx, (void *)d_img, img, npix);
        else
            VO_UPLOAD_2D(ctx, (void *)d_img, img, row_stride, width, height);
        pitch = width;
    }
    if ((rc = vo_stage_reserve(ctx, 2, npix, &p))) return rc;
    uint8_t *d_mask = (uint8_t *)p;
    float *d_corners = corners;
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 7, (size_t)max_corners * 8, &p))) return rc;
        d_corners = (float *)p;
    }
    if ((rc = vo_harris_mask_device(ctx, d_img, height, width, pitch, k, 0, rel_thresh, d_mask))) return rc;
    if ((rc = cc_run(ctx, wk, d_mask, width, nullptr, nullptr, nullptr, 0, d_corners, max_corners))) return rc;
    // the refinement runs over the labels found: meta[0] = n components -> n + 1 corners (background first)
    if ((rc = vo_corner_subpix_device(ctx, d_img, height, width, pitch, d_corners, max_corners, wk.meta, win_w, win_h, max_iters, eps)))
        return rc;
    int n_comp = 0;
    VO_RESULT(ctx, &n_comp, wk.meta, 4);
    if (mem == VO_MEM_HOST)     // the count is not known yet: bring the whole buffer back in the same round trip
        VO_RESULT(ctx, corners, d_corners, (size_t)max_corners * 8);
    VO_RESULTS_FINISH(ctx);
    *n_corners = n_comp + 1;
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_cc(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)cc_init_kernel));
    return VO_OK;
}
