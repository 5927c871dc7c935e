// SIFT detect + describe: replaces cv.SIFT_create(0, 4, 0.03, 10, 1.6).detectAndCompute(img, None) of
// FeatureExtractor.get_kp (vo_pipeline/featureExtraction.py:20-35), which the bootstrap calls on both images
// (vo_pipeline/bootstrap.py:179-181).
//
// Algorithm: OpenCV 4.5.4's SIFT in its scalar form -- image doubled (bilinear) and blurred to sigma, Gaussian
// scale space of layers + 3 images per octave, their differences, 26-neighbour extrema refined by Newton steps
// with the contrast and edge tests, 36-bin orientation histograms (one keypoint per peak >= 0.8 max), keypoints
// sorted and de-duplicated, 4x4x8 trilinear descriptors clamped at 0.2 and stored as saturated bytes in float32.
// Every float expression keeps the operation order of the scalar algorithm (no contraction; exp and 2^x by a
// fixed polynomial instead of a library call), so keypoints and descriptors equal a scalar evaluation bit for bit.
//
// Kernels (all HBM/L2-bound float32 streaming or small per-keypoint work; no contraction that would suit MFMA):
//   sift_upsample_kernel     u8 -> float32, 2x bilinear
//   sift_blur_rows/cols      separable Gaussian, REFLECT_101; rows add their taps left to right, columns in
//                            symmetric pairs (the order of OpenCV's float row and column filters); the column
//                            pass also writes the difference from the layer below
//   sift_halve_kernel        next octave's base = every second pixel
//   sift_extrema_kernel      26-neighbour test over the four middle layers; candidates appended with one atomic
//   sift_refine_kernel       one wavefront per candidate: Newton refinement (every lane, same values), orientation
//                            histogram with lane b adding bin b's samples in raster order (deterministic float sums)
//   sift_descriptor_kernel   four keypoints per wavefront: sixteen lanes form the samples of sixteen window positions at
//                            a time, compacted in raster order; lane c owns inner cell c of the histogram (only the
//                            4x4 inner cells of the 6x6x10 histogram reach the descriptor) and adds the samples that
//                            touch it in that order
//   sift_tail_octaves_kernel the small octaves at the end of the scale space, all layers, in one launch
//   sift_join_lists_kernel   keypoint lists end to end; forms every keypoint's descriptor job on the way
//   sift_order_jobs_kernel   jobs by window radius, largest first (workgroup b places list b)
//   sift_gather_rows_kernel  descriptor rows of the surviving keypoints in output order
// Sorting / de-duplication of the keypoints runs on the host while the device computes the descriptors of all raw
// keypoints; the rows of the survivors are gathered afterwards.
#include "vo_internal.h"
#include <math.h>
#include <algorithm>
#include <vector>

#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_ORI_SIG_FCTR 1.5f
#define SIFT_ORI_RADIUS (3 * SIFT_ORI_SIG_FCTR)
#define SIFT_ORI_PEAK_RATIO 0.8f
#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_DESCR_SCL_FCTR 3.f
#define SIFT_DESCR_MAG_THR 0.2f
#define SIFT_INT_DESCR_FCTR 512.f
#define SIFT_INIT_SIGMA 0.5f
#define SIFT_FLT_EPSILON 1.1920928955078125e-07f
#define SIFT_MAX_OCTAVES 16
#define SIFT_MAX_LAYERS 8
#define SIFT_MAX_TAPS 63

struct SiftKernel { int n; float k[SIFT_MAX_TAPS + 1]; };
struct SiftCand { int octave, layer, r, c; };
struct SiftOctaves {
    int n_oct, layers;
    int w[SIFT_MAX_OCTAVES], h[SIFT_MAX_OCTAVES];
    size_t gauss_off[SIFT_MAX_OCTAVES], dog_off[SIFT_MAX_OCTAVES];   // float offsets into the pool
    int rmax[SIFT_MAX_OCTAVES];                                      // (int)sqrt(w^2 + h^2): cap of a descriptor window's radius
};
// descriptor job of one keypoint: image, rounded centre, rotation already divided by the histogram width
struct SiftJob { int octave, layer, px, py, radius, out; float ori, cos_t, sin_t; };     // out: descriptor row

__host__ __device__ static inline int sift_floor(float v) { int i = (int)v; return i - (i > v); }

// 2^t: integer part by exponent arithmetic, fraction by a degree-6 polynomial (relative error ~2e-7).
__device__ static inline float sift_exp2(float t)
{
    if (t < -126.f) return 0.f;
    float n = floorf(t), f = t - n;
    float p = 1.530276e-4f;
    p = p * f + 1.339887e-3f;
    p = p * f + 9.618437e-3f;
    p = p * f + 5.550357e-2f;
    p = p * f + 2.402265e-1f;
    p = p * f + 6.931472e-1f;
    p = p * f + 1.0f;
    return ldexpf(p, (int)n);
}
__device__ static inline float sift_exp(float x) { return sift_exp2(x * 1.44269504088896341f); }

// cos and sin of a float angle t (radians, 0 <= t <= 2 pi) as OpenCV's descriptor rotation takes them (cosf / sinf of
// the float product angle * (float)(pi / 180)): evaluated in float64 -- argument reduced by multiples of pi/2 given in
// two parts, fdlibm's kernel polynomials on [-pi/4, pi/4] -- in this fixed operation order and rounded to float.  The
// float64 result is within one unit in its last place, so the float is the correctly rounded one except when the exact
// value lies within ~1e-16 (relative) of a float rounding boundary; written out (instead of calling libm) so that a
// CPU and a GPU evaluate the very same expression.
__host__ __device__ static inline void sift_sincos(float t, float *c, float *s)
{
    const double x = (double)t;
    const int k = (int)(x * 0.63661977236758138 + 0.5);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double r2 = r * r;
    const double sn = r + r * r2 * (-1.66666666666666324348e-01 + r2 * (8.33333333332248946124e-03 + r2 * (-1.98412698298579493134e-04 +
                      r2 * (2.75573137070700676789e-06 + r2 * (-2.50507602534068634195e-08 + r2 * 1.58969099521155010221e-10)))));
    const double cs = 1.0 - 0.5 * r2 + r2 * r2 * (4.16666666666666019037e-02 + r2 * (-1.38888888888741095749e-03 + r2 * (2.48015872894767294178e-05 +
                      r2 * (-2.75573143513906633035e-07 + r2 * (2.08757232129817482790e-09 + r2 * -1.13596475577881948265e-11)))));
    switch (k & 3) {
    case 0: *c = (float)cs; *s = (float)sn; break;
    case 1: *c = (float)-sn; *s = (float)cs; break;
    case 2: *c = (float)-cs; *s = (float)-sn; break;
    default: *c = (float)sn; *s = (float)-cs; break;
    }
}

// OpenCV's fastAtan2 polynomial: degrees in [0, 360)
__device__ static inline float sift_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f,
                p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    // one quotient serves both octants: min / (max + eps) is ay / (ax + eps) where ax >= ay and ax / (ay + eps) elsewhere
    const float ax = fabsf(x), ay = fabsf(y);
    const bool steep = ax < ay;
    const float c = (steep ? ax : ay) / ((steep ? ay : ax) + 2.220446049250313e-16f), c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (steep) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ static inline int sift_round_f(float v) { return (int)rintf(v); }     // to nearest, ties to even

__device__ static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__global__ __launch_bounds__(256) void sift_upsample_kernel(const uint8_t *__restrict__ img, int w, int h, int64_t stride,
                                                           float *__restrict__ dst)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = 2 * w, H = 2 * h;
    if (x >= W || y >= H) return;
    float fy = (float)((y + 0.5) * 0.5 - 0.5);
    int sy = sift_floor(fy);
    fy -= sy;
    if (sy < 0) { sy = 0; fy = 0; }
    if (sy >= h - 1) { sy = h - 1; fy = 0; }
    const int sy1 = sy + 1 < h ? sy + 1 : h - 1;
    float fx = (float)((x + 0.5) * 0.5 - 0.5);
    int sx = sift_floor(fx);
    fx -= sx;
    if (sx < 0) { sx = 0; fx = 0; }
    if (sx >= w - 1) { sx = w - 1; fx = 0; }
    const int sx1 = sx + 1 < w ? sx + 1 : w - 1;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float r0 = (float)img[sy * stride + sx] * a0 + (float)img[sy * stride + sx1] * a1;
    const float r1 = (float)img[sy1 * stride + sx] * a0 + (float)img[sy1 * stride + sx1] * a1;
    dst[(size_t)y * W + x] = r0 * b0 + r1 * b1;
}

// Row pass: a workgroup stages SIFT_ROW_SEG + 2r pixels of one row in LDS (reflected at the ends), a lane adds its
// taps left to right.
#ifndef SIFT_ROW_SEG
#define SIFT_ROW_SEG 256
#endif
__global__ __launch_bounds__(SIFT_ROW_SEG) void sift_blur_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                                     SiftKernel K)
{
    __shared__ float row[SIFT_ROW_SEG + SIFT_MAX_TAPS + 1];
    const int y = blockIdx.y, x0 = blockIdx.x * SIFT_ROW_SEG, r = K.n >> 1;
    const float *s = src + (size_t)y * w;
    for (int i = threadIdx.x; i < SIFT_ROW_SEG + 2 * r; i += SIFT_ROW_SEG) row[i] = s[reflect101(x0 - r + i, w)];
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= w) return;
    float acc = row[threadIdx.x] * K.k[0];
    for (int j = 1; j < K.n; ++j) acc += row[threadIdx.x + j] * K.k[j];
    dst[(size_t)y * w + x] = acc;
}

// Column pass: a workgroup stages a 64-column strip of SIFT_COL_ROWS + 2r rows in LDS (rows reflected at the image
// border; 64 consecutive floats per row, so global reads are coalesced and LDS reads conflict-free), then every
// lane finishes SIFT_COL_ROWS / 4 outputs of its column from LDS: centre tap first, then symmetric pairs.  Every
// input row is read (SIFT_COL_ROWS + 2r) / SIFT_COL_ROWS times -- from L2 after the first -- instead of 2r + 1
// times.  Short strips win: the pass is bound by the latency of a workgroup's load-then-compute sequence, not by
// bytes, and 8-row strips put four times the workgroups of 32-row ones in flight (whole SIFT call at 1241 x 376:
// 1.73 / 1.53 / 1.43 / 1.40 ms with 64 / 32 / 16 / 8 rows).  With `diff` given the difference from the layer below
// (the image the row pass read) is written on the way.
#ifndef SIFT_COL_ROWS
#define SIFT_COL_ROWS 8
#endif
__global__ __launch_bounds__(256) void sift_blur_cols_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                            SiftKernel K, const float *__restrict__ below, float *__restrict__ diff)
{
    __shared__ float tile[(SIFT_COL_ROWS + SIFT_MAX_TAPS + 1) * 64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + tx, y0 = blockIdx.y * SIFT_COL_ROWS, r = K.n >> 1;
    const int rows = min(SIFT_COL_ROWS, h - y0) + 2 * r;
    if (x < w)
        for (int i = ty; i < rows; i += 4) tile[i * 64 + tx] = src[(size_t)reflect101(y0 - r + i, h) * w + x];
    __syncthreads();
    if (x >= w) return;
    for (int yy = ty; yy < SIFT_COL_ROWS && y0 + yy < h; yy += 4) {
        const float *c = tile + (yy + r) * 64 + tx;
        float acc = K.k[r] * c[0];
        for (int j = 1; j <= r; ++j) acc += K.k[r + j] * (c[j * 64] + c[-j * 64]);
        const size_t at = (size_t)(y0 + yy) * w + x;
        dst[at] = acc;
        if (diff) diff[at] = acc - below[at];        // difference of Gaussians: this layer minus the one it was blurred from
    }
}

__global__ __launch_bounds__(256) void sift_halve_kernel(const float *__restrict__ src, int sw, float *__restrict__ dst, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < w && y < h) dst[(size_t)y * w + x] = src[(size_t)(2 * y) * sw + 2 * x];
}

// The small octaves at the end of the scale space (a 1241 x 376 frame: 77 x 23 pixels and down) in one launch: a
// launch per blur pass costs more than the pass itself there.  One workgroup walks the octaves and their layers in
// order -- the first image of an octave is every second pixel of the previous octave's layer `layers`, layer i the
// blur of layer i - 1 -- with the row and column sums in the order of sift_blur_rows_kernel / sift_blur_cols_kernel,
// and writes the Gaussian images and their differences.  The current layer lives in LDS with its rows extended by
// the reflected pixels at either end, the row pass's output with the reflected rows above and below, so a tap is
// one LDS read, one multiplication and one addition; a thread owns the same (at most two) pixels throughout.
#define SIFT_TAIL_PIXELS 2048       // an octave of at most this many pixels ...
#define SIFT_TAIL_PADDED 6144       // ... and at most this many with the reflected borders on
// A barrier between phases that exchange data through LDS only: it does not wait for global stores in flight.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
struct SiftTail { int first, halo; SiftKernel K[SIFT_MAX_LAYERS + 3]; };      // K[i]: the blur that makes layer i from layer i - 1
__global__ __launch_bounds__(1024) void sift_tail_octaves_kernel(float *__restrict__ pool, SiftOctaves O, SiftTail T)
{
    __shared__ float buf[2][SIFT_TAIL_PADDED];
    __shared__ float tmp[SIFT_TAIL_PADDED];
    const int tid = threadIdx.x, layers = O.layers, R = T.halo;
    for (int o = T.first; o < O.n_oct; ++o) {
        const int w = O.w[o], h = O.h[o], plane = w * h, pw = O.w[o - 1], wp = w + 2 * R;
        float *g = pool + O.gauss_off[o], *dog = pool + O.dog_off[o];
        const float *prev = pool + O.gauss_off[o - 1] + (size_t)pw * O.h[o - 1] * layers;
        float *cur = buf[0], *nxt = buf[1];
        int at[2], att[2];              // this thread's pixels in the row-extended and in the row-added layout
        bool has[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 1024 * q, y = p / w, x = p - y * w;
            has[q] = p < plane;
            at[q] = y * wp + R + x;
            att[q] = (y + R) * w + x;
            if (has[q]) {
                const float v = prev[(size_t)(2 * y) * pw + 2 * x];
                cur[at[q]] = v;
                g[p] = v;
            }
        }
        __syncthreads();
        for (int i = 1; i < layers + 3; ++i) {
            const SiftKernel &K = T.K[i];
            const int r = K.n >> 1;
            for (int e = tid; e < 2 * r * h; e += 1024) {              // r reflected pixels at either end of every row
                const int y = e / (2 * r), k = e - y * 2 * r, x = k < r ? k - r : w + (k - r);
                cur[y * wp + R + x] = cur[y * wp + R + reflect101(x, w)];
            }
            lds_barrier();
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (has[q]) {
                    const float *s = cur + at[q] - r;
                    float acc = s[0] * K.k[0];
                    for (int j = 1; j < K.n; ++j) acc += s[j] * K.k[j];
                    tmp[att[q]] = acc;
                }
            lds_barrier();
            for (int e = tid; e < 2 * r * w; e += 1024) {              // r reflected rows above and below
                const int k = e / w, x = e - k * w, y = k < r ? k - r : h + (k - r);
                tmp[(y + R) * w + x] = tmp[(reflect101(y, h) + R) * w + x];
            }
            lds_barrier();
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (has[q]) {
                    const float *c = tmp + att[q];
                    float acc = K.k[r] * c[0];
                    for (int j = 1; j <= r; ++j) acc += K.k[r + j] * (c[j * w] + c[-j * w]);
                    const int p = tid + 1024 * q;
                    nxt[at[q]] = acc;
                    g[(size_t)plane * i + p] = acc;
                    dog[(size_t)plane * (i - 1) + p] = acc - cur[at[q]];
                }
            lds_barrier();
            float *t = cur; cur = nxt; nxt = t;
        }
        __syncthreads();        // the next octave reads this one's layer `layers` from memory
    }
}

// A workgroup tests 64 x 4 pixels of one layer.  It first stages the 66 x 6 neighbourhood of the layer and of its two
// neighbours in LDS with coalesced reads that are all in flight at once (testing straight from memory made every
// wavefront wait for three dependent rounds of reads: 185 us at the finest octave of a 1241 x 376 frame).
// Candidates and keypoints are appended to SIFT_LISTS lists, each with its own counter (SIFT_LIST_STRIDE ints apart, a
// cache line each) and 1 / SIFT_LISTS of the buffer: appends through one counter are served one after the other
// (about 14 ns each), which for tens of thousands of them was most of the time of the kernels that make them.
#define SIFT_LISTS 64
#define SIFT_LIST_STRIDE 16
// One workgroup tests a tile of EX_ROWS x 64 pixels in ALL layers of the octave: the tile of every difference plane
// (layers + 2 of them, one halo pixel all round) is staged in LDS once and serves the three layers that read it.
// (One layer per workgroup with 4-row tiles read every plane three times and a 6-row halo for 4 rows: 4.6 staged
// values per tested pixel against 1.9 here, 1.5 being the planes themselves.)
#define EX_ROWS 8
#define EX_PITCH 68
__global__ __launch_bounds__(256) void sift_extrema_kernel(const float *__restrict__ dog, size_t plane, int w, int h, int octave,
                                                          int layers, int threshold, SiftCand *__restrict__ cand,
                                                          int *__restrict__ count, int cap)
{
    extern __shared__ float t[];                  // [layers + 2][EX_ROWS + 2][EX_PITCH]
    __shared__ int s_wave[4], s_base;
    const int c0 = SIFT_IMG_BORDER + blockIdx.x * 64, r0 = SIFT_IMG_BORDER + blockIdx.y * EX_ROWS;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int per_plane = (EX_ROWS + 2) * 66;
    for (int i = threadIdx.x; i < (layers + 2) * per_plane; i += 256) {
        const int pl = i / per_plane, rem = i - pl * per_plane, ry = rem / 66, rx = rem - ry * 66;
        const int y = min(r0 - 1 + ry, h - 1), x = min(c0 - 1 + rx, w - 1);       // the border of 5 keeps y - 1, x - 1 inside
        t[(pl * (EX_ROWS + 2) + ry) * EX_PITCH + rx] = dog[plane * pl + (size_t)y * w + x];
    }
    __syncthreads();
    // a thread's pixels: column tx, rows ty and ty + 4, every layer; bit (layer - 1) * 2 + k of `hits` = row ty + 4k found
    const int c = c0 + tx;
    unsigned hits = 0;
    if (c < w - SIFT_IMG_BORDER) {
        for (int layer = 1; layer <= layers; ++layer)
#pragma unroll
            for (int k = 0; k < EX_ROWS / 4; ++k) {
                const int yy = ty + 4 * k;
                if (r0 + yy >= h - SIFT_IMG_BORDER) continue;
                const float *mid = t + (layer * (EX_ROWS + 2) + yy) * EX_PITCH + tx;
                const float val = mid[EX_PITCH + 1];
                if (!(fabsf(val) > (float)threshold)) continue;
                float hi = -__builtin_inff(), lo = __builtin_inff();
#pragma unroll
                for (int pl = -1; pl <= 1; ++pl)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float b = mid[(pl * (EX_ROWS + 2) + dy) * EX_PITCH + dx];
                            hi = fmaxf(hi, b);
                            lo = fminf(lo, b);
                        }
                if ((val > 0 && val >= hi) || (val < 0 && val <= lo)) hits |= 1u << ((layer - 1) * (EX_ROWS / 4) + k);
            }
    }
    // slots: prefix over the lanes of a wavefront, then over the four wavefronts, one global atomic per workgroup on the
    // list the workgroup's position picks
    const int mine = __popc(hits), lane = threadIdx.x & 63;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[ty] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        before += q < ty ? s_wave[q] : 0;
        total += s_wave[q];
    }
    if (total == 0) return;
    const int list = (blockIdx.x + 5 * blockIdx.y + 23 * octave) & (SIFT_LISTS - 1), list_cap = cap / SIFT_LISTS;
    if (threadIdx.x == 0) s_base = atomicAdd(count + list * SIFT_LIST_STRIDE, total);
    __syncthreads();
    int slot = s_base + before + incl - mine;
    while (hits) {
        const int bit = __ffs(hits) - 1;
        hits &= hits - 1;
        const int layer = 1 + bit / (EX_ROWS / 4), k = bit % (EX_ROWS / 4);
        if (slot < list_cap) cand[(size_t)list * list_cap + slot] = SiftCand{octave, layer, r0 + ty + 4 * k, c};
        ++slot;
    }
}

// The keypoint lists put end to end: block b copies list b behind the lists before it; total[0] = the number of
// keypoints, or cap + 1 when a list overflowed.
#define SIFT_JOB_BINS 1024
__device__ static inline int sift_job_bin(int radius) { return radius < SIFT_JOB_BINS - 1 ? radius : SIFT_JOB_BINS - 1; }

// Descriptor job of raw keypoint k (coordinates of the doubled image, octave index from 0), in the float operations
// OpenCV's calcSIFTDescriptor set-up applies to the keypoint it is handed (coordinates and size halved and the
// octave shifted by -1 first, as the detector's last step does).
__device__ static inline SiftJob sift_make_job(const vo_sift_keypoint &k, int row, const SiftOctaves &O)
{
    constexpr int d = SIFT_DESCR_WIDTH;
    const int first_octave = -1;
    const int shifted = (k.octave & ~255) | ((k.octave + first_octave) & 255);
    const float kx = k.x * 0.5f, ky = k.y * 0.5f, ksize = k.size * 0.5f;
    int octave = shifted & 255;
    const int layer = (shifted >> 8) & 255;
    octave = octave < 128 ? octave : (-128 | octave);
    const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
    const float size = ksize * scale, scl = size * 0.5f;
    const float ptx = kx * scale, pty = ky * scale;
    float angle = 360.f - k.angle;
    if (fabsf(angle - 360.f) < SIFT_FLT_EPSILON) angle = 0.f;
    SiftJob jb;
    jb.octave = octave - first_octave;
    jb.out = row;
    jb.layer = layer;
    jb.px = (int)rint((double)ptx);
    jb.py = (int)rint((double)pty);
    jb.ori = angle;
    float cos_t, sin_t;
    sift_sincos(angle * (float)(3.14159265358979323846 / 180.), &cos_t, &sin_t);
    const float hist_width = SIFT_DESCR_SCL_FCTR * scl;
    const int radius = (int)rint((double)(hist_width * 1.4142135623730951f * (d + 1) * 0.5f));
    const int rmax = O.rmax[jb.octave];
    jb.radius = radius < rmax ? radius : rmax;
    jb.cos_t = cos_t / hist_width;
    jb.sin_t = sin_t / hist_width;
    return jb;
}

// Puts the keypoint lists end to end (workgroup b copies list b) and, with `unordered` given, forms every
// keypoint's descriptor job on the way and counts its list's jobs per window radius (row b of radius_hist;
// sift_order_jobs_kernel sorts by it).
__global__ __launch_bounds__(256) void sift_join_lists_kernel(const vo_sift_keypoint *__restrict__ lists, const int *__restrict__ counts,
                                                             int cap, vo_sift_keypoint *__restrict__ joined, int *__restrict__ total,
                                                             SiftOctaves O, SiftJob *__restrict__ unordered, int *__restrict__ radius_hist)
{
    __shared__ int s_off, s_n;
    __shared__ int hist[SIFT_JOB_BINS];
    const int list_cap = cap / SIFT_LISTS;
    for (int b = threadIdx.x; b < SIFT_JOB_BINS; b += 256) hist[b] = 0;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, raw = counts[lane * SIFT_LIST_STRIDE], mine = min(raw, list_cap);
        int upto = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(upto, d);
            if (lane >= d) upto += o;
        }
        const bool over = __ballot(raw > list_cap) != 0;
        if (lane == (int)blockIdx.x) { s_off = upto - mine; s_n = mine; }
        if (blockIdx.x == 0 && lane == 63) total[0] = over ? cap + 1 : upto;
    }
    __syncthreads();
    const vo_sift_keypoint *src = lists + (size_t)blockIdx.x * list_cap;
    for (int i = threadIdx.x; i < s_n; i += 256) {
        const vo_sift_keypoint k = src[i];
        joined[s_off + i] = k;
        if (unordered) {
            const SiftJob jb = sift_make_job(k, s_off + i, O);
            unordered[s_off + i] = jb;
            atomicAdd(&hist[sift_job_bin(jb.radius)], 1);
        }
    }
    if (!unordered) return;
    __syncthreads();
    for (int b = threadIdx.x; b < SIFT_JOB_BINS; b += 256) radius_hist[blockIdx.x * SIFT_JOB_BINS + b] = hist[b];
}

// One wavefront per candidate.  The kernel waits more than it computes (dependent reads of the Newton steps, the
// barrier between forming and adding a chunk's samples), so small chunks -- less LDS, more wavefronts per CU -- win.
#ifndef SIFT_ORI_CHUNK
#define SIFT_ORI_CHUNK 512
#endif
__global__ __launch_bounds__(64) void sift_refine_kernel(const float *__restrict__ pool, SiftOctaves O, const SiftCand *__restrict__ cand,
                                                        const int *__restrict__ n_cand, int cand_cap, float contrast_thr,
                                                        float edge_thr, float sigma, vo_sift_keypoint *__restrict__ out,
                                                        int *__restrict__ n_out, int out_cap)
{
    __shared__ float s_wm[SIFT_ORI_CHUNK];
    __shared__ int s_bin[SIFT_ORI_CHUNK];
    __shared__ float s_th[SIFT_ORI_HIST_BINS + 4];
    const int lane = threadIdx.x;
    // candidate number blockIdx.x of the lists taken one after the other
    const int list_cap = cand_cap / SIFT_LISTS;
    const int mine = min(n_cand[lane * SIFT_LIST_STRIDE], list_cap);
    int upto = mine;                               // inclusive prefix over the 64 lists
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(upto, d);
        if (lane >= d) upto += o;
    }
    const unsigned long long later = __ballot((int)blockIdx.x < upto);
    if (later == 0) return;
    const int list = __ffsll((long long)later) - 1;
    const int within = (int)blockIdx.x - (__shfl(upto, list) - __shfl(mine, list));
    const SiftCand cd = cand[(size_t)list * list_cap + within];
    const int octv = cd.octave, w = O.w[octv], h = O.h[octv], nl = O.layers;
    const size_t plane = (size_t)w * h;
    const float *dogs = pool + O.dog_off[octv];
    int layer = cd.layer, r = cd.r, c = cd.c;
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < SIFT_MAX_INTERP_STEPS; ++i) {
        const float *img = dogs + plane * layer, *prev = img - plane, *next = img + plane;
        const size_t at = (size_t)r * w + c;
        const float dD0 = (img[at + 1] - img[at - 1]) * deriv_scale, dD1 = (img[at + w] - img[at - w]) * deriv_scale,
                    dD2 = (next[at] - prev[at]) * deriv_scale;
        const float v2 = img[at] * 2;
        const float dxx = (img[at + 1] + img[at - 1] - v2) * second, dyy = (img[at + w] + img[at - w] - v2) * second,
                    dss = (next[at] + prev[at] - v2) * second;
        const float dxy = (img[at + w + 1] - img[at + w - 1] - img[at - w + 1] + img[at - w - 1]) * cross;
        const float dxs = (next[at + 1] - next[at - 1] - prev[at + 1] + prev[at - 1]) * cross;
        const float dys = (next[at + w] - next[at - w] - prev[at + w] + prev[at - w]) * cross;
        const float a00 = dxx, a01 = dxy, a02 = dxs, a10 = dxy, a11 = dyy, a12 = dys, a20 = dxs, a21 = dys, a22 = dss;
        float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
        float X0 = 0, X1 = 0, X2 = 0;
        if (d != 0) {
            d = 1 / d;
            X0 = d * (dD0 * (a11 * a22 - a12 * a21) - a01 * (dD1 * a22 - a12 * dD2) + a02 * (dD1 * a21 - a11 * dD2));
            X1 = d * (a00 * (dD1 * a22 - a12 * dD2) - dD0 * (a10 * a22 - a12 * a20) + a02 * (a10 * dD2 - dD1 * a20));
            X2 = d * (a00 * (a11 * dD2 - dD1 * a21) - a01 * (a10 * dD2 - dD1 * a20) + dD0 * (a10 * a21 - a11 * a20));
        }
        xi = -X2; xr = -X1; xc = -X0;
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(2147483647 / 3) || fabsf(xr) > (float)(2147483647 / 3) || fabsf(xc) > (float)(2147483647 / 3)) return;
        c += sift_round_f(xc); r += sift_round_f(xr); layer += sift_round_f(xi);
        if (layer < 1 || layer > nl || c < SIFT_IMG_BORDER || c >= w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER)
            return;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return;
    vo_sift_keypoint kp;
    {
        const float *img = dogs + plane * layer, *prev = img - plane, *next = img + plane;
        const size_t at = (size_t)r * w + c;
        const float dD0 = (img[at + 1] - img[at - 1]) * deriv_scale, dD1 = (img[at + w] - img[at - w]) * deriv_scale,
                    dD2 = (next[at] - prev[at]) * deriv_scale;
        const float t = dD0 * xc + dD1 * xr + dD2 * xi;
        const float contr = img[at] * img_scale + t * 0.5f;
        if (fabsf(contr) * nl < contrast_thr) return;
        const float v2 = img[at] * 2.f;
        const float dxx = (img[at + 1] + img[at - 1] - v2) * second, dyy = (img[at + w] + img[at - w] - v2) * second;
        const float dxy = (img[at + w + 1] - img[at + w - 1] - img[at - w + 1] + img[at - w - 1]) * cross;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) return;
        kp.x = (c + xc) * (1 << octv);
        kp.y = (r + xr) * (1 << octv);
        // cvRound((xi + 0.5) * 255) is a double expression in the scalar code
        kp.octave = octv + (layer << 8) + ((int)rint(((double)xi + 0.5) * 255) << 16);
        kp.size = sigma * sift_exp2((layer + xi) / nl) * (1 << octv) * 2;
        kp.response = fabsf(contr);
    }
    // ---- orientation histogram on the Gaussian image of the keypoint's layer ----
    const float scl_octv = kp.size * 0.5f / (1 << octv);
    const int radius = sift_round_f(SIFT_ORI_RADIUS * scl_octv);
    const float osig = SIFT_ORI_SIG_FCTR * scl_octv, expf_scale = -1.f / (2.f * osig * osig);
    const float *g = pool + O.gauss_off[octv] + plane * layer;
    const int side = 2 * radius + 1, len = side * side;
    const int n = SIFT_ORI_HIST_BINS;
    float acc = 0.f;                                // lane b < 36: bin b
    for (int base = 0; base < len; base += SIFT_ORI_CHUNK) {
        const int m = min(SIFT_ORI_CHUNK, len - base);
        for (int k = lane; k < m; k += 64) {
            const int idx = base + k, i2 = idx / side - radius, j2 = idx % side - radius;
            const int y = r + i2, x = c + j2;
            int bin = -1;
            float wm = 0.f;
            if (!(y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1)) {
                const float dx = g[(size_t)y * w + x + 1] - g[(size_t)y * w + x - 1];
                const float dy = g[(size_t)(y - 1) * w + x] - g[(size_t)(y + 1) * w + x];
                const float wgt = sift_exp((i2 * i2 + j2 * j2) * expf_scale);
                const float ori = sift_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
                bin = sift_round_f((n / 360.f) * ori);
                if (bin >= n) bin -= n;
                if (bin < 0) bin += n;
                wm = wgt * mag;
            }
            s_wm[k] = wm;
            s_bin[k] = bin;
        }
        __syncthreads();
        if (lane < n)
            for (int k = 0; k < m; ++k)
                if (s_bin[k] == lane) acc += s_wm[k];
        __syncthreads();
    }
    if (lane < n) s_th[2 + lane] = acc;
    __syncthreads();
    if (lane == 0) { s_th[0] = s_th[2 + n - 2]; s_th[1] = s_th[2 + n - 1]; s_th[2 + n] = s_th[2]; s_th[2 + n + 1] = s_th[3]; }
    __syncthreads();
    const float *th = s_th + 2;
    const int jl = lane < n ? lane : 0;
    const float hj = (th[jl - 2] + th[jl + 2]) * (1.f / 16.f) + (th[jl - 1] + th[jl + 1]) * (4.f / 16.f) + th[jl] * (6.f / 16.f);
    float omax = hj;                                // lanes >= 36 repeat bin 0: harmless for a maximum
    for (int m2 = 32; m2 >= 1; m2 >>= 1) omax = fmaxf(omax, __shfl_xor(omax, m2));
    const float mag_thr = omax * SIFT_ORI_PEAK_RATIO;
    const float hl = __shfl(hj, lane > 0 ? lane - 1 : n - 1), hr = __shfl(hj, lane < n - 1 ? lane + 1 : 0);
    if (lane < n && hj > hl && hj > hr && hj >= mag_thr) {
        float bin = lane + 0.5f * (hl - hr) / (hl - 2 * hj + hr);
        bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
        kp.angle = 360.f - (360.f / n) * bin;
        if (fabsf(kp.angle - 360.f) < SIFT_FLT_EPSILON) kp.angle = 0.f;
        const int klist = blockIdx.x & (SIFT_LISTS - 1), klist_cap = out_cap / SIFT_LISTS;
        const int slot = atomicAdd(n_out + klist * SIFT_LIST_STRIDE, 1);
        if (slot < klist_cap) out[(size_t)klist * klist_cap + slot] = kp;
    }
}

// Four keypoints per wavefront, sixteen lanes each.  Only the 4x4 inner cells of the 6x6x10 histogram reach the
// descriptor, so a keypoint needs sixteen owners: lane l of a group owns inner cell (l / 4, l % 4) with its ten
// orientation bins (LDS).  A group walks its keypoint's window sixteen positions at a time, keeps the samples that
// fall inside the rotated 4x4 grid in a small buffer in raster order and marks them in a 64-bit word per owner whose
// cell they touch; when the buffer fills every owner adds the samples marked in its word, lowest slot first -- the
// scalar algorithm's order of additions per bin (a sample touches four of the sixteen cells, so an owner that walked
// the whole buffer would skip three of four).
#ifndef SIFT_DESC_BUF
#define SIFT_DESC_BUF 64                 // waiting samples per keypoint (<= 64: one bit each in an owner's word)
#endif
__global__ __launch_bounds__(64) void sift_descriptor_kernel(const float *__restrict__ pool, SiftOctaves O, const SiftJob *__restrict__ jobs,
                                                            int n_jobs, float *__restrict__ desc)
{
    constexpr int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    // LDS per wavefront decides how many of them a CU holds, and with them how much of the gradient reads' latency is
    // hidden: a waiting sample is kept as its four row/column shares and the orientation fraction (the owner splits a
    // share over the two orientation bins with the same two operations the scalar algorithm uses), and the finished
    // descriptor reuses the sample buffer
    __shared__ float hist[4][d * d * (n + 2)];
    __shared__ float s_v[4][SIFT_DESC_BUF > 26 ? SIFT_DESC_BUF : 26][5];
    __shared__ int s_cell[4][SIFT_DESC_BUF];
    __shared__ float s_scale[4][2];
    float (*s_dst)[d * d * n] = reinterpret_cast<float (*)[d * d * n]>(&s_v[0][0][0]);      // 4 x 128 floats <= 4 x 26 x 5
    __shared__ unsigned long long s_touch[4][16];       // per owner: the waiting samples that touch its cell, one bit each
    __shared__ int s_queue[4][64];                      // positions inside the grid, waiting for a lane: (row offset, column offset) as two shorts
    const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15;
    const int job = blockIdx.x * 4 + g;
    const bool has = job < n_jobs;
    const SiftJob jb = jobs[has ? job : n_jobs - 1];
    const int w = O.w[jb.octave], h = O.h[jb.octave];
    const float *img = pool + O.gauss_off[jb.octave] + (size_t)w * h * jb.layer;
    for (int t = gl; t < d * d * (n + 2); t += 16) hist[g][t] = 0.f;
    s_touch[g][gl] = 0ull;
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f);
    const int radius = jb.radius, side = 2 * radius + 1, len = has ? side * side : 0;
    const int max_len = max(max(__shfl(len, 0), __shfl(len, 16)), max(__shfl(len, 32), __shfl(len, 48)));
    const int my_cell = ((gl >> 2) + 1) * (d + 2) + (gl & 3) + 1;      // index of this lane's cell in the 6x6 grid
    float *mine = hist[g] + gl * (n + 2);
    // Only about half of the window's positions fall inside the rotated 4x4 grid, so the positions are tested first
    // (cheap: the rotation and two range checks) and the survivors queued in raster order; the gradient, weight and
    // angle work then takes sixteen queued positions at a time, one per lane, with no lane idle.
    int filled = 0;                                                     // samples waiting in this group's buffer
    int qn = 0, qhead = 0;                                              // positions waiting in this group's queue (64 slots, circular)
    int wi = gl / side, wj = gl - wi * side;                            // row and column of this lane's next window position
    __builtin_amdgcn_wave_barrier();
    for (int base = 0;;) {
        if (base < max_len && __ballot(qn > 31) == 0) {                 // room for 32 more positions in every group's queue
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int idx = base + 16 * sub + gl;
                const int i = wi - radius, j = wj - radius;
                wj += 16;                           // the next position, sixteen further on in raster order
                while (wj >= side) { wj -= side; ++wi; }
                bool ok = false;
                if (idx < len) {
                    const float c_rot = j * jb.cos_t - i * jb.sin_t, r_rot = j * jb.sin_t + i * jb.cos_t;
                    const float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
                    const int r = jb.py + i, c = jb.px + j;
                    ok = rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1;
                }
                const unsigned gm = (unsigned)(__ballot(ok) >> (16 * g)) & 0xffffu;
                if (ok) s_queue[g][(qhead + qn + __popc(gm & ((1u << gl) - 1u))) & 63] = (i & 0xffff) | (int)((unsigned)j << 16);
                qn += __popc(gm);
            }
            base += 32;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool last = base >= len;              // this group's window has been walked
        const int take = qn >= 16 ? 16 : (last ? qn : 0);
        if (gl < take) {
            const int packed = s_queue[g][(qhead + gl) & 63];
            const int i = (int)(short)(packed & 0xffff), j = packed >> 16;
            const float c_rot = j * jb.cos_t - i * jb.sin_t, r_rot = j * jb.sin_t + i * jb.cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = jb.py + i, c = jb.px + j;
            const float dx = img[(size_t)r * w + c + 1] - img[(size_t)r * w + c - 1];
            const float dy = img[(size_t)(r - 1) * w + c] - img[(size_t)(r + 1) * w + c];
            const float wgt = sift_exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (sift_atan2_deg(dy, dx) - jb.ori) * bins_per_rad;
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            const int r0 = sift_floor(rbin), c0 = sift_floor(cbin);
            int o0 = sift_floor(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const int at = filled + gl;             // raster order: the queue's order
            s_cell[g][at] = (((r0 + 1) * (d + 2) + c0 + 1) << 8) | o0;      // first of the four cells it touches, orientation bin
            s_v[g][at][0] = v_rc00; s_v[g][at][1] = v_rc01; s_v[g][at][2] = v_rc10; s_v[g][at][3] = v_rc11;      // [r][c] -> index 2r + c
            s_v[g][at][4] = obin;
            // tell the owners of the inner cells it touches (bit `at` of their word: order-free, so atomics may land in any order)
#pragma unroll
            for (int dr = 0; dr < 2; ++dr)
#pragma unroll
                for (int dc = 0; dc < 2; ++dc) {
                    const int rr = r0 + 1 + dr, cc = c0 + 1 + dc;
                    if (rr >= 1 && rr <= d && cc >= 1 && cc <= d) atomicOr(&s_touch[g][(rr - 1) * d + cc - 1], 1ull << at);
                }
        }
        qhead = (qhead + take) & 63;
        qn -= take;
        filled += take;
        // when one group's buffer is full (or its window ends) all four groups empty theirs: the owners' loops then
        // run side by side instead of one group at a time
        if (__ballot(filled > SIFT_DESC_BUF - 16 || (last && qn == 0 && filled > 0)) != 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // an owner visits the samples that touch its cell, lowest buffer slot first = raster order
            unsigned long long todo = s_touch[g][gl];
            while (__ballot(todo != 0ull) != 0) {
                if (todo != 0ull) {
                    const int sidx = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int cl = s_cell[g][sidx];
                    // the sample touches cells b, b + 1, b + 6, b + 7 of the 6x6 grid: this one is b + t, t = 6 dr + dc
                    const unsigned t = (unsigned)(my_cell - (cl >> 8));
                    const int o0 = cl & 255, q = (int)((t >> 2) * 2 + (t & 1u));              // 2 * dr + dc
                    const float share = s_v[g][sidx][q], hi = share * s_v[g][sidx][4];
                    mine[o0] += share - hi;
                    mine[o0 + 1] += hi;
                }
            }
            s_touch[g][gl] = 0ull;
            filled = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (base >= max_len && __ballot(qn > 0 || filled > 0) == 0) break;
    }
    // circular orientation bins, then the 128 values in (row, column, bin) order
    mine[0] += mine[n];
    mine[1] += mine[n + 1];
    for (int k = 0; k < n; ++k) s_dst[g][gl * n + k] = mine[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int L = d * d * n;
    if (gl == 0) {
        float nrm2 = 0;
        for (int k = 0; k < L; ++k) nrm2 += s_dst[g][k] * s_dst[g][k];
        const float thr = sqrtf(nrm2) * SIFT_DESCR_MAG_THR;
        nrm2 = 0;
        for (int k = 0; k < L; ++k) {
            const float vv = s_dst[g][k] < thr ? s_dst[g][k] : thr;
            nrm2 += vv * vv;
        }
        const float sq = sqrtf(nrm2);
        s_scale[g][0] = thr;
        s_scale[g][1] = SIFT_INT_DESCR_FCTR / (sq > SIFT_FLT_EPSILON ? sq : SIFT_FLT_EPSILON);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!has) return;
    for (int k = gl; k < L; k += 16) {
        const float vv = s_dst[g][k] < s_scale[g][0] ? s_dst[g][k] : s_scale[g][0];
        int q = sift_round_f(vv * s_scale[g][1]);
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        desc[(size_t)jb.out * L + k] = (float)q;
    }
}

// The jobs ordered by window radius, largest first: a wavefront of the descriptor kernel serves four keypoints and
// lasts as long as the largest of their windows.  Workgroup b places the jobs of list b: from the per-list radius
// counts every workgroup works out where each radius starts (scan over the totals, largest radius first) and how
// many jobs of that radius the lists before its own hold; inside one list and radius the order is whatever the LDS
// atomics give (a job names its output row, so the descriptors do not depend on it).
__global__ __launch_bounds__(256) void sift_order_jobs_kernel(const SiftJob *__restrict__ unordered, const int *__restrict__ counts, int cap,
                                                             const int *__restrict__ radius_hist, SiftJob *__restrict__ jobs)
{
    __shared__ int s_off, s_n;
    __shared__ int base[SIFT_JOB_BINS];
    __shared__ int wave_sum[4];
    const int tid = threadIdx.x, list_cap = cap / SIFT_LISTS;
    if (tid < 64) {
        const int raw = counts[tid * SIFT_LIST_STRIDE], mine = min(raw, list_cap);
        int upto = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(upto, d);
            if (tid >= d) upto += o;
        }
        if (tid == (int)blockIdx.x) { s_off = upto - mine; s_n = mine; }
    }
    // thread t holds bins SIFT_JOB_BINS - 1 - 4t ... - 4t - 3 (descending radius): totals over the lists, and the lists before this one
    int tot[4] = {0, 0, 0, 0}, pre[4] = {0, 0, 0, 0};
    for (int l = 0; l < SIFT_LISTS; ++l) {
        const int4 v = ((const int4 *)(radius_hist + l * SIFT_JOB_BINS))[255 - tid];
        const int q[4] = {v.w, v.z, v.y, v.x};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tot[k] += q[k];
            if (l < (int)blockIdx.x) pre[k] += q[k];
        }
    }
    const int sum4 = tot[0] + tot[1] + tot[2] + tot[3];
    int incl = sum4;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const int up = __shfl_up(incl, sft);
        if ((tid & 63) >= sft) incl += up;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    int start = incl - sum4;
    for (int wv = 0; wv < (tid >> 6); ++wv) start += wave_sum[wv];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        base[SIFT_JOB_BINS - 1 - (4 * tid + k)] = start + pre[k];
        start += tot[k];
    }
    __syncthreads();
    for (int i = tid; i < s_n; i += 256) {
        const SiftJob jb = unordered[s_off + i];
        jobs[atomicAdd(&base[sift_job_bin(jb.radius)], 1)] = jb;
    }
}

// dst row i = src row rows[i] (128 floats each): the descriptors of the keypoints that survive the ordering, in its order
__global__ __launch_bounds__(256) void sift_gather_rows_kernel(const float *__restrict__ src, const int *__restrict__ rows, int n,
                                                              float *__restrict__ dst)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = t >> 7;
    if (row < (size_t)n) dst[t] = src[(size_t)rows[row] * 128 + (t & 127)];
}

// ---------------------------------------------------------------------------------------------------
static int sift_gauss_kernel(double sigma, SiftKernel &K)
{
    int n = (int)lrint(sigma * 8 + 1) | 1;
    VO_REQUIRE(n <= SIFT_MAX_TAPS, "SIFT: sigma %g needs a %d-tap Gaussian (at most %d)", sigma, n, SIFT_MAX_TAPS);
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        K.k[i] = (float)exp(s2 * x * x);
        sum += K.k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) K.k[i] = (float)(K.k[i] * sum);
    K.n = n;
    return VO_OK;
}

// dst = src blurred with sigma; with `diff` also diff = dst - src
static int sift_blur(vo_ctx *ctx, const float *src, float *dst, float *tmp, int w, int h, double sigma, float *diff)
{
    SiftKernel K;
    int rc = sift_gauss_kernel(sigma, K);
    if (rc) return rc;
    hipLaunchKernelGGL(sift_blur_rows_kernel, dim3((w + SIFT_ROW_SEG - 1) / SIFT_ROW_SEG, h), dim3(SIFT_ROW_SEG), 0, ctx->stream, src, tmp, w, h, K);
    hipLaunchKernelGGL(sift_blur_cols_kernel, dim3((w + 63) / 64, (h + SIFT_COL_ROWS - 1) / SIFT_COL_ROWS), dim3(256), 0, ctx->stream,
                       (const float *)tmp, dst, w, h, K, src, diff);
    return VO_OK;
}

struct SiftSpace {
    SiftOctaves O;
    float *pool = nullptr;      // Gaussian and difference images (context slot 1)
    float *tmp = nullptr;       // row-pass output / doubled input (context slot 2)
};

static int sift_num_octaves(int w, int h)
{
    const int m = 2 * (w < h ? w : h);
    return (int)lrint(log((double)m) / log(2.) - 2) + 1;
}

// Scale space of one image (device pointer d_img) in the context's buffers; asynchronous.
static int sift_build(vo_ctx *ctx, const uint8_t *d_img, int height, int width, int64_t pitch, int layers, double sigma, SiftSpace &S)
{
    SiftOctaves &O = S.O;
    O.layers = layers;
    O.n_oct = sift_num_octaves(width, height);
    VO_REQUIRE(O.n_oct >= 1 && O.n_oct <= SIFT_MAX_OCTAVES, "SIFT: %d octaves for a %d x %d image", O.n_oct, width, height);
    const int W = 2 * width, H = 2 * height;
    size_t total = 0;
    for (int o = 0; o < O.n_oct; ++o) {
        O.w[o] = o == 0 ? W : O.w[o - 1] / 2;
        O.h[o] = o == 0 ? H : O.h[o - 1] / 2;
        VO_REQUIRE(O.w[o] >= 1 && O.h[o] >= 1, "SIFT: octave %d of a %d x %d image is empty", o, width, height);
        const size_t plane = (size_t)O.w[o] * O.h[o];
        O.rmax[o] = (int)sqrt((double)O.w[o] * O.w[o] + (double)O.h[o] * O.h[o]);
        O.gauss_off[o] = total; total += plane * (layers + 3);
        O.dog_off[o] = total; total += plane * (layers + 2);
    }
    int rc;
    void *p;
    if ((rc = vo_stage_reserve(ctx, 1, total * 4, &p))) return rc; S.pool = (float *)p;
    if ((rc = vo_stage_reserve(ctx, 2, (size_t)W * H * 8, &p))) return rc; S.tmp = (float *)p;
    float *dbl = S.tmp + (size_t)W * H;
    hipLaunchKernelGGL(sift_upsample_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, ctx->stream, d_img, width, height, pitch, dbl);
    double sig[SIFT_MAX_LAYERS + 3];
    sig[0] = sigma;
    const double k = pow(2., 1. / layers);
    for (int i = 1; i < layers + 3; ++i) {
        const double sp = pow(k, (double)(i - 1)) * sigma, st = sp * k;
        sig[i] = sqrt(st * st - sp * sp);
    }
    double sd = sigma * sigma - (double)(SIFT_INIT_SIGMA * 2) * (SIFT_INIT_SIGMA * 2);
    sd = sqrt(sd > 0.01 ? sd : 0.01);
    // the octaves from `tail` on are small enough for one workgroup (sift_tail_octaves_kernel)
    SiftTail T;
    T.K[0].n = 0;
    T.halo = 0;
    for (int i = 1; i < layers + 3; ++i) {
        if ((rc = sift_gauss_kernel(sig[i], T.K[i]))) return rc;
        T.halo = T.K[i].n / 2 > T.halo ? T.K[i].n / 2 : T.halo;
    }
    int tail = O.n_oct;
    while (tail > 1 && (size_t)O.w[tail - 1] * O.h[tail - 1] <= SIFT_TAIL_PIXELS &&
           (size_t)(O.w[tail - 1] + 2 * T.halo) * O.h[tail - 1] <= SIFT_TAIL_PADDED &&
           (size_t)O.w[tail - 1] * (O.h[tail - 1] + 2 * T.halo) <= SIFT_TAIL_PADDED)
        --tail;
    for (int o = 0; o < tail; ++o) {
        const int ow = O.w[o], oh = O.h[o];
        const size_t plane = (size_t)ow * oh;
        float *g = S.pool + O.gauss_off[o];
        for (int i = 0; i < layers + 3; ++i) {
            float *dst = g + plane * i;
            if (o == 0 && i == 0) { if ((rc = sift_blur(ctx, dbl, dst, S.tmp, ow, oh, sd, nullptr))) return rc; }
            else if (i == 0)
                hipLaunchKernelGGL(sift_halve_kernel, dim3((ow + 63) / 64, (oh + 3) / 4), dim3(256), 0, ctx->stream,
                                   (const float *)(S.pool + O.gauss_off[o - 1] + (size_t)O.w[o - 1] * O.h[o - 1] * layers), O.w[o - 1], dst, ow, oh);
            else if ((rc = sift_blur(ctx, dst - plane, dst, S.tmp, ow, oh, sig[i], S.pool + O.dog_off[o] + plane * (i - 1)))) return rc;
        }
    }
    if (tail < O.n_oct) {
        T.first = tail;
        hipLaunchKernelGGL(sift_tail_octaves_kernel, dim3(1), dim3(1024), 0, ctx->stream, S.pool, O, T);
    }
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

static int sift_check_args(const char *who, vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, int layers,
                           double sigma, int mem)
{
    VO_REQUIRE(ctx && img, "%s: NULL argument", who);
    VO_REQUIRE(height >= 2 && width >= 2 && row_stride >= width && (int64_t)height * width < (1ll << 28), "%s: bad shape h=%d w=%d", who,
               height, width);
    VO_REQUIRE(layers >= 1 && layers <= SIFT_MAX_LAYERS, "%s: n_octave_layers=%d (1..%d)", who, layers, SIFT_MAX_LAYERS);
    VO_REQUIRE(sigma > 0.1 && sigma < 7, "%s: sigma=%g", who, sigma);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "%s: bad mem flag %d", who, mem);
    return VO_OK;
}

static int sift_stage_image(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride, int mem, const uint8_t **d_img,
                            int64_t *pitch)
{
    *d_img = img; *pitch = row_stride;
    if (mem == VO_MEM_DEVICE) return VO_OK;
    void *p;
    int rc = vo_stage_reserve(ctx, 0, (size_t)width * height, &p);
    if (rc) return rc;
    if (row_stride == width)
        VO_UPLOAD(ctx, p, img, (size_t)width * height);
    else
        VO_UPLOAD_2D(ctx, p, img, row_stride, width, height);
    *d_img = (const uint8_t *)p; *pitch = width;
    return VO_OK;
}

extern "C" int vo_sift_scale_space_image(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride,
                                         int n_octave_layers, double sigma, int which, int octave, int index, float *host_out,
                                         int *out_width, int *out_height, int mem)
{
    int rc = sift_check_args("vo_sift_scale_space_image", ctx, img, height, width, row_stride, n_octave_layers, sigma, mem);
    if (rc) return rc;
    VO_REQUIRE(out_width && out_height, "vo_sift_scale_space_image: NULL size outputs");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const uint8_t *d_img;
    int64_t pitch;
    if ((rc = sift_stage_image(ctx, img, height, width, row_stride, mem, &d_img, &pitch))) return rc;
    SiftSpace S;
    if ((rc = sift_build(ctx, d_img, height, width, pitch, n_octave_layers, sigma, S))) return rc;
    VO_REQUIRE(octave >= 0 && octave < S.O.n_oct && index >= 0 && index < n_octave_layers + (which ? 2 : 3),
               "vo_sift_scale_space_image: no image (octave %d of %d, index %d)", octave, S.O.n_oct, index);
    *out_width = S.O.w[octave]; *out_height = S.O.h[octave];
    const size_t plane = (size_t)S.O.w[octave] * S.O.h[octave];
    if (host_out) {
        const float *src = S.pool + (which ? S.O.dog_off[octave] : S.O.gauss_off[octave]) + plane * index;
        VO_HIP_CHECK(hipMemcpyAsync(host_out, src, plane * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_sift_detect_compute(vo_ctx *ctx, const uint8_t *img, int height, int width, int64_t row_stride,
                                      int n_octave_layers, double contrast_threshold, double edge_threshold, double sigma,
                                      vo_sift_keypoint *keypoints, float *descriptors, int max_keypoints, int32_t *n_keypoints,
                                      int mem)
{
    int rc = sift_check_args("vo_sift_detect_compute", ctx, img, height, width, row_stride, n_octave_layers, sigma, mem);
    if (rc) return rc;
    VO_REQUIRE(n_keypoints && max_keypoints >= 0 && (max_keypoints == 0 || keypoints), "vo_sift_detect_compute: NULL output");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const uint8_t *d_img;
    int64_t pitch;
    if ((rc = sift_stage_image(ctx, img, height, width, row_stride, mem, &d_img, &pitch))) return rc;
    SiftSpace S;
    if ((rc = sift_build(ctx, d_img, height, width, pitch, n_octave_layers, sigma, S))) return rc;
    const SiftOctaves &O = S.O;

    // candidates and raw keypoints: counters | candidates | keypoints (context slot 3)
    // counters (candidate lists | keypoint lists | keypoint total) | candidates | keypoint lists | keypoints joined
    // The room for both starts at half a million extrema and a quarter of a million raw keypoints (five times what a
    // dense KITTI-sized texture yields) and belongs to the context: a frame that overflows one of the 64 lists of
    // either kind -- the kernels keep counting past the end without writing there -- has the room grown to what the
    // counters ask for and the detection passes run again (the scale space stays).  VO_SIFT_CAND_CAP / VO_SIFT_KP_CAP
    // in the environment set the starting room (tests).
    if (!ctx->sift_cand_cap) {
        const char *e1 = getenv("VO_SIFT_CAND_CAP"), *e2 = getenv("VO_SIFT_KP_CAP");
        ctx->sift_cand_cap = e1 && atoi(e1) >= SIFT_LISTS ? atoi(e1) / SIFT_LISTS * SIFT_LISTS : 1 << 19;
        ctx->sift_kp_cap = e2 && atoi(e2) >= SIFT_LISTS ? atoi(e2) / SIFT_LISTS * SIFT_LISTS : 1 << 18;
    }
    const size_t cnt_bytes = (size_t)(2 * SIFT_LISTS * SIFT_LIST_STRIDE + 64) * 4;
    const bool describe = descriptors && max_keypoints > 0;
    const int threshold = (int)floor(0.5 * contrast_threshold / n_octave_layers * 255);
    int cand_cap = 0, kp_cap = 0;
    int *d_cnt = nullptr, *d_kcnt = nullptr, *d_ktotal = nullptr, *d_radius_hist = nullptr;
    SiftJob *d_jobs = nullptr;
    SiftCand *d_cand = nullptr;
    vo_sift_keypoint *d_klists = nullptr, *d_kp = nullptr;
    int h_cnt[2] = {0, 0};
    void *p;
    for (int attempt = 0;; ++attempt) {
        VO_REQUIRE(attempt < 8, "vo_sift_detect_compute: the candidate buffers keep overflowing");
        cand_cap = ctx->sift_cand_cap; kp_cap = ctx->sift_kp_cap;
        if ((rc = vo_stage_reserve(ctx, 3, cnt_bytes + (size_t)cand_cap * sizeof(SiftCand) + 2 * (size_t)kp_cap * sizeof(vo_sift_keypoint), &p)))
            return rc;
        d_cnt = (int *)p; d_kcnt = d_cnt + SIFT_LISTS * SIFT_LIST_STRIDE; d_ktotal = d_kcnt + SIFT_LISTS * SIFT_LIST_STRIDE;
        d_radius_hist = nullptr;       // SIFT_LISTS rows of SIFT_JOB_BINS counts behind the job arrays (rewritten whole by every call)
        // descriptor jobs, unordered and ordered, for as many keypoints as the keypoint buffer holds (formed while the lists are joined)
        if (describe) {
            void *q;
            const size_t job_bytes = (2 * (size_t)kp_cap * sizeof(SiftJob) + 255) & ~(size_t)255;
            if ((rc = vo_stage_reserve(ctx, 4, job_bytes + (size_t)SIFT_LISTS * SIFT_JOB_BINS * 4, &q))) return rc;
            d_jobs = (SiftJob *)q;
            d_radius_hist = (int *)((char *)q + job_bytes);
        }
        d_cand = (SiftCand *)((char *)p + cnt_bytes);
        d_klists = (vo_sift_keypoint *)((char *)p + cnt_bytes + (size_t)cand_cap * sizeof(SiftCand));
        d_kp = d_klists + kp_cap;
        VO_HIP_CHECK(hipMemsetAsync(d_cnt, 0, cnt_bytes, ctx->stream));
        for (int o = 0; o < O.n_oct; ++o) {
            const int ow = O.w[o], oh = O.h[o];
            if (ow <= 2 * SIFT_IMG_BORDER || oh <= 2 * SIFT_IMG_BORDER) continue;
            hipLaunchKernelGGL(sift_extrema_kernel, dim3((ow - 2 * SIFT_IMG_BORDER + 63) / 64, (oh - 2 * SIFT_IMG_BORDER + EX_ROWS - 1) / EX_ROWS),
                               dim3(256), (size_t)(n_octave_layers + 2) * (EX_ROWS + 2) * EX_PITCH * 4, ctx->stream,
                               (const float *)(S.pool + O.dog_off[o]), (size_t)ow * oh, ow, oh, o, n_octave_layers, threshold, d_cand, d_cnt,
                               cand_cap);
        }
        VO_HIP_CHECK(hipGetLastError());
        h_cnt[0] = h_cnt[1] = 0;
        int h_lists[SIFT_LISTS * SIFT_LIST_STRIDE];
        VO_RESULT(ctx, h_lists, d_cnt, sizeof(h_lists));
        VO_RESULTS_FINISH(ctx);
        int longest = 0;
        for (int l = 0; l < SIFT_LISTS; ++l) {
            const int n = h_lists[l * SIFT_LIST_STRIDE];
            longest = n > longest ? n : longest;
            h_cnt[0] += n;
        }
        if (longest > cand_cap / SIFT_LISTS) {
            const int64_t want = ((int64_t)longest + longest / 4 + 64) * SIFT_LISTS;
            VO_REQUIRE(want < (1ll << 30), "vo_sift_detect_compute: %d scale-space extrema in one of %d lists", longest, SIFT_LISTS);
            ctx->sift_cand_cap = (int)want;
            continue;
        }
        if (h_cnt[0] > 0) {
            hipLaunchKernelGGL(sift_refine_kernel, dim3(h_cnt[0]), dim3(64), 0, ctx->stream, (const float *)S.pool, O, (const SiftCand *)d_cand,
                               (const int *)d_cnt, cand_cap, (float)contrast_threshold, (float)edge_threshold, (float)sigma, d_klists, d_kcnt, kp_cap);
            hipLaunchKernelGGL(sift_join_lists_kernel, dim3(SIFT_LISTS), dim3(256), 0, ctx->stream, (const vo_sift_keypoint *)d_klists,
                               (const int *)d_kcnt, kp_cap, d_kp, d_ktotal, O, describe ? d_jobs + kp_cap : (SiftJob *)nullptr, d_radius_hist);
            VO_HIP_CHECK(hipGetLastError());
        }
        VO_RESULT(ctx, h_cnt + 1, d_ktotal, 4);
        VO_RESULTS_FINISH(ctx);
        if (h_cnt[1] > kp_cap) {        // a keypoint list overflowed: its counter says by how much
            VO_RESULT(ctx, h_lists, d_kcnt, sizeof(h_lists));
            VO_RESULTS_FINISH(ctx);
            longest = 0;
            for (int l = 0; l < SIFT_LISTS; ++l) longest = h_lists[l * SIFT_LIST_STRIDE] > longest ? h_lists[l * SIFT_LIST_STRIDE] : longest;
            const int64_t want = ((int64_t)longest + longest / 4 + 64) * SIFT_LISTS;
            VO_REQUIRE(want < (1ll << 28) && want > kp_cap, "vo_sift_detect_compute: %d keypoints in one of %d lists", longest, SIFT_LISTS);
            ctx->sift_kp_cap = (int)want;
            continue;
        }
        break;
    }
    const size_t n_raw = (size_t)h_cnt[1];
    if (n_raw == 0) { *n_keypoints = 0; return VO_OK; }
    // Everything the rest of the call needs, before anything is queued (growing a buffer waits for the device):
    // page-locked room for the keypoints coming back and the row order going out; the descriptors of all raw
    // keypoints, the row order and -- for a host caller -- the ordered descriptors on the device.
    const size_t kp_bytes = (n_raw * sizeof(vo_sift_keypoint) + 15) & ~(size_t)15, row_bytes = (n_raw * 4 + 15) & ~(size_t)15;
    void *pin;
    if ((rc = vo_pinned_reserve(ctx, kp_bytes + row_bytes, &pin))) return rc;
    const vo_sift_keypoint *h_raw = (const vo_sift_keypoint *)pin;
    int *h_rows = (int *)((char *)pin + kp_bytes);
    float *d_raw_desc = nullptr, *d_desc = descriptors;
    int *d_rows = nullptr;
    if (describe) {
        if ((rc = vo_stage_reserve(ctx, 6, n_raw * 128 * 4, &p))) return rc;
        d_raw_desc = (float *)p;
        if ((rc = vo_stage_reserve(ctx, 7, row_bytes, &p))) return rc;
        d_rows = (int *)p;
        if (mem == VO_MEM_HOST) {
            const size_t most = n_raw < (size_t)max_keypoints ? n_raw : (size_t)max_keypoints;
            if ((rc = vo_stage_reserve(ctx, 5, most * 128 * 4, &p))) return rc;
            d_desc = (float *)p;
        }
    }
    // The keypoints leave for the host first; the descriptors of ALL raw keypoints are computed while the host puts
    // them in order (a descriptor does not depend on the keypoint's place in the output, only its row does, and the
    // few duplicates the ordering drops cost less than the device idling through it).
    if ((rc = vo_download_async(ctx, pin, d_kp, n_raw * sizeof(vo_sift_keypoint)))) return rc;
    VO_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    if (describe) {
        hipLaunchKernelGGL(sift_order_jobs_kernel, dim3(SIFT_LISTS), dim3(256), 0, ctx->stream, (const SiftJob *)(d_jobs + kp_cap),
                           (const int *)d_kcnt, kp_cap, (const int *)d_radius_hist, d_jobs);
        hipLaunchKernelGGL(sift_descriptor_kernel, dim3((unsigned)((n_raw + 3) / 4)), dim3(64), 0, ctx->stream, (const float *)S.pool, O,
                           (const SiftJob *)d_jobs, (int)n_raw, d_raw_desc);
        VO_HIP_CHECK(hipGetLastError());
    }
    VO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    struct Rec { vo_sift_keypoint k; uint32_t row; };      // a raw keypoint and where it (and its descriptor) stands in the raw list
    std::vector<Rec> kps(n_raw);
    for (size_t i = 0; i < n_raw; ++i) { kps[i].k = h_raw[i]; kps[i].row = (uint32_t)i; }
    // KeyPointsFilter::removeDuplicatedSorted: order by (x, y, size desc, angle, response desc, octave desc), drop repeats.
    {
        const auto full_less = [](const Rec &ra, const Rec &rb) {
            const vo_sift_keypoint &a = ra.k, &b = rb.k;
            if (a.x != b.x) return a.x < b.x;
            if (a.y != b.y) return a.y < b.y;
            if (a.size != b.size) return a.size > b.size;
            if (a.angle != b.angle) return a.angle < b.angle;
            if (a.response != b.response) return a.response > b.response;
            return a.octave > b.octave;
        };
        // x is the leading key and lies in [0, 2 * width): a counting sort on its integer part leaves buckets of a few
        // keypoints each, which are then put in the full order (two linear passes and many tiny sorts instead of one
        // comparison sort over tens of thousands of records)
        const int nbuckets = 2 * width + 2;
        bool plain = true;
        for (const auto &r : kps) plain = plain && r.k.x >= 0 && r.k.x < (float)(nbuckets - 1);   // always true for SIFT; else: full sort
        if (plain && kps.size() > 64) {
            std::vector<uint32_t> start((size_t)nbuckets + 1, 0);
            for (const auto &r : kps) ++start[(size_t)(int)r.k.x + 1];
            for (int b = 0; b < nbuckets; ++b) start[(size_t)b + 1] += start[b];
            std::vector<Rec> sorted(kps.size());
            {
                std::vector<uint32_t> fill(start.begin(), start.end() - 1);
                for (const auto &r : kps) sorted[fill[(size_t)(int)r.k.x]++] = r;
            }
            for (int b = 0; b < nbuckets; ++b)
                if (start[(size_t)b + 1] - start[b] > 1) std::sort(sorted.begin() + start[b], sorted.begin() + start[(size_t)b + 1], full_less);
            kps.swap(sorted);
        } else {
            std::sort(kps.begin(), kps.end(), full_less);
        }
    }
    size_t m = 0;
    for (size_t i = 0; i < kps.size(); ++i) {
        if (m > 0 && kps[i].k.x == kps[m - 1].k.x && kps[i].k.y == kps[m - 1].k.y && kps[i].k.size == kps[m - 1].k.size &&
            kps[i].k.angle == kps[m - 1].k.angle)
            continue;
        kps[m++] = kps[i];
    }
    *n_keypoints = (int32_t)m;
    const size_t n_out = m < (size_t)max_keypoints ? m : (size_t)max_keypoints;
    const int first_octave = -1;
    for (size_t i = 0; i < n_out; ++i) {       // back to the coordinates of the input image (the scale space starts at the doubled image)
        vo_sift_keypoint k = kps[i].k;
        k.octave = (k.octave & ~255) | ((k.octave + first_octave) & 255);
        k.x *= 0.5f; k.y *= 0.5f; k.size *= 0.5f;
        keypoints[i] = k;
        h_rows[i] = (int)kps[i].row;
    }
    if (!describe) return VO_OK;
    if ((rc = vo_copy_pinned_wide_async(ctx, d_rows, h_rows, n_out * 4))) return rc;
    hipLaunchKernelGGL(sift_gather_rows_kernel, dim3((unsigned)((n_out * 128 + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float *)d_raw_desc, (const int *)d_rows, (int)n_out, d_desc);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST)
        VO_HIP_CHECK(hipMemcpyAsync(descriptors, d_desc, n_out * 128 * 4, hipMemcpyDeviceToHost, ctx->stream));
    // the page-locked row order must stay untouched until the kernel has read its copy: wait in both memory modes
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_sift(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)sift_upsample_kernel));
    return VO_OK;
}
