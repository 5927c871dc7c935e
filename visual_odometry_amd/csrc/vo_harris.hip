// Harris / Shi-Tomasi corner response and non-maximum suppression + top-K.
//
// Response: cv.cornerHarris(float32(img), blockSize=2, ksize=3, k) as
// FeatureExtractor.harris calls it (vo_pipeline/featureExtraction.py:37-40):
// float32 Sobel 3x3 scaled by 1/8 (REFLECT_101), 2x2 un-normalised box sum of
// the derivative products with offsets {-1,0} (REFLECT_101 on the product maps),
// R = a*c - b*b - k*(a+c)^2; mode 1 = min-eigenvalue (Shi-Tomasi) response.
// One workgroup stages a (16+3)x(64+3) image tile in LDS and emits 16x64 responses,
// four per lane, computed in registers; image bytes are read once from HBM per tile
// (+halo), the response is written once.
//
// NMS/top-K: per image (1) max reduction, (2) local maxima above rel_thresh*max
// in a (2r+1)^2 window (separable running maximum over LDS tiles; ties go to the
// earlier raster position) are stored as 64-bit keys {ordered(response), ~raster
// index} in per-tile slots, (3) one workgroup radix-selects the K-th largest key,
// gathers the K winners into LDS, bitonic-sorts them and writes (x, y) strongest
// first.  Keys are unique, so the output is deterministic.
#include "vo_internal.h"
#include <limits.h>

#define HR_TW 64
#define HR_TH 16
#define HR_IW (HR_TW + 3)
#define HR_IH (HR_TH + 3)
#define TOPK_MAX 4096
#define TOPK_THREADS 1024
// per-image maxima are kept MAXKEY_STRIDE words apart: atomics on neighbouring words queue up at one
// memory channel (thousands of workgroups report to a few cache lines otherwise)
#define MAXKEY_STRIDE 64

// atomicMax that skips the atomic when the value cannot raise the maximum any more
__device__ __forceinline__ void raise_max(uint32_t *slot, uint32_t v)
{
    if (v > __builtin_nontemporal_load(slot)) atomicMax(slot, v);
}

__device__ __forceinline__ int refl1(int i, int n)
{
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
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

__device__ __forceinline__ float harris_value(float a, float b, float c, float k, int mode)
{
    if (mode == 0) {
        float ac_bb = a * c - b * b;
        float ac = a + c;
        return ac_bb - k * ac * ac;
    }
    float ha = a * 0.5f, hc = c * 0.5f;
    return (ha + hc) - sqrtf((ha - hc) * (ha - hc) + b * b);
}

// One workgroup: 64 x 16 responses.  The (16+3) x (64+3) image tile is staged as floats in LDS
// (interior tiles: unaligned dword reads, four pixels per item, all of a lane's reads in flight
// before the first LDS store; border tiles: bytes with REFLECT_101).  A lane then finishes four
// horizontally adjacent responses entirely in registers: it reads the 4 x 8 floats around them as
// eight 16-byte LDS reads, forms the row differences / row smoothings once per tile row, the 2 x 5
// Sobel derivatives and their products from them, and the 2x2 box sums from the products; the
// expressions and their order are those of the scalar definition, so results are bit-identical
// to it.  Four responses leave as one 16-byte store.
__global__ __launch_bounds__(256) void harris_response_kernel(const uint8_t *__restrict__ img, int w, int h,
                                                              int64_t pitch, int64_t istride, float k, int mode,
                                                              float *__restrict__ R, uint32_t *__restrict__ maxkey)
{
    constexpr int IS = 68;   // row stride of the tile: 67 columns used, rows 16-byte aligned
    __shared__ __attribute__((aligned(16))) float tile[HR_IH * IS];
    __shared__ uint32_t wave_max[4];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * HR_TW, y0 = blockIdx.y * HR_TH;
    const uint8_t *src = img + (int64_t)blockIdx.z * istride;
    float *dst = R + (int64_t)blockIdx.z * w * h;

    if (x0 - 2 >= 0 && y0 - 2 >= 0 && x0 - 2 + IS <= w && y0 - 2 + HR_IH <= h) {
        constexpr int ITEMS = HR_IH * (IS / 4), ROUNDS = (ITEMS + 255) / 256;
        const uint8_t *base = src + (int64_t)(y0 - 2) * pitch + (x0 - 2);
        uint32_t q[ROUNDS];
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int idx = min(t + 256 * i, ITEMS - 1);
            const int r = idx / (IS / 4), c = idx - r * (IS / 4);
            __builtin_memcpy(&q[i], base + (int64_t)r * pitch + 4 * c, 4);
        }
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int idx = min(t + 256 * i, ITEMS - 1);
            const float4 f = {(float)(q[i] & 0xffu), (float)((q[i] >> 8) & 0xffu), (float)((q[i] >> 16) & 0xffu), (float)(q[i] >> 24)};
            *reinterpret_cast<float4 *>(tile + 4 * idx) = f;
        }
    } else {
        constexpr int ITEMS = HR_IH * HR_IW, ROUNDS = (ITEMS + 255) / 256;
        uint8_t v[ROUNDS];
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int idx = min(t + 256 * i, ITEMS - 1);
            const int r = idx / HR_IW, c = idx - r * HR_IW;
            v[i] = src[(int64_t)refl1(y0 - 2 + r, h) * pitch + refl1(x0 - 2 + c, w)];
        }
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int idx = min(t + 256 * i, ITEMS - 1);
            const int r = idx / HR_IW, c = idx - r * HR_IW;
            tile[r * IS + c] = (float)v[i];
        }
    }
    __syncthreads();

    const int tx = (t & 15) * 4, ty = t >> 4;
    const int gx = x0 + tx, gy = y0 + ty;
    const bool inside = gx < w && gy < h;
    float p[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 lo = *reinterpret_cast<const float4 *>(tile + (ty + r) * IS + tx);
        const float4 hi = *reinterpret_cast<const float4 *>(tile + (ty + r) * IS + tx + 4);
        p[r][0] = lo.x; p[r][1] = lo.y; p[r][2] = lo.z; p[r][3] = lo.w;
        p[r][4] = hi.x; p[r][5] = hi.y; p[r][6] = hi.z; p[r][7] = hi.w;
    }
    // derivative (dr, c) sits at image (gx - 1 + c, gy - 1 + dr) and uses tile rows dr..dr+2, columns c..c+2
    float pxx[2][5], pxy[2][5], pyy[2][5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        float hd[4], hs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hd[r] = p[r][c + 2] - p[r][c];
            hs[r] = p[r][c + 1] * 0.25f + (p[r][c] + p[r][c + 2]) * 0.125f;
        }
#pragma unroll
        for (int dr = 0; dr < 2; ++dr) {
            float dx = hd[dr + 1] * 0.25f + (hd[dr] + hd[dr + 2]) * 0.125f;
            float dy = hs[dr + 2] - hs[dr];
            // one pixel left of / above the image the product maps are mirrored: undo the sign the
            // mirrored image gives the odd derivative
            if (gx - 1 + c < 0) dx = -dx;
            if (gy - 1 + dr < 0) dy = -dy;
            pxx[dr][c] = dx * dx; pxy[dr][c] = dx * dy; pyy[dr][c] = dy * dy;
        }
    }
    float out[4];
    uint32_t mk = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = pxx[0][j] + pxx[0][j + 1], b = pxy[0][j] + pxy[0][j + 1], c = pyy[0][j] + pyy[0][j + 1];
        const float ra = pxx[1][j] + pxx[1][j + 1], rb = pxy[1][j] + pxy[1][j + 1], rc = pyy[1][j] + pyy[1][j + 1];
        a = a + ra; b = b + rb; c = c + rc;
        out[j] = harris_value(a, b, c, k, mode);
        if (inside && gx + j < w) mk = max(mk, ordered_key(out[j]));
    }
    if (inside) {
        float *o = dst + (int64_t)gy * w + gx;
        if (gx + 3 < w) {
            __builtin_memcpy(o, out, 16);
        } else {
            for (int j = 0; j < 4 && gx + j < w; ++j) o[j] = out[j];
        }
    }
    if (maxkey) {
        // per-image maximum of the response (as ordered keys) for the detector's relative threshold:
        // at most one atomic per workgroup
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mk = max(mk, (uint32_t)__shfl_xor((int)mk, o));
        if ((t & 63) == 0) wave_max[t >> 6] = mk;
        __syncthreads();
        if (t == 0) raise_max(&maxkey[blockIdx.z * MAXKEY_STRIDE], max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3])));
    }
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
        raise_max(&maxkey[blockIdx.y * MAXKEY_STRIDE], max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3])));
}

#define NMS_TW 64
#define NMS_TH 16
#define NMS_MAX_SLOTS 256      // survivors of one tile at the smallest radius: (65 / 2) * (17 / 2)
#define TOPK_POOL 8192         // candidate keys topk_kernel keeps in LDS

// Window non-maximum suppression in one pass over R.  A workgroup stages a 64x16
// tile with a halo of `radius` pixels (outside the image: -inf) and takes the running
// maximum along rows, then along columns, the rows above the centre separately.  A
// pixel above the threshold survives when it equals its window maximum and every
// earlier raster neighbour (rows above, left part of its own row) is strictly
// smaller: an earlier neighbour wins ties.  Survivors are more than `radius` apart,
// which bounds their number per tile; they are collected in LDS as 64-bit keys
// {ordered(R), ~raster index} and stored in the tile's own slots of the candidate
// array, the unused slots zeroed (zero marks an empty slot): no global atomic and no
// clearing of the array between calls.
// RAD > 0 fixes the radius at compile time: a lane then finishes four neighbouring
// outputs per step from one set of 16-byte LDS reads (row pass) or one column of
// row maxima held in registers (column pass).  RAD == 0 takes the radius at run time.
template <int RAD>
__global__ __launch_bounds__(256) void nms_tile_kernel(const float *__restrict__ R, int w, int h, int radius, float rel_thresh,
                                                       const uint32_t *__restrict__ maxkey,
                                                       unsigned long long *__restrict__ cand, int tile_slots)
{
    extern __shared__ __attribute__((aligned(16))) float nms_lds[];
    __shared__ unsigned long long s_keys[NMS_MAX_SLOTS];
    __shared__ int s_n;
    const int r = RAD > 0 ? RAD : radius, TH = NMS_TH + 2 * r;
    // row stride: whole 16-byte groups, wide enough for the last lane's reads of the row pass
    const int TW = RAD > 0 ? NMS_TW - 4 + 4 * ((4 + 2 * r + 3) / 4) : NMS_TW + 2 * r;
    float *tile = nms_lds, *hm = nms_lds + TH * TW;
    const int b = blockIdx.z, x0 = blockIdx.x * NMS_TW, y0 = blockIdx.y * NMS_TH;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const float *src = R + (int64_t)b * w * h;
    const float ninf = -__builtin_inff();
    if (threadIdx.x == 0) s_n = 0;
    const uint32_t image_max = maxkey[b * MAXKEY_STRIDE];   // issued with the tile reads, needed after them
    const bool interior = RAD > 0 && x0 - r >= 0 && x0 - r + TW <= w && y0 - r >= 0 && y0 - r + TH <= h;
    if (interior) {
        // whole tile inside the image: 16-byte reads (4-byte aligned in memory, 16-byte aligned in LDS),
        // all of a lane's reads in flight before the first LDS store
        constexpr int RADC = RAD > 0 ? RAD : 1;
        constexpr int TWC = NMS_TW - 4 + 4 * ((4 + 2 * RADC + 3) / 4), V4 = TWC / 4;
        constexpr int ITEMS = (NMS_TH + 2 * RADC) * V4, ROUNDS = (ITEMS + 255) / 256;
        const float *base = src + (int64_t)(y0 - r) * w + (x0 - r);
        float4 v[ROUNDS];
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) {
            const int idx = min((int)threadIdx.x + 256 * q, ITEMS - 1);
            const int row = idx / V4, c4 = idx - row * V4;
            __builtin_memcpy(&v[q], base + (int64_t)row * w + 4 * c4, 16);
        }
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) {
            const int idx = min((int)threadIdx.x + 256 * q, ITEMS - 1);
            *reinterpret_cast<float4 *>(tile + 4 * idx) = v[q];
        }
    } else
    // tiles on the image border: sixteen independent reads in flight per lane (eight rows x two
    // column groups), one round for radius <= 8
    for (int yb = ly; yb < TH; yb += 32) {
        float v[8][2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int yy = yb + 4 * q, gy = y0 - r + yy;
            const bool row_in = yy < TH && gy >= 0 && gy < h;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int xx = lx + 64 * c, gx = x0 - r + xx;
                v[q][c] = (row_in && xx < TW && gx >= 0 && gx < w) ? src[(int64_t)gy * w + gx] : ninf;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int yy = yb + 4 * q;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int xx = lx + 64 * c;
                if (yy < TH && xx < TW) tile[yy * TW + xx] = v[q][c];
            }
        }
    }
    __syncthreads();
    if (RAD > 0) {
        constexpr int NV = (4 + 2 * RAD + 3) / 4;       // 16-byte reads covering the 4 + 2r inputs of four outputs
        const int q4 = (threadIdx.x & 15) * 4;
        for (int yy = threadIdx.x >> 4; yy < TH; yy += 16) {
            float in[4 * NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float4 f = *reinterpret_cast<const float4 *>(tile + yy * TW + q4 + 4 * k);
                in[4 * k] = f.x; in[4 * k + 1] = f.y; in[4 * k + 2] = f.z; in[4 * k + 3] = f.w;
            }
            float4 o;
            if constexpr (RAD >= 2) {
                // inputs 3 .. 2r are common to the four windows
                float core = in[3];
#pragma unroll
                for (int d = 4; d <= 2 * RAD; ++d) core = fmaxf(core, in[d]);
                o.x = fmaxf(fmaxf(core, in[2]), fmaxf(in[1], in[0]));
                o.y = fmaxf(fmaxf(core, in[2]), fmaxf(in[1], in[2 * RAD + 1]));
                o.z = fmaxf(fmaxf(core, in[2]), fmaxf(in[2 * RAD + 2], in[2 * RAD + 1]));
                o.w = fmaxf(fmaxf(core, in[2 * RAD + 3]), fmaxf(in[2 * RAD + 2], in[2 * RAD + 1]));
            } else {
                o.x = fmaxf(fmaxf(in[0], in[1]), in[2]);
                o.y = fmaxf(fmaxf(in[1], in[2]), in[3]);
                o.z = fmaxf(fmaxf(in[2], in[3]), in[4]);
                o.w = fmaxf(fmaxf(in[3], in[4]), in[5]);
            }
            *reinterpret_cast<float4 *>(hm + yy * NMS_TW + q4) = o;
        }
    } else {
        for (int yy = ly; yy < TH; yy += 4) {
            const float *t = tile + yy * TW + lx;
            float m = t[0];
            for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, t[d]);
            hm[yy * NMS_TW + lx] = m;
        }
    }
    __syncthreads();
    const float thr = rel_thresh * key_value(image_max);
    const int py0 = ly * 4;  // this lane finishes tile rows py0..py0+3 of column lx
    const float *hc = hm + lx;
    float above[4], rest[4];
    if (RAD > 0) {
        float col[4 + 2 * (RAD > 0 ? RAD : 1)];   // row maxima of window rows py0 .. py0+3+2r
#pragma unroll
        for (int d = 0; d < 4 + 2 * RAD; ++d) col[d] = hc[(py0 + d) * NMS_TW];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            above[j] = ninf; rest[j] = ninf;
#pragma unroll
            for (int d = 0; d < RAD; ++d) above[j] = fmaxf(above[j], col[j + d]);
#pragma unroll
            for (int d = RAD; d <= 2 * RAD; ++d) rest[j] = fmaxf(rest[j], col[j + d]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *hj = hc + (py0 + j) * NMS_TW;
            above[j] = ninf; rest[j] = ninf;
            for (int d = 0; d < r; ++d) above[j] = fmaxf(above[j], hj[d * NMS_TW]);
            for (int d = r; d <= 2 * r; ++d) rest[j] = fmaxf(rest[j], hj[d * NMS_TW]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gx = x0 + lx, gy = y0 + py0 + j;
        const float *c = tile + (py0 + j + r) * TW + lx + r;
        const float v = c[0];
        bool ok = gx < w && gy < h && v > thr && v >= rest[j] && v > above[j];
        if (ok) {
            float left = ninf;
            for (int d = 1; d <= r; ++d) left = fmaxf(left, c[-d]);
            ok = v > left;
        }
        if (ok) {
            const uint32_t idx = (uint32_t)(gy * w + gx);
            const int slot = atomicAdd(&s_n, 1);
            if (slot < NMS_MAX_SLOTS) s_keys[slot] = ((unsigned long long)ordered_key(v) << 32) | (0xffffffffu - idx);
        }
    }
    __syncthreads();
    {
        const int n = min(s_n, NMS_MAX_SLOTS);
        const int64_t tile_id = ((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int slot = threadIdx.x; slot < tile_slots; slot += 256)
            cand[tile_id * tile_slots + slot] = slot < n ? s_keys[slot] : 0ull;
    }
}

template <int RAD>
static void launch_nms_tile(hipStream_t stream, dim3 grid, size_t lds, const float *R, int w, int h, int radius, float rel_thresh,
                            const uint32_t *maxkey, unsigned long long *cand, int tile_slots)
{
    hipLaunchKernelGGL(nms_tile_kernel<RAD>, grid, dim3(256), lds, stream, R, w, h, radius, rel_thresh, maxkey, cand, tile_slots);
}

// Top-K of an image's candidate keys, strongest first.  One workgroup per image: the filled slots are
// collected in LDS when they fit (TOPK_POOL keys; otherwise every pass re-reads the slots from memory), the
// K-th largest key is found by radix selection (8 x 8-bit passes), the winners are gathered and
// bitonic-sorted.  Keys are unique, so the result does not depend on the order of the list.
// Position in a list for every lane with `take` set (-1 otherwise): one LDS atomic per wavefront.
__device__ __forceinline__ int wave_append(bool take, int *counter)
{
    const unsigned long long m = __ballot(take);
    if (m == 0) return -1;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    int base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(counter, __builtin_popcountll(m));
    base = __shfl(base, __builtin_ctzll(m));
    return take ? base + __builtin_popcountll(m & ((1ull << lane) - 1ull)) : -1;
}

template <bool POOLED>
__device__ __forceinline__ void topk_body(const unsigned long long *__restrict__ c, const unsigned long long *pool, int n,
                                          int valid, int w,
                                          int K, unsigned long long *sel, int *hist, int *wave_tot,
                                          unsigned long long *s_prefix, int *s_need,
                                          int *s_nsel, float *__restrict__ xy, float *__restrict__ scores,
                                          int *__restrict__ count, int b)
{
    const int t = threadIdx.x;
    const int m = min(valid, K);
    unsigned long long T = 1;  // K-th largest key (everything >= T is selected); 1 = every filled slot
    if (valid > K) {
        if (t == 0) { *s_prefix = 0; *s_need = K; }
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (t < 256) hist[t] = 0;
            __syncthreads();
            const unsigned long long prefix = *s_prefix;
            const unsigned long long himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
            for (int i = t; i < n; i += TOPK_THREADS) {
                const unsigned long long key = POOLED ? pool[i] : c[i];
                if ((POOLED || key != 0ull) && (key & himask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
            }
            __syncthreads();
            // the bin that holds the need-th largest key: suffix sums of the histogram over the first
            // 256 lanes (shuffles inside a wavefront, the wavefront totals through LDS); bin b is the
            // one with suffix(b) >= need > suffix(b + 1)
            const int need = *s_need;
            int suf = 0;
            if (t < 256) {
                suf = hist[t];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_down(suf, o);
                    if ((t & 63) + o < 64) suf += up;
                }
                if ((t & 63) == 0) wave_tot[t >> 6] = suf;
            }
            __syncthreads();
            if (t < 256) {
                const int own = hist[t];
                for (int wv = (t >> 6) + 1; wv < 4; ++wv) suf += wave_tot[wv];
                if (suf >= need && suf - own < need) {      // exactly one lane (need >= 1, total >= need)
                    *s_need = need - (suf - own);
                    *s_prefix = prefix | ((unsigned long long)t << shift);
                }
            }
            __syncthreads();
        }
        T = *s_prefix;
    }
    if (t == 0) *s_nsel = 0;
    for (int i = t; i < TOPK_MAX; i += TOPK_THREADS) sel[i] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += TOPK_THREADS) {
        const int i = i0 + t;
        const unsigned long long key = i < n ? (POOLED ? pool[i] : c[i]) : 0ull;
        const int pos = wave_append(key >= T && key != 0ull, s_nsel);
        if (pos >= 0 && pos < TOPK_MAX) sel[pos] = key;
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

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const unsigned long long *__restrict__ cand, int cap, int w, int K,
                                                            float *__restrict__ xy, float *__restrict__ scores,
                                                            int *__restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long pool[];   // TOPK_POOL keys
    __shared__ unsigned long long sel[TOPK_MAX];
    __shared__ int hist[256];
    __shared__ int wave_tot[4];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_need, s_nsel;
    const int b = blockIdx.x, t = threadIdx.x;
    const unsigned long long *c = cand + (int64_t)b * cap;
    if (t == 0) s_nsel = 0;
    __syncthreads();
    // one pass over the slots, up to 16 reads in flight per lane: filled ones go to the pool (in no
    // particular order) and are counted
    for (int i0 = 0; i0 < cap; i0 += 16 * TOPK_THREADS) {
        unsigned long long k16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + u * TOPK_THREADS + t;
            k16[u] = i < cap ? c[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int pos = wave_append(k16[u] != 0ull, &s_nsel);
            if (pos >= 0 && pos < TOPK_POOL) pool[pos] = k16[u];
        }
    }
    __syncthreads();
    const int valid = s_nsel;
    __syncthreads();
    if (valid <= TOPK_POOL)
        topk_body<true>(c, pool, valid, valid, w, K, sel, hist, wave_tot, &s_prefix, &s_need, &s_nsel, xy, scores, count, b);
    else
        topk_body<false>(c, pool, cap, valid, w, K, sel, hist, wave_tot, &s_prefix, &s_need, &s_nsel, xy, scores, count, b);
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
            VO_UPLOAD(ctx, (void *)d_img, img, npix * batch);
        } else {
            for (int b = 0; b < batch; ++b)
                VO_UPLOAD_2D(ctx, d_img + b * npix, img + (size_t)b * img_stride, row_stride, width, height);
        }
        pitch = width; istride = (int64_t)npix;
    }
    dim3 grid((width + HR_TW - 1) / HR_TW, (height + HR_TH - 1) / HR_TH, batch);
    VO_REQUIRE(grid.y <= 65535u, "vo_harris_response: image too tall");
    hipLaunchKernelGGL(harris_response_kernel, grid, dim3(256), 0, ctx->stream, d_img, width, height, pitch, istride, k, mode, d_R,
                       (uint32_t *)nullptr);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, R, d_R, npix * batch * 4);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

// Device part of the selection: (max of R unless have_max), tile NMS into per-tile candidate slots, top-K.
// d_max: (batch) ordered keys MAXKEY_STRIDE words apart, zeroed by the caller and already complete when have_max is set.
static int nms_select(vo_ctx *ctx, const float *d_R, int batch, int height, int width, float rel_thresh, int radius, int K,
                      uint32_t *d_max, bool have_max, float *d_xy, float *d_sc, int *d_cnt)
{
    const size_t npix = (size_t)width * height;
    // survivors are more than `radius` apart, so a 64x16 tile holds at most this many
    const int tile_slots = ((NMS_TW + radius) / (radius + 1)) * ((NMS_TH + radius) / (radius + 1));
    dim3 grid((width + NMS_TW - 1) / NMS_TW, (height + NMS_TH - 1) / NMS_TH, batch);
    VO_REQUIRE(grid.y <= 65535u, "corner selection: image too tall");
    const int cap = (int)(grid.x * grid.y) * tile_slots;  // candidate slots per image
    int rc;
    void *p;
    if ((rc = vo_stage_reserve(ctx, 5, (size_t)batch * cap * 8, &p))) return rc;
    unsigned long long *d_cand = (unsigned long long *)p;
    if (!have_max) {
        int mblocks = (int)((npix / 4 + 256 * 4 - 1) / (256 * 4));
        if (mblocks > 512) mblocks = 512;
        if (mblocks < 1) mblocks = 1;
        hipLaunchKernelGGL(max_key_kernel, dim3(mblocks, batch), dim3(256), 0, ctx->stream, d_R, (int64_t)npix, d_max);
    }
    const int tile_w = radius <= 8 ? NMS_TW - 4 + 4 * ((4 + 2 * radius + 3) / 4) : NMS_TW + 2 * radius;   // as in the kernel
    const size_t nms_lds = ((size_t)(NMS_TH + 2 * radius) * tile_w + (size_t)(NMS_TH + 2 * radius) * NMS_TW) * 4;
    {
        typedef void (*launch_fn)(hipStream_t, dim3, size_t, const float *, int, int, int, float, const uint32_t *,
                                  unsigned long long *, int);
        static const launch_fn fixed[9] = {launch_nms_tile<0>, launch_nms_tile<1>, launch_nms_tile<2>, launch_nms_tile<3>,
                                           launch_nms_tile<4>, launch_nms_tile<5>, launch_nms_tile<6>, launch_nms_tile<7>,
                                           launch_nms_tile<8>};
        (radius <= 8 ? fixed[radius] : fixed[0])(ctx->stream, grid, nms_lds, d_R, width, height, radius, rel_thresh,
                                                 (const uint32_t *)d_max, d_cand, tile_slots);
    }
    static bool pool_enabled[64] = {false};   // per device: the kernel's dynamic LDS exceeds the default per-launch limit
    if (ctx->device < 0 || ctx->device >= 64 || !pool_enabled[ctx->device]) {
        VO_HIP_CHECK(hipFuncSetAttribute((const void *)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TOPK_POOL * 8));
        if (ctx->device >= 0 && ctx->device < 64) pool_enabled[ctx->device] = true;
    }
    hipLaunchKernelGGL(topk_kernel, dim3(batch), dim3(TOPK_THREADS), TOPK_POOL * 8, ctx->stream,
                       (const unsigned long long *)d_cand, cap, width, K, d_xy, d_sc, d_cnt);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

static int corners_copy_out(vo_ctx *ctx, int batch, int K, const float *d_xy, const float *d_sc, const int *d_cnt, float *xy,
                            float *scores, int32_t *count)
{
    VO_RESULT(ctx, xy, d_xy, (size_t)batch * K * 8);
    if (scores) VO_RESULT(ctx, scores, d_sc, (size_t)batch * K * 4);
    VO_RESULT(ctx, count, d_cnt, (size_t)batch * 4);
    VO_RESULTS_FINISH(ctx);
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
        VO_UPLOAD(ctx, (void *)d_R, R, npix * batch * 4);
    }
    if ((rc = vo_stage_reserve(ctx, 6, (size_t)batch * MAXKEY_STRIDE * 4, &p))) return rc;
    VO_HIP_CHECK(hipMemsetAsync(p, 0, (size_t)batch * MAXKEY_STRIDE * 4, ctx->stream));
    if ((rc = nms_select(ctx, d_R, batch, height, width, rel_thresh, radius, K, (uint32_t *)p, false, d_xy, d_sc, d_cnt))) return rc;
    if (mem == VO_MEM_HOST) return corners_copy_out(ctx, batch, K, d_xy, d_sc, d_cnt, xy, scores, count);
    return VO_OK;
}

extern "C" int vo_detect_corners(vo_ctx *ctx, const uint8_t *img, int batch, int height, int width, int64_t row_stride,
                                 int64_t img_stride, float k, int mode, float rel_thresh, int radius, int K, float *xy,
                                 float *scores, int32_t *count, int mem)
{
    VO_REQUIRE(ctx && img && xy && count, "vo_detect_corners: NULL argument");
    VO_REQUIRE(batch >= 1 && batch <= 65535 && height >= 3 && width >= 3, "vo_detect_corners: bad shape batch=%d h=%d w=%d", batch, height, width);
    VO_REQUIRE((int64_t)height * width < (1ll << 31), "vo_detect_corners: image too large");
    VO_REQUIRE(mode == 0 || mode == 1, "vo_detect_corners: mode %d (0 = Harris, 1 = min eigenvalue)", mode);
    VO_REQUIRE(row_stride >= width && (batch == 1 || img_stride >= row_stride * height), "vo_detect_corners: bad strides");
    VO_REQUIRE(radius >= 1 && radius <= 32, "vo_detect_corners: radius %d out of [1,32]", radius);
    VO_REQUIRE(K >= 1 && K <= TOPK_MAX, "vo_detect_corners: K=%d out of [1,%d]", K, TOPK_MAX);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_detect_corners: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    int rc;
    void *p;
    const size_t npix = (size_t)width * height;
    const uint8_t *d_img = img;
    int64_t pitch = row_stride, istride = img_stride;
    float *d_xy = xy, *d_sc = scores;
    int *d_cnt = count;
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 0, npix * batch, &p))) return rc; d_img = (const uint8_t *)p;
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)batch * K * 8, &p))) return rc; d_xy = (float *)p;
        if ((rc = vo_stage_reserve(ctx, 3, (size_t)batch * K * 4, &p))) return rc; d_sc = scores ? (float *)p : nullptr;
        if ((rc = vo_stage_reserve(ctx, 4, (size_t)batch * 4, &p))) return rc; d_cnt = (int *)p;
        if (row_stride == width && (batch == 1 || img_stride == (int64_t)npix)) {
            VO_UPLOAD(ctx, (void *)d_img, img, npix * batch);
        } else {
            for (int b = 0; b < batch; ++b)
                VO_UPLOAD_2D(ctx, d_img + b * npix, img + (size_t)b * img_stride, row_stride, width, height);
        }
        pitch = width; istride = (int64_t)npix;
    }
    if ((rc = vo_stage_reserve(ctx, 1, npix * batch * 4, &p))) return rc;     // library-owned response map
    float *d_R = (float *)p;
    if ((rc = vo_stage_reserve(ctx, 6, (size_t)batch * MAXKEY_STRIDE * 4, &p))) return rc;
    uint32_t *d_max = (uint32_t *)p;
    VO_HIP_CHECK(hipMemsetAsync(d_max, 0, (size_t)batch * MAXKEY_STRIDE * 4, ctx->stream));
    dim3 grid((width + HR_TW - 1) / HR_TW, (height + HR_TH - 1) / HR_TH, batch);
    VO_REQUIRE(grid.y <= 65535u, "vo_detect_corners: image too tall");
    hipLaunchKernelGGL(harris_response_kernel, grid, dim3(256), 0, ctx->stream, d_img, width, height, pitch, istride, k, mode, d_R,
                       d_max);
    if ((rc = nms_select(ctx, d_R, batch, height, width, rel_thresh, radius, K, d_max, true, d_xy, d_sc, d_cnt))) return rc;
    if (mem == VO_MEM_HOST) return corners_copy_out(ctx, batch, K, d_xy, d_sc, d_cnt, xy, scores, count);
    return VO_OK;
}

// ---- corner mask of FeatureExtractor.harris ---------------------------------------------------
// mask = (3x3 maximum of R, pixels outside the image ignored) > (float)(rel_thresh * max R):
// cv.dilate + cv.threshold of vo_pipeline/featureExtraction.py:41-42 in one pass over R.
__global__ __launch_bounds__(256) void harris_mask_kernel(const float *__restrict__ R, int w, int h, double rel_thresh,
                                                          const uint32_t *__restrict__ maxkey, uint8_t *__restrict__ mask)
{
    // a lane owns four neighbouring pixels of one row: six reads per row of the 3x3 neighbourhood, 18 per four pixels
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x0 >= w || y >= h) return;
    const float thr = (float)(rel_thresh * (double)key_value(maxkey[b * MAXKEY_STRIDE]));
    const float *img = R + (size_t)b * w * h;
    const float ninf = __uint_as_float(0xff800000u);
    float m[4] = {ninf, ninf, ninf, ninf};
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        const float *row = img + (size_t)yy * w;
        float v[6];
        if (x0 >= 1 && x0 + 5 <= w) {
            __builtin_memcpy(v, row + x0 - 1, 24);        // interior: the six values in one unaligned read
        } else {
            for (int c = 0; c < 6; ++c) {
                const int xx = x0 - 1 + c;
                v[c] = (xx >= 0 && xx < w) ? row[xx] : ninf;
            }
        }
        for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], fmaxf(v[c], fmaxf(v[c + 1], v[c + 2])));
    }
    uint8_t *out = mask + (size_t)b * w * h + (size_t)y * w + x0;
    if (x0 + 4 <= w) {
        const uint32_t packed = (m[0] > thr ? 0xffu : 0u) | (m[1] > thr ? 0xff00u : 0u) | (m[2] > thr ? 0xff0000u : 0u) |
                                (m[3] > thr ? 0xff000000u : 0u);
        __builtin_memcpy(out, &packed, 4);                // rows are not dword-aligned in general: unaligned store
    } else {
        for (int c = 0; c < 4; ++c)
            if (x0 + c < w) out[c] = m[c] > thr ? 255 : 0;
    }
}

// Response + dilation + threshold on device pointers (response map and per-image maxima in context slots 1 and 6).
static int harris_mask_launch(vo_ctx *ctx, const uint8_t *d_img, int batch, int height, int width, int64_t pitch, int64_t istride,
                              float k, int mode, double rel_thresh, uint8_t *d_mask)
{
    int rc;
    void *p;
    const size_t npix = (size_t)width * height;
    if ((rc = vo_stage_reserve(ctx, 1, npix * batch * 4, &p))) return rc;     // library-owned response map
    float *d_R = (float *)p;
    if ((rc = vo_stage_reserve(ctx, 6, (size_t)batch * MAXKEY_STRIDE * 4, &p))) return rc;
    uint32_t *d_max = (uint32_t *)p;
    VO_HIP_CHECK(hipMemsetAsync(d_max, 0, (size_t)batch * MAXKEY_STRIDE * 4, ctx->stream));
    dim3 grid((width + HR_TW - 1) / HR_TW, (height + HR_TH - 1) / HR_TH, batch);
    VO_REQUIRE(grid.y <= 65535u, "vo_harris_mask: image too tall");
    hipLaunchKernelGGL(harris_response_kernel, grid, dim3(256), 0, ctx->stream, d_img, width, height, pitch, istride, k, mode, d_R,
                       d_max);
    dim3 mgrid((width + 255) / 256, (height + 3) / 4, batch);
    VO_REQUIRE(mgrid.y <= 65535u, "vo_harris_mask: image too tall");
    hipLaunchKernelGGL(harris_mask_kernel, mgrid, dim3(256), 0, ctx->stream, d_R, width, height, rel_thresh, d_max, d_mask);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

int vo_harris_mask_device(vo_ctx *ctx, const uint8_t *d_img, int height, int width, int64_t pitch, float k, int mode,
                          double rel_thresh, uint8_t *d_mask)
{
    VO_REQUIRE((int64_t)height * width < (1ll << 31), "harris mask: image too large");
    return harris_mask_launch(ctx, d_img, 1, height, width, pitch, pitch * height, k, mode, rel_thresh, d_mask);
}

extern "C" int vo_harris_mask(vo_ctx *ctx, const uint8_t *img, int batch, int height, int width, int64_t row_stride,
                              int64_t img_stride, float k, int mode, double rel_thresh, uint8_t *mask, int mem)
{
    VO_REQUIRE(ctx && img && mask, "vo_harris_mask: NULL argument");
    VO_REQUIRE(batch >= 1 && batch <= 65535 && height >= 3 && width >= 3, "vo_harris_mask: bad shape batch=%d h=%d w=%d", batch, height, width);
    VO_REQUIRE((int64_t)height * width < (1ll << 31), "vo_harris_mask: image too large");
    VO_REQUIRE(mode == 0 || mode == 1, "vo_harris_mask: mode %d (0 = Harris, 1 = min eigenvalue)", mode);
    VO_REQUIRE(row_stride >= width && (batch == 1 || img_stride >= row_stride * height), "vo_harris_mask: bad strides");
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_harris_mask: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    int rc;
    void *p;
    const size_t npix = (size_t)width * height;
    const uint8_t *d_img = img;
    uint8_t *d_mask = mask;
    int64_t pitch = row_stride, istride = img_stride;
    if (mem == VO_MEM_HOST) {
        if ((rc = vo_stage_reserve(ctx, 0, npix * batch, &p))) return rc; d_img = (const uint8_t *)p;
        if ((rc = vo_stage_reserve(ctx, 2, npix * batch, &p))) return rc; d_mask = (uint8_t *)p;
        if (row_stride == width && (batch == 1 || img_stride == (int64_t)npix)) {
            VO_UPLOAD(ctx, (void *)d_img, img, npix * batch);
        } else {
            for (int b = 0; b < batch; ++b)
                VO_UPLOAD_2D(ctx, d_img + b * npix, img + (size_t)b * img_stride, row_stride, width, height);
        }
        pitch = width; istride = (int64_t)npix;
    }
    if ((rc = harris_mask_launch(ctx, d_img, batch, height, width, pitch, istride, k, mode, rel_thresh, d_mask))) return rc;
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, mask, d_mask, npix * batch);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_harris(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)harris_response_kernel));
    return VO_OK;
}
