// Brute-force two-nearest-neighbour matching of 128-dimensional descriptors.
//
// Replaces FeatureMatcher.match_descriptors -> matcher.knnMatch(des1, des2, k=2)
// (vo_pipeline/featureMatching.py:52-63; exact search, i.e. what cv.BFMatcher()
// of featureMatching.py:45 returns and what the FLANN kd-trees of :46-50
// approximate) and the ratio test of match_filter (featureMatching.py:65-69).
// Distance of a pair = sqrtf(sum_k (a_k - b_k)^2), the float32 NORM_L2 of
// BFMatcher; equal distances go to the lower train index.
//
// Two kernels produce the same candidate keys {float bits of the squared distance,
// train index} and only one of them does work in a call:
//  * knn2_mfma_kernel -- descriptors whose entries are integers in [0, 255] (what
//    SIFT emits: float32 holding saturated bytes).  Such values are exact in bf16,
//    their products and the 128-term sums are exact in float32 (< 2^24), so
//    |a|^2 + |b|^2 - 2 a.b from the matrix cores equals the direct sum bit for bit.
//  * knn2_f32_kernel -- anything else: direct differences, one fused multiply-add
//    per term in ascending k, the summation order of the CPU checker.
// desc_pack_kernel decides which: it converts both sets to bf16, takes the norms
// and raises a device flag at the first entry that is not such an integer.
#include "vo_internal.h"

#define DIM 128
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned long long u64;

#define KEY_NONE (~0ull)

__device__ __forceinline__ void top2_insert(u64 key, u64 &b0, u64 &b1)
{
    const u64 lo = key < b0 ? key : b0;
    const u64 hi = key < b0 ? b0 : key;
    b0 = lo;
    b1 = hi < b1 ? hi : b1;
}

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask)
{
    const unsigned lo = __shfl_xor((unsigned)v, mask, 64);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), mask, 64);
    return ((u64)hi << 32) | lo;
}

// rows [0, n) of a set -> bf16 rows (padding rows zero), squared norms, integer flag.
// 16 lanes per row, 8 entries per lane.
// One launch packs both sets: workgroups [0, blocks1) take the query set, the rest the train set.
__global__ __launch_bounds__(256) void desc_pack_kernel(const float *__restrict__ src1, int n1, int blocks1,
                                                        unsigned short *__restrict__ dst1, float *__restrict__ norm1,
                                                        const float *__restrict__ src2, int n2,
                                                        unsigned short *__restrict__ dst2, float *__restrict__ norm2,
                                                        unsigned *__restrict__ not_bytes)
{
    const bool second = (int)blockIdx.x >= blocks1;
    const float *__restrict__ src = second ? src2 : src1;
    unsigned short *__restrict__ dst = second ? dst2 : dst1;
    float *__restrict__ norm = second ? norm2 : norm1;
    const int n = second ? n2 : n1;
    const float pad_norm = second ? __uint_as_float(0x7f800000u) : 0.f;   // padding trains are infinitely far
    const int row = ((int)blockIdx.x - (second ? blocks1 : 0)) * 16 + (threadIdx.x >> 4);
    const int c = (threadIdx.x & 15) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < n) {
        const float4 a = *(const float4 *)(src + (int64_t)row * DIM + c);
        const float4 b = *(const float4 *)(src + (int64_t)row * DIM + c + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    bool bad = false;
    float s = 0.f;
    unsigned short h[8];
    for (int j = 0; j < 8; ++j) {
        bad |= !(v[j] >= 0.f && v[j] <= 255.f && v[j] == floorf(v[j]));
        s = fmaf(v[j], v[j], s);
        h[j] = (unsigned short)(__float_as_uint(v[j]) >> 16);
    }
    uint4 o;
    o.x = h[0] | ((unsigned)h[1] << 16);
    o.y = h[2] | ((unsigned)h[3] << 16);
    o.z = h[4] | ((unsigned)h[5] << 16);
    o.w = h[6] | ((unsigned)h[7] << 16);
    // fragment order of the 32x32x16 matrix-core product: [tile of 32 rows][k-step s][k-half h][row in tile][8 entries],
    // so that the 64 lanes of a wavefront read one contiguous KB per k-step
    const int chunk = threadIdx.x & 15;
    *(uint4 *)(dst + ((((int64_t)(row >> 5) * 8 + (chunk >> 1)) * 2 + (chunk & 1)) * 32 + (row & 31)) * 8) = o;
    for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 15) == 0) norm[row] = row < n ? s : pad_norm;
    // one wavefront raises the flag; the others see it set and skip the atomic (thousands of atomics on one
    // word took longer than the rest of the kernel).  A stale read only costs a redundant atomic.
    if (__any(bad) && (threadIdx.x & 63) == 0 && *(volatile unsigned *)not_bytes == 0) atomicOr(not_bytes, 1u);
}

// Matrix-core path.  A wavefront owns 64 queries (two operands B, held in registers for the
// whole launch) and walks its split of the train set in tiles of 32 (operand A, read
// straight from global memory, 16 contiguous bytes per lane and k-step; the reads of the
// next tile are issued before the products of the current one).  The result tile has the
// query on the lane and 16 trains in the registers, so each lane keeps the two best keys
// of its query without cross-lane traffic; the two lane halves are merged once at the end.
// A lane visits its trains in ascending order, so a candidate can only enter the pair when
// its squared distance is below the second best: the wavefront tests the minimum of the
// 16 new values first and touches the keys only for elements some lane accepts.  Padding
// trains carry an infinite norm and never pass that test.
struct TrainTile {
    bf16x8 a[8];
    float4 tn[4];
};

__device__ __forceinline__ void load_train_tile(TrainTile &t, const unsigned short *__restrict__ T, const float *__restrict__ nt,
                                                int tile, int lane, int h)
{
    for (int s = 0; s < 8; ++s) t.a[s] = *(const bf16x8 *)(T + (((int64_t)tile * 8 + s) * 64 + lane) * 8);
    for (int g = 0; g < 4; ++g) t.tn[g] = *(const float4 *)(nt + tile * 32 + 8 * g + 4 * h);
}

// 32 trains against the wavefront's 64 queries: 16 products, then the two-best update of each lane's two queries.
__device__ __forceinline__ void match_tile(const TrainTile &t, const bf16x8 (&b)[2][8], const float (&qq)[2], int t0, int h,
                                           u64 (&b0)[2], u64 (&b1)[2], float (&thr)[2])
{
    f32x16 acc[2];
    for (int u = 0; u < 2; ++u)
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;
    for (int s = 0; s < 8; ++s)
        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.a[s], b[u][s], acc[u], 0, 0, 0);
    for (int u = 0; u < 2; ++u) {
        float d2[16];
        for (int g = 0; g < 4; ++g) {
            d2[4 * g] = fmaf(-2.f, acc[u][4 * g], qq[u] + t.tn[g].x);
            d2[4 * g + 1] = fmaf(-2.f, acc[u][4 * g + 1], qq[u] + t.tn[g].y);
            d2[4 * g + 2] = fmaf(-2.f, acc[u][4 * g + 2], qq[u] + t.tn[g].z);
            d2[4 * g + 3] = fmaf(-2.f, acc[u][4 * g + 3], qq[u] + t.tn[g].w);
        }
        float m = fminf(d2[0], d2[1]);
        for (int e = 2; e < 16; e += 2) m = fminf(m, fminf(d2[e], d2[e + 1]));
        if (__any(m < thr[u])) {
            // all 16 wave-wide tests up front (against the threshold before any update: a superset of the hits),
            // so the scalar unit walks a list of masks instead of waiting for one compare per element
            u64 hit[16];
            for (int e = 0; e < 16; ++e) hit[e] = __ballot(d2[e] < thr[u]);
            for (int e = 0; e < 16; ++e) {
                if (hit[e]) {
                    const int tr = t0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const u64 key = d2[e] < thr[u] ? (((u64)__float_as_uint(d2[e]) << 32) | (unsigned)tr) : KEY_NONE;
                    top2_insert(key, b0[u], b1[u]);
                    thr[u] = __uint_as_float(min((unsigned)(b1[u] >> 32), 0x7f800000u));
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void knn2_mfma_kernel(const unsigned short *__restrict__ Q, const float *__restrict__ nq,
                                                        const unsigned short *__restrict__ T, const float *__restrict__ nt,
                                                        int n1, int n2, int tiles_per_split,
                                                        const unsigned *__restrict__ not_bytes, u64 *__restrict__ part)
{
    const unsigned other_path = __builtin_nontemporal_load(not_bytes);   // tested after the operand reads are issued
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qt = (blockIdx.x * 4 + wave) * 2;      // first of the wavefront's two query tiles; n1 is padded to 256
    bf16x8 b[2][8];
    float qq[2];
    for (int u = 0; u < 2; ++u) {
        for (int s = 0; s < 8; ++s) b[u][s] = *(const bf16x8 *)(Q + (((int64_t)(qt + u) * 8 + s) * 64 + lane) * 8);
        qq[u] = nq[(qt + u) * 32 + r];
    }
    const int n_tiles = (n2 + 31) >> 5;
    const int tile0 = blockIdx.y * tiles_per_split;
    const int tile1 = min(tile0 + tiles_per_split, n_tiles);
    u64 b0[2] = {KEY_NONE, KEY_NONE}, b1[2] = {KEY_NONE, KEY_NONE};
    float thr[2] = {__uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u)};
    // two tiles in flight in two register sets (no copies); reads past the split's end repeat its last tile
    TrainTile ta, tb;
    if (tile0 < tile1) load_train_tile(ta, T, nt, tile0, lane, h);
    if (other_path) return;
    for (int tile = tile0; tile < tile1; tile += 2) {
        load_train_tile(tb, T, nt, min(tile + 1, tile1 - 1), lane, h);
        match_tile(ta, b, qq, tile * 32, h, b0, b1, thr);
        load_train_tile(ta, T, nt, min(tile + 2, tile1 - 1), lane, h);
        if (tile + 1 < tile1) match_tile(tb, b, qq, (tile + 1) * 32, h, b0, b1, thr);
    }
    for (int u = 0; u < 2; ++u) {
        const u64 o0 = shfl_xor_u64(b0[u], 32), o1 = shfl_xor_u64(b1[u], 32);
        top2_insert(o0, b0[u], b1[u]);
        top2_insert(o1, b0[u], b1[u]);
        const int q = (qt + u) * 32 + r;
        if (h == 0 && q < n1) {
            part[((int64_t)blockIdx.y * n1 + q) * 2] = b0[u];
            part[((int64_t)blockIdx.y * n1 + q) * 2 + 1] = b1[u];
        }
    }
}

// General float path: 64 queries x 64 trains per workgroup step, both tiles in LDS with a
// row pitch of 132 floats (16-byte reads of 16 different rows land in different banks);
// a lane owns a 4x4 block of pairs and adds the 128 terms of each in ascending k.  The kernel runs at the
// float32 vector rate (one subtract and one fused multiply-add per term); packed float32 instructions
// (v_pk_add_f32 / v_pk_fma_f32 over a transposed query tile) were tried and take the same time.
#define FP 132
__global__ __launch_bounds__(256) void knn2_f32_kernel(const float *__restrict__ D1, const float *__restrict__ D2, int n1, int n2,
                                                       int tiles_per_split, const unsigned *__restrict__ not_bytes,
                                                       u64 *__restrict__ part)
{
    if (!*not_bytes) return;
    __shared__ float qs[64 * FP];
    __shared__ float ts[64 * FP];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.x * 64;
    for (int e = 0; e < 8; ++e) {
        const int i = e * 256 + tid, row = i >> 5, c4 = i & 31;
        const int src = min(q0 + row, n1 - 1);
        *(float4 *)(qs + row * FP + 4 * c4) = *(const float4 *)(D1 + (int64_t)src * DIM + 4 * c4);
    }
    const int n_tiles = (n2 + 63) >> 6;
    const int tile0 = blockIdx.y * tiles_per_split;
    const int tile1 = min(tile0 + tiles_per_split, n_tiles);
    u64 b0[4], b1[4];
    for (int i = 0; i < 4; ++i) { b0[i] = KEY_NONE; b1[i] = KEY_NONE; }
    for (int tile = tile0; tile < tile1; ++tile) {
        const int t0 = tile * 64;
        __syncthreads();  // previous tile fully consumed (and the query tile stored)
        for (int e = 0; e < 8; ++e) {
            const int i = e * 256 + tid, row = i >> 5, c4 = i & 31;
            const int src = min(t0 + row, n2 - 1);
            *(float4 *)(ts + row * FP + 4 * c4) = *(const float4 *)(D2 + (int64_t)src * DIM + 4 * c4);
        }
        __syncthreads();
        float acc[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int k4 = 0; k4 < 32; ++k4) {
            float4 qv[4], tv[4];
            for (int i = 0; i < 4; ++i) qv[i] = *(const float4 *)(qs + (ty * 4 + i) * FP + 4 * k4);
            for (int j = 0; j < 4; ++j) tv[j] = *(const float4 *)(ts + (tx + 16 * j) * FP + 4 * k4);
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    float d;
                    d = qv[i].x - tv[j].x; acc[i][j] = fmaf(d, d, acc[i][j]);
                    d = qv[i].y - tv[j].y; acc[i][j] = fmaf(d, d, acc[i][j]);
                    d = qv[i].z - tv[j].z; acc[i][j] = fmaf(d, d, acc[i][j]);
                    d = qv[i].w - tv[j].w; acc[i][j] = fmaf(d, d, acc[i][j]);
                }
        }
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + tx + 16 * j;
            for (int i = 0; i < 4; ++i) {
                const u64 key = t < n2 ? (((u64)__float_as_uint(acc[i][j]) << 32) | (unsigned)t) : KEY_NONE;
                top2_insert(key, b0[i], b1[i]);
            }
        }
    }
    for (int i = 0; i < 4; ++i) {
        for (int m = 1; m < 16; m <<= 1) {
            const u64 o0 = shfl_xor_u64(b0[i], m), o1 = shfl_xor_u64(b1[i], m);
            top2_insert(o0, b0[i], b1[i]);
            top2_insert(o1, b0[i], b1[i]);
        }
        const int q = q0 + ty * 4 + i;
        if (tx == 0 && q < n1) {
            part[((int64_t)blockIdx.y * n1 + q) * 2] = b0[i];
            part[((int64_t)blockIdx.y * n1 + q) * 2 + 1] = b1[i];
        }
    }
}

// Sixteen lanes per query share the splits; best two keys, distances, ratio test.
__global__ __launch_bounds__(256) void knn2_merge_kernel(const u64 *__restrict__ part, int n1, int splits_bytes,
                                                         int splits_float, const unsigned *__restrict__ not_bytes, double ratio,
                                                         int32_t *__restrict__ idx, float *__restrict__ dist,
                                                         uint8_t *__restrict__ keep)
{
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int splits = *not_bytes ? splits_float : splits_bytes;
    u64 b0 = KEY_NONE, b1 = KEY_NONE;
    if (q < n1)
        for (int s = sub; s < splits; s += 16) {
            const ulonglong2 k = *(const ulonglong2 *)(part + ((int64_t)s * n1 + q) * 2);
            top2_insert(k.x, b0, b1);
            top2_insert(k.y, b0, b1);
        }
    for (int m = 1; m < 16; m <<= 1) {
        const u64 o0 = shfl_xor_u64(b0, m), o1 = shfl_xor_u64(b1, m);
        top2_insert(o0, b0, b1);
        top2_insert(o1, b0, b1);
    }
    if (q >= n1 || sub != 0) return;
    const float inf = __uint_as_float(0x7f800000u);
    const float d0 = b0 == KEY_NONE ? inf : sqrtf(__uint_as_float((unsigned)(b0 >> 32)));
    const float d1 = b1 == KEY_NONE ? inf : sqrtf(__uint_as_float((unsigned)(b1 >> 32)));
    idx[2 * (int64_t)q] = b0 == KEY_NONE ? -1 : (int32_t)(unsigned)b0;
    idx[2 * (int64_t)q + 1] = b1 == KEY_NONE ? -1 : (int32_t)(unsigned)b1;
    dist[2 * (int64_t)q] = d0;
    dist[2 * (int64_t)q + 1] = d1;
    if (keep) keep[q] = (b1 != KEY_NONE && (double)d0 < ratio * (double)d1) ? 1 : 0;
}

static int div_up(int a, int b) { return (a + b - 1) / b; }

extern "C" int vo_knn_match2(vo_ctx *ctx, const float *des1, int n1, const float *des2, int n2, int dim, double ratio,
                             int32_t *idx, float *dist, uint8_t *keep, int mem)
{
    VO_REQUIRE(ctx != nullptr, "vo_knn_match2: ctx is NULL");
    VO_REQUIRE(dim == DIM, "vo_knn_match2: descriptors of %d entries (this library matches 128-entry descriptors)", dim);
    VO_REQUIRE(n1 >= 0 && n2 >= 0, "vo_knn_match2: n1=%d n2=%d", n1, n2);
    VO_REQUIRE(n1 <= (1 << 22) && n2 <= (1 << 22), "vo_knn_match2: at most %d descriptors per set", 1 << 22);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_knn_match2: bad mem flag %d", mem);
    if (n1 == 0) return VO_OK;
    VO_REQUIRE(des1 && idx && dist && (des2 || n2 == 0), "vo_knn_match2: NULL buffer");
    VO_REQUIRE(mem == VO_MEM_HOST || (((uintptr_t)des1 & 15) == 0 && ((uintptr_t)des2 & 15) == 0),
               "vo_knn_match2: device descriptors must be 16-byte aligned");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    void *p;
    const float *d1 = des1, *d2 = des2;
    int32_t *d_idx = idx;
    float *d_dist = dist;
    uint8_t *d_keep = keep;
    if (mem == VO_MEM_HOST) {
        const size_t b1 = (size_t)n1 * DIM * 4, b2 = (size_t)n2 * DIM * 4;
        if ((rc = vo_stage_reserve(ctx, 0, b1, &p))) return rc; d1 = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, b2, &p))) return rc; d2 = (const float *)p;
        // idx (n1,2) int32 | dist (n1,2) float | keep (n1)
        if ((rc = vo_stage_reserve(ctx, 2, (size_t)n1 * 17, &p))) return rc;
        d_idx = (int32_t *)p;
        d_dist = (float *)((char *)p + (size_t)n1 * 8);
        d_keep = keep ? (uint8_t *)p + (size_t)n1 * 16 : nullptr;
        VO_UPLOAD(ctx, (void *)d1, des1, b1);
        if (n2) VO_UPLOAD(ctx, (void *)d2, des2, b2);
    }
    const int n1p = div_up(n1, 256) * 256, n2p = div_up(n2 > 0 ? n2 : 1, 64) * 64;
    // split the train set so that each launch fills the device once: 256 CUs x 2 resident workgroups of either kernel
    const int tiles64 = n2p / 64, tiles32 = n2p / 32;
    int splits = 512 / div_up(n1, 64);
    if (splits < 1) splits = 1;
    if (splits > tiles64) splits = tiles64;
    const int tiles64_per_split = div_up(tiles64, splits);
    splits = div_up(tiles64, tiles64_per_split);
    int splits_m = 512 / (n1p / 256);
    if (splits_m < 1) splits_m = 1;
    if (splits_m > tiles32) splits_m = tiles32;
    const int tiles32_per_split = div_up(tiles32, splits_m);
    splits_m = div_up(tiles32, tiles32_per_split);
    const int max_splits = splits > splits_m ? splits : splits_m;
    // bf16 copies | norms | flag | per-split candidate keys
    const size_t off_t = (size_t)n1p * DIM * 2, off_nq = off_t + (size_t)n2p * DIM * 2, off_nt = off_nq + (size_t)n1p * 4,
                 off_flag = off_nt + (size_t)n2p * 4, off_part = (off_flag + 4 + 15) / 16 * 16,
                 total = off_part + (size_t)max_splits * n1 * 16;
    if ((rc = vo_stage_reserve(ctx, 3, total, &p))) return rc;
    char *w = (char *)p;
    unsigned short *Qh = (unsigned short *)w, *Th = (unsigned short *)(w + off_t);
    float *nq = (float *)(w + off_nq), *nt = (float *)(w + off_nt);
    unsigned *flag = (unsigned *)(w + off_flag);
    u64 *part = (u64 *)(w + off_part);
    VO_HIP_CHECK(hipMemsetAsync(flag, 0, 4, st));
    hipLaunchKernelGGL(desc_pack_kernel, dim3(n1p / 16 + n2p / 16), dim3(256), 0, st, d1, n1, n1p / 16, Qh, nq, d2, n2, Th, nt,
                       flag);
    if (n2 > 0) {
        hipLaunchKernelGGL(knn2_mfma_kernel, dim3(n1p / 256, splits_m), dim3(256), 0, st, Qh, nq, Th, nt, n1, n2,
                           tiles32_per_split, flag, part);
        hipLaunchKernelGGL(knn2_f32_kernel, dim3(div_up(n1, 64), splits), dim3(256), 0, st, d1, d2, n1, n2, tiles64_per_split,
                           flag, part);
    }
    hipLaunchKernelGGL(knn2_merge_kernel, dim3(div_up(n1, 16)), dim3(256), 0, st, part, n1, n2 > 0 ? splits_m : 0,
                       n2 > 0 ? splits : 0, flag, ratio, d_idx,
                       d_dist, d_keep);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, idx, d_idx, (size_t)n1 * 8);
        VO_RESULT(ctx, dist, d_dist, (size_t)n1 * 8);
        if (keep) VO_RESULT(ctx, keep, d_keep, (size_t)n1);
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_match(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)desc_pack_kernel));
    return VO_OK;
}
