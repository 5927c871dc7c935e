// Pyramidal Lucas-Kanade tracker: klt_track3_kernel (three keypoints per 64-lane
// wavefront, the default) and klt_track_kernel (one keypoint per wavefront), described
// first.
//
// Semantics: cv.calcOpticalFlowPyrLK as the reference calls it
// (vo_pipeline/poseEstimation.py:116-133): translation model, 21x21 window,
// 14-bit fixed-point bilinear weights, int16 Scharr derivatives (x32 gain),
// float32 2x2 normal equations, min-eigenvalue test, <= max_iters Gauss-Newton
// steps per level with the epsilon and oscillation stops.
//
// Mapping: lane l < 63 owns window column l % 21 and the 7 rows 7*(l / 21)...;
// its template values (I, Ix, Iy) stay in registers for the whole level.
// Per wave the kernel stages in LDS
//   * the 24x24 previous-image tile as 16-bit pixel pairs {p[x], p[x+1]} (ready
//     for v_dot2_i32_i16 against the packed bilinear weights); the Scharr
//     derivatives are formed from it in registers with packed 16-bit math, two
//     window columns per instruction, and never touch memory;
//   * a (22+2M)x(21+2M) tile of the next image around the current estimate,
//     re-staged only when the estimate leaves the M-pixel margin.
// The row stride (35 dwords) puts the three 7-row strips of a half-wave on
// disjoint LDS banks.  Per-lane sums are exact int32 (v_mad_i32_i24); the 2x2
// system and the right-hand side are reduced over the wave with DPP adds; every
// branch is wave-uniform.
#include "vo_internal.h"
#include <stdlib.h>

#define KLT_WIN 21
#define KLT_HALF 10.0f
#define KLT_M 2                         // margin of the next-image tile (pixels)
#define K3_S_ 25                        // tile row stride of the three-keypoint kernel (dwords)
#define JT_S 35                         // next-image tile row stride (dwords), 7*35 % 32 == 21
#define JT_ROWS (KLT_WIN + 1 + 2 * KLT_M)
#define JT_COLS (KLT_WIN + 2 * KLT_M)   // pair columns
#define PT_S 35                         // previous-image pair tile stride (dwords), 7*35 % 32 == 21
#define PT_ROWS 24
#define LDS_A_DWORDS (JT_S * JT_ROWS)   // one region per wave: previous-image tile, later next-image tile
#define LDS_WAVE_DWORDS ((LDS_A_DWORDS + 3) / 4 * 4)
#define KLT_WAVES 4

typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
typedef short v2i16 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct KltLevel {
    const uint8_t *prev;
    const uint8_t *next;
    int w, h;
    int64_t ppitch, pstride;  // previous image: row pitch, image stride
    int64_t npitch, nstride;  // next image
};

struct KltArgs {
    KltLevel lv[VO_MAX_LEVELS];
    int max_level;
    int n;      // keypoints per image
    int batch;  // images
    int blocks_per_image;
    const float *prev_pts;
    const float *init_pts;
    float *next_pts;
    uint8_t *status;
    float *err;
    int max_iters;
    double eps2;
    float eps2_lo, eps2_hi;  // float brackets of eps2: outside them the float test decides
    float min_eig;
};

__device__ __forceinline__ void wave_sync()
{
    // LDS operations of one wave execute in order; this only stops the compiler
    // from moving LDS accesses across the hand-over point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// Sum over the 64 lanes, returned in every lane (fixed order => deterministic).
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_f<0x140, 0xf>(v);  // row_mirror
    v += dpp_f<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3
    v += dpp_f<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sums of two values over the 64 lanes at the price of one reduction:
// v_permlane32_swap puts x's upper half beside its lower half and y's likewise, so
// after one add lanes 0-31 carry x pair sums and lanes 32-63 carry y pair sums; the
// remaining five steps stay inside the 32-lane halves.
__device__ __forceinline__ void wave_sum2(float x, float y, float &sx, float &sy)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    float v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    v += dpp_f<0xB1, 0xf>(v);
    v += dpp_f<0x4E, 0xf>(v);
    v += dpp_f<0x141, 0xf>(v);
    v += dpp_f<0x140, 0xf>(v);
    // row_bcast15 add in one instruction (rows 1 and 3 += lane 15 of rows 0 and 2)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0" : "+v"(v));
    sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ int refl(int i, int n)
{
    // BORDER_REFLECT_101 for one fold on either side; positions further out are
    // only touched by tile margins that no valid window reads, so they are clamped.
    // = clamp(i < 0 ? -i : (i >= n ? 2n - 2 - i : i), 0, n - 1): with a = |i| the fold is min(a, 2n - 2 - a), which never
    // exceeds n - 1, and max(.., 0) supplies the lower clamp (the compiler joins min and max into one median instruction)
    const int a = max(i, -i);
    return max(min(a, 2 * n - 2 - a), 0);
}

// Packed 14-bit bilinear weights {iw00, iw01} and {iw10, iw11} for the fractional
// offsets (a, b): iw = cvRound(w * 2^14), iw11 = 2^14 - iw00 - iw01 - iw10.  Adding
// 1.5 * 2^23 rounds to nearest-even and leaves the integer in the low mantissa bits,
// so everything stays on full-rate add/and/shift instructions (v_cvt_i32_f32 is half rate).
__device__ __forceinline__ void bilinear_weights(float a, float b, uint32_t &wt, uint32_t &wb)
{
    const float magic = 12582912.0f;               // 0x4B400000
    // w * 2^14 with the factor split over the two offsets: scaling by a power of two commutes with float rounding,
    // so (128 - 128 a) is 128 * fl(1 - a) and (128 a') * (128 b') is 2^14 * fl(a' b') bit for bit, three multiplications fewer
    const float a7 = a * 128.f, b7 = b * 128.f;
    const float ia = 128.f - a7, ib = 128.f - b7;
    const uint32_t t00 = __float_as_uint(ia * ib + magic);
    const uint32_t t01 = __float_as_uint(a7 * ib + magic);
    const uint32_t t10 = __float_as_uint(ia * b7 + magic);
    const uint32_t iw11 = (16384u + 3u * 0x4B400000u) - (t00 + t01 + t10);
    // low halves side by side: one byte shuffle each instead of a mask, a shift and an or
    wt = __builtin_amdgcn_perm(t01, t00, 0x05040100u);
    wb = __builtin_amdgcn_perm(iw11, t10, 0x05040100u);
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Lanes for which p holds.  A condition already lives in a scalar mask register; the compiler's own ballot of it
// is a scalar AND with the execution mask, where __ballot() first turns it into a 0/1 vector and compares that again.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Pixel pair . weight pair + c.  The product is SIGNED: the fourth weight 2^14 - iw00 - iw01 - iw10 is -1 when
// all three roundings go up (a fractional offset within ~3e-5 of an integer), and an unsigned 16-bit lane
// would read that -1 as 65535.  Pixels (<= 255) and the other weights (<= 2^14) are the same in both readings.
// Three-operand encoding written out: from the builtin the compiler picks the accumulate-in-place form
// plus a move for every accumulator that has to survive (the per-sample 256 - 512 I seeds do): +5 % kernel time.
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// the same with a wave-uniform addend, which then needs no vector register
__device__ __forceinline__ uint32_t udot2_s(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2i16, a), __builtin_bit_cast(v2i16, b), c, false);
}

// a . b + c with c left intact: the three-operand encoding, so a chain that starts from a
// constant needs no copy of it (the compiler picks the accumulate-in-place form plus a move)
__device__ __forceinline__ int sdot2_keep(uint32_t a, uint32_t b, int c)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// {low 16 bits of lo, bits [sh+15 : sh] of src}: the shifted value is written straight into the
// upper half of the register (sub-dword destination select), so packing two residuals or two
// derivatives into int16 lanes costs the two shifts and nothing else
__device__ __forceinline__ uint32_t pack_hi_shr(uint32_t lo, uint32_t src, uint32_t sh)
{
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(lo)
        : "v"(sh), "v"(src));
    return lo;
}
// the same with the shift amount in a scalar register (set once, outside the loops)
__device__ __forceinline__ uint32_t pack_hi_shr_s(uint32_t lo, uint32_t src, uint32_t sh)
{
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(lo)
        : "s"(sh), "v"(src));
    return lo;
}
// a . b: the first term of a sum, so that the accumulator needs no clearing move
__device__ __forceinline__ int sdot2_first(uint32_t a, uint32_t b)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Fifteen pixel pairs {p[x], p[x+1]} out of sixteen bytes, to consecutive LDS dwords.
__device__ __forceinline__ void store_pairs(uint32_t *d, const uint32_t (&q)[4])
{
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        const int c = j >> 2, o = j & 3;
        d[j] = o == 0   ? __builtin_amdgcn_perm(0u, q[c], 0x0c010c00u)
               : o == 1 ? __builtin_amdgcn_perm(0u, q[c], 0x0c020c01u)
               : o == 2 ? __builtin_amdgcn_perm(0u, q[c], 0x0c030c02u)
                        : __builtin_amdgcn_perm(q[c + 1], q[c], 0x0c040c03u);
    }
}

// Stage a rows x (pairs) tile of 16-bit pixel pairs {p[x], p[x+1]} with origin
// (x0, y0) at dword lds_off of the workgroup's LDS array; pixels outside the image follow BORDER_REFLECT_101.  An item is half a
// (reflected) row: one unaligned 16-byte read yields 15 pairs through v_perm_b32, and
// the second half starts G1 pixels in so that the two reads end exactly at the last
// pixel the tile needs (the pairs both halves produce are written twice with the same
// value).  One item per lane covers the whole tile.  Only items that straddle the
// left/right image border read their sixteen bytes one by one.
template <int ROWS, int PAIRS, int STRIDE>
__device__ __forceinline__ void stage_pairs(uint32_t *lds, unsigned lds_off, const uint8_t *img, int64_t pitch, int w, int h,
                                            int x0, int y0, int lane)
{
    constexpr int G1 = PAIRS + 1 - 16;  // first pixel of the second half
    static_assert(G1 >= 1 && G1 <= 15, "two 16-byte reads must cover the row");
    static_assert(2 * ROWS <= 64, "one item per lane");
    static_assert(PAIRS <= STRIDE, "tile row too narrow");
    const unsigned pitch32 = (unsigned)pitch;
    // keep the per-lane item coordinates from being hoisted out of the level
    // loop (they would stay live for the whole kernel)
    int l = lane;
    asm volatile("" : "+v"(l));
    const int idx = min(l, 2 * ROWS - 1);  // surplus lanes repeat the last item
    const unsigned r = (unsigned)idx >> 1;
    const unsigned g = (idx & 1) ? (unsigned)G1 : 0u;
    const unsigned y = (unsigned)refl(y0 + (int)r, h);
    const unsigned rowoff = __umul24(y, pitch32);
    const uint8_t *row = img + rowoff;
    const int xs = x0 + (int)g;            // first pixel of this item
    uint32_t q[4];
    if (xs >= 0 && xs + 16 <= w) {
        // one unsigned 32-bit offset from the image's (wave-uniform) base: the read then takes the base from scalar
        // registers and the offset from one vector register, with no 64-bit address arithmetic per lane
        __builtin_memcpy(q, img + (rowoff + (unsigned)xs), 16);
    } else {
        // the item straddles the left or right image border: sixteen byte reads with
        // BORDER_REFLECT_101, assembled into the same four dwords
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint32_t)row[refl(xs + 4 * c + b, w)] << (8 * b);
            q[c] = v;
        }
    }
    // the item's LDS position as one register, region offset included: the fifteen stores then differ by small
    // immediates (left to itself the compiler keeps the region's offset out of the register and spends an add per
    // pair of stores to bring it back in, because a paired store's immediates reach 1 KB only)
    unsigned at = lds_off + __umul24(r, (unsigned)STRIDE) + g;
    asm volatile("" : "+v"(at));
    store_pairs(lds + at, q);
}

// The three-keypoint kernel's staging of a tile that lies inside the image together with the rows and pixels its
// surplus items touch (26 rows from y0, pixels x0 .. x0 + G1 + 15): no reflection, no per-lane border test.  An
// item's place in the tile (LDS dword lane_at, byte offset lane_off from the tile's first pixel) depends on the
// lane only and is formed once per staging site by k3_item(); a tile then costs two adds, one 16-byte read, the
// fifteen byte shuffles and the stores.  Items 48-51 of a 24-row tile fill rows 24 and 25 of the region, which
// nothing reads.
template <int PAIRS>
__device__ __forceinline__ void k3_item(int lane, unsigned pitch32, unsigned &lane_at, unsigned &lane_off)
{
    constexpr int G1 = PAIRS + 1 - 16;
    int l = lane;
    asm volatile("" : "+v"(l));             // not hoisted out of the level loop (registers)
    const unsigned idx = (unsigned)min(l, 2 * (KLT_WIN + 1 + 2 * KLT_M) - 1);
    const unsigned r = idx >> 1, g = (idx & 1u) ? (unsigned)G1 : 0u;
    lane_at = __umul24(r, (unsigned)K3_S_) + g;
    lane_off = __umul24(r, pitch32) + g;
}
__device__ __forceinline__ void k3_stage_interior(uint32_t *lds, unsigned at, const uint8_t *img, unsigned off)
{
    asm volatile("" : "+v"(at));
    uint32_t q[4];
    __builtin_memcpy(q, img + off, 16);
    store_pairs(lds + at, q);
}

__global__ __launch_bounds__(64 * KLT_WAVES, 8) void klt_track_kernel(KltArgs A)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[KLT_WAVES * LDS_WAVE_DWORDS];

    const int lane = threadIdx.x & 63;
    const int wid = uniform(threadIdx.x >> 6);

    // blocks b and b+8 share an XCD (round-robin dispatch): keep one image's
    // keypoints on one XCD so its pyramid stays in that XCD's L2.
    int bid = blockIdx.x, img, kb;
    if ((A.batch & 7) == 0) {
        int xcd = bid & 7, j = bid >> 3;
        img = (j / A.blocks_per_image) * 8 + xcd;
        kb = j % A.blocks_per_image;
    } else {
        img = bid / A.blocks_per_image;
        kb = bid % A.blocks_per_image;
    }
    const int kp = kb * KLT_WAVES + wid;
    if (kp >= A.n) return;
    const int64_t pidx = (int64_t)img * A.n + kp;

    const unsigned lds_off = wid * LDS_WAVE_DWORDS;
    uint32_t *ldsA = lds_all + lds_off;

    // lane -> window column and 7-row strip; lane 63 shadows lane 0 with zero weight
    const bool active = lane < 63;
    const int wl = active ? lane : 0;
    const int col = wl % KLT_WIN;
    const int row0 = (wl / KLT_WIN) * 7;

    const int lane_jt = row0 * JT_S + col;
    const float px0 = A.prev_pts[2 * pidx], py0 = A.prev_pts[2 * pidx + 1];
    float outx = 0.f, outy = 0.f;
    int status = 1;
    float errv = 0.f;
    const float FLT_SCALE = 1.f / (1 << 20);

    for (int level = A.max_level; level >= 0; --level) {
        const KltLevel L = A.lv[level];
        const uint8_t *pimg = L.prev + (int64_t)img * L.pstride;
        const uint8_t *nimg = L.next + (int64_t)img * L.nstride;
        const float sc = __uint_as_float((unsigned)(127 - level) << 23);   // 2^-level, built from its exponent (a float64 division otherwise)
        float px = px0 * sc, py = py0 * sc;
        float nx, ny;
        if (level == A.max_level) {
            if (A.init_pts) { nx = A.init_pts[2 * pidx] * sc; ny = A.init_pts[2 * pidx + 1] * sc; }
            else { nx = px; ny = py; }
        } else {
            nx = outx * 2.f; ny = outy * 2.f;
        }
        outx = nx; outy = ny;

        px -= KLT_HALF; py -= KLT_HALF;
        const float fpx = floorf(px), fpy = floorf(py);
        const int ipx = uniform((int)fpx), ipy = uniform((int)fpy);
        if (ipx < -KLT_WIN || ipx >= L.w || ipy < -KLT_WIN || ipy >= L.h) {
            if (level == 0) { status = 0; errv = 0.f; }
            continue;
        }
        uint32_t wt, wb;
        bilinear_weights(px - fpx, py - fpy, wt, wb);

        // ---- template: previous-image tile -> Scharr pairs in registers -> bilinear samples ----
        wave_sync();  // region A may still be read as the coarser level's next-image tile
        stage_pairs<PT_ROWS, 23, PT_S>(lds_all, lds_off, pimg, L.ppitch, L.w, L.h, ipx - 1, ipy - 1, lane);
        wave_sync();
        uint32_t Ic[7];  // 256 - 512*I: accumulator seed that folds rounding and "- I" into the dot product
        v2f Ixy[7];      // template derivatives (Ix, Iy) as float pairs for packed FMAs
        v2f s12 = {0.f, 0.f};
        float sA22 = 0.f;
        // lane 63 shadows lane 0: zero derivative weights make its Ix, Iy (and all sums) vanish
        const uint32_t wtd = active ? wt : 0u, wbd = active ? wb : 0u;
        {
            // A lane reads tile rows row0..row0+9 at pair columns col..col+2.  With pixel
            // pairs as packed 16-bit lanes, one packed op yields the Scharr terms of window
            // columns col and col+1 at once -- exactly the {d[x], d[x+1]} operand the
            // bilinear dot product wants, so derivatives never leave registers.
            const uint32_t *t = ldsA + row0 * PT_S + col;
            uint32_t pr[8];
            s16x2 h0[10], h1[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const uint32_t q0 = t[k * PT_S], q1 = t[k * PT_S + 1], q2 = t[k * PT_S + 2];
                const u16x2 pl = __builtin_bit_cast(u16x2, q0), pc = __builtin_bit_cast(u16x2, q1),
                            prt = __builtin_bit_cast(u16x2, q2);
                h0[k] = __builtin_bit_cast(s16x2, (u16x2)(prt - pl));                       // horizontal difference
                h1[k] = __builtin_bit_cast(s16x2, (u16x2)((pl + prt) * (unsigned short)3 + pc * (unsigned short)10));  // horizontal smoothing
                if (k >= 1 && k <= 8) pr[k - 1] = q1;
            }
            uint32_t dxp[8], dyp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const s16x2 dx = (h0[k] + h0[k + 2]) * (short)3 + h0[k + 1] * (short)10;
                const s16x2 dy = h1[k + 2] - h1[k];
                dxp[k] = __builtin_bit_cast(uint32_t, dx);
                dyp[k] = __builtin_bit_cast(uint32_t, dy);
            }
            const bool interior = ipx >= 0 && ipx + KLT_WIN < L.w && ipy >= 0 && ipy + KLT_WIN < L.h;
            if (!interior) {
                // the derivative image is zero outside the image
                const uint32_t cm = ((unsigned)(ipx + col) < (unsigned)L.w ? 0xffffu : 0u) |
                                    ((unsigned)(ipx + col + 1) < (unsigned)L.w ? 0xffff0000u : 0u);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t m = (unsigned)(ipy + row0 + k) < (unsigned)L.h ? cm : 0u;
                    dxp[k] &= m;
                    dyp[k] &= m;
                }
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int ival = (int)(udot2(pr[k], wt, udot2_s(pr[k + 1], wb, 1u << 8)) >> 9);
                const int ixval = sdot2(dxp[k], wtd, sdot2(dxp[k + 1], wbd, 1 << 13)) >> 14;
                const int iyval = sdot2(dyp[k], wtd, sdot2(dyp[k + 1], wbd, 1 << 13)) >> 14;
                Ic[k] = 256u - ((uint32_t)ival << 9);
                const float fx = (float)ixval, fy = (float)iyval;
                Ixy[k] = (v2f){fx, fy};
                // integer products below 2^24 are exact in float, so each fused step rounds
                // exactly like "convert the product, then add"
                s12 = __builtin_elementwise_fma((v2f){fx, fx}, Ixy[k], s12);   // (sum Ix*Ix, sum Ix*Iy)
                sA22 = __builtin_fmaf(fy, fy, sA22);
            }
        }
        float A11, A12;
        wave_sum2(s12.x, s12.y, A11, A12);
        A11 *= FLT_SCALE; A12 *= FLT_SCALE;
        const float A22 = wave_sum(sA22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        // hardware sqrt / reciprocal (1 ulp): the sums above already differ from a scalar
        // evaluation in the last bits through their reduction order
        const float minEig = (A22 + A11 - __builtin_amdgcn_sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) *
                             (1.f / (float)(2 * KLT_WIN * KLT_WIN));
        errv = minEig;
        if (minEig < A.min_eig || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = __builtin_amdgcn_rcpf(D);

        // ---- Gauss-Newton iterations on the next image ----
        nx -= KLT_HALF; ny -= KLT_HALF;
        float pdx = 0.f, pdy = 0.f;
        int ox = 0, oy = 0;
        bool have_tile = false;
        for (int j = 0; j < A.max_iters; ++j) {
            const float fnx = floorf(nx), fny = floorf(ny);
            const int inx = uniform((int)fnx), iny = uniform((int)fny);
            if (inx < -KLT_WIN || inx >= L.w || iny < -KLT_WIN || iny >= L.h) {
                if (level == 0) status = 0;
                break;
            }
            int rx = inx - ox, ry = iny - oy;
            if (!have_tile || rx < 0 || rx > 2 * KLT_M || ry < 0 || ry > 2 * KLT_M) {
                ox = inx - KLT_M; oy = iny - KLT_M;
                rx = KLT_M; ry = KLT_M;
                wave_sync();
                stage_pairs<JT_ROWS, JT_COLS, JT_S>(lds_all, lds_off, nimg, L.npitch, L.w, L.h, ox, oy, lane);
                wave_sync();
                have_tile = true;
            }
            bilinear_weights(nx - fnx, ny - fny, wt, wb);

            const uint32_t *jt = ldsA + lane_jt + (ry * JT_S + rx);
            uint32_t q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = jt[k * JT_S];
            v2f sb = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                // ((S + 256) >> 9) - I == (S + 256 - 512*I) >> 9 with an arithmetic shift
                const float diff = (float)((int)udot2(q[k], wt, udot2(q[k + 1], wb, Ic[k])) >> 9);
                sb = __builtin_elementwise_fma((v2f){diff, diff}, Ixy[k], sb);
            }
            float b1, b2;
            wave_sum2(sb.x, sb.y, b1, b2);
            b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            outx = nx + KLT_HALF; outy = ny + KLT_HALF;
            // delta.ddot(delta) <= eps2 is a float64 test; the float32 value decides it
            // unless it falls inside a narrow bracket around eps2
            const float dd = __builtin_fmaf(dx, dx, dy * dy);
            bool conv = dd < A.eps2_lo;
            if (!conv && !(dd > A.eps2_hi)) conv = (double)dx * dx + (double)dy * dy <= A.eps2;
            if (conv) break;
            // |x| < 0.01 (float64) for a float x  <=>  |x| <= 0x3C23D70A, the float just below 0.01
            if (j > 0 && fabsf(dx + pdx) <= 0.00999999977648258209228515625f &&
                fabsf(dy + pdy) <= 0.00999999977648258209228515625f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
    }

    if (lane == 0) {
        A.next_pts[2 * pidx] = outx;
        A.next_pts[2 * pidx + 1] = outy;
        A.status[pidx] = (uint8_t)status;
        if (A.err) A.err[pidx] = errv;
    }
}

// ---------------------------------------------------------------------------
// Variant with three keypoints per wavefront: lane l < 63 owns window column
// l % 21 of keypoint l / 21 and walks all 21 rows of it, so the per-iteration
// work that does not depend on the window size (bilinear weights, the 2x2 solve,
// the stopping tests, loop control) is issued once for three keypoints.  Sums over
// a keypoint's 21 lanes come from a DPP prefix scan over the wave and two
// ds_bpermute reads of the scan at the group's ends.  The template keeps
// 256 - 512*I per sample and the derivatives as int16 pairs of vertically
// adjacent samples, matched by pairs of residuals so that v_dot2_i32_i16
// accumulates b1 and b2 exactly in int32.
// ---------------------------------------------------------------------------
#define K3_S K3_S_                     // tile row stride (dwords)
// One keypoint's tile region (dwords).  A dword read serves lanes 0-31 and 32-63 in one LDS cycle each when
// the 32 lanes of a half hit 32 different banks (bank = dword address mod 32).  A half holds 21 lanes of one
// keypoint (consecutive columns = consecutive banks) and 10-11 lanes of the middle keypoint: the region
// stride is padded from 26 * 25 = 650 to 661 dwords (661 mod 32 = 21, 2 * 661 mod 32 = 10), so that at equal
// tile offsets keypoint 1's columns 0-10 follow keypoint 0's 21 banks, and keypoint 2's columns 0-20 follow
// keypoint 1's columns 11-20: no two lanes of a half share a bank.  (With 650 the middle keypoint's lanes
// sat on banks of their neighbours in both halves and every tile read took two cycles per half.)
#define K3_REGION 661
static_assert(K3_REGION >= (KLT_WIN + 1 + 2 * KLT_M) * K3_S, "region holds a next-image tile");
static_assert(K3_REGION % 32 == 21, "bank offset of the middle keypoint");
#define K3_WAVES 4
#define K3_KP_WAVE 3
#define K3_KP_BLOCK (K3_WAVES * K3_KP_WAVE)

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_fz(float v)
{
    // lanes without a valid source, or masked out, see 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true));
}

// Sum of v over the 21 lanes of this lane's keypoint, in every lane of the group.
// e_addr: byte address (lane * 4) of the group's last lane (the lane before its first one is
// 84 bytes lower); first: the group starts at lane 0.
__device__ __forceinline__ float group_sum21(float v, int e_addr, bool first)
{
    float p = v + dpp_fz<0x111, 0xf, 0xf>(v);      // row_shr:1
    p += dpp_fz<0x112, 0xf, 0xf>(v);               // row_shr:2
    p += dpp_fz<0x113, 0xf, 0xf>(v);               // row_shr:3
    // masked steps as single DPP adds: lanes outside the masks keep p
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
                 : "+v"(p));
    const float e = __int_as_float(__builtin_amdgcn_ds_bpermute(e_addr, __float_as_int(p)));
    const float s = __int_as_float(__builtin_amdgcn_ds_bpermute(max(e_addr - 84, 0), __float_as_int(p)));
    return e - (first ? 0.f : s);
}

template <bool MASKED>
__device__ __forceinline__ void k3_template(const uint32_t *t, uint32_t wt, uint32_t wb, uint32_t wtd, uint32_t wbd,
                                            int ipx, int ipy, int col, int w, int h, uint32_t (&Ic)[21],
                                            uint32_t (&IxP)[11], uint32_t (&IyP)[11], int &a11, int &a12, int &a22)
{
    // rows r = 0..23 of the tile; derivative row k = r - 2 (k = 0..21) sits at tile row k + 1;
    // sample j (0..20) interpolates rows j, j + 1 of the derivative / centre-pixel rows
    s16x2 h0a, h0b, h1a, h1b;
    uint32_t prev_centre = 0, pr_k1 = 0, dx_k1 = 0, dy_k1 = 0;
    int ixd_even = 0, iyd_even = 0;
    int bias = 1 << 13;
    asm volatile("" : "+v"(bias));   // keep the rounding constant in one register
    uint32_t sh14 = 14;
    asm volatile("" : "+v"(sh14));
    // MASKED: derivatives outside the image count as zero.  Columns: a derivative pair {d[x], d[x+1]} only ever
    // meets the weight pair it is multiplied with, so the column test is folded into the weights once.  Rows:
    // bit k of rowbits says whether derivative row k lies inside the image; a signed one-bit field extract
    // turns it into an all-ones / zero mask.
    uint32_t rowbits = 0;
    if (MASKED) {
        const uint32_t cm = ((unsigned)(ipx + col) < (unsigned)w ? 0xffffu : 0u) | ((unsigned)(ipx + col + 1) < (unsigned)w ? 0xffff0000u : 0u);
        wtd &= cm;
        wbd &= cm;
        const int lo = min(max(-ipy, 0), 22), hi = min(max(h - ipy, 0), 22);     // rows lo .. hi-1 are inside
        rowbits = hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
    }
    a11 = 0; a12 = 0; a22 = 0;
#pragma unroll
    for (int r = 0; r < 24; ++r) {
        const uint32_t q0 = t[r * K3_S], q1 = t[r * K3_S + 1], q2 = t[r * K3_S + 2];
        const u16x2 pl = __builtin_bit_cast(u16x2, q0), pc = __builtin_bit_cast(u16x2, q1), prt = __builtin_bit_cast(u16x2, q2);
        const s16x2 h0c = __builtin_bit_cast(s16x2, (u16x2)(prt - pl));
        const s16x2 h1c = __builtin_bit_cast(s16x2, (u16x2)((pl + prt) * (unsigned short)3 + pc * (unsigned short)10));
        if (r >= 2) {
            const int k = r - 2;
            uint32_t dx = __builtin_bit_cast(uint32_t, (s16x2)((h0a + h0c) * (short)3 + h0b * (short)10));
            uint32_t dy = __builtin_bit_cast(uint32_t, (s16x2)(h1c - h1a));
            if (MASKED) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)rowbits, k, 1);
                dx &= m;
                dy &= m;
            }
            const uint32_t pr_k = prev_centre;   // centre pairs of tile row k + 1
            if (k >= 1) {
                const int j = k - 1;
                const uint32_t si = udot2(pr_k1, wt, udot2_s(pr_k, wb, 1u << 8));   // 512 * I + rounding bits
                const int ixd = sdot2(dx_k1, wtd, sdot2_keep(dx, wbd, bias));     // Ix << 14
                const int iyd = sdot2(dy_k1, wtd, sdot2_keep(dy, wbd, bias));
                Ic[j] = 256u - (si & ~0x1ffu);
                if ((j & 1) == 0) {
                    ixd_even = ixd; iyd_even = iyd;
                    if (j == 20) {
                        IxP[10] = (uint32_t)(ixd >> 14) & 0xffffu;
                        IyP[10] = (uint32_t)(iyd >> 14) & 0xffffu;
                    }
                } else {
                    // {Ix_j-1, Ix_j} as int16 lanes: low half from the even sample, high half from this one
                    IxP[j >> 1] = pack_hi_shr((uint32_t)(ixd_even >> 14), (uint32_t)ixd, sh14);
                    IyP[j >> 1] = pack_hi_shr((uint32_t)(iyd_even >> 14), (uint32_t)iyd, sh14);
                }
                if ((j & 1) == 1 || j == 20) {
                    const uint32_t xp = IxP[j >> 1], yp = IyP[j >> 1];
                    a11 = sdot2(xp, xp, a11);
                    a12 = sdot2(xp, yp, a12);
                    a22 = sdot2(yp, yp, a22);
                }
            }
            pr_k1 = pr_k; dx_k1 = dx; dy_k1 = dy;
        }
        h0a = h0b; h0b = h0c;
        h1a = h1b; h1b = h1c;
        prev_centre = q1;
    }
}

__global__ __launch_bounds__(64 * K3_WAVES, 5) void klt_track3_kernel(KltArgs A)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[K3_WAVES * K3_KP_WAVE * K3_REGION];

    const int wid = uniform(threadIdx.x >> 6);
    int bid = blockIdx.x, img, kb;
    if ((A.batch & 7) == 0) {
        int xcd = bid & 7, j = bid >> 3;
        img = (j / A.blocks_per_image) * 8 + xcd;
        kb = j % A.blocks_per_image;
    } else {
        img = bid / A.blocks_per_image;
        kb = bid % A.blocks_per_image;
    }
    const int kp_wave = kb * K3_KP_BLOCK + wid * K3_KP_WAVE;
    if (kp_wave >= A.n) return;
    // index of a lane's keypoint = pbase + kp_wave + slot (the launch checks that batch * n fits
    // 31 bits).  Everything per lane that only depends on the lane number (slot, column, tile
    // region, scan addresses) is re-derived from the lane counter at the top of each level:
    // holding it across the kernel would cost registers at the 96-VGPR limit of 5 waves/SIMD.
    const int pbase = img * A.n, plast = A.n - 1 - kp_wave;   // wave-uniform
    const unsigned wave_off = wid * (K3_KP_WAVE * K3_REGION);
    uint32_t *lds_wave = lds_all + wave_off;

    float outx = 0.f, outy = 0.f;
    int status = 1;
    float errv = 0.f;
    const float FLT_SCALE = 1.f / (1 << 20);

    for (int level = A.max_level; level >= 0; --level) {
        const KltLevel L = A.lv[level];
        const uint8_t *pimg = L.prev + (int64_t)img * L.pstride;
        const uint8_t *nimg = L.next + (int64_t)img * L.nstride;
        const float sc = __uint_as_float((unsigned)(127 - level) << 23);   // 2^-level, built from its exponent (a float64 division otherwise)
        const bool roomy = L.w >= JT_ROWS && L.h >= JT_ROWS;    // a tile can lie inside this level at all (its limits below are then >= 0)
        // lane -> keypoint slot g and window column; lane 63 shadows lane 62 (it lies outside every
        // group's scan range, so whatever it computes is never summed).  The barrier keeps these values
        // and the 64-bit addresses below from being hoisted out of the level loop, where they would
        // occupy registers for the whole kernel.
        int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane));   // not loop-invariant as far as the optimiser can tell
        const int g = lane < 21 ? 0 : (lane < 42 ? 1 : 2);
        const int col = lane < 63 ? lane - 21 * g : 20;
        const bool have = g <= plast;
        const int e_addr = 4 * (21 * g + 20);   // the lane before the group's first one is e_addr - 84 (groups 1, 2)
        const bool first = g == 0;
        const uint32_t *region = lds_wave + g * K3_REGION;
        const int pi = pbase + kp_wave + min(g, plast);
        float px = A.prev_pts[2 * (int64_t)pi] * sc, py = A.prev_pts[2 * (int64_t)pi + 1] * sc;
        float nx, ny;
        if (level == A.max_level) {
            if (A.init_pts) { nx = A.init_pts[2 * (int64_t)pi] * sc; ny = A.init_pts[2 * (int64_t)pi + 1] * sc; }
            else { nx = px; ny = py; }
        } else {
            nx = outx * 2.f; ny = outy * 2.f;
        }
        outx = nx; outy = ny;

        px -= KLT_HALF; py -= KLT_HALF;
        const float fpx = floorf(px), fpy = floorf(py);
        const int ipx = (int)fpx, ipy = (int)fpy;
        const bool lvl_ok = have && !(ipx < -KLT_WIN || ipx >= L.w || ipy < -KLT_WIN || ipy >= L.h);
        if (have && !lvl_ok && level == 0) { status = 0; errv = 0.f; }
        const unsigned long long okm = ballot64(lvl_ok);
        if (okm == 0) continue;
        uint32_t wt, wb;
        bilinear_weights(px - fpx, py - fpy, wt, wb);

        // ---- template ----
        wave_sync();  // the regions may still be read as the coarser level's next-image tiles
        {
            unsigned lane_at, lane_off;
            k3_item<23>(lane, (unsigned)L.ppitch, lane_at, lane_off);
#pragma unroll
            for (int gi = 0; gi < K3_KP_WAVE; ++gi) {
                if ((okm >> (21 * gi)) & 1ull) {
                    const int x0 = __builtin_amdgcn_readlane(ipx, 21 * gi) - 1, y0 = __builtin_amdgcn_readlane(ipy, 21 * gi) - 1;
                    // 0 <= x0 <= w - 24 and 0 <= y0 <= h - 26: one unsigned compare each
                    if (roomy & ((unsigned)x0 <= (unsigned)(L.w - 24)) & ((unsigned)y0 <= (unsigned)(L.h - JT_ROWS)))
                        k3_stage_interior(lds_all, wave_off + gi * K3_REGION + lane_at, pimg,
                                          (unsigned)y0 * (unsigned)L.ppitch + (unsigned)x0 + lane_off);
                    else
                        stage_pairs<PT_ROWS, 23, K3_S>(lds_all, wave_off + gi * K3_REGION, pimg, L.ppitch, L.w, L.h, x0, y0, lane);
                }
            }
        }
        wave_sync();
        uint32_t Ic[21], IxP[11], IyP[11];
        int a11, a12, a22;
        const bool interior = ipx >= 0 && ipx + KLT_WIN < L.w && ipy >= 0 && ipy + KLT_WIN < L.h;
        if (ballot64(lvl_ok && !interior) == 0)
            k3_template<false>(region + col, wt, wb, wt, wb, ipx, ipy, col, L.w, L.h, Ic, IxP, IyP, a11, a12, a22);
        else
            k3_template<true>(region + col, wt, wb, wt, wb, ipx, ipy, col, L.w, L.h, Ic, IxP, IyP, a11, a12, a22);
        const float A11 = group_sum21((float)a11, e_addr, first) * FLT_SCALE;
        const float A12 = group_sum21((float)a12, e_addr, first) * FLT_SCALE;
        const float A22 = group_sum21((float)a22, e_addr, first) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - __builtin_amdgcn_sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) *
                             (1.f / (float)(2 * KLT_WIN * KLT_WIN));
        if (lvl_ok) errv = minEig;
        bool run = lvl_ok && !(minEig < A.min_eig || D < 1.1920928955078125e-07f);
        if (lvl_ok && !run && level == 0) status = 0;
        D = __builtin_amdgcn_rcpf(D);

        // ---- Gauss-Newton iterations on the next image ----
        nx -= KLT_HALF; ny -= KLT_HALF;
        float pdx = 0.f, pdy = 0.f;
        int ox = 0, oy = 0;
        bool have_tile = false;
        for (int j = 0; j < A.max_iters; ++j) {
            if (ballot64(run) == 0) break;
            const float fnx = floorf(nx), fny = floorf(ny);
            const int inx = (int)fnx, iny = (int)fny;
            if (run && (inx < -KLT_WIN || inx >= L.w || iny < -KLT_WIN || iny >= L.h)) {
                if (level == 0) status = 0;
                run = false;
            }
            int rx = inx - ox, ry = iny - oy;
            const bool need = run && (!have_tile || rx < 0 || rx > 2 * KLT_M || ry < 0 || ry > 2 * KLT_M);
            const unsigned long long needm = ballot64(need);
            if (needm != 0) {
                wave_sync();
                unsigned lane_at, lane_off;
                k3_item<JT_COLS>(lane, (unsigned)L.npitch, lane_at, lane_off);
#pragma unroll
                for (int gi = 0; gi < K3_KP_WAVE; ++gi) {
                    if ((needm >> (21 * gi)) & 1ull) {
                        const int sx = __builtin_amdgcn_readlane(inx, 21 * gi) - KLT_M, sy = __builtin_amdgcn_readlane(iny, 21 * gi) - KLT_M;
                        if (roomy & ((unsigned)sx <= (unsigned)(L.w - JT_ROWS)) & ((unsigned)sy <= (unsigned)(L.h - JT_ROWS)))
                            k3_stage_interior(lds_all, wave_off + gi * K3_REGION + lane_at, nimg,
                                              (unsigned)sy * (unsigned)L.npitch + (unsigned)sx + lane_off);
                        else
                            stage_pairs<JT_ROWS, JT_COLS, K3_S>(lds_all, wave_off + gi * K3_REGION, nimg, L.npitch, L.w, L.h, sx, sy, lane);
                    }
                }
                wave_sync();
                if (need) { ox = inx - KLT_M; oy = iny - KLT_M; rx = KLT_M; ry = KLT_M; have_tile = true; }
            }
            bilinear_weights(nx - fnx, ny - fny, wt, wb);

            // ry, rx <= 4: the 24-bit multiply (a 32-bit product is a quarter-rate 64-bit multiply-add on this target)
            const uint32_t *jt = region + col + (run ? (int)__umul24((unsigned)ry, (unsigned)K3_S) + rx : 0);
            uint32_t q[22];
#pragma unroll
            for (int k = 0; k < 22; ++k) q[k] = jt[k * K3_S];
            int b1i = 0, b2i = 0;
            const uint32_t sh9 = 9;
#pragma unroll
            for (int p = 0; p < 11; ++p) {
                // ((S + 256) >> 9) - I == (S + 256 - 512*I) >> 9 with an arithmetic shift; the residuals of two
                // vertically adjacent samples share a dword as int16 lanes
                const uint32_t s0 = udot2(q[2 * p], wt, udot2(q[2 * p + 1], wb, Ic[2 * p]));
                uint32_t d2;
                if (2 * p + 1 < 21) {
                    const uint32_t s1 = udot2(q[2 * p + 1], wt, udot2(q[2 * p + 2], wb, Ic[2 * p + 1]));
                    d2 = pack_hi_shr_s(s0 >> 9, s1, sh9);
                } else {
                    d2 = (s0 >> 9) & 0xffffu;
                }
                if (p == 0) {
                    b1i = sdot2_first(d2, IxP[p]);
                    b2i = sdot2_first(d2, IyP[p]);
                } else {
                    b1i = sdot2(d2, IxP[p], b1i);
                    b2i = sdot2(d2, IyP[p], b2i);
                }
            }
            const float b1 = group_sum21((float)b1i, e_addr, first) * FLT_SCALE;
            const float b2 = group_sum21((float)b2i, e_addr, first) * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            if (run) {
                nx += dx; ny += dy;
                outx = nx + KLT_HALF; outy = ny + KLT_HALF;
                // delta.ddot(delta) <= eps2 is a float64 test; the float32 value decides it
                // unless it falls inside a narrow bracket around eps2
                const float dd = __builtin_fmaf(dx, dx, dy * dy);
                bool conv = dd < A.eps2_lo;
                if (!conv && !(dd > A.eps2_hi)) conv = (double)dx * dx + (double)dy * dy <= A.eps2;
                // |x| < 0.01 (float64) for a float x  <=>  |x| <= 0x3C23D70A, the float just below 0.01
                if (!conv && j > 0 && fabsf(dx + pdx) <= 0.00999999977648258209228515625f &&
                    fabsf(dy + pdy) <= 0.00999999977648258209228515625f) {
                    outx -= dx * 0.5f; outy -= dy * 0.5f;
                    conv = true;
                }
                if (conv) run = false;
                pdx = dx; pdy = dy;
            }
        }
    }

    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int g_e = lane_e / 21;
    if (lane_e == 21 * g_e && g_e < K3_KP_WAVE && g_e <= plast) {
        const int pidx = pbase + kp_wave + g_e;
        A.next_pts[2 * (int64_t)pidx] = outx;
        A.next_pts[2 * (int64_t)pidx + 1] = outy;
        A.status[pidx] = (uint8_t)status;
        if (A.err) A.err[pidx] = errv;
    }
}

static int fill_levels(KltArgs &a, const vo_pyr *prev, const vo_pyr *next)
{
    VO_REQUIRE(prev->built && next->built, "vo_klt_track: pyramids must be built first");
    VO_REQUIRE(prev->batch == next->batch && prev->h[0] == next->h[0] && prev->w[0] == next->w[0] &&
                   prev->max_level == next->max_level,
               "vo_klt_track: pyramids differ in geometry");
    VO_REQUIRE(prev->win == KLT_WIN && next->win == KLT_WIN, "vo_klt_track: only the 21x21 window is implemented (got %d)", prev->win);
    a.max_level = prev->max_level;
    a.batch = prev->batch;
    for (int l = 0; l <= prev->max_level; ++l) {
        a.lv[l].prev = prev->lvl[l];
        a.lv[l].next = next->lvl[l];
        a.lv[l].w = prev->w[l];
        a.lv[l].h = prev->h[l];
        a.lv[l].ppitch = prev->pitch[l];
        a.lv[l].pstride = prev->img_stride[l];
        a.lv[l].npitch = next->pitch[l];
        a.lv[l].nstride = next->img_stride[l];
        // OpenCV's level rule (every level larger than the window).  It is also what the reflect helper needs: a
        // valid window reads pixels -21 .. n + 20, one fold on either side once n >= 22.
        VO_REQUIRE(prev->w[l] > KLT_WIN && prev->h[l] > KLT_WIN, "vo_klt_track: level %d (%dx%d) too small", l, prev->w[l], prev->h[l]);
    }
    return VO_OK;
}

// Two tracker kernels with identical semantics.  Three keypoints per wavefront issues the fewest
// instructions per keypoint and is the throughput kernel; one keypoint per wavefront has the
// shorter critical path (a wavefront walks the levels of one keypoint, not three) and wins when
// all keypoints of a call fit on the device at once, i.e. for single-frame, latency-bound calls.
// VO_KLT_VARIANT=1 or =3 in the environment forces one of them (comparison runs, parity tests).
#define KLT_ONE_PER_WAVE_MAX 8192     // 1024 SIMDs x 8 wavefronts
#define KLT_READ_IN_PLACE_MAX 16384   // host keypoints up to this many are read from page-locked memory in place
static int klt_variant(int64_t keypoints)
{
    const char *e = getenv("VO_KLT_VARIANT");
    if (e && e[0] == '1') return 1;
    if (e && e[0] == '3') return 3;
    return keypoints <= KLT_ONE_PER_WAVE_MAX ? 1 : 3;
}

static int launch_klt(vo_ctx *ctx, KltArgs &a)
{
    if (klt_variant((int64_t)a.n * a.batch) == 3) {
        a.blocks_per_image = (a.n + K3_KP_BLOCK - 1) / K3_KP_BLOCK;
        VO_REQUIRE((int64_t)a.n * a.batch < (1ll << 30), "vo_klt_track: too many keypoints (%lld)", (long long)a.n * a.batch);
        int64_t blocks3 = (int64_t)a.blocks_per_image * a.batch;
        VO_REQUIRE(blocks3 < (1ll << 31), "vo_klt_track: too many keypoints (%lld workgroups)", (long long)blocks3);
        hipLaunchKernelGGL(klt_track3_kernel, dim3((unsigned)blocks3), dim3(64 * K3_WAVES), 0, ctx->stream, a);
        VO_HIP_CHECK(hipGetLastError());
        return VO_OK;
    }
    a.blocks_per_image = (a.n + KLT_WAVES - 1) / KLT_WAVES;
    int64_t blocks = (int64_t)a.blocks_per_image * a.batch;
    VO_REQUIRE(blocks < (1ll << 31), "vo_klt_track: too many keypoints (%lld workgroups)", (long long)blocks);
    hipLaunchKernelGGL(klt_track_kernel, dim3((unsigned)blocks), dim3(64 * KLT_WAVES), 0, ctx->stream, a);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

static void set_params(KltArgs &a, const vo_klt_params *p)
{
    int it = p->max_iters;
    it = it < 0 ? 0 : (it > 100 ? 100 : it);
    double e = (double)p->eps;
    e = e < 0 ? 0 : (e > 10 ? 10 : e);
    a.max_iters = it;
    a.eps2 = e * e;
    a.eps2_lo = (float)(a.eps2 * (1.0 - 1e-5));
    a.eps2_hi = (float)(a.eps2 * (1.0 + 1e-5));
    if (a.eps2 == 0) { a.eps2_lo = 0.f; a.eps2_hi = 0.f; }
    a.min_eig = p->min_eig;
}

extern "C" int vo_klt_track(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts,
                            const float *init_pts, int n, const vo_klt_params *params, float *next_pts,
                            uint8_t *status, float *err, int mem)
{
    VO_REQUIRE(ctx && prev && next && params, "vo_klt_track: NULL argument");
    VO_REQUIRE(n >= 0, "vo_klt_track: n=%d", n);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_klt_track: bad mem flag %d", mem);
    if (n == 0) return VO_OK;
    VO_REQUIRE(prev_pts && next_pts && status, "vo_klt_track: NULL point/status buffer");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    KltArgs a = {};
    int rc = fill_levels(a, prev, next);
    if (rc) return rc;
    set_params(a, params);
    a.n = n;
    const size_t cnt = (size_t)n * prev->batch;
    if (mem == VO_MEM_DEVICE) {
        a.prev_pts = prev_pts; a.init_pts = init_pts; a.next_pts = next_pts; a.status = status; a.err = err;
        return launch_klt(ctx, a);
    }
    // Host arrays travel through the context's page-locked buffer: [results (13 B per keypoint) | keypoints | start
    // positions].  The results come back with one copy and one wait (a copy into pageable memory would block the caller
    // until it has landed); the inputs of a single-frame call are read by the tracker in place (each keypoint once, over
    // the bus: cheaper than the ~20 us a copy engine's hand-over to the compute queue costs), larger ones are copied.
    void *pin, *d_res, *d_in = nullptr, *d_init = nullptr;
    const size_t in_off = (cnt * 13 + 255) & ~(size_t)255, in_bytes = (cnt * 8 + 255) & ~(size_t)255;
    if ((rc = vo_pinned_reserve(ctx, in_off + 2 * in_bytes, &pin))) return rc;
    char *hp = (char *)pin;
    if ((rc = vo_stage_reserve(ctx, 2, cnt * 13, &d_res))) return rc;
    memcpy(hp + in_off, prev_pts, cnt * 8);
    if (init_pts) memcpy(hp + in_off + in_bytes, init_pts, cnt * 8);
    if (cnt <= KLT_READ_IN_PLACE_MAX) {
        d_in = hp + in_off;
        if (init_pts) d_init = hp + in_off + in_bytes;
    } else {
        if ((rc = vo_stage_reserve(ctx, 0, cnt * 8, &d_in))) return rc;
        if (init_pts && (rc = vo_stage_reserve(ctx, 1, cnt * 8, &d_init))) return rc;
        VO_HIP_CHECK(hipMemcpyAsync(d_in, hp + in_off, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
        if (init_pts) VO_HIP_CHECK(hipMemcpyAsync(d_init, hp + in_off + in_bytes, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    a.prev_pts = (const float *)d_in; a.init_pts = (const float *)d_init; a.next_pts = (float *)d_res;
    a.err = (float *)((char *)d_res + cnt * 8); a.status = (uint8_t *)d_res + cnt * 12;
    if ((rc = launch_klt(ctx, a))) return rc;
    if ((rc = vo_download_async(ctx, hp, d_res, cnt * 13))) return rc;
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(next_pts, hp, cnt * 8);
    if (err) memcpy(err, hp + cnt * 8, cnt * 4);
    memcpy(status, hp + cnt * 12, cnt);
    return VO_OK;
}

// Bidirectional acceptance: |p - back| < lambda in both axes and both passes tracked.
__global__ void klt_fb_check_kernel(const float *__restrict__ prev_pts, const float *__restrict__ back_pts,
                                    const uint8_t *__restrict__ st_f, const uint8_t *__restrict__ st_b,
                                    float lambda, float *__restrict__ fwd_pts, uint8_t *__restrict__ mask, int64_t cnt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    float ex = fabsf(prev_pts[2 * i] - back_pts[2 * i]);
    float ey = fabsf(prev_pts[2 * i + 1] - back_pts[2 * i + 1]);
    bool ok = st_f[i] && st_b[i] && ex < lambda && ey < lambda;
    mask[i] = ok ? 1 : 0;
    if (!ok) { fwd_pts[2 * i] = 0.f; fwd_pts[2 * i + 1] = 0.f; }
}

extern "C" int vo_klt_track_fb(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts, int n,
                               const vo_klt_params *params, float lambda, float *next_pts, uint8_t *mask, int mem)
{
    VO_REQUIRE(ctx && prev && next && params, "vo_klt_track_fb: NULL argument");
    VO_REQUIRE(n >= 0, "vo_klt_track_fb: n=%d", n);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_klt_track_fb: bad mem flag %d", mem);
    if (n == 0) return VO_OK;
    VO_REQUIRE(prev_pts && next_pts && mask, "vo_klt_track_fb: NULL point/mask buffer");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t cnt = (size_t)n * prev->batch;
    int rc;
    void *d_in, *d_fwd, *d_back, *d_sf, *d_sb, *d_mask, *pin = nullptr;
    if ((rc = vo_stage_reserve(ctx, 5, cnt * 8, &d_back))) return rc;
    if ((rc = vo_stage_reserve(ctx, 6, cnt, &d_sf))) return rc;
    if ((rc = vo_stage_reserve(ctx, 7, cnt, &d_sb))) return rc;
    const size_t in_off = (cnt * 9 + 255) & ~(size_t)255;
    if (mem == VO_MEM_DEVICE) {
        d_in = const_cast<float *>(prev_pts); d_fwd = next_pts; d_mask = mask;
    } else {
        // page-locked buffer: [results (points, mask) | keypoints]; see vo_klt_track
        if ((rc = vo_pinned_reserve(ctx, in_off + cnt * 8, &pin))) return rc;
        if ((rc = vo_stage_reserve(ctx, 2, cnt * 9, &d_fwd))) return rc;
        d_mask = (char *)d_fwd + cnt * 8;
        memcpy((char *)pin + in_off, prev_pts, cnt * 8);
        if (cnt <= KLT_READ_IN_PLACE_MAX) {
            d_in = (char *)pin + in_off;
        } else {
            if ((rc = vo_stage_reserve(ctx, 0, cnt * 8, &d_in))) return rc;
            VO_HIP_CHECK(hipMemcpyAsync(d_in, (char *)pin + in_off, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    KltArgs f = {};
    if ((rc = fill_levels(f, prev, next))) return rc;
    set_params(f, params);
    f.n = n;
    f.prev_pts = (const float *)d_in; f.next_pts = (float *)d_fwd; f.status = (uint8_t *)d_sf;
    if ((rc = launch_klt(ctx, f))) return rc;
    KltArgs g = {};
    if ((rc = fill_levels(g, next, prev))) return rc;
    set_params(g, params);
    g.n = n;
    g.prev_pts = (const float *)d_fwd; g.next_pts = (float *)d_back; g.status = (uint8_t *)d_sb;
    if ((rc = launch_klt(ctx, g))) return rc;
    unsigned blocks = (unsigned)((cnt + 255) / 256);
    hipLaunchKernelGGL(klt_fb_check_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const float *)d_in,
                       (const float *)d_back, (const uint8_t *)d_sf, (const uint8_t *)d_sb, lambda, (float *)d_fwd,
                       (uint8_t *)d_mask, (int64_t)cnt);
    VO_HIP_CHECK(hipGetLastError());
    if (mem == VO_MEM_HOST) {
        char *hp = (char *)pin;
        if ((rc = vo_download_async(ctx, hp, d_fwd, cnt * 9))) return rc;
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        memcpy(next_pts, hp, cnt * 8);
        memcpy(mask, hp + cnt * 8, cnt);
    }
    return VO_OK;
}

// ---------------------------------------------------------------------------
// Forward-backward tracking for callers that cannot allocate (tight serving loops,
// harnesses that forbid hipMalloc): the same three launches as vo_klt_track_fb, with the intermediate
// results -- back-tracked points and the two status arrays, 10 bytes per keypoint -- in
// a fixed device array instead of the context's growable staging buffers.
// Every GPU holds its own copy of the array at its own address; the address is looked up
// once per device (vo_klt_noalloc_setup, called by vo_ctx_create with that device
// current) and only that pointer is ever handed to kernels.  Calls on one GPU share the
// array: a call from another stream while the previous call's kernels are still running is
// refused (VO_ESTATE; calls on one stream are ordered anyway).  The check is made by the
// calling host thread: two host threads calling at once are the caller's to serialise.  Nothing in it
// has to start at zero -- each call's kernels write every entry they later read.
// ---------------------------------------------------------------------------
#define FB_MAX_KP VO_KLT_FB_NOALLOC_MAX_KEYPOINTS
#define FB_OFF_BACK 0
#define FB_OFF_SF ((size_t)FB_MAX_KP * 8)
#define FB_OFF_SB (FB_OFF_SF + (size_t)FB_MAX_KP)
#define FB_SCRATCH_BYTES (FB_OFF_SB + (size_t)FB_MAX_KP)
static __device__ __align__(256) unsigned char scratch[FB_SCRATCH_BYTES];

#define FB_MAX_DEVICES 64
static unsigned char *g_scratch_ptr[FB_MAX_DEVICES];   // per device: address of that GPU's copy of `scratch`
// Who used a device's array last: an event recorded behind that call's kernels (created with the address, once per
// device) and the stream it ran on.  A call from another stream while that event is still pending would write
// the array under the earlier call's kernels; it is refused instead.
static hipEvent_t g_scratch_done[FB_MAX_DEVICES];
static hipStream_t g_scratch_stream[FB_MAX_DEVICES];
static bool g_scratch_used[FB_MAX_DEVICES];

int vo_klt_noalloc_setup(int device)
{
    VO_REQUIRE(device >= 0 && device < FB_MAX_DEVICES, "vo_klt_noalloc_setup: device %d", device);
    if (g_scratch_ptr[device]) return VO_OK;
    void *p = nullptr;
    VO_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));   // the current device's copy
    VO_REQUIRE(p != nullptr, "vo_klt_noalloc_setup: no device address for the scratch array");
    VO_HIP_CHECK(hipEventCreateWithFlags(&g_scratch_done[device], hipEventDisableTiming));
    g_scratch_ptr[device] = (unsigned char *)p;
    return VO_OK;
}

extern "C" int vo_klt_track_fb_noalloc(vo_ctx *ctx, const vo_pyr *prev, const vo_pyr *next, const float *prev_pts, int n,
                                       const vo_klt_params *params, float lambda, float *next_pts, uint8_t *mask)
{
    VO_REQUIRE(ctx && prev && next && params, "vo_klt_track_fb_noalloc: NULL argument");
    VO_REQUIRE(n >= 0, "vo_klt_track_fb_noalloc: n=%d", n);
    if (n == 0) return VO_OK;
    VO_REQUIRE(prev_pts && next_pts && mask, "vo_klt_track_fb_noalloc: NULL point/mask buffer");
    const int64_t cnt = (int64_t)n * prev->batch;
    VO_REQUIRE(cnt <= FB_MAX_KP, "vo_klt_track_fb_noalloc: %lld keypoints (%d images x %d) exceed the fixed scratch of %d; "
               "use vo_klt_track_fb", (long long)cnt, prev->batch, n, FB_MAX_KP);
    VO_REQUIRE(ctx->device >= 0 && ctx->device < FB_MAX_DEVICES && g_scratch_ptr[ctx->device],
               "vo_klt_track_fb_noalloc: scratch of device %d was not set up", ctx->device);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    unsigned char *base = g_scratch_ptr[ctx->device];
    if (g_scratch_used[ctx->device] && g_scratch_stream[ctx->device] != ctx->stream) {
        // same stream: ordered behind the earlier call.  Another stream: only once that call's kernels are through.
        const hipError_t q = hipEventQuery(g_scratch_done[ctx->device]);
        if (q == hipErrorNotReady) {
            vo_set_error("vo_klt_track_fb_noalloc: the scratch array of device %d is still in use by a call on another stream; "
                         "serialise the calls (or use vo_klt_track_fb)", ctx->device);
            return VO_ESTATE;
        }
        VO_HIP_CHECK(q);
    }
    float *d_back = (float *)(base + FB_OFF_BACK);
    uint8_t *d_sf = base + FB_OFF_SF, *d_sb = base + FB_OFF_SB;
    int rc;
    KltArgs f = {};
    if ((rc = fill_levels(f, prev, next))) return rc;
    set_params(f, params);
    f.n = n;
    f.prev_pts = prev_pts; f.next_pts = next_pts; f.status = d_sf;
    if ((rc = launch_klt(ctx, f))) return rc;
    KltArgs g = {};
    if ((rc = fill_levels(g, next, prev))) return rc;
    set_params(g, params);
    g.n = n;
    g.prev_pts = next_pts; g.next_pts = d_back; g.status = d_sb;
    if ((rc = launch_klt(ctx, g))) return rc;
    hipLaunchKernelGGL(klt_fb_check_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, prev_pts,
                       (const float *)d_back, (const uint8_t *)d_sf, (const uint8_t *)d_sb, lambda, next_pts, mask, cnt);
    VO_HIP_CHECK(hipGetLastError());
    VO_HIP_CHECK(hipEventRecord(g_scratch_done[ctx->device], ctx->stream));
    g_scratch_stream[ctx->device] = ctx->stream;
    g_scratch_used[ctx->device] = true;
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_klt(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)klt_track_kernel));
    return VO_OK;
}
