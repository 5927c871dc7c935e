// Bundle-adjustment residual, analytic Jacobian blocks and normal equations,
// one observation per lane, float64.
//
// Residual: BundleAdjustment._rotate/_reproject/_fun of the reference
// (vo_pipeline/bundleAdjustment.py:14-45), including the theta == 0 branch
// (v = 0, point unchanged).  The reference hands scipy a finite-difference
// Jacobian through the sparsity of bundleAdjustment.py:47-67 and never forms
// J^T J; here the 2x6 camera and 2x3 point blocks are analytic and the normal
// equations are accumulated on the fly:
//   Hcc (F,6,6) += Jc^T Jc   Hpp (P,3,3) += Jp^T Jp   Hcp (O,6,3) = Jc^T Jp
//   g (6F+3P) = J^T r        cost = 0.5 sum r^2 (all observations)
// Observations i < obs_start (those of the frozen first cameras) keep their
// residual but contribute no Jacobian rows, as in the reference's sparsity.
//
// HBM-bound by construction (~200 B per observation, dominated by the Hcp
// store); camera blocks are reduced across the wave with DPP before one atomic
// per wave and entry, point blocks use per-lane float64 atomics.
#include "vo_internal.h"
#include <stdlib.h>

#define BA_THREADS 256
#define BA_REPLICAS 16  // copies of the camera blocks that the per-wave atomics are spread over
// Partial sums of the cost, chosen by workgroup: one float64 atomic per wave on a single address is served one at a
// time (about 14 ns each: 0.7 ms at 50 000 waves, the whole time of a 3.2 M-observation launch)
#define BA_COST_SLOTS 256

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v)
{
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes; valid in lane 63.
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<0xB1, 0xf>(v);
    v += dpp_d<0x4E, 0xf>(v);
    v += dpp_d<0x141, 0xf>(v);
    v += dpp_d<0x140, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    return v;
}

struct BaK { double k[9]; };

// Levenberg-Marquardt bookkeeping of vo_ba_solve, kept on the device so that trial steps can be queued without the
// host reading a cost in between: the damping the next trial step uses, the cost at the current point, counters and
// the stop flag.  Kernels of a trial step that are handed the record leave at once when `done` is set.
struct BaCtl {
    double lambda, cost, prev, ftol, cost_start;
    int done, copy_now, tries, iters, max_iters, trials, accepted, reason, flat_reject, pad;
};

template <bool WITH_J>
__device__ __forceinline__ void ba_obs(const double *cam, const double *X, const BaK &K, const double *ob,
                                       double *r, double *Jc, double *Jp)
{
    const double rx = cam[0], ry = cam[1], rz = cam[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    double vx = 0, vy = 0, vz = 0;
    if (theta != 0) { vx = rx / theta; vy = ry / theta; vz = rz / theta; }
    const double dot = X[0] * vx + X[1] * vy + X[2] * vz;
    double s, c;
    sincos(theta, &s, &c);
    const double cxx = vy * X[2] - vz * X[1], cxy = vz * X[0] - vx * X[2], cxz = vx * X[1] - vy * X[0];
    const double fx_ = c * X[0] + s * cxx + dot * (1 - c) * vx;
    const double fy_ = c * X[1] + s * cxy + dot * (1 - c) * vy;
    const double fz_ = c * X[2] + s * cxz + dot * (1 - c) * vz;
    const double Xc = fx_ + cam[3], Yc = fy_ + cam[4], Zc = fz_ + cam[5];
    const double p0 = K.k[0] * Xc + K.k[1] * Yc + K.k[2] * Zc;
    const double p1 = K.k[3] * Xc + K.k[4] * Yc + K.k[5] * Zc;
    const double p2 = K.k[6] * Xc + K.k[7] * Yc + K.k[8] * Zc;
    r[0] = p0 / p2 - ob[0];
    r[1] = p1 / p2 - ob[1];
    if (!WITH_J) return;

    const double ip2 = 1.0 / p2;
    const double a0 = ip2, a2 = -p0 * ip2 * ip2, b1 = ip2, b2 = -p1 * ip2 * ip2;
    double G[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        G[0][k] = a0 * K.k[k] + a2 * K.k[6 + k];
        G[1][k] = b1 * K.k[3 + k] + b2 * K.k[6 + k];
    }
    const double v[3] = {vx, vy, vz};
    double R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = (1 - c) * v[i] * v[j] + (i == j ? c : 0);
    R[0][1] -= s * vz; R[0][2] += s * vy;
    R[1][0] += s * vz; R[1][2] -= s * vx;
    R[2][0] -= s * vy; R[2][1] += s * vx;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) Jp[a * 3 + j] = G[a][0] * R[0][j] + G[a][1] * R[1][j] + G[a][2] * R[2][j];

    double D[3][3];
    if (theta == 0) {
        D[0][0] = 0; D[0][1] = X[2]; D[0][2] = -X[1];
        D[1][0] = -X[2]; D[1][1] = 0; D[1][2] = X[0];
        D[2][0] = X[1]; D[2][1] = -X[0]; D[2][2] = 0;
    } else {
        const double cr[3] = {cxx, cxy, cxz};
        double dth[3], B[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) dth[i] = -s * X[i] + c * cr[i] + s * dot * v[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) B[i][j] = (1 - c) * (v[i] * X[j] + (i == j ? dot : 0));
        B[0][1] += s * X[2]; B[0][2] -= s * X[1];
        B[1][0] -= s * X[2]; B[1][2] += s * X[0];
        B[2][0] += s * X[1]; B[2][1] -= s * X[0];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double Bv = B[i][0] * v[0] + B[i][1] * v[1] + B[i][2] * v[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) D[i][j] = dth[i] * v[j] + (B[i][j] - Bv * v[j]) / theta;
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jc[a * 6 + j] = G[a][0] * D[0][j] + G[a][1] * D[1][j] + G[a][2] * D[2][j];
            Jc[a * 6 + 3 + j] = G[a][j];
        }
}

// GATHER: the point Jacobian (2x3) of every observation is stored to Jp_out instead of
// being summed into Hpp / g with atomics; ba_point_gather_kernel finishes the landmark
// blocks from landmark-major observation lists.
template <bool WITH_J, bool GATHER = false>
__global__ __launch_bounds__(BA_THREADS) void ba_kernel(const double *__restrict__ params, int F, int P,
                                                        const int64_t *__restrict__ ci, const int64_t *__restrict__ pi,
                                                        const double *__restrict__ obs, int O, BaK K, int obs_start,
                                                        double *__restrict__ r, double *__restrict__ cost,
                                                        double *__restrict__ Hcc, double *__restrict__ Hpp,
                                                        double *__restrict__ Hcp, double *__restrict__ g,
                                                        double *__restrict__ cam_rep, double *__restrict__ cost_slots,
                                                        double *__restrict__ Jp_out = nullptr,
                                                        const int *__restrict__ list_start = nullptr,
                                                        const int *__restrict__ list_slot = nullptr, int *__restrict__ list_obs = nullptr,
                                                        const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl && ctl->done) return;
    const int i = blockIdx.x * BA_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < O;
    int64_t c = 0, p = 0;
    bool ok = false;
    if (valid) {
        c = ci[i]; p = pi[i];
        ok = c >= 0 && c < F && p >= 0 && p < P;   // out-of-range indices never touch memory
    }
    double rr[2] = {0, 0}, Jc[12], Jp[6];
    const bool active = WITH_J && ok && i >= obs_start;
    if (ok) {
        double cam[6], X[3], ob[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) cam[k] = params[6 * c + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = params[6 * (int64_t)F + 3 * p + k];
        const double2 o2 = *(const double2 *)(obs + 2 * (int64_t)i);
        ob[0] = o2.x; ob[1] = o2.y;
        ba_obs<WITH_J>(cam, X, K, ob, rr, Jc, Jp);
        *(double2 *)(r + 2 * (int64_t)i) = double2{rr[0], rr[1]};
    } else if (valid) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        r[2 * (int64_t)i] = qnan; r[2 * (int64_t)i + 1] = qnan;
        rr[0] = qnan;
    }
    // cost: wave sum, one atomic per wave
    const double cs = wave_sum_d(rr[0] * rr[0] + rr[1] * rr[1]);
    double *cost_at = cost_slots ? cost_slots + blockIdx.x % BA_COST_SLOTS : cost;   // slots: ba_cam_sum_kernel adds them up
    if (!WITH_J) {
        if (lane == 63) unsafeAtomicAdd(cost_at, 0.5 * cs);
        return;
    }

    if (!active) {
#pragma unroll
        for (int k = 0; k < 12; ++k) Jc[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) Jp[k] = 0;
    }
    // camera blocks: when every active lane of the wave sees one camera, reduce over the
    // wave first (the caller's observations are sorted by camera).  The sums go to one of
    // BA_REPLICAS copies of the camera blocks chosen by workgroup, so that the float64
    // atomics of the many waves of one camera do not queue on the same few cache lines;
    // ba_cam_sum_kernel adds the copies up.
    // (this section comes before the global stores and atomics of the wave: its wavefront fence waits for
    // every outstanding memory operation of the wave, and behind those writes it cost 25 us of 60 at one window)
    // One pass per camera present among the wave's active lanes (the caller's observations are sorted by camera, so
    // that is one pass, two for the wave that straddles a camera boundary): the other cameras' lanes add zeros.
    // A wave that left its mixed lanes to per-lane atomics put up to 64 atomics of one instruction on the same
    // address, which the memory system takes one after the other -- nine such waves of 784 were half of the
    // kernel's time at one window.
    unsigned long long todo = __ballot(active);
    if constexpr (WITH_J) {
        __shared__ double red[BA_THREADS / 64][8][65];
        double(*mine)[65] = red[threadIdx.x >> 6];
        constexpr int TA[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
        constexpr int TB[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
        const int v = lane & 7, seg = lane >> 3;
        for (int pass = 0; todo != 0 && pass < 4; ++pass) {
            const int first = __ffsll((long long)todo) - 1;
            const long long c0 = __shfl((long long)c, first);
            const bool here = active && c == c0;
            todo &= ~__ballot(here);
            double *rep = cam_rep + (size_t)(blockIdx.x % BA_REPLICAS) * F * 42;
            double *hc = rep + 42 * c0;
            // 21 upper-triangle entries of Jc^T Jc and 6 of Jc^T r, eight terms at a time: every lane parks its
            // terms in LDS (row per term, column per lane), a lane adds up an eighth of one row, three shuffles
            // join the eighths, and eight lanes issue the atomics (mirroring the off-diagonal entries).  Eight
            // rows per wave rather than all 27 keep the workgroup at 16 KB of LDS: with 56 KB only two waves
            // fitted a SIMD, and at many windows the kernel waited on memory with nothing else to run.
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = 8 * ch + k, et = e < 21 ? e : 0, eg = e < 21 ? 0 : (e < 27 ? e - 21 : 0);
                    if (e < 21) mine[k][lane] = here ? Jc[TA[et]] * Jc[TB[et]] + Jc[6 + TA[et]] * Jc[6 + TB[et]] : 0.0;
                    else if (e < 27) mine[k][lane] = here ? Jc[eg] * rr[0] + Jc[6 + eg] * rr[1] : 0.0;
                }
                // the rows are read by other lanes of the same wavefront: only the LDS writes have to have
                // landed (a wavefront-scope fence would also wait for the wave's global stores)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const double *row = &mine[v][seg * 8];
                double part = ((row[0] + row[1]) + (row[2] + row[3])) + ((row[4] + row[5]) + (row[6] + row[7]));
                part += __shfl_xor(part, 8);
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                const int e = 8 * ch + v;
                if (seg == 0 && e < 27) {
                    if (e < 21) {
                        // term index -> (a, b) of the upper triangle, row-major
                        const int ta = e < 6 ? 0 : e < 11 ? 1 : e < 15 ? 2 : e < 18 ? 3 : e < 20 ? 4 : 5;
                        const int first_of_row = ta == 0 ? 0 : ta == 1 ? 6 : ta == 2 ? 11 : ta == 3 ? 15 : ta == 4 ? 18 : 20;
                        const int tb = ta + (e - first_of_row);
                        unsafeAtomicAdd(hc + ta * 6 + tb, part);
                        if (tb != ta) unsafeAtomicAdd(hc + tb * 6 + ta, part);
                    } else {
                        unsafeAtomicAdd(hc + 36 + (e - 21), part);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this chunk's reads are in before the next one's writes
                __builtin_amdgcn_wave_barrier();
            }
        }
        // observations in no camera order: after four cameras the rest of the wave adds its terms lane by lane
        if (todo != 0 && active && ((todo >> lane) & 1ull)) {
            double *hc = cam_rep + (size_t)(blockIdx.x % BA_REPLICAS) * F * 42 + 42 * c;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = 0; b < 6; ++b) unsafeAtomicAdd(hc + a * 6 + b, Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
                unsafeAtomicAdd(hc + 36 + a, Jc[a] * rr[0] + Jc[6 + a] * rr[1]);
            }
        }
    }

    if (valid) {
        // an observation's block is 144 bytes at a multiple of 144: nine 16-byte stores rather than eighteen of 8
        double hv[18];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) hv[a * 3 + b] = Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b];
        double2 *hx = (double2 *)(Hcp + 18 * (int64_t)i);
#pragma unroll
        for (int k = 0; k < 9; ++k) hx[k] = double2{hv[2 * k], hv[2 * k + 1]};
    }
    if (GATHER) {
        if (valid) {
#pragma unroll
            for (int a = 0; a < 3; ++a) ((double2 *)(Jp_out + 6 * (int64_t)i))[a] = double2{Jp[2 * a], Jp[2 * a + 1]};
            // landmark-major observation lists built on the device: this observation's place in its landmark's list
            if (list_obs && ok) list_obs[list_start[p] + list_slot[i]] = i;
        }
    } else if (active) {  // point blocks: per-lane atomics
        // upper triangle only; ba_cam_sum_kernel mirrors it afterwards
        double *hp = Hpp + 9 * p, *gp = g + 6 * (int64_t)F + 3 * p;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = a; b < 3; ++b) unsafeAtomicAdd(hp + a * 3 + b, Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b]);
            unsafeAtomicAdd(gp + a, Jp[a] * rr[0] + Jp[3 + a] * rr[1]);
        }
    }
    if (lane == 63) unsafeAtomicAdd(cost_at, 0.5 * cs);
}

// Hcc[c] and g[6c..6c+5] = sum over the replicas filled by ba_kernel (overwrites both); with
// Hpp given, the lower triangles of the landmark blocks are filled from the upper ones.
__global__ __launch_bounds__(256) void ba_cam_sum_kernel(int F, const double *__restrict__ cam_rep, double *__restrict__ Hcc,
                                                         double *__restrict__ g, int P, double *__restrict__ Hpp,
                                                         const double *__restrict__ cost_slots, double *__restrict__ cost,
                                                         const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl && ctl->done) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (cost_slots && t < 64) {                   // cost = sum of the slots (overwrites)
        double s = 0;
#pragma unroll
        for (int k = 0; k < BA_COST_SLOTS / 64; ++k) s += cost_slots[t + 64 * k];
        s = wave_sum_d(s);
        if (t == 63) cost[0] = s;
    }
    if (Hpp && t < P) {
        double *hp = Hpp + 9 * (int64_t)t;
        hp[3] = hp[1]; hp[6] = hp[2]; hp[7] = hp[5];
    }
    if (t >= F * 42) return;
    double s = 0;
#pragma unroll
    for (int r = 0; r < BA_REPLICAS; ++r) s += cam_rep[(size_t)r * F * 42 + t];
    const int c = t / 42, e = t - 42 * c;
    if (e < 36) Hcc[36 * (int64_t)c + e] = s;
    else g[6 * (int64_t)c + (e - 36)] = s;
}

// ---- landmark-major observation lists built on the device (large calls of vo_ba_normal_eq) -------------
// slot[i] = how many observations of landmark pi[i] were counted before this one (any order: integer atomics);
// cnt[p] = observations of landmark p.  Out-of-range indices get slot -1 and are left out.
__global__ __launch_bounds__(256) void ba_list_count_kernel(const int64_t *__restrict__ ci, const int64_t *__restrict__ pi, int O, int F,
                                                           int P, int *__restrict__ cnt, int *__restrict__ slot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= O) return;
    const int64_t c = ci[i], p = pi[i];
    slot[i] = (c >= 0 && c < F && p >= 0 && p < P) ? atomicAdd(cnt + p, 1) : -1;
}

#define BA_SCAN 1024
__device__ __forceinline__ int ba_block_scan(int v, int &total)       // inclusive scan over the workgroup
{
    __shared__ int part[BA_SCAN];
    __syncthreads();
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < BA_SCAN; d <<= 1) {
        const int add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    total = part[BA_SCAN - 1];
    return part[threadIdx.x];
}

__global__ __launch_bounds__(BA_SCAN) void ba_scan_sums_kernel(const int *__restrict__ cnt, int P, int *__restrict__ bsum)
{
    const int i = blockIdx.x * BA_SCAN + threadIdx.x;
    int total;
    ba_block_scan(i < P ? cnt[i] : 0, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(BA_SCAN) void ba_scan_offsets_kernel(int *__restrict__ bsum, int nblk)      // exclusive, in place
{
    int carry = 0;
    for (int base = 0; base < nblk; base += BA_SCAN) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? bsum[i] : 0;
        int total;
        const int inc = ba_block_scan(v, total);
        if (i < nblk) bsum[i] = carry + inc - v;
        carry += total;
    }
}

__global__ __launch_bounds__(BA_SCAN) void ba_scan_finish_kernel(const int *__restrict__ cnt, int P, const int *__restrict__ boff,
                                                                 int *__restrict__ start)
{
    const int i = blockIdx.x * BA_SCAN + threadIdx.x;
    const int v = i < P ? cnt[i] : 0;
    int total;
    const int inc = ba_block_scan(v, total);
    if (i < P) start[i] = boff[blockIdx.x] + inc - v;
    if (i == P - 1) start[P] = boff[blockIdx.x] + inc;
}

// Hpp[p], g_p from the landmark's observation list.  The list's order is the order in which the counting
// atomics happened to land; lists of up to 64 entries are walked in ascending observation index instead
// (repeated minimum search over the cached list), so that the float64 sums come out the same in every run.
__global__ __launch_bounds__(256) void ba_point_gather_ordered_kernel(int P, int F, const int *__restrict__ start,
                                                                      const int *__restrict__ list, const double *__restrict__ Jp,
                                                                      const double *__restrict__ r, int obs_start,
                                                                      double *__restrict__ Hpp, double *__restrict__ g)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    double h[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0};
    const int o0 = start[p], k = start[p + 1] - o0;
    int last = -1;
    for (int t = 0; t < k; ++t) {
        int i;
        if (k <= 64) {
            i = 0x7fffffff;
            for (int u = 0; u < k; ++u) {
                const int c = list[o0 + u];
                if (c > last && c < i) i = c;
            }
            last = i;
        } else {
            i = list[o0 + t];
        }
        if (i < obs_start) continue;                      // frozen cameras: no Jacobian rows
        const double *j = Jp + 6 * (int64_t)i;
        const double j0 = j[0], j1 = j[1], j2 = j[2], j3 = j[3], j4 = j[4], j5 = j[5];
        const double r0 = r[2 * (int64_t)i], r1 = r[2 * (int64_t)i + 1];
        h[0] += j0 * j0 + j3 * j3; h[1] += j0 * j1 + j3 * j4; h[2] += j0 * j2 + j3 * j5;
        h[3] += j1 * j1 + j4 * j4; h[4] += j1 * j2 + j4 * j5; h[5] += j2 * j2 + j5 * j5;
        gp[0] += j0 * r0 + j3 * r1; gp[1] += j1 * r0 + j4 * r1; gp[2] += j2 * r0 + j5 * r1;
    }
    double *hp = Hpp + 9 * (int64_t)p;
    hp[0] = h[0]; hp[1] = h[1]; hp[2] = h[2];
    hp[3] = h[1]; hp[4] = h[3]; hp[5] = h[4];
    hp[6] = h[2]; hp[7] = h[4]; hp[8] = h[5];
    double *gq = g + 6 * (int64_t)F + 3 * (int64_t)p;
    gq[0] = gp[0]; gq[1] = gp[1]; gq[2] = gp[2];
}

// From this many observations on, vo_ba_normal_eq builds landmark-major lists and gathers the landmark blocks;
// below it the nine float64 atomics per observation are cheaper than the five extra launches.
#define BA_GATHER_MIN_OBS 150000
static int ba_gather_min_obs()
{
    const char *e = getenv("VO_BA_GATHER_MIN_OBS");       // comparison runs and tests
    return e ? atoi(e) : BA_GATHER_MIN_OBS;
}

template <bool WITH_J>
static int ba_common(vo_ctx *ctx, const double *params, int F, int P, const int64_t *ci, const int64_t *pi,
                     const double *obs, int O, const double *K, int obs_start, double *r, double *cost, double *Hcc,
                     double *Hpp, double *Hcp, double *g, int mem, const char *who)
{
    VO_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    VO_REQUIRE(F >= 0 && P >= 0 && O >= 0, "%s: F=%d P=%d O=%d", who, F, P, O);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "%s: bad mem flag %d", who, mem);
    VO_REQUIRE(K != nullptr && cost != nullptr, "%s: NULL K/cost", who);
    VO_REQUIRE(O == 0 || (params && ci && pi && obs && r), "%s: NULL input buffer", who);
    VO_REQUIRE(!WITH_J || (Hcc && Hpp && g && (O == 0 || Hcp)), "%s: NULL output buffer", who);
    VO_REQUIRE(obs_start >= 0, "%s: obs_start=%d", who, obs_start);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t np_ = 6 * (size_t)F + 3 * (size_t)P;
    int rc;
    BaK Kv;
    memcpy(Kv.k, K, sizeof(Kv.k));
    const double *d_params = params, *d_obs = obs;
    const int64_t *d_ci = ci, *d_pi = pi;
    double *d_r = r, *d_cost = cost, *d_Hcc = Hcc, *d_Hpp = Hpp, *d_Hcp = Hcp, *d_g = g;
    if (mem == VO_MEM_HOST) {
        for (int i = 0; i < O; ++i)
            VO_REQUIRE(ci[i] >= 0 && ci[i] < F && pi[i] >= 0 && pi[i] < P, "%s: observation %d has index out of range (camera %lld, point %lld)",
                       who, i, (long long)ci[i], (long long)pi[i]);
        // one staging block: params | obs | ci | pi | r | cost | Hcc | Hpp | g | Hcp
        size_t off[11], tot = 0;
        const size_t sz[10] = {np_ * 8, (size_t)O * 16, (size_t)O * 8, (size_t)O * 8, (size_t)O * 16, 8,
                               WITH_J ? (size_t)F * 288 : 0, WITH_J ? (size_t)P * 72 : 0, WITH_J ? np_ * 8 : 0,
                               WITH_J ? (size_t)O * 144 : 0};
        for (int k = 0; k < 10; ++k) { off[k] = tot; tot += (sz[k] + 255) / 256 * 256; }
        void *base;
        if ((rc = vo_stage_reserve(ctx, 0, tot, &base))) return rc;
        char *b = (char *)base;
        d_params = (double *)(b + off[0]); d_obs = (double *)(b + off[1]);
        d_ci = (int64_t *)(b + off[2]); d_pi = (int64_t *)(b + off[3]);
        d_r = (double *)(b + off[4]); d_cost = (double *)(b + off[5]);
        d_Hcc = (double *)(b + off[6]); d_Hpp = (double *)(b + off[7]); d_g = (double *)(b + off[8]);
        d_Hcp = (double *)(b + off[9]);
        if (np_) VO_UPLOAD(ctx, (void *)d_params, params, np_ * 8);
        if (O) {
            VO_UPLOAD(ctx, (void *)d_obs, obs, (size_t)O * 16);
            VO_UPLOAD(ctx, (void *)d_ci, ci, (size_t)O * 8);
            VO_UPLOAD(ctx, (void *)d_pi, pi, (size_t)O * 8);
        }
    }
    VO_HIP_CHECK(hipMemsetAsync(d_cost, 0, 8, ctx->stream));
    double *d_rep = nullptr, *d_slots = nullptr;
    const bool gather = WITH_J && O >= ba_gather_min_obs() && P > 0;
    // few workgroups add their cost straight to the result; many go through the slots
    const bool cost_slots = O > BA_COST_SLOTS * BA_THREADS;
    if (WITH_J || cost_slots) {
        void *rp;
        const size_t cam_bytes = WITH_J ? (size_t)BA_REPLICAS * (F ? F : 1) * 42 * 8 : 0;
        const size_t rep_bytes = cam_bytes + (size_t)BA_COST_SLOTS * 8;
        if ((rc = vo_stage_reserve(ctx, 1, rep_bytes, &rp))) return rc;
        d_rep = (double *)rp;
        if (cost_slots) d_slots = (double *)((char *)rp + cam_bytes);
        VO_HIP_CHECK(hipMemsetAsync(d_rep, 0, rep_bytes, ctx->stream));
    }
    if (WITH_J) {
        if (!gather) {
            if (P) VO_HIP_CHECK(hipMemsetAsync(d_Hpp, 0, (size_t)P * 72, ctx->stream));
            if (P) VO_HIP_CHECK(hipMemsetAsync(d_g + 6 * (size_t)F, 0, (size_t)P * 24, ctx->stream));
        }
    }
    if (gather) {
        // work block (context slot 2): point Jacobians | slot | list | cnt | start | block sums
        const int nblk = (P + BA_SCAN - 1) / BA_SCAN;
        size_t off[7], tot = 0;
        const size_t sz[6] = {(size_t)O * 48, (size_t)O * 4, (size_t)O * 4, (size_t)P * 4, ((size_t)P + 1) * 4, (size_t)nblk * 4};
        for (int k = 0; k < 6; ++k) { off[k] = tot; tot += (sz[k] + 255) / 256 * 256; }
        void *wp;
        if ((rc = vo_stage_reserve(ctx, 2, tot, &wp))) return rc;
        char *b = (char *)wp;
        double *d_Jp = (double *)(b + off[0]);
        int *d_slot = (int *)(b + off[1]), *d_list = (int *)(b + off[2]), *d_cnt = (int *)(b + off[3]), *d_start = (int *)(b + off[4]),
            *d_bsum = (int *)(b + off[5]);
        VO_HIP_CHECK(hipMemsetAsync(d_cnt, 0, (size_t)P * 4, ctx->stream));
        hipLaunchKernelGGL(ba_list_count_kernel, dim3((O + 255) / 256), dim3(256), 0, ctx->stream, d_ci, d_pi, O, F, P, d_cnt, d_slot);
        hipLaunchKernelGGL(ba_scan_sums_kernel, dim3(nblk), dim3(BA_SCAN), 0, ctx->stream, (const int *)d_cnt, P, d_bsum);
        hipLaunchKernelGGL(ba_scan_offsets_kernel, dim3(1), dim3(BA_SCAN), 0, ctx->stream, d_bsum, nblk);
        hipLaunchKernelGGL(ba_scan_finish_kernel, dim3(nblk), dim3(BA_SCAN), 0, ctx->stream, (const int *)d_cnt, P, (const int *)d_bsum, d_start);
        const unsigned blocks = (unsigned)((O + BA_THREADS - 1) / BA_THREADS);
        hipLaunchKernelGGL((ba_kernel<WITH_J, true>), dim3(blocks), dim3(BA_THREADS), 0, ctx->stream, d_params, F, P, d_ci, d_pi, d_obs, O, Kv,
                           obs_start, d_r, d_cost, d_Hcc, d_Hpp, d_Hcp, d_g, d_rep, d_slots, d_Jp, (const int *)d_start, (const int *)d_slot, d_list);
        hipLaunchKernelGGL(ba_point_gather_ordered_kernel, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, P, F, (const int *)d_start,
                           (const int *)d_list, (const double *)d_Jp, (const double *)d_r, obs_start, d_Hpp, d_g);
        hipLaunchKernelGGL(ba_cam_sum_kernel, dim3((F * 42 + 255) / 256), dim3(256), 0, ctx->stream, F, d_rep, d_Hcc, d_g, 0, (double *)nullptr,
                           (const double *)d_slots, d_cost);
        VO_HIP_CHECK(hipGetLastError());
    } else {
    if (O) {
        const unsigned blocks = (unsigned)((O + BA_THREADS - 1) / BA_THREADS);
        hipLaunchKernelGGL(ba_kernel<WITH_J>, dim3(blocks), dim3(BA_THREADS), 0, ctx->stream, d_params, F, P, d_ci, d_pi,
                           d_obs, O, Kv, obs_start, d_r, d_cost, d_Hcc, d_Hpp, d_Hcp, d_g, d_rep, d_slots);
        VO_HIP_CHECK(hipGetLastError());
    }
    if (WITH_J && (F || P)) {
        const int work = F * 42 > P ? F * 42 : P;
        hipLaunchKernelGGL(ba_cam_sum_kernel, dim3((work + 255) / 256), dim3(256), 0, ctx->stream, F, d_rep, d_Hcc, d_g, P, d_Hpp,
                           (const double *)d_slots, d_cost);
        VO_HIP_CHECK(hipGetLastError());
    } else if (d_slots) {
        hipLaunchKernelGGL(ba_cam_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, 0, d_rep, d_Hcc, d_g, 0, (double *)nullptr,
                           (const double *)d_slots, d_cost);
        VO_HIP_CHECK(hipGetLastError());
    }
    }
    if (mem == VO_MEM_HOST) {
        // through the context's page-locked result area where they fit (vo_result), one wait for all of them
        if (O) VO_RESULT(ctx, r, d_r, (size_t)O * 16);
        VO_RESULT(ctx, cost, d_cost, 8);
        if (WITH_J) {
            if (F) VO_RESULT(ctx, Hcc, d_Hcc, (size_t)F * 288);
            if (P) VO_RESULT(ctx, Hpp, d_Hpp, (size_t)P * 72);
            if (np_) VO_RESULT(ctx, g, d_g, np_ * 8);
            if (O) VO_RESULT(ctx, Hcp, d_Hcp, (size_t)O * 144);
        }
        VO_RESULTS_FINISH(ctx);
    }
    return VO_OK;
}

extern "C" int vo_ba_residual(vo_ctx *ctx, const double *params, int F, int P, const int64_t *ci, const int64_t *pi,
                              const double *obs, int O, const double *K, double *r, double *cost, int mem)
{
    return ba_common<false>(ctx, params, F, P, ci, pi, obs, O, K, 0, r, cost, nullptr, nullptr, nullptr, nullptr, mem,
                            "vo_ba_residual");
}

extern "C" int vo_ba_normal_eq(vo_ctx *ctx, const double *params, int F, int P, const int64_t *ci, const int64_t *pi,
                               const double *obs, int O, const double *K, int obs_start, double *r, double *cost,
                               double *Hcc, double *Hpp, double *Hcp, double *g, int mem)
{
    return ba_common<true>(ctx, params, F, P, ci, pi, obs, O, K, obs_start, r, cost, Hcc, Hpp, Hcp, g, mem,
                           "vo_ba_normal_eq");
}

// ---------------------------------------------------------------------------
// Pose-only Gauss-Newton refinement in one launch: the refit on the inliers that
// follows the RANSAC loop inside cv.solvePnPRansac (vo_pipeline/poseEstimation.py:61-69;
// OpenCV refits with its own PnP solver, this minimises the same reprojection error).
// One workgroup: every iteration sums the 6x6 normal equations over the selected
// correspondences (every lane's partial sums through LDS, eight lanes per sum), and every
// lane solves the damped system for itself by LDL^T.  Stops when the cost rises (the previous pose is returned), when it stops
// improving by more than 1e-10 relative, when a step falls below 1e-5 of the pose (that step is
// taken; the cost reported is the one before it), or after max_iters.
// ---------------------------------------------------------------------------
#define REFINE_THREADS 256
#define REFINE_STEP_TOL 1e-5
#define REFINE_RED 8          // lanes per sum in the reduction (28 * REFINE_RED <= REFINE_THREADS)
#define REFINE_PITCH (REFINE_THREADS + REFINE_RED)
#define REFINE_CACHED 4096   // correspondences kept in LDS (20 bytes + a flag each, beside the partial sums)
struct Pose6 { double v[6]; };

// Rotation vector of a rotation matrix (row-major 3x3 inside a 3x4 [R|t], row pitch 4) via the unit quaternion,
// largest component first for accuracy near 0 and pi.
__device__ static void rotvec_of_Rt(const double *Rt, double *rv)
{
    const double R0 = Rt[0], R1 = Rt[1], R2 = Rt[2], R3 = Rt[4], R4 = Rt[5], R5 = Rt[6], R6 = Rt[8], R7 = Rt[9], R8 = Rt[10];
    const double tr = R0 + R4 + R8;
    double q0, q1, q2, q3;   // x, y, z, w
    if (tr > R0 && tr > R4 && tr > R8) {
        q3 = 1. + tr; q0 = R7 - R5; q1 = R2 - R6; q2 = R3 - R1;
    } else if (R0 >= R4 && R0 >= R8) {
        q0 = 1. + 2. * R0 - tr; q1 = R1 + R3; q2 = R2 + R6; q3 = R7 - R5;
    } else if (R4 >= R8) {
        q1 = 1. + 2. * R4 - tr; q0 = R1 + R3; q2 = R5 + R7; q3 = R2 - R6;
    } else {
        q2 = 1. + 2. * R8 - tr; q0 = R2 + R6; q1 = R5 + R7; q3 = R3 - R1;
    }
    const double nq = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    q0 /= nq; q1 /= nq; q2 /= nq; q3 /= nq;
    if (q3 < 0) { q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3; }
    const double sn = sqrt(q0 * q0 + q1 * q1 + q2 * q2);
    const double angle = 2. * atan2(sn, q3);
    const double scale = sn < 1e-12 ? 2. : angle / sn;     // angle / sin(angle / 2) -> 2
    rv[0] = scale * q0; rv[1] = scale * q1; rv[2] = scale * q2;
}

// Rotation matrix of a rotation vector and its derivatives with respect to the vector's three components
// (dR/dw_k = [w_k w + w x ((I - R) e_k)]x R / |w|^2; [e_k]x at w = 0), row-major.  Per pose, not per point: the
// refit's point loop then needs one matrix-vector product per derivative.
__device__ static void rotation_and_derivatives(const double *w, double *R, double (*dR)[9])
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    if (th == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { R[k] = (k % 4 == 0); dR[0][k] = 0; dR[1][k] = 0; dR[2][k] = 0; }
        dR[0][5] = -1; dR[0][7] = 1;
        dR[1][2] = 1; dR[1][6] = -1;
        dR[2][1] = -1; dR[2][3] = 1;
        return;
    }
    double s, c;
    sincos(th, &s, &c);
    // two reciprocals instead of twelve divisions: this chain runs on every lane before the point loop can start
    const double ith = 1.0 / th, ith2 = ith * ith;
    const double v[3] = {w[0] * ith, w[1] * ith, w[2] * ith};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[3 * i + j] = (1 - c) * v[i] * v[j] + (i == j ? c : 0);
    R[1] -= s * v[2]; R[2] += s * v[1];
    R[3] += s * v[2]; R[5] -= s * v[0];
    R[6] -= s * v[1]; R[7] += s * v[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double u[3] = {(k == 0) - R[k], (k == 1) - R[3 + k], (k == 2) - R[6 + k]};      // (I - R) e_k
        const double m[3] = {(w[k] * w[0] + (w[1] * u[2] - w[2] * u[1])) * ith2, (w[k] * w[1] + (w[2] * u[0] - w[0] * u[2])) * ith2,
                             (w[k] * w[2] + (w[0] * u[1] - w[1] * u[0])) * ith2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {           // column j of [m]x R = m x (column j of R)
            dR[k][j] = m[1] * R[6 + j] - m[2] * R[3 + j];
            dR[k][3 + j] = m[2] * R[j] - m[0] * R[6 + j];
            dR[k][6 + j] = m[0] * R[3 + j] - m[1] * R[j];
        }
    }
}

// Start pose: `start` (rotation vector, translation), or -- when Rt_src is given -- the [R|t] at
// Rt_src + 12 * (sel_dev ? *sel_dev : sel), converted on the device (a negative index: nothing to refine, NaN out).
// n_dev (optional): the number of correspondences lives on the device.  mask_row (optional): correspondence i is
// selected by mask[mask_row[i]] instead of mask[i].
__global__ __launch_bounds__(REFINE_THREADS) void refine_pose_kernel(const float *__restrict__ X, const float *__restrict__ x,
                                                                     const uint8_t *__restrict__ mask, int n, BaK K, Pose6 start,
                                                                     int max_iters, double *__restrict__ out,
                                                                     const double *__restrict__ Rt_src, int sel,
                                                                     const int *__restrict__ sel_dev, const int *__restrict__ n_dev,
                                                                     const int *__restrict__ mask_row, vo_batch_strides bs)
{
    // A kernel of the per-frame pose chain: a few short wavefronts that, with many streams in flight, share their SIMDs
    // with another launch group's tracker (five wavefronts issuing back to back).  Raised issue priority lets them through.
    __builtin_amdgcn_s_setprio(3);
    // blockIdx.x: the problem (a stream of a batched frame), one workgroup each; its arrays start bs.work / bs.out /
    // bs.mask bytes after the problem before
    if (bs.count > 1) {
        const size_t b = blockIdx.x;
        X = (const float *)((const char *)X + b * bs.work);
        x = (const float *)((const char *)x + b * bs.work);
        if (mask) mask += b * bs.mask;
        out = (double *)((char *)out + b * bs.out);
        if (Rt_src) Rt_src = (const double *)((const char *)Rt_src + b * bs.work);
        if (sel_dev) sel_dev = (const int *)((const char *)sel_dev + b * bs.out);
        if (n_dev) n_dev = (const int *)((const char *)n_dev + b * bs.out);
        if (mask_row) mask_row = (const int *)((const char *)mask_row + b * bs.work);
    }
    // every lane's 28 partial sums, one row per sum: consecutive lanes of a row are consecutive doubles, and a row starts
    // REFINE_RED doubles further into the 256-byte LDS line than the row before it, so the rows that the reducing lanes of
    // a half-wavefront read at once fall on different banks
    __shared__ double part[28][REFINE_PITCH];
    __shared__ double tot[28];
    __shared__ float pts[5][REFINE_CACHED];
    __shared__ uint8_t pts_sel[REFINE_CACHED];
    const int tid = threadIdx.x;
    if (n_dev) n = *n_dev;
    if (Rt_src) {
        if (sel_dev) sel = *sel_dev;
        if (sel < 0) {
            if (tid < 8) out[tid] = __builtin_nan("");
            return;
        }
        // every lane converts the candidate for itself (same reads, same arithmetic): no hand-over through LDS
        const double *Rt = Rt_src + 12 * (int64_t)sel;
        rotvec_of_Rt(Rt, start.v);
        start.v[3] = Rt[3]; start.v[4] = Rt[7]; start.v[5] = Rt[11];
    }
    double cam[6], prev_cam[6], prev_cost = __builtin_inf(), last_cost = __builtin_inf();
#pragma unroll
    for (int k = 0; k < 6; ++k) { cam[k] = start.v[k]; prev_cam[k] = start.v[k]; }
    // The first REFINE_CACHED correspondences are fetched once, all of a lane's reads in flight together, and kept in
    // LDS with their selection flag: in the passes a lane's reads (row map, mask, landmark, pixel) are three
    // dependent global-memory latencies per correspondence, which was half of the kernel's time.
    const int n_cached = min(n, REFINE_CACHED);
    for (int i0 = tid; i0 < n_cached; i0 += 4 * REFINE_THREADS) {
        int sel[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * REFINE_THREADS;
            sel[u] = i < n_cached ? (mask_row ? mask_row[i] : i) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sel[u] = sel[u] < 0 ? 0 : (mask ? (int)mask[sel[u]] : 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * REFINE_THREADS;
            if (i < n_cached) {
                pts_sel[i] = (uint8_t)(sel[u] != 0);
                pts[0][i] = X[3 * (int64_t)i]; pts[1][i] = X[3 * (int64_t)i + 1]; pts[2][i] = X[3 * (int64_t)i + 2];
                pts[3][i] = x[2 * (int64_t)i]; pts[4][i] = x[2 * (int64_t)i + 1];
            }
        }
    }
    __syncthreads();
    int iters_done = 0;
    for (int it = 0; it <= max_iters; ++it) {
        double acc[28];
#pragma unroll
        for (int k = 0; k < 28; ++k) acc[k] = 0;
        double R[9], dR[3][9];
        rotation_and_derivatives(cam, R, dR);
        for (int i = tid; i < n; i += REFINE_THREADS) {
            float Xf0, Xf1, Xf2, xf0, xf1;
            if (i < n_cached) {
                if (!pts_sel[i]) continue;
                Xf0 = pts[0][i]; Xf1 = pts[1][i]; Xf2 = pts[2][i]; xf0 = pts[3][i]; xf1 = pts[4][i];
            } else {
                if (mask && !mask[mask_row ? mask_row[i] : i]) continue;
                Xf0 = X[3 * (int64_t)i]; Xf1 = X[3 * (int64_t)i + 1]; Xf2 = X[3 * (int64_t)i + 2];
                xf0 = x[2 * (int64_t)i]; xf1 = x[2 * (int64_t)i + 1];
            }
            const double X0 = (double)Xf0, X1 = (double)Xf1, X2 = (double)Xf2;
            const double Y0 = R[0] * X0 + R[1] * X1 + R[2] * X2 + cam[3], Y1 = R[3] * X0 + R[4] * X1 + R[5] * X2 + cam[4],
                         Y2 = R[6] * X0 + R[7] * X1 + R[8] * X2 + cam[5];
            const double p0 = K.k[0] * Y0 + K.k[1] * Y1 + K.k[2] * Y2, p1 = K.k[3] * Y0 + K.k[4] * Y1 + K.k[5] * Y2,
                         p2 = K.k[6] * Y0 + K.k[7] * Y1 + K.k[8] * Y2;
            const double ip2 = 1.0 / p2;
            double rr[2], Jc[12];
            rr[0] = p0 * ip2 - (double)xf0;
            rr[1] = p1 * ip2 - (double)xf1;
            const double a2 = -p0 * ip2 * ip2, b2 = -p1 * ip2 * ip2;
#pragma unroll
            for (int k = 0; k < 3; ++k) {        // d(residual) / d(camera-frame point): also the translation columns
                Jc[3 + k] = ip2 * K.k[k] + a2 * K.k[6 + k];
                Jc[9 + k] = ip2 * K.k[3 + k] + b2 * K.k[6 + k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {        // rotation columns: through dR/dw_k applied to the landmark
                const double d0 = dR[k][0] * X0 + dR[k][1] * X1 + dR[k][2] * X2, d1 = dR[k][3] * X0 + dR[k][4] * X1 + dR[k][5] * X2,
                             d2 = dR[k][6] * X0 + dR[k][7] * X1 + dR[k][8] * X2;
                Jc[k] = Jc[3] * d0 + Jc[4] * d1 + Jc[5] * d2;
                Jc[6 + k] = Jc[9] * d0 + Jc[10] * d1 + Jc[11] * d2;
            }
            int e = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[e++] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] += Jc[a] * rr[0] + Jc[6 + a] * rr[1];
            acc[27] += 0.5 * (rr[0] * rr[0] + rr[1] * rr[1]);
        }
#pragma unroll
        for (int k = 0; k < 28; ++k) part[k][tid] = acc[k];
        __syncthreads();
        if (tid < 28 * REFINE_RED) {
            // REFINE_RED lanes per sum: lane j adds entries j, j + REFINE_RED, ..., then exchanges inside the group
            const int k = tid / REFINE_RED, j = tid % REFINE_RED;
            double s = 0;
#pragma unroll 8
            for (int i = j; i < REFINE_THREADS; i += REFINE_RED) s += part[k][i];
#pragma unroll
            for (int d = 1; d < REFINE_RED; d *= 2) s += __shfl_xor(s, d);
            if (j == 0) tot[k] = s;
        }
        __syncthreads();
        // every lane takes the decision and solves the 6x6 system for itself from the same 28 sums: no broadcast of the
        // outcome, no further barrier (the next pass writes `part`, and `tot` only after its own first barrier)
        int stop = 0;
        {
            const double cost = tot[27];
            if (!(cost == cost) || fabs(cost) == __builtin_inf() || cost > prev_cost * (1 + 1e-12)) {
                // diverged: fall back to the pose before the last step
#pragma unroll
                for (int k = 0; k < 6; ++k) cam[k] = prev_cam[k];
                stop = 1;
            } else {
                last_cost = cost;
                if (it == max_iters || (prev_cost < __builtin_inf() && prev_cost - cost < 1e-10 * fmax(cost, 1e-300))) stop = 1;
            }
            if (!stop) {
                // every loop below has constant bounds and is fully unrolled, so the 6x6 system lives in
                // registers (indexed through run-time loops it went to scratch memory, and the solve
                // of one lane dominated the iteration)
                double A[6][6], g[6];
                double tr = 0;
                {
                    int e = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int b = a; b < 6; ++b) { A[a][b] = tot[e]; A[b][a] = tot[e]; ++e; }
                }
#pragma unroll
                for (int a = 0; a < 6; ++a) { tr += A[a][a]; g[a] = -tot[21 + a]; }
#pragma unroll
                for (int a = 0; a < 6; ++a) A[a][a] += 1e-9 * tr / 6;
                // L D L^T (unit lower triangle L below the diagonal of A, d_k on it): six reciprocals are the only
                // divisions -- this lane's dependent float64 chain is what every other wavefront waits for
                bool ok = true;
                double inv_d[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    double d = A[k][k];
#pragma unroll
                    for (int m = 0; m < k; ++m) d -= A[k][m] * A[k][m] * A[m][m];
                    ok = ok && d > 0;
                    A[k][k] = ok ? d : 1.0;
                    inv_d[k] = 1.0 / A[k][k];
#pragma unroll
                    for (int i = k + 1; i < 6; ++i) {
                        double v = A[i][k];
#pragma unroll
                        for (int m = 0; m < k; ++m) v -= A[i][m] * A[k][m] * A[m][m];
                        A[i][k] = v * inv_d[k];
                    }
                }
                if (!ok) stop = 1;
                else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {        // L z = g
                        double v = g[k];
#pragma unroll
                        for (int m = 0; m < k; ++m) v -= A[k][m] * g[m];
                        g[k] = v;
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) g[k] *= inv_d[k];    // D y = z
#pragma unroll
                    for (int k = 5; k >= 0; --k) {       // L^T x = y
                        double v = g[k];
#pragma unroll
                        for (int m = k + 1; m < 6; ++m) v -= A[m][k] * g[m];
                        g[k] = v;
                    }
                    prev_cost = cost;
                    double gmax = 0, cmax = 1;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        prev_cam[k] = cam[k];
                        cam[k] += g[k];
                        gmax = fmax(gmax, fabs(g[k]));
                        cmax = fmax(cmax, fabs(cam[k]));
                    }
                    iters_done = it + 1;
                    // a step five digits below the pose: converged.  From a RANSAC winner the steps shrink by four to
                    // five orders of magnitude each (2e-3, 7e-6, 8e-11 of the pose on tracked data), so the pose is then
                    // good to ~1e-9 and the pass that would only confirm it -- a third of the kernel's time -- is saved
                    if (gmax <= REFINE_STEP_TOL * cmax) stop = 1;
                }
            }
        }
        if (stop) break;
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = cam[k];
        out[6] = last_cost;
        out[7] = (double)iters_done;
    }
}

// Device pointers; launches the refit and enqueues the copy of {pose (6), cost, iterations} into host_out.
// The caller synchronises the stream before reading host_out (context slot 3 holds the device copy).
int vo_refine_pose_async(vo_ctx *ctx, const float *dX, const float *dx, int n, const uint8_t *dmask, const double *K,
                         const double *pose_in, int max_iters, double *host_out)
{
    BaK Kv;
    memcpy(Kv.k, K, sizeof(Kv.k));
    Pose6 start;
    memcpy(start.v, pose_in, sizeof(start.v));
    int rc;
    void *p;
    if ((rc = vo_stage_reserve(ctx, 3, 64, &p))) return rc;
    double *d_out = (double *)p;
    hipLaunchKernelGGL(refine_pose_kernel, dim3(1), dim3(REFINE_THREADS), 0, ctx->stream, dX, dx, dmask, n, Kv, start, max_iters, d_out,
                       (const double *)nullptr, 0, (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, vo_batch_strides());
    VO_HIP_CHECK(hipGetLastError());
    VO_HIP_CHECK(hipMemcpyAsync(host_out, d_out, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return VO_OK;
}

// The refit started from a candidate that lives on the device: [R|t] number sel (or *sel_dev) of dRt; the number of
// correspondences may live on the device too (n_dev, bounded by n); correspondence i is selected by
// dmask[mask_row ? mask_row[i] : i].  {pose (6), cost, iterations} go to d_out when given (8 doubles on the device,
// left there for the caller to fetch), else through a context buffer into host_out (page-locked) with the copy
// enqueued here; no synchronisation.
int vo_refine_pose_from_Rt_async(vo_ctx *ctx, const float *dX, const float *dx, int n, const int *n_dev, const uint8_t *dmask,
                                 const int *mask_row, const double *K, const double *dRt, int sel, const int *sel_dev,
                                 int max_iters, double *host_out, double *d_out, const vo_batch_strides &bs)
{
    BaK Kv;
    memcpy(Kv.k, K, sizeof(Kv.k));
    Pose6 start = {};
    int rc;
    void *p;
    const bool own = d_out == nullptr;
    VO_REQUIRE(bs.count == 1 || !own, "vo_refine_pose: batched refits leave their results on the device");
    if (bs.count <= 0) return VO_OK;
    if (own) {
        if ((rc = vo_stage_reserve(ctx, 3, 64, &p))) return rc;
        d_out = (double *)p;
    }
    hipLaunchKernelGGL(refine_pose_kernel, dim3(bs.count), dim3(REFINE_THREADS), 0, ctx->stream, dX, dx, dmask, n, Kv, start, max_iters, d_out,
                       dRt, sel, sel_dev, n_dev, mask_row, bs);
    VO_HIP_CHECK(hipGetLastError());
    if (own) VO_HIP_CHECK(hipMemcpyAsync(host_out, d_out, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return VO_OK;
}

extern "C" int vo_refine_pose(vo_ctx *ctx, const float *X, const float *x, int n, const uint8_t *mask, const double *K,
                              const double *pose_in, int max_iters, double *pose_out, double *cost_out, int mem)
{
    VO_REQUIRE(ctx && K && pose_in && pose_out, "vo_refine_pose: NULL argument");
    VO_REQUIRE(n >= 0 && max_iters >= 0, "vo_refine_pose: n=%d max_iters=%d", n, max_iters);
    VO_REQUIRE(n == 0 || (X && x), "vo_refine_pose: NULL point buffers");
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_refine_pose: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const float *dX = X, *dx = x;
    const uint8_t *dm = mask;
    int rc;
    void *p;
    if (mem == VO_MEM_HOST && n) {
        if ((rc = vo_stage_reserve(ctx, 0, (size_t)n * 12, &p))) return rc; dX = (const float *)p;
        if ((rc = vo_stage_reserve(ctx, 1, (size_t)n * 8, &p))) return rc; dx = (const float *)p;
        VO_UPLOAD(ctx, (void *)dX, X, (size_t)n * 12);
        VO_UPLOAD(ctx, (void *)dx, x, (size_t)n * 8);
        if (mask) {
            if ((rc = vo_stage_reserve(ctx, 2, (size_t)n, &p))) return rc; dm = (const uint8_t *)p;
            VO_UPLOAD(ctx, (void *)dm, mask, (size_t)n);
        }
    }
    double h[8];
    if ((rc = vo_refine_pose_async(ctx, dX, dx, n, dm, K, pose_in, max_iters, h))) return rc;
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(pose_out, h, 6 * sizeof(double));
    if (cost_out) *cost_out = h[6];
    return VO_OK;
}

// ---------------------------------------------------------------------------
// Device-resident Levenberg-Marquardt state for one bundle-adjustment window.
//
// The reference solves the window with scipy's least_squares (trf + lsmr,
// vo_pipeline/bundleAdjustment.py:96-97).  Here a step is: Schur complement of
// the landmark blocks (one wavefront per landmark), Cholesky solve of the
// 6F x 6F reduced camera system in one workgroup, back-substitution (one lane
// per landmark), then residual + normal equations at the candidate.  Current and
// candidate linearisations are double-buffered, so accepting a step is a swap.
// Damping: D = H + lambda * diag(H) (+1e-12 on camera diagonals); blocks that
// received no Jacobian rows (frozen cameras, unseen landmarks) become identity
// and their step is zero.
// ---------------------------------------------------------------------------
struct vo_ba {
    vo_ctx *ctx = nullptr;
    int F = 0, P = 0, O = 0, n = 0, obs_start = 0;
    BaK K;
    char *pool = nullptr;
    size_t pool_bytes = 0;
    int64_t *ci = nullptr, *pi = nullptr;
    int *cam32 = nullptr, *pptr = nullptr, *pobs = nullptr, *flag = nullptr;
    double *obs = nullptr, *r = nullptr, *Jp = nullptr, *cam_rep = nullptr, *Dpinv = nullptr, *S = nullptr, *rhs = nullptr, *dc = nullptr;
    double *x[2] = {nullptr, nullptr}, *cost[2] = {nullptr, nullptr}, *Hcc[2] = {nullptr, nullptr},
           *Hpp[2] = {nullptr, nullptr}, *Hcp[2] = {nullptr, nullptr}, *g[2] = {nullptr, nullptr};
    int cur = 0;
    bool linearized = false, have_candidate = false;
    bool solve_pending = false;   // between vo_ba_solve_begin and vo_ba_solve_finish
    int solve_cap = 0, solve_queued = 0;     // trial steps the loop may take in all / queued so far
    BaCtl *ctl = nullptr;      // vo_ba_solve's bookkeeping record
    size_t slot_bytes = 0;     // one slot's {x, cost, Hcc, Hpp, Hcp, g} are one contiguous block of this size
    size_t lds_bytes = 0;  // n*n+2n doubles when the reduced system is kept in LDS, else 0
    size_t solve6_lds = 0; // LDS of the camera-at-a-time solve of such a system (0: the column-at-a-time kernel)
    int solve_w = 0;       // panel width of the blocked solve of a system that is not (0: the column-at-a-time kernel)
    size_t solve_lds = 0;
    int cam_first = 0;     // cameras below this one have no observation with Jacobian rows
    bool panels = false;   // factorise with one launch per panel (ba_solve_panel_kernel)
    double *Lf = nullptr, *zf = nullptr, *dinv = nullptr;      // its factor, eliminated right-hand side, 1 / pivots
    int *pcam = nullptr;   // camera of every entry of the landmark-major lists (pobs)
    int2 *bcam = nullptr;  // (lowest, highest) camera in the lists of landmarks 64 b .. 64 b + 63; (INT_MAX, -1) when they are empty
    bool lists_sorted = false;   // cameras ascend along every landmark's list (observations sorted by camera)
    bool pairs = false;    // strip-wise Schur complement with a block pair per lane (ba_schur_pairs_kernel)
    int n_strips = 0;    // strips of the strip-wise Schur complement of such a system (0: global atomics)
    int strip_budget = 0;  // doubles of LDS a strip may take
    size_t strip_lds = 0;
};

#define SCHUR_THREADS 1024
#define SOLVE_THREADS 256

__device__ __forceinline__ void damped_point_inverse(const double *__restrict__ H, double lambda, double *inv, bool &dead)
{
    double a[9];
    double sabs = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) { a[k] = H[k]; sabs += fabs(a[k]); }
    dead = sabs == 0;
    if (dead) {
#pragma unroll
        for (int k = 0; k < 9; ++k) inv[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    a[0] += lambda * a[0]; a[4] += lambda * a[4]; a[8] += lambda * a[8];
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const double id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = (a[2] * a[7] - a[1] * a[8]) * id; inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    inv[3] = c01 * id; inv[4] = (a[0] * a[8] - a[2] * a[6]) * id; inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    inv[6] = c02 * id; inv[7] = (a[1] * a[6] - a[0] * a[7]) * id; inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// S -= sum_p E_p Dp^-1 E_p^T, rhs += sum_p E_p Dp^-1 g_p with E_p the Hcp blocks of
// landmark p's observations.  One wavefront per landmark (grid-stride).  A lane owns
// one column of the reduced system, (observation b, row j of its block): it keeps
// z = Dp^-1 Hcp_b[j,:]^T, and for every observation a of the landmark (wave-uniform,
// so Hcp_a comes through scalar loads) adds the six entries Hcp_a[i,:] . z -- those on or
// below the diagonal of the reduced system, the half the factorisation reads.  With LDS
// the workgroup sums into a private copy of the reduced system first.
template <bool LDS>
__global__ __launch_bounds__(SCHUR_THREADS) void ba_schur_kernel(int n, int P, int F, const int *__restrict__ pptr,
                                                                 const int *__restrict__ pobs, const int *__restrict__ cam,
                                                                 const double *__restrict__ Hpp, const double *__restrict__ Hcp,
                                                                 const double *__restrict__ g, double lambda,
                                                                 double *__restrict__ Dpinv, double *__restrict__ S,
                                                                 double *__restrict__ rhs, const BaCtl *__restrict__ ctl = nullptr)
{
    extern __shared__ double sl[];
    if (ctl) { if (ctl->done) return; lambda = ctl->lambda; }
    double *Sa = LDS ? sl : S, *ra = LDS ? sl + (size_t)n * n : rhs;
    if (LDS) {
        for (int i = threadIdx.x; i < n * n + n; i += SCHUR_THREADS) sl[i] = 0.0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int waves = SCHUR_THREADS / 64;
    for (int p = blockIdx.x * waves + (threadIdx.x >> 6); p < P; p += gridDim.x * waves) {
        double inv[9];
        bool dead;
        damped_point_inverse(Hpp + 9 * (int64_t)p, lambda, inv, dead);
        if (lane < 9) Dpinv[9 * (int64_t)p + lane] = inv[lane];
        if (dead) continue;  // no Jacobian rows: every Hcp block of this landmark is zero
        const int o0 = pptr[p], k = pptr[p + 1] - o0;
        const double *gp = g + 6 * (int64_t)F + 3 * (int64_t)p;
        const double u0 = inv[0] * gp[0] + inv[1] * gp[1] + inv[2] * gp[2];
        const double u1 = inv[3] * gp[0] + inv[4] * gp[1] + inv[5] * gp[2];
        const double u2 = inv[6] * gp[0] + inv[7] * gp[1] + inv[8] * gp[2];
        for (int t0 = 0; t0 < 6 * k; t0 += 64) {
            const int t = t0 + lane;
            const bool on = t < 6 * k;
            const int b = on ? t / 6 : 0, j = on ? t - 6 * b : 0;
            const int ob = pobs[o0 + b];
            const double *y = Hcp + 18 * (int64_t)ob + 3 * j;
            const double y0 = y[0], y1 = y[1], y2 = y[2];
            const int cb = cam[ob], col = 6 * cb + j;
            if (on) unsafeAtomicAdd(ra + col, y0 * u0 + y1 * u1 + y2 * u2);
            const double z0 = inv[0] * y0 + inv[1] * y1 + inv[2] * y2;
            const double z1 = inv[3] * y0 + inv[4] * y1 + inv[5] * y2;
            const double z2 = inv[6] * y0 + inv[7] * y1 + inv[8] * y2;
            // the landmark's observations and their cameras, 64 at a time in a register each: the loop below then takes
            // them with a lane read instead of two dependent memory reads per step
            for (int a0 = 0; a0 < k; a0 += 64) {
                const int na = min(64, k - a0);
                const int my_oa = lane < na ? pobs[o0 + a0 + lane] : 0;
                const int my_ca = lane < na ? cam[my_oa] : 0;
                for (int a = 0; a < na; ++a) {
                    const int oa = __builtin_amdgcn_readlane(my_oa, a);
                    const int ca = __builtin_amdgcn_readlane(my_ca, a);
                    const double *x = Hcp + 18 * (int64_t)oa;
                    double *dst = Sa + (size_t)(6 * ca) * n + col;
                    // the solver reads the lower triangle only (row >= column): the other half is not formed
                    if (ca < cb) continue;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const double v = x[3 * i] * z0 + x[3 * i + 1] * z1 + x[3 * i + 2] * z2;
                        if (on && v != 0.0 && (ca > cb || i >= j)) unsafeAtomicAdd(dst + (size_t)i * n, -v);
                    }
                }
            }
        }
    }
    if (LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < n * n; i += SCHUR_THREADS)
            if (sl[i] != 0.0) unsafeAtomicAdd(S + i, sl[i]);
        for (int i = threadIdx.x; i < n; i += SCHUR_THREADS)
            if (sl[(size_t)n * n + i] != 0.0) unsafeAtomicAdd(rhs + i, sl[(size_t)n * n + i]);
    }
}


// The Schur complement of a reduced system too large for one LDS copy (more than 23 cameras): the system is cut into
// strips of cameras' rows (ba_strip_bounds), and a workgroup sums one strip's share of its landmarks in LDS -- the same per-landmark
// work as ba_schur_kernel, with the row blocks outside the strip skipped -- before adding it to global memory once.
// blockIdx.y = strip, blockIdx.x = slice of the landmarks.  With the whole system in global memory every one of the
// ~36 k^2 / 2 entries of a landmark seen k times is a float64 atomic on L2 (0.54 ms at 60 cameras and 20 000 landmarks);
// here those land in LDS, and global memory sees one add per non-zero entry of a strip and workgroup.
// Strip 0's workgroups also form Dp^-1 and the right-hand side.
// Strip `want` of the cameras cam_first .. F-1: each strip takes as many cameras as keep its rows, from the column of
// cam_first up to its own diagonal blocks, within `budget` doubles -- early strips are short in columns and so tall in
// rows, and all hold about the same number of entries.  Returns false past the last strip.
__host__ __device__ static inline bool ba_strip_bounds(int cam_first, int F, int budget, int want, int &c0, int &c1)
{
    c0 = cam_first;
    for (int s = 0; c0 < F; ++s) {
        int h = 1;
        while (c0 + h < F && 36 * (h + 1) * (c0 + h + 1 - cam_first) <= budget) ++h;
        c1 = c0 + h;
        if (s == want) return true;
        c0 = c1;
    }
    return false;
}

__global__ __launch_bounds__(SCHUR_THREADS) void ba_schur_strip_kernel(int n, int P, int F, int cam_first, int budget, const int *__restrict__ pptr,
                                                                       const int *__restrict__ pobs, const int *__restrict__ cam,
                                                                       const double *__restrict__ Hpp, const double *__restrict__ Hcp,
                                                                       const double *__restrict__ g, double lambda,
                                                                       double *__restrict__ Dpinv, double *__restrict__ S,
                                                                       double *__restrict__ rhs, const BaCtl *__restrict__ ctl = nullptr)
{
    extern __shared__ double sl[];
    if (ctl) { if (ctl->done) return; lambda = ctl->lambda; }
    int c0, c1;                                                            // cameras of this strip
    if (!ba_strip_bounds(cam_first, F, budget, blockIdx.y, c0, c1)) return;
    // only columns from the first camera with unknowns up to the strip's own diagonal blocks are needed
    const int colbase = 6 * cam_first, rows = 6 * (c1 - c0), cols = 6 * c1 - colbase;
    for (int i = threadIdx.x; i < rows * cols; i += SCHUR_THREADS) sl[i] = 0.0;
    __syncthreads();
    const bool first = blockIdx.y == 0;
    const int lane = threadIdx.x & 63;
    const int waves = SCHUR_THREADS / 64;
    for (int p = blockIdx.x * waves + (threadIdx.x >> 6); p < P; p += gridDim.x * waves) {
        const int o0 = pptr[p], k = pptr[p + 1] - o0;
        if (!first) {
            // a landmark is seen by a run of neighbouring cameras, so most strips hold none of its rows: look at its
            // cameras first and leave before the inverse and the block reads (the first strip also writes the
            // inverses and the right-hand side of every landmark, so it visits them all)
            bool hit = false;
            for (int a0 = 0; a0 < k; a0 += 64)
                if (a0 + lane < k) {
                    const int ca = cam[pobs[o0 + a0 + lane]];
                    hit = hit || (ca >= c0 && ca < c1);
                }
            if (__ballot(hit) == 0) continue;
        }
        double inv[9];
        bool dead;
        damped_point_inverse(Hpp + 9 * (int64_t)p, lambda, inv, dead);
        if (first && lane < 9) Dpinv[9 * (int64_t)p + lane] = inv[lane];
        if (dead) continue;  // no Jacobian rows: every Hcp block of this landmark is zero
        const double *gp = g + 6 * (int64_t)F + 3 * (int64_t)p;
        const double u0 = inv[0] * gp[0] + inv[1] * gp[1] + inv[2] * gp[2];
        const double u1 = inv[3] * gp[0] + inv[4] * gp[1] + inv[5] * gp[2];
        const double u2 = inv[6] * gp[0] + inv[7] * gp[1] + inv[8] * gp[2];
        for (int t0 = 0; t0 < 6 * k; t0 += 64) {
            const int t = t0 + lane;
            const bool on = t < 6 * k;
            const int b = on ? t / 6 : 0, j = on ? t - 6 * b : 0;
            const int ob = pobs[o0 + b];
            const double *y = Hcp + 18 * (int64_t)ob + 3 * j;
            const double y0 = y[0], y1 = y[1], y2 = y[2];
            const int cb = cam[ob], col = 6 * cb + j;
            if (first && on) unsafeAtomicAdd(rhs + col, y0 * u0 + y1 * u1 + y2 * u2);
            if (__ballot(on && cb < c1) == 0) continue;         // none of these columns reaches the strip
            const double z0 = inv[0] * y0 + inv[1] * y1 + inv[2] * y2;
            const double z1 = inv[3] * y0 + inv[4] * y1 + inv[5] * y2;
            const double z2 = inv[6] * y0 + inv[7] * y1 + inv[8] * y2;
            for (int a0 = 0; a0 < k; a0 += 64) {
                const int na = min(64, k - a0);
                const int my_oa = lane < na ? pobs[o0 + a0 + lane] : 0;
                const int my_ca = lane < na ? cam[my_oa] : 0;
                for (int a = 0; a < na; ++a) {
                    const int ca = __builtin_amdgcn_readlane(my_ca, a);
                    if (ca < c0 || ca >= c1) continue;               // wave-uniform: another strip's rows
                    const int oa = __builtin_amdgcn_readlane(my_oa, a);
                    const double *x = Hcp + 18 * (int64_t)oa;
                    double *dst = sl + (size_t)(6 * (ca - c0)) * cols + col - colbase;
                    if (!on || ca < cb) continue;                    // above the diagonal: not formed
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const double v = x[3 * i] * z0 + x[3 * i + 1] * z1 + x[3 * i + 2] * z2;
                        if (v != 0.0 && (ca > cb || i >= j)) unsafeAtomicAdd(dst + (size_t)i * cols, -v);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * cols; i += SCHUR_THREADS) {
        const double v = sl[i];
        if (v != 0.0) {
            const int r = i / cols, c = i - r * cols;
            unsafeAtomicAdd(S + (size_t)(6 * c0 + r) * n + colbase + c, v);
        }
    }
}

// Dp^-1 of every landmark for the damping of this trial step (identity for a landmark without Jacobian rows).
__global__ __launch_bounds__(256) void ba_point_inverse_kernel(int P, const double *__restrict__ Hpp, double lambda,
                                                               double *__restrict__ Dpinv, const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl) { if (ctl->done) return; lambda = ctl->lambda; }
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    double inv[9];
    bool dead;
    damped_point_inverse(Hpp + 9 * (int64_t)p, lambda, inv, dead);
#pragma unroll
    for (int k = 0; k < 9; ++k) Dpinv[9 * (int64_t)p + k] = inv[k];
}

// First index of the ascending array v[0..k) whose value is not below key.
__device__ __forceinline__ int ba_lower_bound(const int *__restrict__ v, int k, int key)
{
    int lo = 0, hi = k;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// The strip-wise Schur complement with a block pair per lane.  ba_schur_strip_kernel gives a landmark to a wavefront,
// which walks the landmark's observations one after the other through scalar loads: three quarters of its cycles are
// waits for those loads, and a landmark seen by sixty cameras keeps its wavefront a hundred times as long as one seen
// by six.  Here the unit of work is a pair (a, b <= a) of observations of one landmark -- one 6 x 6 block
// E_a Dp^-1 E_b^T of the reduced system: a wavefront takes 64 landmarks, a lane each, finds by two binary searches
// which of its landmark's observations fall on the strip's cameras (the lists ascend in camera), a prefix sum numbers
// the strip's pairs of all 64, and the lanes then take 64 pairs at a time whatever landmark they belong to: every lane
// loads its two blocks and the landmark's inverse itself, so all of a wavefront's loads are in flight together, and
// adds its 36 products (21 for a pair with itself) to the strip's tile in LDS.  The right-hand side rows of the strip
// come from the pairs (a, a).  Needs the lists' cameras ascending (pcam), the cameras a block of 64 landmarks spans
// (bcam: first, last) and Dp^-1 from ba_point_inverse_kernel.
__global__ __launch_bounds__(SCHUR_THREADS) void ba_schur_pairs_kernel(int n, int P, int F, int cam_first, int budget,
                                                                       const int *__restrict__ pptr, const int *__restrict__ pobs,
                                                                       const int *__restrict__ pcam, const int2 *__restrict__ bcam,
                                                                       const double *__restrict__ Hcp,
                                                                       const double *__restrict__ g, const double *__restrict__ Dpinv,
                                                                       double *__restrict__ S, double *__restrict__ rhs,
                                                                       const BaCtl *__restrict__ ctl = nullptr)
{
    extern __shared__ double sl[];
    if (ctl && ctl->done) return;
    int c0, c1;                                                            // cameras of this strip
    if (!ba_strip_bounds(cam_first, F, budget, blockIdx.y, c0, c1)) return;
    const int colbase = 6 * cam_first, rows = 6 * (c1 - c0), cols = 6 * c1 - colbase;
    double *rl = sl + (size_t)rows * cols;                                 // the strip's rows of the right-hand side
    for (int i = threadIdx.x; i < rows * cols + rows; i += SCHUR_THREADS) sl[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int waves = SCHUR_THREADS / 64;
    const int nblk = (P + 63) / 64, wave = threadIdx.x >> 6;
    // A workgroup takes a block of 64 landmarks at a time and its wavefronts share the block's pairs: landmarks that
    // start at one keyframe are neighbours in number and see the same cameras, so a strip's pairs sit in a few dozen
    // blocks, and with a block per wavefront those few wavefronts did the whole strip.  Most blocks hold nothing for a
    // strip; the range of cameras a block's landmarks see (bcam) tells without touching their lists.
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int2 span = bcam[blk];
        if (span.y < c0 || span.x >= c1) continue;
        const int p = blk * 64 + lane;
        int o0 = 0, k = 0;
        if (p < P) { o0 = pptr[p]; k = pptr[p + 1] - o0; }
        int a_lo = 0, a_hi = 0;                                            // the landmark's observations on the strip's cameras
        if (k > 0 && pcam[o0 + k - 1] >= c0 && pcam[o0] < c1) {
            a_lo = ba_lower_bound(pcam + o0, k, c0);
            a_hi = ba_lower_bound(pcam + o0, k, c1);
        }
        // pairs (a, b) with a_lo <= a < a_hi, b <= a, numbered row after row: pair r of the whole triangle is
        // (a, r - a (a + 1) / 2) with a the last row that starts at or before r
        const int tlo = a_lo * (a_lo + 1) / 2;
        const int cnt = a_hi * (a_hi + 1) / 2 - tlo;
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        const int excl = incl - cnt;
        const int total = __shfl(incl, 63);
        for (int q = 64 * wave; q < total; q += 64 * waves) {
            const int e = q + lane;
            // the landmark of pair e: the last lane whose first pair is at or before e (lanes without pairs share
            // their successor's start, so the last of equals is the one that owns it)
            int L = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int cand = L + step;
                const int first_of = __shfl(excl, cand & 63);
                if (cand < 64 && first_of <= e) L = cand;
            }
            const int base = __shfl(excl, L), tl = __shfl(tlo, L), oL = __shfl(o0, L);
            if (e < total) {
                const int r = e - base + tl;
                int a = (int)((sqrtf(8.0f * (float)r + 1.0f) - 1.0f) * 0.5f);
                while (a * (a + 1) / 2 > r) --a;
                while ((a + 1) * (a + 2) / 2 <= r) ++a;
                const int b = r - a * (a + 1) / 2;
                const int oa = pobs[oL + a], ob = pobs[oL + b], ca = pcam[oL + a], cb = pcam[oL + b];
                const int64_t pL = (int64_t)blk * 64 + L;
                double inv[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) inv[t] = Dpinv[9 * pL + t];
                // an observation's block is 144 bytes at a multiple of 144: 16-byte loads
                const double2 *y2 = (const double2 *)(Hcp + 18 * (int64_t)ob);
                double yv[18];
#pragma unroll
                for (int t = 0; t < 9; ++t) { const double2 two = y2[t]; yv[2 * t] = two.x; yv[2 * t + 1] = two.y; }
                double W[6][3];                                            // Dp^-1 E_b^T, a column of three per row j of E_b
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double y0 = yv[3 * j], y1 = yv[3 * j + 1], y2_ = yv[3 * j + 2];
                    W[j][0] = inv[0] * y0 + inv[1] * y1 + inv[2] * y2_;
                    W[j][1] = inv[3] * y0 + inv[4] * y1 + inv[5] * y2_;
                    W[j][2] = inv[6] * y0 + inv[7] * y1 + inv[8] * y2_;
                }
                const bool same = a == b, diag = ca == cb;
                double u0 = 0, u1 = 0, u2 = 0;
                if (same) {
                    const double *gp = g + 6 * (int64_t)F + 3 * pL;
                    u0 = inv[0] * gp[0] + inv[1] * gp[1] + inv[2] * gp[2];
                    u1 = inv[3] * gp[0] + inv[4] * gp[1] + inv[5] * gp[2];
                    u2 = inv[6] * gp[0] + inv[7] * gp[1] + inv[8] * gp[2];
                }
                const double2 *x2 = (const double2 *)(Hcp + 18 * (int64_t)oa);
                const int row0 = 6 * (ca - c0);
                double *tile = sl + (size_t)row0 * cols + 6 * cb - colbase;
#pragma unroll
                for (int i2 = 0; i2 < 3; ++i2) {                           // two rows of E_a at a time: three 16-byte loads
                    const double2 xa = x2[3 * i2], xb = x2[3 * i2 + 1], xc = x2[3 * i2 + 2];
                    const double xr[2][3] = {{xa.x, xa.y, xb.x}, {xb.y, xc.x, xc.y}};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = 2 * i2 + h;
                        const double x0 = xr[h][0], x1 = xr[h][1], x2_ = xr[h][2];
                        if (same) unsafeAtomicAdd(rl + row0 + i, x0 * u0 + x1 * u1 + x2_ * u2);
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            double v = x0 * W[j][0] + x1 * W[j][1] + x2_ * W[j][2];
                            if (v == 0.0) continue;
                            if (!diag) unsafeAtomicAdd(tile + (size_t)i * cols + j, -v);
                            else if (same) {
                                if (i >= j) unsafeAtomicAdd(tile + (size_t)i * cols + j, -v);    // the block is symmetric: its lower half
                            } else {
                                // two observations of the landmark by one camera: the block and its transpose both fall
                                // on the diagonal block, of which the lower half is kept
                                if (i == j) v += v;
                                unsafeAtomicAdd(tile + (size_t)(i > j ? i : j) * cols + (i > j ? j : i), -v);
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * cols; i += SCHUR_THREADS) {
        const double v = sl[i];
        if (v != 0.0) {
            const int r = i / cols, c = i - r * cols;
            unsafeAtomicAdd(S + (size_t)(6 * c0 + r) * n + colbase + c, v);
        }
    }
    for (int i = threadIdx.x; i < rows; i += SCHUR_THREADS)
        if (rl[i] != 0.0) unsafeAtomicAdd(rhs + 6 * c0 + i, rl[i]);
}

// Adds the damped camera blocks, then solves S dc = rhs - g_c in one workgroup by a
// square-root-free Cholesky (S = L D L^T).  Columns stay unscaled (c_ik = L_ik d_k) and
// the right-hand side is eliminated along as an extra row, so a column step is one
// rank-1 update and one barrier; back-substitution is one barrier per column.
// flag[0] = 1 on success, 0 when a pivot is not positive.
template <bool LDS>
__global__ __launch_bounds__(SOLVE_THREADS) void ba_solve_kernel(int n, int F, double *__restrict__ S, double *__restrict__ rhs,
                                                                 const double *__restrict__ Hcc, const double *__restrict__ g,
                                                                 double lambda, double *__restrict__ dc, int *__restrict__ flag,
                                                                 const BaCtl *__restrict__ ctl = nullptr)
{
    extern __shared__ double sl[];
    if (ctl) { if (ctl->done) return; lambda = ctl->lambda; }
    __shared__ int fail;
    __shared__ unsigned char dead_cam[256];
    double *A = LDS ? sl : S, *b = LDS ? sl + (size_t)n * n : rhs, *dinv = LDS ? sl + (size_t)n * n + n : dc;
    const int tid = threadIdx.x;
    if (tid == 0) fail = 0;
    for (int c = tid; c < F; c += SOLVE_THREADS) {
        double sabs = 0;
        for (int k = 0; k < 36; ++k) sabs += fabs(Hcc[36 * (int64_t)c + k]);
        dead_cam[c] = sabs == 0;
    }
    if (LDS) {
        for (int i = tid; i < n * n; i += SOLVE_THREADS) A[i] = S[i];
        for (int i = tid; i < n; i += SOLVE_THREADS) b[i] = rhs[i];
    }
    __syncthreads();
    for (int t = tid; t < F * 36; t += SOLVE_THREADS) {
        const int c = t / 36, e = t - c * 36, i = e / 6, j = e - i * 6;
        const bool dead = dead_cam[c];
        double v = Hcc[t];
        if (i == j) v += lambda * v + 1e-12;
        if (dead) v = (i == j) ? 1.0 : 0.0;
        const size_t at = (size_t)(6 * c + i) * n + 6 * c + j;
        A[at] = dead ? v : A[at] + v;
    }
    for (int i = tid; i < n; i += SOLVE_THREADS) b[i] -= g[i];
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    for (int k = 0; k < n; ++k) {
        const double d = A[(size_t)k * n + k];  // final since the previous step's barrier; same in every thread
        if (!(d > 0.0)) {
            if (tid == 0) fail = 1;
            break;
        }
        const double id = 1.0 / d;
        if (tid == 0) dinv[k] = id;
        for (int i = k + 1 + ty; i <= n; i += 16) {  // row n is the right-hand side
            const double f = (i < n ? A[(size_t)i * n + k] : b[k]) * id;
            double *row = i < n ? A + (size_t)i * n : b;
            const int jmax = i < n ? i : n - 1;
            for (int j = k + 1 + tx; j <= jmax; j += 16) row[j] -= f * A[(size_t)j * n + k];
        }
        __syncthreads();
    }
    __syncthreads();
    if (fail) {
        __syncthreads();  // dinv may alias dc
        if (tid == 0) flag[0] = 0;
        for (int i = tid; i < n; i += SOLVE_THREADS) dc[i] = 0.0;
        return;
    }
    // b holds z = L^-1 rhs; y = D^-1 z; L^T x = y
    for (int i = tid; i < n; i += SOLVE_THREADS) b[i] *= dinv[i];
    __syncthreads();
    for (int k = n - 1; k > 0; --k) {
        const double xk = b[k];
        for (int i = tid; i < k; i += SOLVE_THREADS) b[i] -= A[(size_t)k * n + i] * dinv[i] * xk;
        __syncthreads();
    }
    double out[(256 * 6 + SOLVE_THREADS - 1) / SOLVE_THREADS];
    int m = 0;
    for (int i = tid; i < n; i += SOLVE_THREADS) out[m++] = dead_cam[i / 6] ? 0.0 : b[i];
    __syncthreads();  // every read of dinv (possibly aliasing dc) is done
    m = 0;
    for (int i = tid; i < n; i += SOLVE_THREADS) dc[i] = out[m++];
    if (tid == 0) flag[0] = 1;
}


// ba_solve_kernel<true> one camera (six columns) at a time.  A column step of the kernel above is a rank-1 update of
// the whole trailing matrix and a barrier, 6F of them, and the back-substitution another 6F barriers; LDS latency and
// the barriers are its time, not arithmetic.  Here a step takes a camera's six columns: every lane factorises the 6x6
// diagonal block for itself in registers (no hand-over), a lane per row eliminates that row's six entries against it
// and parks the scaled row beside the matrix, and the trailing matrix takes one rank-6 update in 2x2 register tiles:
// two barriers per camera.  The back-substitution is blocked the same way, one barrier per camera.  Same LDL^T, same
// unscaled columns, same treatment of cameras without Jacobian rows and of a non-positive pivot.
#define SOLVE6_LD(n) ((n) | 1)              // odd row pitch: a column's entries fall on different LDS banks
__global__ __launch_bounds__(SOLVE_THREADS) void ba_solve6_kernel(int n, int F, const double *__restrict__ S,
                                                                  const double *__restrict__ rhs, const double *__restrict__ Hcc,
                                                                  const double *__restrict__ g, double lambda,
                                                                  double *__restrict__ dc, int *__restrict__ flag,
                                                                  const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl) { if (ctl->done) return; lambda = ctl->lambda; }
    extern __shared__ double sl[];
    __shared__ unsigned char dead_cam[256];
    const int ld = SOLVE6_LD(n), tid = threadIdx.x;
    // rows 0 .. n-1 of the matrix, row n = the right-hand side; then 1 / pivots; then the scaled rows of the current camera
    double *A = sl, *dinv = sl + (size_t)(n + 1) * ld, *Lp = dinv + n;
    for (int c = tid; c < F; c += SOLVE_THREADS) {
        double sabs = 0;
        for (int k = 0; k < 36; ++k) sabs += fabs(Hcc[36 * (int64_t)c + k]);
        dead_cam[c] = sabs == 0;
    }
    for (int i = tid; i < n * n; i += SOLVE_THREADS) {
        const int r = i / n, c = i - r * n;
        A[(size_t)r * ld + c] = S[i];
    }
    for (int i = tid; i < n; i += SOLVE_THREADS) A[(size_t)n * ld + i] = rhs[i] - g[i];
    __syncthreads();
    for (int t = tid; t < F * 36; t += SOLVE_THREADS) {
        const int c = t / 36, e = t - c * 36, i = e / 6, j = e - i * 6;
        const bool dead = dead_cam[c];
        double v = Hcc[t];
        if (i == j) v += lambda * v + 1e-12;
        if (dead) v = (i == j) ? 1.0 : 0.0;
        const size_t at = (size_t)(6 * c + i) * ld + 6 * c + j;
        A[at] = dead ? v : A[at] + v;
    }
    __syncthreads();
    bool ok = true;
    for (int kb = 0; kb < n && ok; kb += 6) {
        // the camera's diagonal block, lower triangle, in every lane's registers: q[r][c] = unscaled column entry
        double q[6][6], id[6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) q[r][c] = A[(size_t)(kb + r) * ld + kb + c];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double d = q[c][c];
            ok = ok && d > 0.0;
            id[c] = 1.0 / (ok ? d : 1.0);
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                const double f = q[r][c] * id[c];
#pragma unroll
                for (int c2 = c + 1; c2 <= r; ++c2) q[r][c2] -= f * q[c2][c];
            }
        }
        if (!ok) break;                                      // the same in every lane
        // rows below the block (and the right-hand side): eliminate the six entries, keep them (unscaled) in place and
        // the scaled ones in Lp
        for (int i = kb + 6 + tid; i <= n; i += SOLVE_THREADS) {
            double x[6];
            double *row = A + (size_t)i * ld + kb;
#pragma unroll
            for (int c = 0; c < 6; ++c) x[c] = row[c];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double f = x[c] * id[c];
#pragma unroll
                for (int c2 = c + 1; c2 < 6; ++c2) x[c2] -= f * q[c2][c];
                Lp[(size_t)i * 6 + c] = f;
                row[c] = x[c];
            }
        }
        __syncthreads();
        // every lane has read the diagonal block by now: its factored entries (the back-substitution reads them; the
        // update below touches rows and columns after the block only) and the reciprocal pivots go back
        if (tid < 6) {
            double dv = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) dv = c == tid ? id[c] : dv;
            dinv[kb + tid] = dv;
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (c <= tid) {
                    double v = 0;
#pragma unroll
                    for (int r = c; r < 6; ++r) v = r == tid ? q[r][c] : v;
                    A[(size_t)(kb + tid) * ld + kb + c] = v;
                }
        }
        // trailing rank-6 update, 2 x 2 tiles of the lower triangle (row n = right-hand side takes part as a row)
        const int m = n + 1 - (kb + 6), tiles = (m + 1) / 2;     // rows / columns left, in tiles of two
        for (int t = tid; t < tiles * tiles; t += SOLVE_THREADS) {
            const int ti = t / tiles, tj = t - ti * tiles;
            if (tj > ti) continue;
            const int i0 = kb + 6 + 2 * ti, j0 = kb + 6 + 2 * tj;
            const int i1 = min(i0 + 1, n), j1 = min(j0 + 1, n - 1);
            if (j0 > n - 1) continue;                             // the right-hand side is a row only
            double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double l0 = Lp[(size_t)i0 * 6 + c], l1 = Lp[(size_t)i1 * 6 + c];
                const double u0 = A[(size_t)j0 * ld + kb + c], u1 = A[(size_t)j1 * ld + kb + c];
                a00 += l0 * u0; a01 += l0 * u1; a10 += l1 * u0; a11 += l1 * u1;
            }
            A[(size_t)i0 * ld + j0] -= a00;
            if (j1 != j0 && j1 <= i0) A[(size_t)i0 * ld + j1] -= a01;
            if (i1 != i0) {
                A[(size_t)i1 * ld + j0] -= a10;
                if (j1 != j0 && j1 <= i1) A[(size_t)i1 * ld + j1] -= a11;
            }
        }
        __syncthreads();
    }
    if (!ok) {
        if (tid == 0) flag[0] = 0;
        for (int i = tid; i < n; i += SOLVE_THREADS) dc[i] = 0.0;
        return;
    }
    // row n holds z = L^-1 rhs; y = D^-1 z; L^T x = y, a camera at a time from the last
    double *b = A + (size_t)n * ld;
    for (int i = tid; i < n; i += SOLVE_THREADS) b[i] *= dinv[i];
    __syncthreads();
    for (int kb = n - 6; kb >= 0; kb -= 6) {
        // the camera's six unknowns, in every lane: x_k = y_k - sum_{i > k in the block} L_ik x_i, L_ik = A[i][k] / d_k
        double x[6];
#pragma unroll
        for (int c = 5; c >= 0; --c) {
            double v = b[kb + c];
#pragma unroll
            for (int r = c + 1; r < 6; ++r) v -= A[(size_t)(kb + r) * ld + kb + c] * dinv[kb + c] * x[r];
            x[c] = v;
        }
        __syncthreads();                                      // every lane has read b[kb ..] before it is overwritten
        if (tid < 6) {
            double v = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) v = c == tid ? x[c] : v;
            b[kb + tid] = v;
        }
        for (int i = tid; i < kb; i += SOLVE_THREADS) {
            double acc = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) acc += A[(size_t)(kb + c) * ld + i] * x[c];
            b[i] -= acc * dinv[i];
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += SOLVE_THREADS) dc[i] = dead_cam[i / 6] ? 0.0 : b[i];
    if (tid == 0) flag[0] = 1;
}

// The same factorisation for reduced systems that do not fit in LDS (more than 23 cameras: the reference's window
// reaches back to its 20th-last keyframe and takes every frame since), still one workgroup, but blocked: a step
// of ba_solve_kernel<false> touches the whole trailing matrix in global memory once per column and waits for it, 6F
// times.  Here W columns at a time are staged in LDS as a panel (column-major: pan[c * ld + r] = A[k0 + r][k0 + c],
// the right-hand side as the last row): the panel's own triangle is factorised in registers, every row
// below it then follows from that triangle alone, a row per thread, and the panel is applied to
// the trailing matrix in one pass on the matrix cores: A[i][j] -= sum_c (pan[i][c] / d_c) pan[j][c].  The back-substitution is blocked the same way: W rows of L^T at a
// time come in with coalesced reads, the W x W triangle is solved by one wave, the rest of the vector is updated in one
// pass.  flag[0] = 1 on success, 0 when a pivot is not positive.
#define SOLVEB_THREADS 512
typedef double vo_d4 __attribute__((ext_vector_type(4)));
// the value lane l (the same for the whole wave) holds, whatever lanes are active
__device__ __forceinline__ double lane_read(double x, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
#define SOLVEB_MAX_W 16
// block number -> (row, column) of a 32 x 32 block on or below the diagonal, rows first
__device__ __forceinline__ void solveb_block_of(int blk, int &bi, int &bj)
{
    bi = (int)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
    while (bi * (bi + 1) / 2 > blk) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
    bj = blk - bi * (bi + 1) / 2;
}
// a block of the trailing matrix (rows up to n: row n is the right-hand side b) in the result layout of the f64 product
__device__ __forceinline__ void solveb_read_block(const double *A, const double *b, int n, int t0, int bi, int bj, int l4, int l15,
                                                  double (&dst)[2][2][4])
{
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = t0 + 32 * bi + 16 * a + l4 + 4 * e, j = t0 + 32 * bj + 16 * q + l15;
                const bool ok = i <= n && j < n && j <= i;
                dst[a][q][e] = ok ? (i < n ? A[(size_t)i * n + j] : b[j]) : 0.0;
            }
}
// A barrier between phases that exchange data through LDS only: waits for this wave's LDS traffic, not for global
// reads still in flight (__syncthreads would wait those out too, and with them any prefetch).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// 1 / d for a pivot: the hardware's estimate and two Newton steps (full precision for the normal values a
// positive pivot takes; the division it replaces is several times as long a chain).
__device__ __forceinline__ double fast_reciprocal(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
#define PANEL_MAX_W 32       // columns per launch of ba_solve_panel_kernel
// LDL^T of a w x w triangle held a row per lane (tv[c] = entry c of row `lane`, c <= lane), columns left unscaled;
// idr[c] = 1 / d_c.  Every lane runs the same steps, the pivot row's entries come by lane reads.  False when a
// pivot is not positive.
__device__ __forceinline__ bool solveb_triangle(double (&tv)[SOLVEB_MAX_W], double (&idr)[SOLVEB_MAX_W], int w, int lane)
{
    bool bad = false;
#pragma unroll
    for (int c = 0; c < SOLVEB_MAX_W; ++c) {
        idr[c] = 0.0;
        if (c < w && !bad) {
            const double d = lane_read(tv[c], c);
            if (!(d > 0.0)) bad = true;
            else {
                const double id = 1.0 / d;
                idr[c] = id;
                const double f = tv[c] * id;
#pragma unroll
                for (int c2 = c + 1; c2 < SOLVEB_MAX_W; ++c2) {
                    const double pv = lane_read(tv[c], c2);          // row c2's entry of column c
                    if (c2 <= lane) tv[c2] -= f * pv;
                }
            }
        }
    }
    return !bad;
}

// The same factorisation spread over the chip, one launch per panel (ba_solve_prep_kernel first,
// ba_solve_blocked_kernel in its back-substitution mode last).  A wave owns one 32 x 32 block on or below the
// diagonal of the trailing matrix and needs no other wave: it factorises the panel's triangle for itself, brings
// the panel entries of its block's 32 rows (lanes 0-31) and 32 columns (lanes 32-63) up to date from that triangle,
// a row per lane, passes them through its own LDS tile into the operand layout of v_mfma_f64_16x16x4_f64 and
// updates the block in place.  The panel's columns of A and entries of b are only read in a launch (the update
// touches columns right of the panel), so no wave sees another's writes; the factor's columns go to Lf, the
// eliminated right-hand side to zf and the pivots' reciprocals to dinv, each written by one owner: the wave of
// block 0 for the triangle, the diagonal block of a row range for the rows below.  A pivot that is not positive
// clears flag[0], and later launches return at once.
// FULL: the panel has PANEL_MAX_W columns and rows start on 16-byte boundaries (n and k0 even), which every panel but a
// window's last is: the column tests fold away and the rows come and go two entries per access.
template <bool FULL>
__global__ __launch_bounds__(64) void ba_solve_panel_kernel(int n, int k0, int w_arg, double *A, double *b, double *__restrict__ Lf,
                                                            double *__restrict__ zf, double *__restrict__ dinv,
                                                            int *__restrict__ flag, const BaCtl *__restrict__ ctl = nullptr)
{
    __shared__ double tile[PANEL_MAX_W * 64];
    // whether there is anything to do is looked at after the panel's reads have been issued: tested first, the two
    // flags are a trip to memory of their own before every launch's real reads (20 launches a trial step)
    const int stop = (ctl ? ctl->done : 0) | (flag[0] == 0);
    const int w = FULL ? PANEL_MAX_W : w_arg;
    const int lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    const int t0 = k0 + w, m = n - t0;          // trailing rows 0 .. m (row m: the right-hand side), columns 0 .. m-1
    int bi, bj;
    solveb_block_of(blockIdx.x, bi, bj);
    double tv[PANEL_MAX_W], v[PANEL_MAX_W];
    const int lr = lane < 32 ? 32 * bi + lane : 32 * bj + lane - 32;
    const bool valid = lr <= m;
    const int i = t0 + lr;
    const double *src = i < n ? A + (size_t)i * n + k0 : b + k0;
    if (FULL) {
        // a triangle row is read whole (its entries right of the diagonal lie inside the matrix and are dropped here)
        const double *trow = A + (size_t)(k0 + (lane < PANEL_MAX_W ? lane : 0)) * n + k0;
#pragma unroll
        for (int c = 0; c < PANEL_MAX_W; c += 2) {
            const double2 two = *(const double2 *)(trow + c);
            tv[c] = (lane < PANEL_MAX_W && c <= lane) ? two.x : 0.0;
            tv[c + 1] = (lane < PANEL_MAX_W && c + 1 <= lane) ? two.y : 0.0;
        }
        const double *vrow = valid ? src : A + (size_t)k0 * n + k0;
#pragma unroll
        for (int c = 0; c < PANEL_MAX_W; c += 2) {
            const double2 two = *(const double2 *)(vrow + c);
            v[c] = valid ? two.x : 0.0;
            v[c + 1] = valid ? two.y : 0.0;
        }
    } else {
#pragma unroll
        for (int c = 0; c < PANEL_MAX_W; ++c) tv[c] = (lane < w && c <= lane) ? A[(size_t)(k0 + lane) * n + k0 + c] : 0.0;
#pragma unroll
        for (int c = 0; c < PANEL_MAX_W; ++c) v[c] = (valid && c < w) ? src[c] : 0.0;
    }
    double cur[2][2][4];
    solveb_read_block(A, b, n, t0, bi, bj, l4, l15, cur);
    if (stop) return;
    // Entry c of every row -- the triangle's (tv, a row per lane) and this block's (v) -- follows from the row's
    // earlier entries and row c of the triangle: x_c -= sum_q (x_q / d_q) T[c][q].  T[c][q] is final after step q
    // and comes by a lane read that serves both; the columns stay unscaled (c_ik = L_ik d_k).
    double ft[PANEL_MAX_W], f[PANEL_MAX_W];
    bool bad = false;
    double my_id = 0.0;                                         // 1 / d_lane, for the lane that owns pivot `lane`
#pragma unroll
    for (int c = 0; c < PANEL_MAX_W; ++c) {
        ft[c] = f[c] = 0.0;
        if (c < w) {
            // fused multiply-adds, written out (the file is built without contraction): half the vector
            // instructions of this chain, which is what a launch's duration consists of
            double at = tv[c], av = v[c];
#pragma unroll
            for (int q = 0; q < c; ++q) {
                const double t = lane_read(tv[q], c);
                at = fma(-ft[q], t, at);
                av = fma(-f[q], t, av);
            }
            tv[c] = at;
            v[c] = av;
            const double d = lane_read(at, c);
            if (!(d > 0.0)) bad = true;
            const double id = fast_reciprocal(d);
            ft[c] = at * id;
            f[c] = av * id;
            my_id = lane == c ? id : my_id;
            tile[c * 64 + lane] = lane < 32 ? -f[c] : av;       // rows' side scaled by -1/d: the product comes out negative
        }
    }
    if (bad) {
        if (blockIdx.x == 0 && lane == 0) flag[0] = 0;
        return;
    }
    // What leaves the launch is written after the chain (inside it every entry was a store of its own behind a
    // branch).  Owners: block 0 for the triangle and the pivots, the diagonal block of a row range for its rows (the
    // factor's entries scaled, L_ik; the right-hand side as it is).
    if (blockIdx.x == 0 && lane < w) {
        dinv[k0 + lane] = my_id;
        double *tri_out = Lf + (size_t)(k0 + lane) * n + k0;
#pragma unroll
        for (int c = 0; c < PANEL_MAX_W; c += 2) {
            if (FULL && c + 1 < lane) *(double2 *)(tri_out + c) = double2{ft[c], ft[c + 1]};
            else {
                if (c < lane) tri_out[c] = ft[c];
                if (c + 1 < lane) tri_out[c + 1] = ft[c + 1];
            }
        }
    }
    if (bi == bj && lane < 32 && valid) {
        const bool fac = i < n;
        double *row_out = fac ? Lf + (size_t)i * n + k0 : zf + k0;
#pragma unroll
        for (int c = 0; c < PANEL_MAX_W; c += 2) {
            const double o0 = fac ? f[c] : v[c], o1 = fac ? f[c + 1] : v[c + 1];
            if (FULL) *(double2 *)(row_out + c) = double2{o0, o1};
            else {
                if (c < w) row_out[c] = o0;
                if (c + 1 < w) row_out[c + 1] = o1;
            }
        }
    }
    __syncthreads();
    vo_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[a][q] = vo_d4{0.0, 0.0, 0.0, 0.0};
    for (int kc = 0; kc < w; kc += 4) {
        const int k = kc + l4;
        const bool kok = k < w;
        double fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            fa[a] = kok ? tile[k * 64 + 16 * a + l15] : 0.0;
            fb[a] = kok ? tile[k * 64 + 32 + 16 * a + l15] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (!(bi == bj && q > a))              // wholly above the diagonal
                    acc[a][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[q], acc[a][q], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ri = t0 + 32 * bi + 16 * a + l4 + 4 * e, cj = t0 + 32 * bj + 16 * q + l15;
                if (ri <= n && cj < n && cj <= ri) {
                    const double out = cur[a][q][e] + acc[a][q][e];
                    if (ri < n) A[(size_t)ri * n + cj] = out;
                    else b[cj] = out;
                }
            }
}

// Before the panels: the damped camera blocks onto the diagonal of S (identity for a camera without unknowns),
// rhs - g_c, unit pivots and a zero right-hand side below `first`, flag[0] = 1.
__global__ __launch_bounds__(256) void ba_solve_prep_kernel(int n, int F, int first, double *__restrict__ A, double *__restrict__ b,
                                                            const double *__restrict__ Hcc, const double *__restrict__ g,
                                                            double lambda, double *__restrict__ zf, double *__restrict__ dinv,
                                                            int *__restrict__ flag, const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl) { if (ctl->done) return; lambda = ctl->lambda; }
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) flag[0] = 1;
    if (t < F * 36) {
        const int c = t / 36, e = t - c * 36, i = e / 6, j = e - i * 6;
        double sabs = 0;
        for (int k = 0; k < 36; ++k) sabs += fabs(Hcc[36 * (int64_t)c + k]);
        const bool dead = sabs == 0;
        double v = Hcc[t];
        if (i == j) v += lambda * v + 1e-12;
        if (dead) v = (i == j) ? 1.0 : 0.0;
        const size_t at = (size_t)(6 * c + i) * n + 6 * c + j;
        A[at] = dead ? v : A[at] + v;
    }
    if (t < n) b[t] -= g[t];
    if (t < first) {
        dinv[t] = 1.0;
        zf[t] = 0.0;
    }
}

// Forward part only: leaves the factor in A, the eliminated right-hand side in b and the pivots' reciprocals in
// dinv_g for ba_backsolve_kernel; flag[0] = 0 when a pivot is not positive, else 1.
__global__ __launch_bounds__(SOLVEB_THREADS) void ba_solve_blocked_kernel(int n, int F, int first, int W, double *__restrict__ A,
                                                                          double *__restrict__ b, const double *__restrict__ Hcc,
                                                                          const double *__restrict__ g, double lambda,
                                                                          int *__restrict__ flag, double *__restrict__ dinv_g,
                                                                          const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl) { if (ctl->done) return; lambda = ctl->lambda; }
    extern __shared__ double sl[];
    __shared__ unsigned char dead_cam[256];
    const int ld = n + 1;                       // panel column length (rows k0 .. n, the last one the right-hand side)
    double *pan = sl, *dinv = sl + (size_t)W * ld;
    const int tid = threadIdx.x;
    bool fail = false;              // every thread sees the same pivots: uniform
    for (int c = tid; c < F; c += SOLVEB_THREADS) {
        double sabs = 0;
        for (int k = 0; k < 36; ++k) sabs += fabs(Hcc[36 * (int64_t)c + k]);
        dead_cam[c] = sabs == 0;
    }
    __syncthreads();
    for (int t = tid; t < F * 36; t += SOLVEB_THREADS) {
        const int c = t / 36, e = t - c * 36, i = e / 6, j = e - i * 6;
        const bool dead = dead_cam[c];
        double v = Hcc[t];
        if (i == j) v += lambda * v + 1e-12;
        if (dead) v = (i == j) ? 1.0 : 0.0;
        const size_t at = (size_t)(6 * c + i) * n + 6 * c + j;
        A[at] = dead ? v : A[at] + v;
    }
    for (int i = tid; i < n; i += SOLVEB_THREADS) b[i] -= g[i];
    // rows and columns below `first` belong to cameras without unknowns: identity there, nothing to eliminate
    for (int i = tid; i < first; i += SOLVEB_THREADS) dinv[i] = 1.0;
    __syncthreads();
    const int lane = tid & 63;
    for (int k0 = first; k0 < n; k0 += W) {
        const int w = min(W, n - k0), rows = n - k0 + 1;
        // The panel's own w x w triangle, a row per lane in registers.  Every wave factorises it for itself: the
        // rows below then take its entries by lane reads, with no barrier and no trip through LDS in between.
        double tv[SOLVEB_MAX_W], idr[SOLVEB_MAX_W], v[SOLVEB_MAX_W];
#pragma unroll
        for (int c = 0; c < SOLVEB_MAX_W; ++c) tv[c] = (lane < w && c <= lane) ? A[(size_t)(k0 + lane) * n + k0 + c] : 0.0;
        // this thread's first row below the triangle (the right-hand side is the last row), read while the triangle is worked on
        int r = w + tid;
        double *src = r < rows ? (k0 + r < n ? A + (size_t)(k0 + r) * n + k0 : b + k0) : nullptr;
        if (src) {
#pragma unroll
            for (int c = 0; c < SOLVEB_MAX_W; ++c) v[c] = c < w ? src[c] : 0.0;
        }
        if (!solveb_triangle(tv, idr, w, lane)) {                   // the same in every wave
            fail = true;
            break;
        }
        if (tid < w) {
#pragma unroll
            for (int c = 0; c < SOLVEB_MAX_W; ++c) {
                if (c == tid) dinv[k0 + c] = idr[c];
                if (c < tid) A[(size_t)(k0 + tid) * n + k0 + c] = tv[c] * idr[c];      // scaled: L_ik
            }
        }
        // entry c of a row takes the row's earlier entries and the triangle only (the factor's columns stay
        // unscaled: c_ik = L_ik d_k)
        while (src) {
            double f[SOLVEB_MAX_W];
#pragma unroll
            for (int c = 0; c < SOLVEB_MAX_W; ++c) {
                if (c < w) {
                    double acc = v[c];
#pragma unroll
                    for (int q = 0; q < c; ++q) acc -= f[q] * lane_read(tv[q], c);
                    f[c] = acc * idr[c];
                    src[c] = k0 + r < n ? f[c] : acc;        // the factor scaled (L_ik), the right-hand side as it is
                    pan[c * ld + r] = acc;
                }
            }
            r += SOLVEB_THREADS;
            src = r < rows ? (k0 + r < n ? A + (size_t)(k0 + r) * n + k0 : b + k0) : nullptr;
            if (src) {
#pragma unroll
                for (int c = 0; c < SOLVEB_MAX_W; ++c) v[c] = c < w ? src[c] : 0.0;
            }
        }
        __syncthreads();
        // Trailing update on the matrix cores: a wave takes 32 x 32 blocks on or below the diagonal as 2 x 2 products of
        // v_mfma_f64_16x16x4_f64 (operands: lane l holds row l & 15 of the panel at column 4 s + (l >> 4) of step s;
        // results: column l & 15, rows (l >> 4) + 4 e), the rows' side scaled by -1/d so the sum comes out negative
        // and is added to the block read at the start.  Rows run over the matrix rows and the right-hand side.
        const int t0 = k0 + w, m = n - t0;
        if (m > 0) {
            const int nb = (m + 1 + 31) / 32, nblk = nb * (nb + 1) / 2, l15 = lane & 15, l4 = lane >> 4;
            // the block a wave takes next is read while the current one is computed: one CU alone has too few
            // reads in flight otherwise, and every block would wait out its own trip to the L2
            int blk = tid >> 6, bi = 0, bj = 0;
            double cur[2][2][4], nxt[2][2][4];
            if (blk < nblk) {
                solveb_block_of(blk, bi, bj);
                solveb_read_block(A, b, n, t0, bi, bj, l4, l15, cur);
            }
            while (blk < nblk) {
                const int blk2 = blk + SOLVEB_THREADS / 64;
                int bi2 = 0, bj2 = 0;
                if (blk2 < nblk) {
                    solveb_block_of(blk2, bi2, bj2);
                    solveb_read_block(A, b, n, t0, bi2, bj2, l4, l15, nxt);
                }
                vo_d4 acc[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[a][q] = vo_d4{0.0, 0.0, 0.0, 0.0};
                for (int kc = 0; kc < w; kc += 4) {
                    const int k = kc + l4;
                    const bool kok = k < w;
                    const double nid = kok ? -dinv[k0 + k] : 0.0;
                    double fa[2], fb[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int ra = w + 32 * bi + 16 * a + l15, rb = w + 32 * bj + 16 * a + l15;
                        fa[a] = (kok && ra < rows) ? pan[k * ld + ra] * nid : 0.0;
                        fb[a] = (kok && rb < rows - 1) ? pan[k * ld + rb] : 0.0;
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            if (!(bi == bj && q > a))              // wholly above the diagonal
                                acc[a][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[q], acc[a][q], 0, 0, 0);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = t0 + 32 * bi + 16 * a + l4 + 4 * e, j = t0 + 32 * bj + 16 * q + l15;
                            if (i <= n && j < n && j <= i) {
                                const double out = cur[a][q][e] + acc[a][q][e];
                                if (i < n) A[(size_t)i * n + j] = out;
                                else b[j] = out;
                            }
                            cur[a][q][e] = nxt[a][q][e];
                        }
                blk = blk2;
                bi = bi2;
                bj = bj2;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < n; i += SOLVEB_THREADS) dinv_g[i] = dinv[i];
    if (tid == 0) flag[0] = fail ? 0 : 1;
}

// Back-substitution for the blocked factorisations: A holds the factor L below the diagonal, b the eliminated right-hand side z = L^-1 rhs, dinv_g the pivots' reciprocals; y = D^-1 z, L^T x = y,
// W rows of L^T at a time from the bottom: the rows come in with coalesced reads, the W x W triangle is solved by one
// wave in registers, the rest of the vector is updated in one pass.  One workgroup.  dc = x, zero for cameras
// without unknowns and when flag[0] says the factorisation failed.
template <int PF>
__global__ __launch_bounds__(SOLVEB_THREADS) void ba_backsolve_kernel(int n, int F, int first, int W, const double *__restrict__ A,
                                                                      const double *__restrict__ b, const double *__restrict__ dinv_g,
                                                                      const double *__restrict__ Hcc, double *__restrict__ dc,
                                                                      const int *__restrict__ flag, const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl && ctl->done) return;
    extern __shared__ double sl[];
    __shared__ unsigned char dead_cam[256];
    __shared__ double xs[SOLVEB_MAX_W];
    const int ld = n + 1;
    double *pan = sl, *bs = sl + (size_t)W * ld;
    const int tid = threadIdx.x;
    if (flag[0] == 0) {
        for (int i = tid; i < n; i += SOLVEB_THREADS) dc[i] = 0.0;
        return;
    }
    for (int c = tid; c < F; c += SOLVEB_THREADS) {
        double sabs = 0;
        for (int k = 0; k < 36; ++k) sabs += fabs(Hcc[36 * (int64_t)c + k]);
        dead_cam[c] = sabs == 0;
    }
    for (int i = tid; i < n; i += SOLVEB_THREADS) bs[i] = b[i] * dinv_g[i];
    __syncthreads();
    const int last = first < n ? first + ((n - 1 - first) / W) * W : -1;
    // Rows k0 .. k0+w-1 of the factor, entries first <= i < k0 + r of row k0 + r: 32 threads to a row, each taking
    // every 32nd entry.  They do not depend on the unknowns, so the next block's are read into registers while the
    // current block is solved: a block then costs its barriers and arithmetic, not a trip to memory.
    static_assert(SOLVEB_MAX_W * 32 <= SOLVEB_THREADS, "a row of the block per 32 threads");
    // PF: entries of a row a thread may hold, 32 PF >= n - first (the host picks the smallest of 8, 16, 24, 48: the
    // loops below are unrolled PF times with a test each, which at 48 was a third of a block's time).  A variant with an
    // unknown per thread (no LDS staging, one barrier per block, four blocks in flight) that read whole rows took twice
    // as long, and warming the L2 with the factor from other workgroups changed nothing.
    const int pr = tid >> 5, px = tid & 31;
    double pf[PF];
    auto fetch = [&](int kf) {
        const int wf = min(W, n - kf), colsf = kf + wf - 1 - first;
        // every read is made, wanted or not (the others are left out when the values go to LDS): a read under a
        // condition makes the compiler wait for all earlier ones first.  A row's reads may run up to 31 entries past
        // its diagonal, so past the end of the matrix by as much in the last row: the caller keeps that much readable.
        const double *row = A + (size_t)(kf + min(pr, wf - 1)) * n + first + px;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (32 * u < colsf) pf[u] = row[32 * u];
        }
    };
    if (last >= first) fetch(last);
    for (int k0 = last; k0 >= first; k0 -= W) {
        const int w = min(W, n - k0);
        // into LDS: pan[r * ld + i] = L[k0 + r][i]; what lands at or right of a row's diagonal is never read
        const int cols = k0 + w - 1 - first;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (32 * u < cols) {
                const int i = first + px + 32 * u;
                if (pr < w && i < n) pan[pr * ld + i] = pf[u];
            }
        }
        if (k0 - W >= first) fetch(k0 - W);
        lds_barrier();
        if (tid < 64) {                             // the block's own triangle: one wave, an unknown per lane
            double bv = tid < w ? bs[k0 + tid] : 0.0;
            double col[SOLVEB_MAX_W];               // this unknown's column of the triangle, read before the serial part
#pragma unroll
            for (int r = 0; r < SOLVEB_MAX_W; ++r) col[r] = (r < w && tid < r) ? pan[r * ld + k0 + tid] : 0.0;
#pragma unroll
            for (int r = SOLVEB_MAX_W - 1; r >= 0; --r)
                if (r < w) bv -= col[r] * lane_read(bv, r);
            if (tid < w) {
                xs[tid] = bv;
                bs[k0 + tid] = bv;
            }
        }
        lds_barrier();
        for (int i = first + tid; i < k0; i += SOLVEB_THREADS) {
            double v = bs[i];
#pragma unroll
            for (int r = 0; r < SOLVEB_MAX_W; ++r)
                if (r < w) v -= pan[r * ld + i] * xs[r];
            bs[i] = v;
        }
        lds_barrier();
    }
    for (int i = tid; i < n; i += SOLVEB_THREADS) dc[i] = dead_cam[i / 6] ? 0.0 : bs[i];
}

// Lanes per landmark in the two kernels below that walk a landmark's observation list: with a lane per landmark a
// wavefront lasts as long as its longest list (sixty observations where most have six), every step a dependent read.
#define BA_LIST_LANES 8
// Sum over the BA_LIST_LANES neighbouring lanes of a landmark; every one of them gets the total.
__device__ __forceinline__ double list_lanes_sum(double v)
{
#pragma unroll
    for (int d = 1; d < BA_LIST_LANES; d <<= 1) v += __shfl_xor(v, d);
    return v;
}

// dp = Dp^-1 (-g_p - sum_obs Hcp^T dc[camera]); candidate = current + (dc, dp).
__global__ __launch_bounds__(256) void ba_backsub_kernel(int n, int P, int F, const int *__restrict__ pptr,
                                                         const int *__restrict__ pobs, const int *__restrict__ cam,
                                                         const double *__restrict__ Hpp, const double *__restrict__ Hcp,
                                                         const double *__restrict__ g, const double *__restrict__ Dpinv,
                                                         const double *__restrict__ dc, const double *__restrict__ x,
                                                         double *__restrict__ x_new, const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl && ctl->done) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) x_new[t] = x[t] + dc[t];
    const int p = t / BA_LIST_LANES, sub = t % BA_LIST_LANES;
    const bool live = p < P;                           // the lanes of a landmark are all live or all not
    double bk[3] = {0, 0, 0};
    if (live)
        for (int o = pptr[p] + sub; o < pptr[p + 1]; o += BA_LIST_LANES) {
            const int ob = pobs[o];
            const double *h = Hcp + 18 * (int64_t)ob, *q = dc + 6 * cam[ob];
            for (int a = 0; a < 6; ++a) {
                bk[0] -= h[3 * a] * q[a];
                bk[1] -= h[3 * a + 1] * q[a];
                bk[2] -= h[3 * a + 2] * q[a];
            }
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) bk[a] = list_lanes_sum(bk[a]);
    if (!live || sub != 0) return;
    double sabs = 0;
    for (int k = 0; k < 9; ++k) sabs += fabs(Hpp[9 * (int64_t)p + k]);
    const int64_t at = 6 * (int64_t)F + 3 * (int64_t)p;
    double d[3] = {0, 0, 0};
    if (sabs != 0) {
        for (int a = 0; a < 3; ++a) bk[a] -= g[at + a];
        const double *iv = Dpinv + 9 * (int64_t)p;
        for (int a = 0; a < 3; ++a) d[a] = iv[3 * a] * bk[0] + iv[3 * a + 1] * bk[1] + iv[3 * a + 2] * bk[2];
    }
    for (int a = 0; a < 3; ++a) x_new[at + a] = x[at + a] + d[a];
}

// Hpp[p] = sum Jp^T Jp, g_p = sum Jp^T r over the observations of landmark p (BA_LIST_LANES lanes per
// landmark, each taking every BA_LIST_LANES-th entry of the list; observations without Jacobian rows carry Jp = 0).
__global__ __launch_bounds__(256) void ba_point_gather_kernel(int P, int F, const int *__restrict__ pptr,
                                                              const int *__restrict__ pobs, const double *__restrict__ Jp,
                                                              const double *__restrict__ r, double *__restrict__ Hpp,
                                                              double *__restrict__ g, const BaCtl *__restrict__ ctl = nullptr)
{
    if (ctl && ctl->done) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = t / BA_LIST_LANES, sub = t % BA_LIST_LANES;
    const bool live = p < P;
    double h[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0};
    if (live)
        for (int o = pptr[p] + sub; o < pptr[p + 1]; o += BA_LIST_LANES) {
            const int64_t i = pobs[o];
            const double *j = Jp + 6 * i;
            const double j0 = j[0], j1 = j[1], j2 = j[2], j3 = j[3], j4 = j[4], j5 = j[5];
            const double r0 = r[2 * i], r1 = r[2 * i + 1];
            h[0] += j0 * j0 + j3 * j3; h[1] += j0 * j1 + j3 * j4; h[2] += j0 * j2 + j3 * j5;
            h[3] += j1 * j1 + j4 * j4; h[4] += j1 * j2 + j4 * j5; h[5] += j2 * j2 + j5 * j5;
            gp[0] += j0 * r0 + j3 * r1; gp[1] += j1 * r0 + j4 * r1; gp[2] += j2 * r0 + j5 * r1;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) h[a] = list_lanes_sum(h[a]);
#pragma unroll
    for (int a = 0; a < 3; ++a) gp[a] = list_lanes_sum(gp[a]);
    if (!live || sub != 0) return;
    double *hp = Hpp + 9 * (int64_t)p;
    hp[0] = h[0]; hp[1] = h[1]; hp[2] = h[2];
    hp[3] = h[1]; hp[4] = h[3]; hp[5] = h[4];
    hp[6] = h[2]; hp[7] = h[4]; hp[8] = h[5];
    double *gq = g + 6 * (int64_t)F + 3 * (int64_t)p;
    gq[0] = gp[0]; gq[1] = gp[1]; gq[2] = gp[2];
}

static int ba_linearize_into(vo_ba *ba, int slot, const BaCtl *ctl = nullptr)
{
    vo_ctx *ctx = ba->ctx;
    const size_t np_ = (size_t)ba->n + 3 * (size_t)ba->P;
    VO_HIP_CHECK(hipMemsetAsync(ba->cost[slot], 0, 8, ctx->stream));
    (void)np_;
    const size_t cam_doubles = (size_t)BA_REPLICAS * ba->F * 42;
    VO_HIP_CHECK(hipMemsetAsync(ba->cam_rep, 0, (cam_doubles + BA_COST_SLOTS) * 8, ctx->stream));
    if (ba->O) {
        const unsigned blocks = (unsigned)((ba->O + BA_THREADS - 1) / BA_THREADS);
        hipLaunchKernelGGL((ba_kernel<true, true>), dim3(blocks), dim3(BA_THREADS), 0, ctx->stream, ba->x[slot], ba->F, ba->P,
                           ba->ci, ba->pi, ba->obs, ba->O, ba->K, ba->obs_start, ba->r, ba->cost[slot], ba->Hcc[slot],
                           ba->Hpp[slot], ba->Hcp[slot], ba->g[slot], ba->cam_rep, ba->cam_rep + cam_doubles, ba->Jp,
                           (const int *)nullptr, (const int *)nullptr, (int *)nullptr, ctl);
        VO_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(ba_cam_sum_kernel, dim3((ba->F * 42 + 255) / 256), dim3(256), 0, ctx->stream, ba->F, ba->cam_rep,
                       ba->Hcc[slot], ba->g[slot], 0, (double *)nullptr, (const double *)(ba->cam_rep + cam_doubles), ba->cost[slot], ctl);
    VO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(ba_point_gather_kernel, dim3((unsigned)(((size_t)ba->P * BA_LIST_LANES + 255) / 256)), dim3(256), 0, ctx->stream, ba->P, ba->F, ba->pptr,
                       ba->pobs, ba->Jp, ba->r, ba->Hpp[slot], ba->g[slot], ctl);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

// The solver's device pool for F cameras, P landmarks and room for O observations.  One pool is kept by the context
// between solver objects (a window's solver is built and destroyed at every keyframe; hipMalloc + hipFree of tens of MB
// were a quarter of building one).
static int ba_alloc(vo_ctx *ctx, int F, int P, int O, const double *K, vo_ba **out)
{
    vo_ba *ba = new vo_ba();
    ba->ctx = ctx; ba->F = F; ba->P = P; ba->O = O; ba->n = 6 * F;
    memcpy(ba->K.k, K, sizeof(ba->K.k));
    const size_t n = (size_t)ba->n, np_ = n + 3 * (size_t)P, Oz = (size_t)(O ? O : 1);
    size_t tot = 0;
    auto take = [&](size_t bytes) { size_t at = tot; tot += (bytes + 255) / 256 * 256; return at; };
    const size_t o_ci = take(Oz * 8), o_pi = take(Oz * 8), o_cam = take(Oz * 4), o_pptr = take(((size_t)P + 1) * 4),
                 o_pobs = take(Oz * 4), o_pcam = take(Oz * 4), o_bcam = take(((size_t)P / 64 + 1) * 8), o_flag = take(4), o_obs = take(Oz * 16), o_r = take(Oz * 16), o_jp = take(Oz * 48), o_rep = take(((size_t)BA_REPLICAS * F * 42 + BA_COST_SLOTS) * 8),
                 o_dpi = take((size_t)P * 72), o_S = take(n * n * 8), o_rhs = take(n * 8), o_dc = take(n * 8),
                 o_Lf = take(n * n * 8), o_zf = take(n * 8), o_dinv = take(n * 8), o_ctl = take(sizeof(BaCtl));
    size_t o_x[2], o_cost[2], o_Hcc[2], o_Hpp[2], o_Hcp[2], o_g[2];
    for (int s = 0; s < 2; ++s) {
        o_x[s] = take(np_ * 8); o_cost[s] = take(8); o_Hcc[s] = take((size_t)F * 288); o_Hpp[s] = take((size_t)P * 72);
        o_Hcp[s] = take(Oz * 144); o_g[s] = take(np_ * 8);
    }
    if (ctx->ba_pool && ctx->ba_pool_bytes >= tot) {
        ba->pool = ctx->ba_pool;
        ba->pool_bytes = ctx->ba_pool_bytes;
        ctx->ba_pool = nullptr;
        ctx->ba_pool_bytes = 0;
    } else {
        if (ctx->ba_pool) {
            // too small for this window (windows grow as the run goes on): let it go, so that this solver's pool is
            // the one kept when it is destroyed
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(ctx->ba_pool);
            ctx->ba_pool = nullptr;
            ctx->ba_pool_bytes = 0;
        }
        // windows grow from keyframe to keyframe: the next ones fit without another allocation.  Never below 128 MB
        // (a window of 100 cameras and 150 000 observations takes 70): of a card's 288 GB that is nothing, and a run
        // that starts with a three-camera window would otherwise regrow the pool six times on its way there
        size_t room = 2 * tot;
        if (room < ((size_t)128 << 20)) room = (size_t)128 << 20;
        hipError_t e = hipMalloc((void **)&ba->pool, room);
        if (e != hipSuccess) {
            e = hipMalloc((void **)&ba->pool, tot);
            if (e != hipSuccess) {
                vo_set_error("vo_ba_create: hipMalloc(%zu) failed: %s", tot, hipGetErrorString(e));
                delete ba;
                return VO_ENOMEM;
            }
            ba->pool_bytes = tot;
        } else {
            ba->pool_bytes = room;
        }
    }
    char *b = ba->pool;
    ba->ci = (int64_t *)(b + o_ci); ba->pi = (int64_t *)(b + o_pi); ba->cam32 = (int *)(b + o_cam);
    ba->pptr = (int *)(b + o_pptr); ba->pobs = (int *)(b + o_pobs); ba->pcam = (int *)(b + o_pcam); ba->bcam = (int2 *)(b + o_bcam); ba->flag = (int *)(b + o_flag);
    ba->obs = (double *)(b + o_obs); ba->r = (double *)(b + o_r); ba->Jp = (double *)(b + o_jp); ba->cam_rep = (double *)(b + o_rep); ba->Dpinv = (double *)(b + o_dpi);
    ba->S = (double *)(b + o_S); ba->rhs = (double *)(b + o_rhs); ba->dc = (double *)(b + o_dc);
    ba->Lf = (double *)(b + o_Lf); ba->zf = (double *)(b + o_zf); ba->dinv = (double *)(b + o_dinv);
    ba->ctl = (BaCtl *)(b + o_ctl);
    ba->slot_bytes = o_x[1] - o_x[0];
    for (int s = 0; s < 2; ++s) {
        ba->x[s] = (double *)(b + o_x[s]); ba->cost[s] = (double *)(b + o_cost[s]); ba->Hcc[s] = (double *)(b + o_Hcc[s]);
        ba->Hpp[s] = (double *)(b + o_Hpp[s]); ba->Hcp[s] = (double *)(b + o_Hcp[s]); ba->g[s] = (double *)(b + o_g[s]);
    }
    *out = ba;
    return VO_OK;
}

static void ba_release_pool(vo_ba *ba)
{
    if (!ba->pool) return;
    vo_ctx *ctx = ba->ctx;
    if (!ctx->ba_pool) { ctx->ba_pool = ba->pool; ctx->ba_pool_bytes = ba->pool_bytes; }
    else (void)hipFree(ba->pool);
    ba->pool = nullptr;
}

// Where the reduced system lives and which kernels solve it: decided once per solver from its size and from the
// first camera with unknowns.
static void ba_configure(vo_ba *ba)
{
    const size_t n = (size_t)ba->n;
    const int F = ba->F, cam_first = ba->cam_first;
    // keep the reduced system in LDS when it fits (up to the 160 KB of a gfx950 CU)
    const size_t need = (n * n + 2 * n) * 8;
    ba->lds_bytes = 0;
    const char *nolds = getenv("VO_BA_NO_LDS");                 // comparison runs: keep even a small reduced system in global memory
    if (nolds && nolds[0] == '1') { /* fall through to the global-memory kernels */ }
    else if (need <= 60 * 1024) ba->lds_bytes = need;
    else if (need + 1024 <= 160 * 1024 &&
             hipFuncSetAttribute((const void *)ba_solve_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) == hipSuccess &&
             hipFuncSetAttribute((const void *)ba_schur_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) == hipSuccess)
        ba->lds_bytes = need;
    (void)hipGetLastError();
    if (ba->lds_bytes) {
        // matrix + right-hand side row at an odd pitch, 1 / pivots, six scaled entries per row
        const size_t s6 = ((size_t)(n + 1) * SOLVE6_LD(n) + n + 6 * (n + 1)) * 8;
        const char *env = getenv("VO_BA_SOLVE");              // "columns": the column-at-a-time kernel (comparison runs, tests)
        if (!(env && env[0] == 'c') && s6 <= 160 * 1024 - 512 &&
            (s6 <= 60 * 1024 || hipFuncSetAttribute((const void *)ba_solve6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s6) == hipSuccess))
            ba->solve6_lds = s6;
        (void)hipGetLastError();
    }
    if (!ba->lds_bytes) {
        // panel (W columns of n + 1 rows), the pivots' reciprocals and the right-hand side in LDS
        const long budget = 150 * 1024 - 16 * (long)n;
        long w = budget / (8 * ((long)n + 1));
        if (w > SOLVEB_MAX_W) w = SOLVEB_MAX_W;
        const size_t lds = ((size_t)w * (n + 1) + 2 * (size_t)n) * 8;
        if (w >= 4 && hipFuncSetAttribute((const void *)ba_solve_blocked_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds) == hipSuccess &&
            hipFuncSetAttribute((const void *)ba_backsolve_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
            hipFuncSetAttribute((const void *)ba_backsolve_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
            hipFuncSetAttribute((const void *)ba_backsolve_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
            hipFuncSetAttribute((const void *)ba_backsolve_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
            ba->solve_w = (int)w;
            ba->solve_lds = lds;
        }
        (void)hipGetLastError();
        // strips of the Schur complement: as many cameras' rows as fit in LDS beside nothing else
        // strips of the Schur complement: LDS for one camera's rows over the whole live width at least
        const size_t live = n - 6 * (size_t)cam_first;
        size_t sbudget = (150 * 1024 - 8 * n) / 8;              // beside the tile: the strip's rows of the right-hand side
        if (sbudget < 6 * live) sbudget = 0;
        int strips = 0, sc0, sc1;
        while (sbudget && ba_strip_bounds(cam_first, F, (int)sbudget, strips, sc0, sc1)) ++strips;
        if (strips >= 1 && hipFuncSetAttribute((const void *)ba_schur_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(sbudget * 8)) == hipSuccess) {
            ba->n_strips = strips;
            ba->strip_budget = (int)sbudget;
            ba->strip_lds = sbudget * 8;
            ba->pairs = ba->lists_sorted &&
                        hipFuncSetAttribute((const void *)ba_schur_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(sbudget * 8 + 8 * n)) == hipSuccess;
        }
        (void)hipGetLastError();
        ba->panels = ba->solve_w > 0;
        const char *env = getenv("VO_BA_SOLVE");          // earlier kernels (comparison runs, tests): "columns", or
        if (env && env[0] == 'c') { ba->solve_w = 0; ba->n_strips = 0; ba->panels = false; }
        if (env && env[0] == 'b') ba->panels = false;     // "blocked": the blocked factorisation in one workgroup
        const char *env2 = getenv("VO_BA_SCHUR");         // "global": sums straight into global memory
        if (env2 && env2[0] == 'g') ba->n_strips = 0;
        if (env2 && env2[0] == 's') ba->pairs = false;    // "strips": a landmark per wavefront (ba_schur_strip_kernel)
    }
}

// Between vo_ba_solve_begin and vo_ba_solve_finish the solver's two slots and its record belong to the queued loop.
#define BA_REFUSE_WHILE_SOLVING(ba, who)                                                                          \
    do {                                                                                                          \
        if ((ba)->solve_pending) {                                                                                \
            vo_set_error("%s: a solve begun on this solver has not been finished (vo_ba_solve_finish)", who);     \
            return VO_ESTATE;                                                                                     \
        }                                                                                                         \
    } while (0)

extern "C" void vo_ba_destroy(vo_ba *ba)
{
    if (!ba) return;
    if (ba->pool && vo_ctx_is_live(ba->ctx)) {
        (void)hipSetDevice(ba->ctx->device);
        (void)hipStreamSynchronize(ba->ctx->stream);
        ba_release_pool(ba);
    }
    delete ba;
}

extern "C" int vo_ba_create(vo_ctx *ctx, int F, int P, int O, const int64_t *ci, const int64_t *pi, const double *obs,
                            const double *K, int obs_start, int mem, vo_ba **out)
{
    VO_REQUIRE(ctx && out && K, "vo_ba_create: NULL argument");
    *out = nullptr;
    VO_REQUIRE(F >= 1 && P >= 1 && O >= 0, "vo_ba_create: F=%d P=%d O=%d", F, P, O);
    VO_REQUIRE(F <= 256, "vo_ba_create: at most 256 cameras per window (F=%d)", F);
    VO_REQUIRE(obs_start >= 0, "vo_ba_create: obs_start=%d", obs_start);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_ba_create: bad mem flag %d", mem);
    VO_REQUIRE(O == 0 || (ci && pi && obs), "vo_ba_create: NULL observation buffers");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    // host copies of the index arrays: range check and landmark-major observation lists
    int64_t *hci = nullptr, *hpi = nullptr;
    const int64_t *sci = ci, *spi = pi;
    if (mem == VO_MEM_DEVICE && O) {
        hci = (int64_t *)malloc((size_t)O * 8);
        hpi = (int64_t *)malloc((size_t)O * 8);
        if (!hci || !hpi) { free(hci); free(hpi); vo_set_error("vo_ba_create: out of host memory"); return VO_ENOMEM; }
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(hci, ci, (size_t)O * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hpi, pi, (size_t)O * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(hci); free(hpi); vo_set_error("vo_ba_create: index download failed: %s", hipGetErrorString(e)); return VO_EHIP; }
        sci = hci; spi = hpi;
    }
    int *cam32 = (int *)malloc(((size_t)O + 1) * 4), *pptr = (int *)calloc((size_t)P + 2, 4), *pobs = (int *)malloc(((size_t)O + 1) * 4);
    int rc = VO_OK, cam_first = 0;
    if (!cam32 || !pptr || !pobs) { rc = VO_ENOMEM; vo_set_error("vo_ba_create: out of host memory"); }
    for (int i = 0; rc == VO_OK && i < O; ++i) {
        if (sci[i] < 0 || sci[i] >= F || spi[i] < 0 || spi[i] >= P) {
            vo_set_error("vo_ba_create: observation %d has index out of range (camera %lld, point %lld)", i, (long long)sci[i],
                         (long long)spi[i]);
            rc = VO_EINVAL;
        }
    }
    vo_ba *ba = nullptr;
    if (rc == VO_OK) {
        // the lists hold the observations with Jacobian rows only (index >= obs_start): the others add nothing to
        // the normal equations, and the cameras below the first one seen among them have no unknowns to solve for
        cam_first = F;
        for (int i = 0; i < O; ++i) {
            cam32[i] = (int)sci[i];
            if (i < obs_start) continue;
            pptr[spi[i] + 1]++;
            if (cam32[i] < cam_first) cam_first = cam32[i];
        }
        for (int p = 0; p < P; ++p) pptr[p + 1] += pptr[p];
        int *fill = (int *)malloc(((size_t)P + 1) * 4);
        if (!fill) { rc = VO_ENOMEM; vo_set_error("vo_ba_create: out of host memory"); }
        else {
            memcpy(fill, pptr, ((size_t)P + 1) * 4);
            for (int i = obs_start < O ? obs_start : O; i < O; ++i) pobs[fill[spi[i]]++] = i;
            free(fill);
        }
    }
    // the camera of every list entry, and whether the cameras ascend along every list
    int *pcam = (int *)malloc(((size_t)O + 1) * 4);
    const size_t nblk64 = (size_t)P / 64 + 1;
    int *bcam = (int *)malloc(nblk64 * 8);
    bool sorted = true;
    if (rc == VO_OK && (!pcam || !bcam)) { rc = VO_ENOMEM; vo_set_error("vo_ba_create: out of host memory"); }
    if (rc == VO_OK) {
        for (size_t b = 0; b < nblk64; ++b) { bcam[2 * b] = 0x7fffffff; bcam[2 * b + 1] = -1; }
        for (int p = 0; p < P; ++p)
            for (int e = pptr[p]; e < pptr[p + 1]; ++e) {
                const int c = pcam[e] = cam32[pobs[e]];
                if (e > pptr[p] && c < pcam[e - 1]) sorted = false;
                if (c < bcam[2 * (p / 64)]) bcam[2 * (p / 64)] = c;
                if (c > bcam[2 * (p / 64) + 1]) bcam[2 * (p / 64) + 1] = c;
            }
    }
    if (rc == VO_OK) rc = ba_alloc(ctx, F, P, O, K, &ba);
    if (rc == VO_OK) {
        ba->obs_start = obs_start; ba->cam_first = cam_first; ba->lists_sorted = sorted;
        const hipMemcpyKind kind = mem == VO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        hipError_t e = hipSuccess;
        if (O) {
            e = hipMemcpy(ba->ci, sci, (size_t)O * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(ba->pi, spi, (size_t)O * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(ba->cam32, cam32, (size_t)O * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(ba->pobs, pobs, (size_t)O * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(ba->pcam, pcam, (size_t)O * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(ba->obs, obs, (size_t)O * 16, kind);
        }
        if (e == hipSuccess) e = hipMemcpy(ba->bcam, bcam, nblk64 * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ba->pptr, pptr, ((size_t)P + 1) * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            vo_set_error("vo_ba_create: upload failed: %s", hipGetErrorString(e));
            ba_release_pool(ba); delete ba; ba = nullptr; rc = VO_EHIP;
        }
    }
    if (ba) ba_configure(ba);
    free(hci); free(hpi); free(cam32); free(pptr); free(pobs); free(pcam); free(bcam);
    *out = ba;
    return rc;
}

// ---- a window's observation arrays from the per-frame lists (host) ----------------------------------------------
extern "C" int vo_ba_window_assemble(int F, const int32_t *counts, const float *const *kps, const int32_t *const *ids, int64_t *ci,
                                     int64_t *pi, float *obs, int32_t *traj_of_point, int32_t *n_points)
{
    VO_REQUIRE(F >= 0 && n_points && (F == 0 || (counts && kps && ids)), "vo_ba_window_assemble: NULL argument or F=%d", F);
    *n_points = 0;
    int64_t O = 0;
    int32_t lo = INT32_MAX, hi = -1;
    for (int f = 0; f < F; ++f) {
        VO_REQUIRE(counts[f] >= 0 && (counts[f] == 0 || (kps[f] && ids[f])), "vo_ba_window_assemble: frame %d: bad count or NULL list", f);
        for (int k = 0; k < counts[f]; ++k) {
            const int32_t t = ids[f][k];
            lo = t < lo ? t : lo;
            hi = t > hi ? t : hi;
        }
        O += counts[f];
    }
    if (O == 0) return VO_OK;
    VO_REQUIRE(ci && pi && obs && traj_of_point, "vo_ba_window_assemble: NULL output");
    VO_REQUIRE(lo >= 0, "vo_ba_window_assemble: negative trajectory id %d", lo);
    const int64_t span = (int64_t)hi - lo + 1;
    if (span > 8 * O + 1024) return 1;                      // too sparse for a table: the caller sorts
    int32_t *number = (int32_t *)malloc((size_t)span * 4);
    if (!number) { vo_set_error("vo_ba_window_assemble: out of host memory"); return VO_ENOMEM; }
    memset(number, 0xff, (size_t)span * 4);
    int32_t P = 0;
    int64_t at = 0;
    for (int f = 0; f < F; ++f) {
        const int32_t n = counts[f];
        if (n) memcpy(obs + 2 * at, kps[f], (size_t)n * 8);
        for (int k = 0; k < n; ++k, ++at) {
            int32_t &slot = number[ids[f][k] - lo];
            if (slot < 0) {
                slot = P;
                traj_of_point[P++] = ids[f][k];
            }
            ci[at] = f;
            pi[at] = slot;
        }
    }
    free(number);
    *n_points = P;
    return VO_OK;
}

// ---- the window's poses as the solver's camera parameters (host) -------------------------------------------------
// BundleAdjustment.bundle_adjustment turns every pose [R|t] into (rotation vector, t) with one cv.Rodrigues call per
// frame (vo_pipeline/bundleAdjustment.py:86-91).  Here: unit quaternion by the largest-pivot rule (the trace or the
// largest diagonal entry decides which component is formed from a sum, so nothing is divided by a small number up to
// and including half turns), angle = 2 atan2(|v|, w) in [0, pi], vector = v * angle / sin(angle / 2) with the series
// of that factor near zero.
extern "C" int vo_ba_camera_params(int F, const double *M, double *params)
{
    VO_REQUIRE(F >= 0 && (F == 0 || (M && params)), "vo_ba_camera_params: NULL argument or F=%d", F);
    for (int f = 0; f < F; ++f) {
        const double *m = M + 12 * (size_t)f;            // rows of [R|t], four entries each
        // cv.Rodrigues first replaces the block by the nearest rotation, U V^T of its singular value decomposition (the
        // window's poses have been through float32: orthonormal to 1e-7 only).  That is the orthogonal factor of the
        // polar decomposition, reached here by Newton's iteration R <- (R + R^-T) / 2: quadratic from a near-rotation
        // (two or three rounds), still convergent from any nonsingular block.
        double r[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
        bool ok = true;
        for (int it = 0; it < 60; ++it) {
            const double c00 = r[4] * r[8] - r[5] * r[7], c01 = r[5] * r[6] - r[3] * r[8], c02 = r[3] * r[7] - r[4] * r[6];
            const double det = r[0] * c00 + r[1] * c01 + r[2] * c02;
            if (!(det > 0.0) || !std::isfinite(det)) { ok = false; break; }
            // cofactors / det = inverse transposed
            const double cof[9] = {c00, c01, c02,
                                   r[2] * r[7] - r[1] * r[8], r[0] * r[8] - r[2] * r[6], r[1] * r[6] - r[0] * r[7],
                                   r[1] * r[5] - r[2] * r[4], r[2] * r[3] - r[0] * r[5], r[0] * r[4] - r[1] * r[3]};
            double change = 0.0;
            for (int e = 0; e < 9; ++e) {
                const double next = 0.5 * (r[e] + cof[e] / det);
                change = fmax(change, fabs(next - r[e]));
                r[e] = next;
            }
            if (change < 1e-15) break;
        }
        if (!ok) { vo_set_error("vo_ba_camera_params: pose %d: the rotation block is singular, a reflection or not finite", f); return VO_EINVAL; }
        const double m00 = r[0], m01 = r[1], m02 = r[2], m10 = r[3], m11 = r[4], m12 = r[5], m20 = r[6], m21 = r[7], m22 = r[8];
        const double tr = m00 + m11 + m22;
        double x, y, z, w;
        if (tr >= m00 && tr >= m11 && tr >= m22) { x = m21 - m12; y = m02 - m20; z = m10 - m01; w = 1.0 + tr; }
        else if (m00 >= m11 && m00 >= m22) { x = 1.0 - tr + 2.0 * m00; y = m10 + m01; z = m20 + m02; w = m21 - m12; }
        else if (m11 >= m22) { x = m01 + m10; y = 1.0 - tr + 2.0 * m11; z = m21 + m12; w = m02 - m20; }
        else { x = m02 + m20; y = m12 + m21; z = 1.0 - tr + 2.0 * m22; w = m10 - m01; }
        const double norm = sqrt(x * x + y * y + z * z + w * w);
        if (!(norm > 0.0) || !std::isfinite(norm)) { vo_set_error("vo_ba_camera_params: pose %d: the rotation block is not finite", f); return VO_EINVAL; }
        const double s = (w < 0 ? -1.0 : 1.0) / norm;
        x *= s; y *= s; z *= s; w *= s;
        const double angle = 2.0 * atan2(sqrt(x * x + y * y + z * z), w), a2 = angle * angle;
        const double factor = angle <= 1e-3 ? 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0 : angle / sin(0.5 * angle);
        double *o = params + 6 * (size_t)f;
        o[0] = x * factor; o[1] = y * factor; o[2] = z * factor;
        o[3] = m[3]; o[4] = m[7]; o[5] = m[11];
    }
    return VO_OK;
}

// ---- a window's solver built from raw observations: outlier pre-filter, compaction and lists on the device ----
// BundleAdjustment._preprocess_data (vo_pipeline/bundleAdjustment.py:111-134) evaluates the residual at the start,
// drops the observations whose squared reprojection error exceeds a threshold, and hands the rest to the solver.  Done
// on the host that is three passes over PCIe (observations up, residuals down, kept observations up again) and a
// fancy-indexing copy per array.  Here the observations go up once: residual at x0 (the solver's own kernel), keep
// flags, an exclusive scan, a stable compaction straight into the solver's arrays, and the landmark-major lists of
// the kept observations with Jacobian rows (count per landmark, scan, fill, each list sorted by observation index
// so that the sums that walk them are the same in every run and equal to those of host-built lists).
struct BaMeta { int kept, frozen, cam_first, bad, unsorted; };

__global__ __launch_bounds__(256) void ba_keep_kernel(const double *__restrict__ r, const int64_t *__restrict__ ci,
                                                      const int64_t *__restrict__ pi, int O, int F, int P, double max_sq,
                                                      int *__restrict__ keep, BaMeta *__restrict__ meta)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { meta->frozen = 0; meta->cam_first = F; }
    if (i >= O) return;
    const int64_t c = ci[i], p = pi[i];
    if (c < 0 || c >= F || p < 0 || p >= P) {
        meta->bad = i + 1;                 // any offender will do for the message
        keep[i] = 0;
        return;
    }
    const double u = r[2 * (int64_t)i], v = r[2 * (int64_t)i + 1];
    keep[i] = (u * u + v * v > max_sq) ? 0 : 1;     // the reference drops on "difference > threshold": NaN stays
}

__global__ __launch_bounds__(256) void ba_compact_kernel(const int *__restrict__ keep, const int *__restrict__ pos,
                                                         const int64_t *__restrict__ ci, const int64_t *__restrict__ pi,
                                                         const double *__restrict__ obs, int O, int n_frozen,
                                                         int64_t *__restrict__ ci2, int64_t *__restrict__ pi2,
                                                         int *__restrict__ cam32, double *__restrict__ obs2,
                                                         BaMeta *__restrict__ meta)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool frozen = false;
    if (i < O && keep[i]) {
        const int j = pos[i];
        const int64_t c = ci[i];
        ci2[j] = c; pi2[j] = pi[i]; cam32[j] = (int)c;
        *(double2 *)(obs2 + 2 * (int64_t)j) = *(const double2 *)(obs + 2 * (int64_t)i);
        frozen = c < n_frozen;
    }
    const unsigned long long b = __ballot(frozen);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&meta->frozen, __popcll(b));
    if (i == 0) meta->kept = pos[O];
}

// observations kept .. with Jacobian rows: position >= meta->frozen (the caller's observations are sorted by camera)
__global__ __launch_bounds__(256) void ba_lists_count_kernel(const int64_t *__restrict__ pi2, const int *__restrict__ cam32,
                                                             BaMeta *__restrict__ meta, int *__restrict__ cnt,
                                                             int *__restrict__ slot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < meta->kept && i >= meta->frozen;
    int cam = 0x7fffffff;
    if (live) {
        slot[i] = atomicAdd(cnt + pi2[i], 1);
        cam = cam32[i];
    }
    // the smallest camera with unknowns: one atomic per wavefront (one per observation would queue on a single address)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cam = min(cam, __shfl_xor(cam, d));
    if ((threadIdx.x & 63) == 0 && cam != 0x7fffffff) atomicMin(&meta->cam_first, cam);
}

__global__ __launch_bounds__(256) void ba_lists_fill_kernel(const int64_t *__restrict__ pi2, const BaMeta *__restrict__ meta,
                                                            const int *__restrict__ pptr, const int *__restrict__ slot,
                                                            int *__restrict__ pobs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= meta->kept || i < meta->frozen) return;
    pobs[pptr[pi2[i]] + slot[i]] = i;
}

// Ascending observation index within every landmark's list.  A wavefront takes a list: lane l holds entry l, counts
// the entries below its own (the indices are distinct, so that count is its place) and stores it there.  Lists
// longer than a wavefront (more than 64 cameras see the landmark) are sorted by one lane.
__global__ __launch_bounds__(256) void ba_lists_sort_kernel(int P, const int *__restrict__ pptr, int *__restrict__ pobs)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int o0 = pptr[p], k = pptr[p + 1] - o0;
    if (k <= 1) return;
    if (k <= 64) {
        const int v = lane < k ? pobs[o0 + lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < k; ++j) rank += __shfl(v, j) < v ? 1 : 0;
        if (lane < k) pobs[o0 + rank] = v;        // every lane has read its entry before any lane writes (one wavefront)
        return;
    }
    if (lane == 0)
        for (int a = 1; a < k; ++a) {
            const int v = pobs[o0 + a];
            int b = a - 1;
            while (b >= 0 && pobs[o0 + b] > v) { pobs[o0 + b + 1] = pobs[o0 + b]; --b; }
            pobs[o0 + b + 1] = v;
        }
}

// pcam[e] = camera of list entry e, bcam[b] = (lowest, highest) camera in the lists of landmarks 64 b .. 64 b + 63 (a
// wavefront each); meta->unsorted is set when the cameras of some list do not ascend.
__global__ __launch_bounds__(256) void ba_list_cams_kernel(int P, const int *__restrict__ pptr, const int *__restrict__ pobs,
                                                           const int *__restrict__ cam32, int *__restrict__ pcam,
                                                           int2 *__restrict__ bcam, BaMeta *__restrict__ meta)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    int prev = -1, lo = 0x7fffffff, hi = -1;
    bool bad = false;
    if (p < P)
        for (int e = pptr[p]; e < pptr[p + 1]; ++e) {
            const int c = cam32[pobs[e]];
            pcam[e] = c;
            bad = bad || c < prev;
            prev = c;
            lo = min(lo, c);
            hi = max(hi, c);
        }
    if (bad) meta->unsorted = 1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, __shfl_xor(lo, d));
        hi = max(hi, __shfl_xor(hi, d));
    }
    if ((threadIdx.x & 63) == 0 && p - (p & 63) <= P) bcam[p >> 6] = int2{lo, hi};
}

__global__ __launch_bounds__(256) void ba_widen_kernel(const float *__restrict__ src, double *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

extern "C" int vo_ba_create_filtered(vo_ctx *ctx, int F, int P, int O, const int64_t *ci, const int64_t *pi, const void *obs_in,
                                     int obs_f32, const double *K, const double *x0, double max_sq_error, int n_frozen_cams,
                                     int mem, vo_ba **out, int32_t *n_kept, int32_t *obs_start)
{
    const double *obs = (const double *)obs_in;
    VO_REQUIRE(ctx && out && K && x0 && n_kept && obs_start, "vo_ba_create_filtered: NULL argument");
    *out = nullptr;
    VO_REQUIRE(F >= 1 && P >= 1 && O >= 0, "vo_ba_create_filtered: F=%d P=%d O=%d", F, P, O);
    VO_REQUIRE(F <= 256, "vo_ba_create_filtered: at most 256 cameras per window (F=%d)", F);
    VO_REQUIRE(n_frozen_cams >= 0, "vo_ba_create_filtered: n_frozen_cams=%d", n_frozen_cams);
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_ba_create_filtered: bad mem flag %d", mem);
    VO_REQUIRE(O == 0 || (ci && pi && obs), "vo_ba_create_filtered: NULL observation buffers");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    vo_ba *ba = nullptr;
    int rc = ba_alloc(ctx, F, P, O, K, &ba);
    if (rc) return rc;
    auto fail = [&](int code) { ba_release_pool(ba); delete ba; return code; };
    const size_t np_ = (size_t)ba->n + 3 * (size_t)P;
    // inputs on the device: the caller's arrays, or a staging block (observations | camera indices | landmark indices)
    const int64_t *d_ci = ci, *d_pi = pi;
    const double *d_obs = obs;
    const int nblk_o = (O + BA_SCAN - 1) / BA_SCAN, nblk_p = (P + BA_SCAN - 1) / BA_SCAN;
    void *wp;
    {
        const size_t in_bytes = mem == VO_MEM_HOST ? ((size_t)O * 40 + 1024) : (obs_f32 ? (size_t)O * 16 + 256 : 0);
        const size_t work = ((size_t)(nblk_o > nblk_p ? nblk_o : nblk_p) + 1) * 4 + 256 + sizeof(BaMeta) + 256 + 3 * ((size_t)O + 64) * 4;
        if ((rc = vo_stage_reserve(ctx, 0, in_bytes + work, &wp))) return fail(rc);
    }
    char *wb = (char *)wp;
    int *d_bsum = (int *)wb;
    BaMeta *d_meta = (BaMeta *)(wb + (((size_t)(nblk_o > nblk_p ? nblk_o : nblk_p) + 1) * 4 + 255) / 256 * 256);
    // keep flags | positions (O + 1) | list slots
    int *d_keep = (int *)((char *)d_meta + 256), *d_pos = d_keep + ((size_t)O + 64), *d_slot = d_pos + ((size_t)O + 64);
    char *in_base = (char *)(d_slot + ((size_t)O + 64));
    if (O && (mem == VO_MEM_HOST || obs_f32)) {
        // float32 observations (what the pipeline holds) travel as they are and are widened on the device
        double *t_obs = (double *)in_base;
        const float *d_f32 = (const float *)obs_in;
        if (mem == VO_MEM_HOST) {
            int64_t *t_ci = (int64_t *)(in_base + ((size_t)O * 16 + 255) / 256 * 256);
            int64_t *t_pi = (int64_t *)((char *)t_ci + ((size_t)O * 8 + 255) / 256 * 256);
            float *t_f32 = (float *)((char *)t_pi + ((size_t)O * 8 + 255) / 256 * 256);
            if (obs_f32) { if ((rc = vo_upload(ctx, t_f32, obs_in, (size_t)O * 8))) return fail(rc); d_f32 = t_f32; }
            else if ((rc = vo_upload(ctx, t_obs, obs_in, (size_t)O * 16))) return fail(rc);
            if ((rc = vo_upload(ctx, t_ci, ci, (size_t)O * 8)) || (rc = vo_upload(ctx, t_pi, pi, (size_t)O * 8))) return fail(rc);
            d_ci = t_ci; d_pi = t_pi;
        }
        if (obs_f32)
            hipLaunchKernelGGL(ba_widen_kernel, dim3((unsigned)(((size_t)O * 2 + 255) / 256)), dim3(256), 0, ctx->stream, d_f32, t_obs,
                               (size_t)O * 2);
        d_obs = t_obs;
    }
    if (mem == VO_MEM_HOST) { if ((rc = vo_upload(ctx, ba->x[0], x0, np_ * 8))) return fail(rc); }
    else if (hipMemcpyAsync(ba->x[0], x0, np_ * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        vo_set_error("vo_ba_create_filtered: copy of x0 failed");
        return fail(VO_EHIP);
    }
    // per-landmark counts in the pool's damped-inverse block (free until the first trial step)
    int *d_cnt = (int *)ba->Dpinv;
    hipError_t e = hipMemsetAsync(d_meta, 0, sizeof(BaMeta), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ba->cost[0], 0, 8, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, (size_t)P * 4, ctx->stream);
    if (e != hipSuccess) { vo_set_error("vo_ba_create_filtered: memset failed: %s", hipGetErrorString(e)); return fail(VO_EHIP); }
    const unsigned ob = (unsigned)((O + 255) / 256);
    if (O) {
        hipLaunchKernelGGL((ba_kernel<false>), dim3((unsigned)((O + BA_THREADS - 1) / BA_THREADS)), dim3(BA_THREADS), 0, ctx->stream,
                           (const double *)ba->x[0], F, P, d_ci, d_pi, d_obs, O, ba->K, 0, ba->r, ba->cost[0], (double *)nullptr,
                           (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL(ba_keep_kernel, dim3(ob), dim3(256), 0, ctx->stream, (const double *)ba->r, d_ci, d_pi, O, F, P, max_sq_error,
                           d_keep, d_meta);
        hipLaunchKernelGGL(ba_scan_sums_kernel, dim3(nblk_o), dim3(BA_SCAN), 0, ctx->stream, (const int *)d_keep, O, d_bsum);
        hipLaunchKernelGGL(ba_scan_offsets_kernel, dim3(1), dim3(BA_SCAN), 0, ctx->stream, d_bsum, nblk_o);
        hipLaunchKernelGGL(ba_scan_finish_kernel, dim3(nblk_o), dim3(BA_SCAN), 0, ctx->stream, (const int *)d_keep, O, (const int *)d_bsum, d_pos);
        hipLaunchKernelGGL(ba_compact_kernel, dim3(ob), dim3(256), 0, ctx->stream, (const int *)d_keep, (const int *)d_pos, d_ci, d_pi, d_obs, O,
                           n_frozen_cams, ba->ci, ba->pi, ba->cam32, ba->obs, d_meta);
        hipLaunchKernelGGL(ba_lists_count_kernel, dim3(ob), dim3(256), 0, ctx->stream, (const int64_t *)ba->pi, (const int *)ba->cam32, d_meta,
                           d_cnt, d_slot);
    }
    hipLaunchKernelGGL(ba_scan_sums_kernel, dim3(nblk_p), dim3(BA_SCAN), 0, ctx->stream, (const int *)d_cnt, P, d_bsum);
    hipLaunchKernelGGL(ba_scan_offsets_kernel, dim3(1), dim3(BA_SCAN), 0, ctx->stream, d_bsum, nblk_p);
    hipLaunchKernelGGL(ba_scan_finish_kernel, dim3(nblk_p), dim3(BA_SCAN), 0, ctx->stream, (const int *)d_cnt, P, (const int *)d_bsum, ba->pptr);
    if (O) {
        hipLaunchKernelGGL(ba_lists_fill_kernel, dim3(ob), dim3(256), 0, ctx->stream, (const int64_t *)ba->pi, (const BaMeta *)d_meta,
                           (const int *)ba->pptr, (const int *)d_slot, ba->pobs);
        hipLaunchKernelGGL(ba_lists_sort_kernel, dim3((P + 3) / 4), dim3(256), 0, ctx->stream, P, (const int *)ba->pptr, ba->pobs);
        hipLaunchKernelGGL(ba_list_cams_kernel, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, P, (const int *)ba->pptr,
                           (const int *)ba->pobs, (const int *)ba->cam32, ba->pcam, ba->bcam, d_meta);
    }
    e = hipGetLastError();
    if (e != hipSuccess) { vo_set_error("vo_ba_create_filtered: launch failed: %s", hipGetErrorString(e)); return fail(VO_EHIP); }
    BaMeta meta = {};
    if ((rc = vo_result(ctx, &meta, d_meta, sizeof(meta))) || (rc = vo_results_finish(ctx))) return fail(rc);
    if (meta.bad) {
        vo_set_error("vo_ba_create_filtered: observation %d has an index out of range", meta.bad - 1);
        return fail(VO_EINVAL);
    }
    ba->O = O ? meta.kept : 0;
    ba->obs_start = O ? meta.frozen : 0;
    ba->cam_first = (O && meta.kept > meta.frozen) ? meta.cam_first : F;
    ba->lists_sorted = !meta.unsorted;
    ba_configure(ba);
    *n_kept = ba->O;
    *obs_start = ba->obs_start;
    *out = ba;
    return VO_OK;
}

extern "C" int vo_ba_linearize(vo_ctx *ctx, vo_ba *ba, const double *x, int mem, double *cost)
{
    VO_REQUIRE(ctx && ba && x && cost, "vo_ba_linearize: NULL argument");
    VO_REQUIRE(ba->ctx == ctx, "vo_ba_linearize: solver belongs to another context");
    BA_REFUSE_WHILE_SOLVING(ba, "vo_ba_linearize");
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_ba_linearize: bad mem flag %d", mem);
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t np_ = (size_t)ba->n + 3 * (size_t)ba->P;
    // a host vector goes through the context's page-locked ring (a copy from pageable memory blocks, and the first one
    // of a process sets up the runtime's own staging: 5 ms)
    if (mem == VO_MEM_HOST) VO_UPLOAD(ctx, ba->x[ba->cur], x, np_ * 8);
    else VO_HIP_CHECK(hipMemcpyAsync(ba->x[ba->cur], x, np_ * 8, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = ba_linearize_into(ba, ba->cur);
    if (rc) return rc;
    VO_RESULT(ctx, cost, ba->cost[ba->cur], 8);
    VO_RESULTS_FINISH(ctx);
    ba->linearized = true;
    ba->have_candidate = false;
    return VO_OK;
}

// ba_backsolve_kernel with the smallest per-thread row buffer that covers the columns with unknowns.
static void ba_backsolve_launch(vo_ba *ba, int first, const double *A, const double *b, const double *dinv, const double *Hcc,
                                const BaCtl *ctl)
{
    vo_ctx *ctx = ba->ctx;
    const int n = ba->n, need = (n - first + 31) / 32;
#define BA_BACKSOLVE(PF) hipLaunchKernelGGL(ba_backsolve_kernel<PF>, dim3(1), dim3(SOLVEB_THREADS), ba->solve_lds, ctx->stream, n, ba->F, \
                                            first, ba->solve_w, A, b, dinv, Hcc, ba->dc, (const int *)ba->flag, ctl)
    if (need <= 8) BA_BACKSOLVE(8);
    else if (need <= 16) BA_BACKSOLVE(16);
    else if (need <= 24) BA_BACKSOLVE(24);
    else BA_BACKSOLVE(48);
#undef BA_BACKSOLVE
}

// One trial step queued on the context's stream: candidate = current + LM step for the damping `lambda` (or, with
// ctl, the damping the device-side record holds; every kernel then leaves at once when the record says done),
// linearised into the other slot.
static int ba_try_step_enqueue(vo_ba *ba, double lambda, const BaCtl *ctl)
{
    vo_ctx *ctx = ba->ctx;
    const int c = ba->cur, o = 1 - c, n = ba->n;
    VO_HIP_CHECK(hipMemsetAsync(ba->S, 0, ((size_t)n * n + 0) * 8, ctx->stream));
    VO_HIP_CHECK(hipMemsetAsync(ba->rhs, 0, (size_t)n * 8, ctx->stream));
    const int waves = SCHUR_THREADS / 64;
    int blocks = (ba->P + waves - 1) / waves;
    if (blocks > 128) blocks = 128;
    if (ba->lds_bytes)
        hipLaunchKernelGGL(ba_schur_kernel<true>, dim3(blocks), dim3(SCHUR_THREADS), ba->lds_bytes, ctx->stream, n, ba->P, ba->F,
                           ba->pptr, ba->pobs, ba->cam32, ba->Hpp[c], ba->Hcp[c], ba->g[c], lambda, ba->Dpinv, ba->S, ba->rhs, ctl);
    else if (ba->n_strips) {
        const int strips = ba->n_strips;
        int slices = (ba->P + waves - 1) / waves;
        // about one workgroup per CU in all: the kernel runs at the rate of the CUs' float64 LDS atomics (about one
        // lane every two cycles), so fewer workgroups are slower in proportion, whatever their flush saves
        const int want = (256 + strips - 1) / strips;
        if (slices > want) slices = want;
        if (ba->pairs) {
            hipLaunchKernelGGL(ba_point_inverse_kernel, dim3((ba->P + 255) / 256), dim3(256), 0, ctx->stream, ba->P, ba->Hpp[c], lambda,
                               ba->Dpinv, ctl);
            int pslices = (ba->P + 63) / 64;                                   // a workgroup takes 64 landmarks at a time
            if (pslices > want) pslices = want;
            hipLaunchKernelGGL(ba_schur_pairs_kernel, dim3(pslices, strips), dim3(SCHUR_THREADS), ba->strip_lds + 8 * (size_t)n,
                               ctx->stream, n, ba->P, ba->F, ba->cam_first, ba->strip_budget, ba->pptr, ba->pobs, ba->pcam, (const int2 *)ba->bcam, ba->Hcp[c],
                               ba->g[c], (const double *)ba->Dpinv, ba->S, ba->rhs, ctl);
        } else
        hipLaunchKernelGGL(ba_schur_strip_kernel, dim3(slices, strips), dim3(SCHUR_THREADS), ba->strip_lds, ctx->stream, n, ba->P,
                           ba->F, ba->cam_first, ba->strip_budget, ba->pptr, ba->pobs, ba->cam32, ba->Hpp[c], ba->Hcp[c], ba->g[c], lambda, ba->Dpinv,
                           ba->S, ba->rhs, ctl);
    } else
        hipLaunchKernelGGL(ba_schur_kernel<false>, dim3(blocks), dim3(SCHUR_THREADS), 0, ctx->stream, n, ba->P, ba->F, ba->pptr,
                           ba->pobs, ba->cam32, ba->Hpp[c], ba->Hcp[c], ba->g[c], lambda, ba->Dpinv, ba->S, ba->rhs, ctl);
    VO_HIP_CHECK(hipGetLastError());
    if (ba->lds_bytes && ba->solve6_lds)
        hipLaunchKernelGGL(ba_solve6_kernel, dim3(1), dim3(SOLVE_THREADS), ba->solve6_lds, ctx->stream, n, ba->F, (const double *)ba->S,
                           (const double *)ba->rhs, ba->Hcc[c], ba->g[c], lambda, ba->dc, ba->flag, ctl);
    else if (ba->lds_bytes)
        hipLaunchKernelGGL(ba_solve_kernel<true>, dim3(1), dim3(SOLVE_THREADS), ba->lds_bytes, ctx->stream, n, ba->F, ba->S,
                           ba->rhs, ba->Hcc[c], ba->g[c], lambda, ba->dc, ba->flag, ctl);
    else if (ba->panels) {
        const int first = 6 * ba->cam_first, W = PANEL_MAX_W;
        const int work = ba->F * 36 > n ? ba->F * 36 : n;
        hipLaunchKernelGGL(ba_solve_prep_kernel, dim3((work + 255) / 256), dim3(256), 0, ctx->stream, n, ba->F, first, ba->S, ba->rhs,
                           ba->Hcc[c], ba->g[c], lambda, ba->zf, ba->dinv, ba->flag, ctl);
        for (int k0 = first; k0 < n; k0 += W) {
            const int w = n - k0 < W ? n - k0 : W, m = n - k0 - w, nb = (m + 1 + 31) / 32;
            if (w == W && ((n | k0) & 1) == 0)
                hipLaunchKernelGGL(ba_solve_panel_kernel<true>, dim3(nb * (nb + 1) / 2), dim3(64), 0, ctx->stream, n, k0, w, ba->S,
                                   ba->rhs, ba->Lf, ba->zf, ba->dinv, ba->flag, ctl);
            else
                hipLaunchKernelGGL(ba_solve_panel_kernel<false>, dim3(nb * (nb + 1) / 2), dim3(64), 0, ctx->stream, n, k0, w, ba->S,
                                   ba->rhs, ba->Lf, ba->zf, ba->dinv, ba->flag, ctl);
        }
        ba_backsolve_launch(ba, first, ba->Lf, ba->zf, ba->dinv, ba->Hcc[c], ctl);
    } else if (ba->solve_w) {
        hipLaunchKernelGGL(ba_solve_blocked_kernel, dim3(1), dim3(SOLVEB_THREADS), ba->solve_lds, ctx->stream, n, ba->F,
                           6 * ba->cam_first, ba->solve_w, ba->S, ba->rhs, ba->Hcc[c], ba->g[c], lambda, ba->flag, ba->dinv, ctl);
        ba_backsolve_launch(ba, 6 * ba->cam_first, ba->S, ba->rhs, ba->dinv, ba->Hcc[c], ctl);
    }
    else
        hipLaunchKernelGGL(ba_solve_kernel<false>, dim3(1), dim3(SOLVE_THREADS), 0, ctx->stream, n, ba->F, ba->S, ba->rhs,
                           ba->Hcc[c], ba->g[c], lambda, ba->dc, ba->flag, ctl);
    VO_HIP_CHECK(hipGetLastError());
    const size_t work = (size_t)ba->P * BA_LIST_LANES > (size_t)n ? (size_t)ba->P * BA_LIST_LANES : (size_t)n;
    hipLaunchKernelGGL(ba_backsub_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, ctx->stream, n, ba->P, ba->F, ba->pptr,
                       ba->pobs, ba->cam32, ba->Hpp[c], ba->Hcp[c], ba->g[c], ba->Dpinv, ba->dc, ba->x[c], ba->x[o], ctl);
    VO_HIP_CHECK(hipGetLastError());
    return ba_linearize_into(ba, o, ctl);
}

extern "C" int vo_ba_try_step(vo_ctx *ctx, vo_ba *ba, double lambda, double *cost_new, int *solved)
{
    VO_REQUIRE(ctx && ba && cost_new && solved, "vo_ba_try_step: NULL argument");
    VO_REQUIRE(ba->ctx == ctx, "vo_ba_try_step: solver belongs to another context");
    BA_REFUSE_WHILE_SOLVING(ba, "vo_ba_try_step");
    VO_REQUIRE(lambda >= 0, "vo_ba_try_step: lambda=%g", lambda);
    if (!ba->linearized) { vo_set_error("vo_ba_try_step: call vo_ba_linearize first"); return VO_ESTATE; }
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ba_try_step_enqueue(ba, lambda, nullptr);
    if (rc) return rc;
    VO_RESULT(ctx, cost_new, ba->cost[1 - ba->cur], 8);
    VO_RESULT(ctx, solved, ba->flag, 4);
    VO_RESULTS_FINISH(ctx);
    ba->have_candidate = true;
    return VO_OK;
}

// ---- the whole Levenberg-Marquardt loop with the decisions on the device ------------------------------------
// After a trial step one thread compares the candidate's cost with the current point's and updates the record: an
// accepted step lowers the damping to a third and asks for the candidate to be copied over the current point (the two
// slots are contiguous blocks; the copy is a few microseconds at HBM speed and keeps every kernel's pointers fixed), a
// rejected one raises it tenfold; the stop rules are those of the host loop (gain below ftol after an accepted
// step, a rejected step that moved the cost by less than ftol, ten rejected steps in a row, the iteration cap).
__global__ void ba_lm_init_kernel(BaCtl *ctl, const double *__restrict__ cost, double lambda0, double ftol, int max_iters, int flat_reject)
{
    if (threadIdx.x || blockIdx.x) return;
    ctl->lambda = lambda0;
    ctl->cost = ctl->prev = ctl->cost_start = cost[0];
    ctl->ftol = ftol;
    ctl->done = max_iters <= 0;
    ctl->copy_now = 0;
    ctl->tries = ctl->iters = ctl->trials = ctl->accepted = ctl->reason = 0;
    ctl->max_iters = max_iters;
    ctl->flat_reject = flat_reject;
    ctl->pad = 0;
}

__global__ void ba_lm_control_kernel(BaCtl *ctl, const double *__restrict__ cost_new_p, const int *__restrict__ flag)
{
    if (threadIdx.x || blockIdx.x) return;
    ctl->copy_now = 0;
    if (ctl->done) return;
    const double cost_new = cost_new_p[0], cost = ctl->cost;
    const bool usable = flag[0] != 0 && isfinite(cost_new);
    ctl->trials++;
    if (usable && cost_new < cost) {
        ctl->copy_now = 1;
        ctl->accepted++;
        ctl->cost = cost_new;
        ctl->lambda = fmax(ctl->lambda / 3, 1e-9);
        ctl->tries = 0;
        ctl->iters++;
        if ((ctl->prev - cost_new) < ctl->ftol * ctl->prev) { ctl->done = 1; ctl->reason = 1; }
        else if (ctl->iters >= ctl->max_iters) { ctl->done = 1; ctl->reason = 4; }
        ctl->prev = cost_new;
    } else if (ctl->flat_reject && usable && fabs(cost_new - cost) < ctl->ftol * cost) {
        ctl->done = 1;
        ctl->reason = 2;
    } else {
        ctl->lambda *= 10;
        if (++ctl->tries >= 10) { ctl->done = 1; ctl->reason = 3; }
    }
}

__global__ __launch_bounds__(256) void ba_lm_copy_kernel(const BaCtl *__restrict__ ctl, uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    if (!ctl->copy_now) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// One round of queued trial steps behind the loop's record (each followed by the decision and the conditional copy).
static int ba_solve_enqueue_round(vo_ba *ba, int batch)
{
    vo_ctx *ctx = ba->ctx;
    const int c = ba->cur, o = 1 - c;
    const size_t n16 = ba->slot_bytes / 16;
    unsigned copy_blocks = (unsigned)((n16 + 255) / 256);
    if (copy_blocks > 4096) copy_blocks = 4096;
    int rc;
    for (int k = 0; k < batch; ++k) {
        if ((rc = ba_try_step_enqueue(ba, 0.0, ba->ctl))) return rc;
        hipLaunchKernelGGL(ba_lm_control_kernel, dim3(1), dim3(64), 0, ctx->stream, ba->ctl, (const double *)ba->cost[o], (const int *)ba->flag);
        hipLaunchKernelGGL(ba_lm_copy_kernel, dim3(copy_blocks), dim3(256), 0, ctx->stream, (const BaCtl *)ba->ctl, (uint4 *)ba->x[c],
                           (const uint4 *)ba->x[o], n16);
        VO_HIP_CHECK(hipGetLastError());
    }
    return VO_OK;
}

extern "C" int vo_ba_solve_begin(vo_ctx *ctx, vo_ba *ba, const double *x0, int max_iters, double ftol, double lambda0,
                                 int stop_on_flat_reject, int first_round, int mem)
{
    VO_REQUIRE(ctx && ba && x0, "vo_ba_solve_begin: NULL argument");
    VO_REQUIRE(ba->ctx == ctx, "vo_ba_solve_begin: solver belongs to another context");
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_ba_solve_begin: bad mem flag %d", mem);
    VO_REQUIRE(max_iters >= 0 && ftol >= 0 && lambda0 > 0, "vo_ba_solve_begin: max_iters=%d ftol=%g lambda0=%g", max_iters, ftol, lambda0);
    VO_REQUIRE(first_round >= 0 && first_round <= 64, "vo_ba_solve_begin: first_round=%d", first_round);
    if (ba->solve_pending) { vo_set_error("vo_ba_solve_begin: the previous solve of this solver has not been finished"); return VO_ESTATE; }
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t np_ = (size_t)ba->n + 3 * (size_t)ba->P;
    const int c = ba->cur;
    if (mem == VO_MEM_HOST) VO_UPLOAD(ctx, ba->x[c], x0, np_ * 8);
    else VO_HIP_CHECK(hipMemcpyAsync(ba->x[c], x0, np_ * 8, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = ba_linearize_into(ba, c);
    if (rc) return rc;
    ba->linearized = true;
    ba->have_candidate = false;
    hipLaunchKernelGGL(ba_lm_init_kernel, dim3(1), dim3(64), 0, ctx->stream, ba->ctl, (const double *)ba->cost[c], lambda0, ftol,
                       ba->O == 0 ? 0 : max_iters, stop_on_flat_reject);
    VO_HIP_CHECK(hipGetLastError());
    // Trial steps are queued several at a time and the record read back once per round: a window of the frame loop
    // needs four to six of them, five most often.  A surplus step behind the one that ended the loop costs its
    // launches only (every kernel leaves on `done`), but at 30-40 launches that is 80 us.
    ba->solve_cap = ba->O == 0 ? 0 : 10 * (max_iters > 0 ? max_iters : 1);
    // (a caller with other work to do meanwhile asks for a longer first round: steps behind the end of the loop then
    // cost nothing it would see, and a second round would not be hidden)
    if (first_round <= 0) first_round = 5;
    ba->solve_queued = ba->solve_cap < first_round ? ba->solve_cap : first_round;
    if ((rc = ba_solve_enqueue_round(ba, ba->solve_queued))) return rc;
    ba->solve_pending = true;
    return VO_OK;
}

extern "C" int vo_ba_solve_finish(vo_ctx *ctx, vo_ba *ba, double *x_out, int mem, vo_ba_lm_stats *stats)
{
    VO_REQUIRE(ctx && ba && x_out, "vo_ba_solve_finish: NULL argument");
    VO_REQUIRE(ba->ctx == ctx, "vo_ba_solve_finish: solver belongs to another context");
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_ba_solve_finish: bad mem flag %d", mem);
    if (!ba->solve_pending) { vo_set_error("vo_ba_solve_finish: no solve was begun on this solver"); return VO_ESTATE; }
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    ba->solve_pending = false;
    const size_t np_ = (size_t)ba->n + 3 * (size_t)ba->P;
    const int c = ba->cur;
    BaCtl h = {};
    int rc;
    for (;;) {
        VO_RESULT(ctx, &h, ba->ctl, sizeof(BaCtl));
        VO_RESULTS_FINISH(ctx);
        if (h.done || ba->solve_queued >= ba->solve_cap) break;
        int batch = 3;
        if (batch > ba->solve_cap - ba->solve_queued) batch = ba->solve_cap - ba->solve_queued;
        if ((rc = ba_solve_enqueue_round(ba, batch))) return rc;
        ba->solve_queued += batch;
    }
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, x_out, ba->x[c], np_ * 8);
        VO_RESULTS_FINISH(ctx);
    } else {
        VO_HIP_CHECK(hipMemcpyAsync(x_out, ba->x[c], np_ * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (stats) {
        stats->cost_start = h.cost_start; stats->cost_final = h.cost; stats->lambda = h.lambda;
        stats->trial_steps = h.trials; stats->accepted = h.accepted; stats->iterations = h.iters; stats->stop_reason = h.reason;
    }
    return VO_OK;
}

extern "C" int vo_ba_solve(vo_ctx *ctx, vo_ba *ba, const double *x0, int max_iters, double ftol, double lambda0,
                           int stop_on_flat_reject, double *x_out, int mem, vo_ba_lm_stats *stats)
{
    VO_REQUIRE(x_out != nullptr, "vo_ba_solve: NULL argument");
    int rc = vo_ba_solve_begin(ctx, ba, x0, max_iters, ftol, lambda0, stop_on_flat_reject, 0, mem);
    if (rc) return rc;
    return vo_ba_solve_finish(ctx, ba, x_out, mem, stats);
}

extern "C" int vo_ba_accept(vo_ctx *ctx, vo_ba *ba)
{
    VO_REQUIRE(ctx && ba, "vo_ba_accept: NULL argument");
    VO_REQUIRE(ba->ctx == ctx, "vo_ba_accept: solver belongs to another context");
    BA_REFUSE_WHILE_SOLVING(ba, "vo_ba_accept");
    if (!ba->have_candidate) { vo_set_error("vo_ba_accept: no candidate step (call vo_ba_try_step first)"); return VO_ESTATE; }
    ba->cur = 1 - ba->cur;
    ba->have_candidate = false;
    return VO_OK;
}

extern "C" int vo_ba_params(vo_ctx *ctx, vo_ba *ba, int candidate, double *x, int mem)
{
    VO_REQUIRE(ctx && ba && x, "vo_ba_params: NULL argument");
    VO_REQUIRE(ba->ctx == ctx, "vo_ba_params: solver belongs to another context");
    BA_REFUSE_WHILE_SOLVING(ba, "vo_ba_params");
    VO_REQUIRE(mem == VO_MEM_HOST || mem == VO_MEM_DEVICE, "vo_ba_params: bad mem flag %d", mem);
    if (!ba->linearized || (candidate && !ba->have_candidate)) { vo_set_error("vo_ba_params: nothing to return yet"); return VO_ESTATE; }
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t np_ = (size_t)ba->n + 3 * (size_t)ba->P;
    const int slot = candidate ? 1 - ba->cur : ba->cur;
    if (mem == VO_MEM_HOST) {
        VO_RESULT(ctx, x, ba->x[slot], np_ * 8);
        VO_RESULTS_FINISH(ctx);
    } else {
        VO_HIP_CHECK(hipMemcpyAsync(x, ba->x[slot], np_ * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return VO_OK;
}

// Makes the runtime load this file's device code now (vo_ctx_create calls it once per device) instead of at the
// first launch of one of its kernels.
int vo_load_code_ba(void)
{
    hipFuncAttributes attr;
    VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)ba_cam_sum_kernel));
    return VO_OK;
}
