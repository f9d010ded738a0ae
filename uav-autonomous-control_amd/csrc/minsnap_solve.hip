// Minimum-snap time allocation, row counting and the PIVOTED (banded LU) form of the coefficient solve
// (gfx950).  The solver the API uses by default is the lane-per-mission block-Thomas recurrence of
// minsnap_solve_bt.hip; the wave-per-mission kernel below is kept as an independent cross-check
// (uavac_minsnap_solve_banded_dev) and documents the formulation both share.
//
// Replaces uav_ac/planning/minimum_snap.py (upstream paths):
//   _generate_time_per_spline            :311-321   -> row_counts_kernel
//   len(np.arange(0.0, T, dt))           :104       -> row_counts_kernel (ceil(T/dt) in fp64)
//   _create_polynom_matrices / _create_snap_cost_matrix / _compute_spline_parameters
//                                        :138-255   -> minsnap_solve_kernel
//
// The reference solves one dense (14m+2)^2 KKT system per mission in the monomial
// basis (cond ~1e10).  Here the same QP is restated in knot-derivative coordinates:
// unknowns are (v, a, j) at the m-1 interior knots, C1..C3 continuity and the
// position / rest constraints hold by construction, and the only equality left is
// continuity of the 4th derivative at each interior knot.  Per segment everything
// is a fixed 8x8 map scaled by powers of T (tau = t/T):
//     cost_s   = T^-7 * e^T Q1 e ,  e = diag(1,T,T^2,T^3,1,T,T^2,T^3) d
//     snap(0)  = T^-4 * S0 . e ,    snap(T) = T^-4 * S1 . e
//     c_tau    = W e ,              c_t[i]  = c_tau[i] T^-i
// with d = [p,v,a,j]@start (+) [p,v,a,j]@end and Q1, S0, S1, W exact small rationals
// (W = inverse of the README's 8x8 boundary matrix at T=1, Q1 = W^T H1 W).
// The KKT system has order 4(m-1), is block-tridiagonal in knot order (half
// bandwidth 7) and has cond ~4e4; it is factorised by banded LU with partial
// pivoting, one wavefront per mission, entirely in LDS, 3 right-hand sides (x,y,z).
// Unique optimum => identical coefficients to the reference's KKT solve.

#include "uavac_internal.h"

namespace {

constexpr int KL = 7;             // sub-diagonals of the knot-ordered KKT matrix
constexpr int BW = 22;            // stored band per row: columns [i-7, i+14] (fill-in of partial pivoting)
constexpr int RS = 25;            // row stride in doubles: band + 3 right-hand sides (odd: spreads LDS banks)

// Q1 = W^T H1 W: snap cost of a unit-duration septic in endpoint-derivative coordinates.
__constant__ double kQ1[64] = {
    100800, 50400, 10080, 840, -100800, 50400, -10080, 840,
    50400, 25920, 5400, 480, -50400, 24480, -4680, 360,
    10080, 5400, 1200, 120, -10080, 4680, -840, 60,
    840, 480, 120, 16, -840, 360, -60, 4,
    -100800, -50400, -10080, -840, 100800, -50400, 10080, -840,
    50400, 24480, 4680, 360, -50400, 25920, -5400, 480,
    -10080, -4680, -840, -60, 10080, -5400, 1200, -120,
    840, 360, 60, 4, -840, 480, -120, 16};
// 4th derivative at tau=0 / tau=1 as a function of the endpoint derivatives.
__constant__ double kS0[8] = {-840, -480, -120, -16, 840, -360, 60, -4};
__constant__ double kS1[8] = {840, 360, 60, 4, -840, 480, -120, 16};
// W = M1^-1: endpoint derivatives -> ascending monomial coefficients (rows 4..7; rows 0..3 are 1,1,1/2,1/6 diag).
__constant__ double kW[32] = {
    -35, -20, -5, -2.0 / 3.0, 35, -15, 2.5, -1.0 / 6.0,
    84, 45, 10, 1, -84, 39, -7, 0.5,
    -70, -36, -7.5, -2.0 / 3.0, 70, -34, 6.5, -0.5,
    20, 10, 2, 1.0 / 6.0, -20, 10, -2, 1.0 / 6.0};

// ------------------------------------------------------------------------------------------
// Times and row counts: one thread per mission, 256 missions per workgroup, which also leaves the tile's row total.
// T = np.linalg.norm(wp[i+1] - wp[i]) / velocity [* 1.5] (minimum_snap.py:315-320).  For a 3-vector NumPy's norm is
// sqrt(x.dot(x)) and BLAS ddot accumulates with fused multiply-adds: sqrt(fma(dz, dz, fma(dy, dy, dx * dx))) -- the
// committed reference times are reproduced bit for bit by this form only (the plain sum of squares differs in the
// last bit for ~8 % of the segments, which moves ceil(T/dt) by one row whenever T/dt sits on an integer).  No other
// contraction: the division, the factor and ceil(T/dt) are the plain IEEE sequence of len(np.arange(0, T, dt)).
__device__ __forceinline__ int64_t block_inclusive_scan_256(int64_t v, int64_t *wsum /* [4] shared */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    int64_t base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    return v + base;
}

// RAGGED: mission b has m_b = seg_offsets[b + 1] - seg_offsets[b] segments (1 .. m, m = the batch's maximum) and m_b + 1
// waypoints; waypoints, times and row counts of the batch lie back to back (mission b's first waypoint is waypoint
// seg_offsets[b] + b, its first segment is segment seg_offsets[b]).  A count outside 1 .. m raises flag 0 and is clamped.
template <bool RAGGED>
__global__ void __launch_bounds__(256) row_counts_kernel(const double *__restrict__ wp, int B, int m_uniform, double velocity,
                                                         double dt, double *__restrict__ times,
                                                         int32_t *__restrict__ seg_rows, int32_t *__restrict__ totals,
                                                         int64_t *__restrict__ tile_sum, int32_t *__restrict__ flags,
                                                         const int64_t *__restrict__ seg_offsets) {
#pragma clang fp contract(off)
    __shared__ int64_t wsum[4];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = 0;
    if (b < B) {
        int m = m_uniform;
        size_t seg0 = (size_t)b * m_uniform;
        const double *w = wp + (size_t)b * (m_uniform + 1) * 3;
        bool bad = false;
        if (RAGGED) {
            seg0 = (size_t)seg_offsets[b];
            const int64_t mb = seg_offsets[b + 1] - seg_offsets[b];
            bad = mb < 1 || mb > m_uniform;
            m = (int)(mb < 1 ? 1 : (mb > m_uniform ? m_uniform : mb));
            w = wp + (seg0 + (size_t)b) * 3;
        }
        times += seg0;
        seg_rows += seg0;
        double x0 = w[0], y0 = w[1], z0 = w[2];
        bad = bad || !(isfinite(x0) && isfinite(y0) && isfinite(z0));
        // (one wave per SIMD at B = 65 536 and a square root and two divisions per segment, each a chain of dependent Newton
        // steps: four segments side by side fill the gaps)
#pragma unroll 4
        for (int s = 0; s < m; ++s) {
            double x1 = w[3 * s + 3], y1 = w[3 * s + 4], z1 = w[3 * s + 5];
            double dx = x1 - x0, dy = y1 - y0, dz = z1 - z0;
            double T = sqrt(fma(dz, dz, fma(dy, dy, dx * dx))) / velocity;
            if (s == 0 || s == m - 1) T = T * 1.5;      // START_END_TIME_FACTOR, minimum_snap.py:10,318-320
            bad = bad || !isfinite(T);
            double q = ceil(T / dt);
            int rows = (isfinite(q) && q > 0.0 && q < 2.0e9) ? (int)q : 0;
            times[s] = T;
            seg_rows[s] = rows;
            total += rows;
            x0 = x1; y0 = y1; z0 = z1;
        }
        if (total > 2147483647LL) { atomicOr(&flags[3], 1); total = 0; }     // a mission's rows are indexed with int
        totals[b] = (int32_t)total;
        if (bad) atomicOr(&flags[0], 1);
    }
    const int64_t inc = block_inclusive_scan_256(total, wsum);
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = inc;
}

// Exclusive prefix sum of totals[B] -> row_offsets[B+1] (int64): every workgroup adds up the sums of the tiles before
// its own (B / 256 values at most: 256 at B = 65 536) and scans its tile on top of that.  Two launches in all, every
// access coalesced, no inter-workgroup waiting.
__global__ void __launch_bounds__(256) row_offsets_kernel(const int32_t *__restrict__ totals, int B,
                                                          const int64_t *__restrict__ tile_sum,
                                                          int64_t *__restrict__ row_offsets) {
    __shared__ int64_t wsum[4];
    __shared__ int64_t base_s;
    int64_t part = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) part += tile_sum[t];
    const int64_t before = block_inclusive_scan_256(part, wsum);
    if (threadIdx.x == 255) base_s = before;
    __syncthreads();
    const int64_t base = base_s;
    __syncthreads();                               // wsum is reused below
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int64_t v = i < B ? totals[i] : 0;
    const int64_t inc = block_inclusive_scan_256(v, wsum) + base;
    if (i < B) row_offsets[i] = inc - v;
    if (i == B - 1) row_offsets[B] = inc;
}

// Per-mission row totals from per-segment row counts that exist already (a gathered plan): the tail of row_counts_kernel.
template <bool RAGGED>
__global__ void __launch_bounds__(256) seg_totals_kernel(const int32_t *__restrict__ seg_rows, int B, int m_uniform,
                                                         int32_t *__restrict__ totals, int64_t *__restrict__ tile_sum,
                                                         int32_t *__restrict__ flags, const int64_t *__restrict__ seg_offsets) {
    __shared__ int64_t wsum[4];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = 0;
    if (b < B) {
        int m = m_uniform;
        size_t seg0 = (size_t)b * m_uniform;
        if (RAGGED) {
            seg0 = (size_t)seg_offsets[b];
            const int64_t mb = seg_offsets[b + 1] - seg_offsets[b];
            if (mb < 1 || mb > m_uniform) atomicOr(&flags[0], 1);
            m = (int)(mb < 1 ? 1 : (mb > m_uniform ? m_uniform : mb));
        }
        for (int s = 0; s < m; ++s) {
            const int32_t r = seg_rows[seg0 + s];
            total += r > 0 ? r : 0;
        }
        if (total > 2147483647LL) { atomicOr(&flags[3], 1); total = 0; }
        totals[b] = (int32_t)total;
    }
    const int64_t inc = block_inclusive_scan_256(total, wsum);
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = inc;
}

__global__ void __launch_bounds__(256) plan_commit_kernel(const double *__restrict__ times_s, const int32_t *__restrict__ seg_rows_s,
                                                          const int64_t *__restrict__ row_offsets_s, int B, size_t n_seg,
                                                          int64_t capacity_rows, double *__restrict__ times,
                                                          int32_t *__restrict__ seg_rows, int64_t *__restrict__ row_offsets) {
    if (row_offsets_s[B] > capacity_rows) return;          // refused as a whole (the sampler raises flag 2)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_seg; i += stride) {
        times[i] = times_s[i];
        seg_rows[i] = seg_rows_s[i];
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= (size_t)B; i += stride) row_offsets[i] = row_offsets_s[i];
}

// ------------------------------------------------------------------------------------------
// Local (per segment) KKT contribution.  Local index l: 0..3 = (v,a,j,lambda) at the segment's start
// knot, 4..7 = the same at its end knot.  ip[e] = T^-e.
__device__ __forceinline__ double local_entry(int la, int lb, const double *__restrict__ ip,
                                              const double *__restrict__ Q1, const double *__restrict__ S0,
                                              const double *__restrict__ S1) {
    int ca = la & 3, cb = lb & 3;
    if (ca == 3 && cb == 3) return 0.0;
    if (ca == 3 || cb == 3) {
        int ll = (ca == 3) ? la : lb;          // the multiplier
        int ld = (ca == 3) ? lb : la;          // the derivative it couples to
        int d = (ld & 4) + (ld & 3) + 1;       // index into the 8-vector of endpoint derivatives
        int od = (ld & 3) + 1;                 // derivative order
        double v = (ll & 4) ? S1[d] : -S0[d];  // knot constraint: snap_end(prev) - snap_start(next) = 0
        return v * ip[4 - od];
    }
    int a = (la & 4) + ca + 1, b = (lb & 4) + cb + 1;
    return Q1[a * 8 + b] * ip[7 - (ca + 1) - (cb + 1)];
}

__device__ __forceinline__ double local_rhs(int la, double p0, double p1, const double *__restrict__ ip,
                                            const double *__restrict__ Q1, const double *__restrict__ S0,
                                            const double *__restrict__ S1) {
    int ca = la & 3;
    if (ca == 3) {
        if (la & 4) return -(S1[0] * p0 + S1[4] * p1) * ip[4];
        return (S0[0] * p0 + S0[4] * p1) * ip[4];
    }
    int a = (la & 4) + ca + 1;
    return -(Q1[a * 8 + 0] * p0 + Q1[a * 8 + 4] * p1) * ip[7 - (ca + 1)];
}

// One wavefront (= one 64-thread workgroup) per mission.
__global__ void __launch_bounds__(64) minsnap_solve_kernel(const double *__restrict__ wp,
                                                          const double *__restrict__ times, int B, int m,
                                                          double *__restrict__ coeffs, int32_t *__restrict__ status,
                                                          int32_t *__restrict__ flags) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int n = 4 * (m - 1);
    double *band = lds;                       // [n][RS]
    double *ipw = band + (size_t)n * RS;      // [m][8]   T^-e, e = 0..7
    double *ppw = ipw + m * 8;                // [m][4]   T^e,  e = 0..3
    double *wpl = ppw + m * 4;                // [(m+1)*3]
    double *Q1 = wpl + (m + 1) * 3;           // [64]
    double *S0 = Q1 + 64;                     // [8]
    double *S1 = S0 + 8;                      // [8]
    double *Wl = S1 + 8;                      // [32]

    for (int i = lane; i < 64; i += 64) Q1[i] = kQ1[i];
    if (lane < 8) { S0[lane] = kS0[lane]; S1[lane] = kS1[lane]; }
    if (lane < 32) Wl[lane] = kW[lane];
    for (int i = lane; i < (m + 1) * 3; i += 64) wpl[i] = wp[(size_t)b * (m + 1) * 3 + i];
    for (int s = lane; s < m; s += 64) {
        double T = times[(size_t)b * m + s];
        double r = 1.0 / T, acc = 1.0, pacc = 1.0;
        for (int e = 0; e < 8; ++e) { ipw[s * 8 + e] = acc; acc *= r; }
        for (int e = 0; e < 4; ++e) { ppw[s * 4 + e] = pacc; pacc *= T; }
    }
    __syncthreads();

    // ---- assemble the banded KKT matrix and the 3 right-hand sides -----------------------------
    for (int e = lane; e < n * RS; e += 64) {
        int i = e / RS, off = e - i * RS;
        int ki = (i >> 2) + 1, ci = i & 3;     // knot 1..m-1, component
        double v = 0.0;
        if (off < 15) {
            int j = i + off - KL;
            if (j >= 0 && j < n) {
                int kj = (j >> 2) + 1, cj = j & 3;
                int dk = kj - ki;
                // segment ki-1 (this knot is its end) and segment ki (this knot is its start)
                if (dk == 0) {
                    v = local_entry(4 + ci, 4 + cj, ipw + (ki - 1) * 8, Q1, S0, S1) +
                        local_entry(ci, cj, ipw + ki * 8, Q1, S0, S1);
                } else if (dk == -1) {
                    v = local_entry(4 + ci, cj, ipw + (ki - 1) * 8, Q1, S0, S1);
                } else if (dk == 1) {
                    v = local_entry(ci, 4 + cj, ipw + ki * 8, Q1, S0, S1);
                }
            }
        } else if (off >= BW) {
            int r = off - BW;
            v = local_rhs(4 + ci, wpl[(ki - 1) * 3 + r], wpl[ki * 3 + r], ipw + (ki - 1) * 8, Q1, S0, S1) +
                local_rhs(ci, wpl[ki * 3 + r], wpl[(ki + 1) * 3 + r], ipw + ki * 8, Q1, S0, S1);
        }
        band[e] = v;
    }
    __syncthreads();

    // ---- banded LU with partial pivoting, right-hand sides carried along -----------------------
    bool singular = false;
    for (int k = 0; k < n; ++k) {
        // pivot search over rows k..k+7 of column k (lanes 0..7)
        int row = k + (lane & 7);
        double av = (row < n) ? fabs(band[row * RS + (k - row + KL)]) : -1.0;
        int pr = row;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            double ov = __shfl_xor(av, d);
            int orow = __shfl_xor(pr, d);
            if (ov > av || (ov == av && orow < pr)) { av = ov; pr = orow; }
        }
        av = __shfl(av, 0);
        pr = __shfl(pr, 0);
        if (!(av > 0.0) || !isfinite(av)) { singular = true; break; }
        // row exchange on columns k..k+14 and the right-hand sides (lanes 0..17)
        if (pr != k && lane < 18) {
            int ok, op;
            if (lane < 15) { ok = lane + KL; op = k + lane - pr + KL; }
            else { ok = BW + lane - 15; op = ok; }
            bool live = (lane >= 15) || (k + lane < n);
            if (live) {
                double a = band[k * RS + ok], c = band[pr * RS + op];
                band[k * RS + ok] = c;
                band[pr * RS + op] = a;
            }
        }
        __syncthreads();
        double rinv = 1.0 / band[k * RS + KL];
        // rank-1 update of rows k+1..k+7, columns k+1..k+14 and the right-hand sides: 7 x 17 pairs
        for (int e = lane; e < 7 * 17; e += 64) {
            int r = e / 17 + 1, c = e - (r - 1) * 17;      // c: 0..13 -> column k+1+c ; 14..16 -> rhs
            int ri = k + r;
            if (ri < n) {
                double mult = band[ri * RS + (KL - r)] * rinv;
                int offk, offr;
                bool live;
                if (c < 14) { offk = KL + 1 + c; offr = KL + 1 + c - r; live = (k + 1 + c) < n; }
                else { offk = BW + c - 14; offr = offk; live = true; }
                if (live) band[ri * RS + offr] -= mult * band[k * RS + offk];
            }
        }
        __syncthreads();
    }

    if (singular) {
        if (lane == 0) {
            if (status) status[b] = 1;
            atomicOr(&flags[1], 1);
        }
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int f = lane; f < 24 * m; f += 64) coeffs[(size_t)b * 24 * m + f] = qnan;
        return;
    }

    // ---- back substitution, column oriented: x_i, then rows i-14..i-1 shed their U[.,i] x_i ----
    for (int i = n - 1; i >= 0; --i) {
        double dinv = 1.0 / band[i * RS + KL];
        if (lane < 3) band[i * RS + BW + lane] *= dinv;
        __syncthreads();
        if (lane < 42) {
            int rr = lane / 3 + 1, r = lane - (rr - 1) * 3;
            int ri = i - rr;
            if (ri >= 0) band[ri * RS + BW + r] -= band[ri * RS + (KL + rr)] * band[i * RS + BW + r];
        }
        __syncthreads();
    }
    if (lane == 0 && status) status[b] = 0;

    // ---- endpoint derivatives -> monomial coefficients, ascending powers, layout [8m][3] --------
    for (int f = lane; f < 24 * m; f += 64) {
        int s = f / 24, rem = f - s * 24;
        int i = rem / 3, r = rem - i * 3;
        const double *pp = ppw + s * 4;
        double d[8];
        d[0] = wpl[s * 3 + r];
        d[4] = wpl[(s + 1) * 3 + r];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d[1 + c] = (s >= 1) ? band[(4 * (s - 1) + c) * RS + BW + r] * pp[c + 1] : 0.0;
            d[5 + c] = (s + 1 <= m - 1) ? band[(4 * s + c) * RS + BW + r] * pp[c + 1] : 0.0;
        }
        double ct;
        if (i == 0) ct = d[0];
        else if (i == 1) ct = d[1];
        else if (i == 2) ct = 0.5 * d[2];
        else if (i == 3) ct = d[3] * (1.0 / 6.0);
        else {
            const double *w = Wl + (i - 4) * 8;
            ct = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) ct += w[q] * d[q];
        }
        coeffs[(size_t)b * 24 * m + f] = ct * ipw[s * 8 + i];
    }
}

}  // namespace

// totals [B] i32 followed by the tile sums [ceil(B/256)] i64 (8-byte aligned offset): scratch of the row-offset scan
static int ensure_totals(uavac_ctx *ctx, int B, int64_t **tiles) {
    const int n_tiles = (B + 255) / 256;
    if ((size_t)B > ctx->totals_cap) {
        if (ctx->d_totals) UAVAC_HIP(ctx, hipFree(ctx->d_totals));
        ctx->d_totals = nullptr;
        ctx->totals_cap = 0;
        const size_t bytes = (((size_t)B * 4 + 7) & ~(size_t)7) + (size_t)n_tiles * 8;
        UAVAC_HIP(ctx, hipMalloc(&ctx->d_totals, bytes));
        ctx->totals_cap = (size_t)B;
    }
    *tiles = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(ctx->d_totals) +
                                         (((size_t)ctx->totals_cap * 4 + 7) & ~(size_t)7));
    return UAVAC_OK;
}

int uavac_ensure_totals(uavac_ctx *ctx, int B, int32_t **totals, int64_t **tile_sums) {
    if (int rc = ensure_totals(ctx, B, tile_sums)) return rc;
    *totals = ctx->d_totals;
    return UAVAC_OK;
}

int uavac_launch_totals_scan(uavac_ctx *ctx, int B, int64_t *out) {
    const int n_tiles = (B + 255) / 256;
    int64_t *tiles = nullptr;
    if (int rc = ensure_totals(ctx, B, &tiles)) return rc;
    hipLaunchKernelGGL(row_offsets_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, ctx->d_totals, B, tiles, out);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_launch_row_counts(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt,
                            double *times, int32_t *seg_rows, int64_t *row_offsets, const int64_t *seg_offsets) {
    const int n_tiles = (B + 255) / 256;
    int64_t *tiles = nullptr;
    if (int rc = ensure_totals(ctx, B, &tiles)) return rc;
    if (seg_offsets)
        hipLaunchKernelGGL(row_counts_kernel<true>, dim3(n_tiles), dim3(256), 0, ctx->stream, wp, B, m, velocity, dt,
                           times, seg_rows, ctx->d_totals, tiles, ctx->d_flags, seg_offsets);
    else
        hipLaunchKernelGGL(row_counts_kernel<false>, dim3(n_tiles), dim3(256), 0, ctx->stream, wp, B, m, velocity, dt,
                           times, seg_rows, ctx->d_totals, tiles, ctx->d_flags, seg_offsets);
    hipLaunchKernelGGL(row_offsets_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, ctx->d_totals, B, tiles,
                       row_offsets);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

// row_offsets of a plan whose per-segment row counts exist already (they arrived from another rank,
// uavac_gather_plan_dev): the second half of uavac_launch_row_counts on its own.
int uavac_launch_row_offsets(uavac_ctx *ctx, const int32_t *seg_rows, int B, int m, int64_t *row_offsets,
                             const int64_t *seg_offsets) {
    const int n_tiles = (B + 255) / 256;
    int64_t *tiles = nullptr;
    if (int rc = ensure_totals(ctx, B, &tiles)) return rc;
    if (seg_offsets)
        hipLaunchKernelGGL(seg_totals_kernel<true>, dim3(n_tiles), dim3(256), 0, ctx->stream, seg_rows, B, m, ctx->d_totals,
                           tiles, ctx->d_flags, seg_offsets);
    else
        hipLaunchKernelGGL(seg_totals_kernel<false>, dim3(n_tiles), dim3(256), 0, ctx->stream, seg_rows, B, m, ctx->d_totals,
                           tiles, ctx->d_flags, seg_offsets);
    hipLaunchKernelGGL(row_offsets_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, ctx->d_totals, B, tiles,
                       row_offsets);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

// uavac_minsnap_plan_dev computes times / row counts / offsets into ctx scratch first; this kernel moves them into the
// caller's arrays only when the plan fits the caller's row buffer -- a refused plan leaves every output as it was.
int uavac_launch_plan_commit(uavac_ctx *ctx, const double *times_s, const int32_t *seg_rows_s, const int64_t *row_offsets_s,
                             int B, int m, int64_t capacity_rows, double *times, int32_t *seg_rows, int64_t *row_offsets) {
    const size_t n = (size_t)B * m;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(plan_commit_kernel, dim3(grid), dim3(256), 0, ctx->stream, times_s, seg_rows_s, row_offsets_s, B, n,
                       capacity_rows, times, seg_rows, row_offsets);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_launch_solve(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                       int32_t *status) {
    int n = 4 * (m - 1);
    size_t lds = sizeof(double) * ((size_t)n * RS + (size_t)m * 12 + (size_t)(m + 1) * 3 + 64 + 8 + 8 + 32);
    hipLaunchKernelGGL(minsnap_solve_kernel, dim3(B), dim3(64), lds, ctx->stream, wp, times, B, m, coeffs, status,
                       ctx->d_flags);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
