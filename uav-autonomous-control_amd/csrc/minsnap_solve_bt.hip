// Minimum-snap coefficient solve, thread-per-mission block-Thomas form (gfx950) -- ONE-ENDED: the solver of rounds 1-4, since round 5
// the cross-check of the two-ended form in minsnap_solve_tw.hip (ctx option "solve_order" = 0 selects this file; same system, other
// elimination order, other rounding).
//
// Same QP and same knot-derivative coordinates as minsnap_solve.hip (see its header for the derivation
// and for the reference lines it replaces: uav_ac/planning/minimum_snap.py:138-255).  Ordered by knot,
// the KKT matrix of order 4(m-1) is block tridiagonal with 4x4 blocks, unknowns (v, a, j, lambda) per
// interior knot:
//     D_k = C_{k-1} + A_k ,  U_k = B_k ,  L_k = B_{k-1}^T
// where segment s contributes the symmetric 8x8 local block [[A_s, B_s], [B_s^T, C_s]] (start knot /
// end knot), every entry a fixed small integer times a power of T_s.  The wave-per-mission banded LU of
// minsnap_solve.hip spends ~16 k wave instructions per mission on it (pivot search, row swaps and
// workgroup syncs on a matrix that is 85 % structural zeros).  Here one LANE owns a mission and runs the
// block-Thomas recurrence entirely in registers:
//     S_k = D_k - B_{k-1}^T Ut_{k-1} ;  [Ut_k | rt_k] = S_k^{-1} [B_k | r_k - B_{k-1}^T rt_{k-1}]
//     x_k = rt_k - Ut_k x_{k+1}
// about 6 k scalar fp64 operations per mission and no LDS traffic, no syncs.  Each S_k is a saddle block
// [[G, c], [c^T, -e]] with G positive definite (the Hessian of the cost-to-go in the knot's derivatives)
// and e >= 0, so natural-order elimination needs no pivoting; measured against the dense pivoted solve of
// the reference formulation the sampled trajectories agree to 3e-12 on the SURVEY 8(d) distribution and
// 3e-11 on segment lengths U(1,6) m (tests/test_gpu_planner.py).  A zero or non-finite pivot (repeated
// waypoints) flags the mission exactly like the banded kernel.
//
// HBM: the forward sweep parks [Ut_k | rt_k] (28 doubles per knot) in a [m-1][28][B] workspace (coalesced
// across lanes) for the backward sweep; coefficients leave through a 64 x 24 LDS transpose so that the
// reference's per-mission (8m, 3) layout is written in 192-byte runs.  Loads are issued a knot / segment ahead of
// their use and ahead of that step's stores (see the kernel).

#include "uavac_internal.h"

#pragma clang fp contract(off)

namespace {

// Q1 = W^T H1 W, S0/S1 = end snaps, W rows 4..7: see minsnap_solve.hip
constexpr double Q1c[8][8] = {
    {100800, 50400, 10080, 840, -100800, 50400, -10080, 840},
    {50400, 25920, 5400, 480, -50400, 24480, -4680, 360},
    {10080, 5400, 1200, 120, -10080, 4680, -840, 60},
    {840, 480, 120, 16, -840, 360, -60, 4},
    {-100800, -50400, -10080, -840, 100800, -50400, 10080, -840},
    {50400, 24480, 4680, 360, -50400, 25920, -5400, 480},
    {-10080, -4680, -840, -60, 10080, -5400, 1200, -120},
    {840, 360, 60, 4, -840, 480, -120, 16}};
constexpr double S0c[8] = {-840, -480, -120, -16, 840, -360, 60, -4};
constexpr double S1c[8] = {840, 360, 60, 4, -840, 480, -120, 16};
constexpr double Wc[4][8] = {
    {-35, -20, -5, -2.0 / 3.0, 35, -15, 2.5, -1.0 / 6.0},
    {84, 45, 10, 1, -84, 39, -7, 0.5},
    {-70, -36, -7.5, -2.0 / 3.0, 70, -34, 6.5, -0.5},
    {20, 10, 2, 1.0 / 6.0, -20, 10, -2, 1.0 / 6.0}};

// Local 8x8 KKT entry (la, lb) of a segment as coefficient * T^-e.  Local index: 0..3 = (v, a, j, lambda)
// at the start knot, 4..7 at the end knot.  Both functions fold to literals once la, lb are unrolled.
__device__ __forceinline__ constexpr double loc_coef(int la, int lb) {
    const int ca = la & 3, cb = lb & 3;
    if (ca == 3 && cb == 3) return 0.0;
    if (ca == 3 || cb == 3) {
        const int ll = (ca == 3) ? la : lb, ld = (ca == 3) ? lb : la;
        const int d = (ld & 4) + (ld & 3) + 1;
        return (ll & 4) ? S1c[d] : -S0c[d];        // knot constraint: snap_end(prev) - snap_start(next) = 0
    }
    return Q1c[(la & 4) + ca + 1][(lb & 4) + cb + 1];
}
__device__ __forceinline__ constexpr int loc_exp(int la, int lb) {
    const int ca = la & 3, cb = lb & 3;
    if (ca == 3 && cb == 3) return 0;
    if (ca == 3) return 4 - (cb + 1);
    if (cb == 3) return 4 - (ca + 1);
    return 7 - (ca + 1) - (cb + 1);
}
// right-hand side of local row la: coefficient of p_start / p_end, times T^-e
__device__ __forceinline__ constexpr double rhs_c0(int la) {
    const int ca = la & 3;
    if (ca == 3) return (la & 4) ? -S1c[0] : S0c[0];
    return -Q1c[(la & 4) + ca + 1][0];
}
__device__ __forceinline__ constexpr double rhs_c1(int la) {
    const int ca = la & 3;
    if (ca == 3) return (la & 4) ? -S1c[4] : S0c[4];
    return -Q1c[(la & 4) + ca + 1][4];
}
__device__ __forceinline__ constexpr int rhs_exp(int la) { return ((la & 3) == 3) ? 4 : 7 - ((la & 3) + 1); }

struct Seg {
    double A[4][4], B[4][4], C[4][4];     // start-start, start-end, end-end blocks
    double rs[4][3], re[4][3];            // right-hand side rows of the start / end knot, per axis
    double ip[8];                         // T^-e
};

__device__ __forceinline__ void build_segment(Seg &g, double T, const double p0[3], const double p1[3]) {
    const double r = 1.0 / T;
    g.ip[0] = 1.0;
#pragma unroll
    for (int e = 1; e < 8; ++e) g.ip[e] = g.ip[e - 1] * r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g.A[i][j] = loc_coef(i, j) * g.ip[loc_exp(i, j)];
            g.B[i][j] = loc_coef(i, 4 + j) * g.ip[loc_exp(i, 4 + j)];
            g.C[i][j] = loc_coef(4 + i, 4 + j) * g.ip[loc_exp(4 + i, 4 + j)];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g.rs[i][a] = (rhs_c0(i) * p0[a] + rhs_c1(i) * p1[a]) * g.ip[rhs_exp(i)];
            g.re[i][a] = (rhs_c0(4 + i) * p0[a] + rhs_c1(4 + i) * p1[a]) * g.ip[rhs_exp(4 + i)];
        }
}

// Solve S X = R (4x4, 7 right-hand sides) in natural order; returns false on a zero / non-finite pivot.
__device__ __forceinline__ bool solve4(double S[4][4], double R[4][7]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double piv = S[j][j];
        ok = ok && (fabs(piv) > 0.0) && isfinite(piv);
        const double inv = 1.0 / piv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            const double l = S[i][j] * inv;
#pragma unroll
            for (int c = j + 1; c < 4; ++c) S[i][c] = fma(-l, S[j][c], S[i][c]);
#pragma unroll
            for (int c = 0; c < 7; ++c) R[i][c] = fma(-l, R[j][c], R[i][c]);
        }
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const double inv = 1.0 / S[i][i];
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            double s = R[i][c];
#pragma unroll
            for (int q = i + 1; q < 4; ++q) s = fma(-S[i][q], R[q][c], s);
            R[i][c] = s * inv;
        }
    }
    return ok;
}

// 24 monomial coefficients (ascending powers, [8][3]) of one segment from its knot data
__device__ __forceinline__ void segment_coeffs(const double ip[8], double T, const double p0[3], const double p1[3],
                                               const double x0[3][3], const double x1[3][3], double out[8][3]) {
    const double T2 = T * T, T3 = T2 * T;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // e = diag(1, T, T^2, T^3, 1, T, T^2, T^3) [p v a j]_start (+) [p v a j]_end
        const double e[8] = {p0[a], T * x0[0][a], T2 * x0[1][a], T3 * x0[2][a],
                             p1[a], T * x1[0][a], T2 * x1[1][a], T3 * x1[2][a]};
        out[0][a] = p0[a];
        out[1][a] = x0[0][a];
        out[2][a] = 0.5 * x0[1][a];
        out[3][a] = x0[2][a] * (1.0 / 6.0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) s = fma(Wc[i][q], e[q], s);
            out[4 + i][a] = s * ip[4 + i];
        }
    }
}

constexpr int TB = 64;          // lanes (missions) per workgroup = one wave

// Memory discipline (gfx9 counts loads AND stores in vmcnt, and a wait cannot tell them apart): a wave that has stores in
// flight pays their whole write latency at its next wait for a load.  So every load is issued a full knot / segment before
// its use and BEFORE the stores of the step it is issued in: when the next step waits for it, the youngest stores in
// flight are a step old.  Forward sweep, knot k: issue the loads of knot k + 1's inputs, compute, park 28 doubles.
// Backward sweep, segment s: issue the coefficient stores of segment s + 1 (from the LDS stage) and the loads of knot
// s - 2 and of segment s - 1's inputs, then compute segment s from what was loaded a segment ago.  (As first written --
// inputs loaded where they were used, the parked block read in the compiler's order, four at a time between the FMAs --
// the kernel took 160 us at B = 65 536, m = 12 for ~30 us of arithmetic.)  No workgroup barrier anywhere: one wave,
// whose LDS operations execute in order (`lds_wave_fence`; `__syncthreads` would wait for the global stores too).
// RAGGED: mission b has m_b = seg_offsets[b + 1] - seg_offsets[b] segments (clamped to 1 .. m_uniform = the batch's maximum);
// waypoints, times and coefficients of the batch lie back to back.  Lanes then run different numbers of knots; the backward
// sweep counts segments from each mission's own end (lane l handles segment m_l - 1 - i in step i), so that the 64 x 24
// transpose still moves one segment of every mission that has one left.
// NREG: [Ut | rt] of the first NREG knots stay in registers (uniform batches; see `kept`).
// PARK_LDS: the forward sweep parks [Ut | rt] in the wave's own LDS ([m - 1][28][64] doubles, dynamic) instead of the HBM
// workspace -- when it fits (m <= 8 at 100 KB per wave) and the batch is small enough for that to pay (see the launcher).
template <bool RAGGED, bool PARK_LDS = false, int NACT = TB, int NREG = 0>
__global__ void __launch_bounds__(TB) minsnap_solve_bt_kernel(const double *__restrict__ wp,
                                                             const double *__restrict__ times, int B, int m_uniform,
                                                             double *__restrict__ ws, double *__restrict__ coeffs,
                                                             int32_t *__restrict__ status,
                                                             int32_t *__restrict__ flags,
                                                             const int64_t *__restrict__ seg_offsets,
                                                             const int64_t *__restrict__ guard_rows, int64_t guard_capacity,
                                                             const int32_t *__restrict__ active) {
    // part of a planning chain whose rows would not fit the caller's buffer (uavac_minsnap_plan_dev): the plan is refused as
    // a whole and the coefficients stay what they were (uniform over the launch; one scalar load)
    if (guard_rows && *guard_rows > guard_capacity) return;
    // obstacle loop (uavac_minsnap_obstacle_round_dev): only the missions still being corrected are re-solved -- a wave none
    // of whose missions is active leaves at once (the others solve all 64: their coefficients are simply the same as before)
    if (active) {
        const int b_ = blockIdx.x * NACT + threadIdx.x;
        if (!__any((int)threadIdx.x < NACT && b_ < B && active[b_] != 0)) return;
    }
    __shared__ double stage[TB * 25];                 // one segment's 24 coefficients per mission (+1 pad)
    extern __shared__ double park_lds[];              // PARK_LDS: [m_uniform - 1][28][NACT]
    __shared__ int64_t seg0_of[RAGGED ? TB : 1];      // ragged: first segment and segment count of every mission of the wave
    __shared__ int m_of[RAGGED ? TB : 1];
    // NACT < 64: only the first NACT lanes carry a mission -- twice (four times) the waves for the same batch.  The kernel is
    // bound by the latency of its dependent chains at one wave per SIMD (B = 65 536: 1 024 waves), not by issue: two half-full
    // waves per SIMD hide each other's latencies (the launcher picks; option "solve_lanes")
    const int lane = threadIdx.x;
    const int b0 = blockIdx.x * NACT;
    const int b = b0 + lane;
    const bool live = lane < NACT && b < B;
    const int bb = live ? b : B - 1;
    const size_t sB = (size_t)B;
    // where knot k's parked block lives and how far apart its 28 values are
    const size_t pst = PARK_LDS ? (size_t)NACT : sB;
    // (NREG > 0: knot NREG -- the first one past the registers -- goes to a [28][64] slab of the wave's LDS: launched with 14 KB)
    auto park_stride = [&](int k_) -> size_t { return (NREG > 0 && k_ == NREG) ? (size_t)TB : pst; };
    auto park_at = [&](int k_) -> double * {
        if (NREG > 0 && k_ == NREG) return park_lds + threadIdx.x;
        return PARK_LDS ? park_lds + (size_t)k_ * 28 * NACT + (threadIdx.x < NACT ? threadIdx.x : 0) : ws + ((size_t)k_ * 28) * sB + bb;
    };
    int m = m_uniform;
    const double *w = wp + (size_t)bb * (m_uniform + 1) * 3;
    const double *tm = times + (size_t)bb * m_uniform;
    int m_top = m_uniform;                             // steps of the backward sweep: the longest mission of the wave
    if (RAGGED) {
        const int64_t s0 = seg_offsets[bb], mb = seg_offsets[bb + 1] - s0;
        m = (int)(mb < 1 ? 1 : (mb > m_uniform ? m_uniform : mb));
        w = wp + ((size_t)s0 + (size_t)bb) * 3;
        tm = times + (size_t)s0;
        seg0_of[lane] = s0;
        m_of[lane] = m;
        m_top = live ? m : 1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m_top = max(m_top, __shfl_xor(m_top, d));
        lds_wave_fence();
    }
    const int nk = m - 1;
    bool ok = true;
    // NREG > 0 (uniform batches): [Ut | rt] of the first NREG knots stay in the lane's registers (the compiler places them in
    // the accumulation half of the register file: one wave per SIMD has 512 registers per lane) instead of the HBM workspace
    double kept[NREG > 0 ? NREG : 1][28];

    // ------------------------------------------------------------------ forward sweep over interior knots
    // ([Ut | rt] of a lane's LAST knot is what its backward sweep starts from: it stays in these registers, never parked)
    double Ut[4][4], rt[4][3];
    {
        Seg prev, cur;
        double p0[3] = {w[0], w[1], w[2]}, p1[3] = {w[3], w[4], w[5]};
        build_segment(prev, tm[0], p0, p1);
        // inputs of the knot after this one (clamped reads past the end are never used)
        double nw[3] = {w[6 <= 3 * m ? 6 : 3], w[6 <= 3 * m ? 7 : 4], w[6 <= 3 * m ? 8 : 5]}, nt = tm[m > 1 ? 1 : 0];
        for (int kk = 0; kk < nk; ++kk) {
            const int k = kk + 1;                      // knot k joins segments k-1 (prev) and k (cur)
            const double T = nt;
#pragma unroll
            for (int a = 0; a < 3; ++a) { p0[a] = p1[a]; p1[a] = nw[a]; }
            {
                const int kn = (k + 2 <= m) ? k + 2 : m, tn = (k + 1 <= m - 1) ? k + 1 : m - 1;
#pragma unroll
                for (int a = 0; a < 3; ++a) nw[a] = w[3 * kn + a];
                nt = tm[tn];
            }
            build_segment(cur, T, p0, p1);
            double S[4][4], R[4][7];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    S[i][j] = prev.C[i][j] + cur.A[i][j];
                    R[i][j] = cur.B[i][j];
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) R[i][4 + a] = prev.re[i][a] + cur.rs[i][a];
            }
            if (kk > 0) {
                // subtract B_{k-1}^T [Ut_{k-1} | rt_{k-1}]   (B_{k-1} = prev.B couples knot k-1 to knot k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double s = S[i][j];
#pragma unroll
                        for (int l = 0; l < 4; ++l) s = fma(-prev.B[l][i], Ut[l][j], s);
                        S[i][j] = s;
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double s = R[i][4 + a];
#pragma unroll
                        for (int l = 0; l < 4; ++l) s = fma(-prev.B[l][i], rt[l][a], s);
                        R[i][4 + a] = s;
                    }
                }
            }
            ok = solve4(S, R) && ok;
            double *o = park_at(kk);
            const size_t ost = park_stride(kk);
            const bool in_regs = (NREG > 0 && kk < NREG) || kk == nk - 1;      // (kk < NREG is uniform: every lane of a uniform batch is at the same knot)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { Ut[i][j] = R[i][j]; if (!in_regs && (PARK_LDS ? lane < NACT : live)) o[(size_t)(i * 4 + j) * ost] = R[i][j]; }
#pragma unroll
                for (int a = 0; a < 3; ++a) { rt[i][a] = R[i][4 + a]; if (!in_regs && (PARK_LDS ? lane < NACT : live)) o[(size_t)(16 + i * 3 + a) * ost] = R[i][4 + a]; }
            }
            if (NREG > 0 && kk < NREG) {
#pragma unroll
                for (int q = 0; q < NREG; ++q)
                    if (kk == q) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) kept[q][i * 4 + j] = R[i][j];
#pragma unroll
                            for (int a = 0; a < 3; ++a) kept[q][16 + i * 3 + a] = R[i][4 + a];
                        }
                    }
            }
            prev = cur;
        }
    }
    if (live) {
        if (!ok) atomicOr(&flags[1], 1);
        if (status) status[b] = ok ? 0 : 1;
    }

    // ------------------------------------------------ backward sweep + coefficients, last segment first
    double xn[4][3];                                    // unknowns of knot s+1 (zero at the goal)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) xn[i][a] = 0.0;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    // 64 missions x 24 doubles of one step leave the stage as 192-byte runs: mission q's segment sq = m_q - 1 - step, at
    // coeffs[(first segment of q + sq) * 24]
    auto flush = [&](int step) {
        for (int e = lane; e < NACT * 24; e += TB) {
            const int q = e / 24, j = e - q * 24;
            const int sq = (RAGGED ? m_of[q] : m_uniform) - 1 - step;
            const size_t first = RAGGED ? (size_t)seg0_of[q] : (size_t)(b0 + q) * m_uniform;
            if (b0 + q < B && sq >= 0) coeffs[(first + (size_t)sq) * 24 + j] = stage[q * 25 + j];
        }
    };
    // on their way while the segment before is computed: [Ut | rt] of knot s - 1, start waypoint and duration of segment s
    double nxt[28], nw[3], nt;
    {
        if (nk >= 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[i * 4 + j] = Ut[i][j];
#pragma unroll
                for (int a = 0; a < 3; ++a) nxt[16 + i * 3 + a] = rt[i][a];
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) nw[a] = w[3 * (m - 1) + a];
        nt = tm[m - 1];
    }
    double p1[3] = {w[3 * m], w[3 * m + 1], w[3 * m + 2]};
    for (int step = 0; step < m_top; ++step) {
        const int s = m - 1 - step;                     // this lane's segment; < 0: its mission is finished
        double cur[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) cur[i] = nxt[i];   // (the first use waits for the loads -- and for stores issued a segment ago)
        const double T = nt;
        const double p0[3] = {nw[0], nw[1], nw[2]};
        if (step > 0) flush(step - 1);                  // reads the stage before this step overwrites it (LDS is in order)
        if (s >= 2) {
            if (NREG > 0 && s - 2 < NREG) {
#pragma unroll
                for (int q = 0; q < NREG; ++q)
                    if (s - 2 == q) {
#pragma unroll
                        for (int i = 0; i < 28; ++i) nxt[i] = kept[q][i];
                    }
            } else {
                const double *o = park_at(s - 2);
                const size_t ost = park_stride(s - 2);
#pragma unroll
                for (int i = 0; i < 28; ++i) nxt[i] = o[(size_t)i * ost];
            }
        }
        if (s >= 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) nw[a] = w[3 * (s - 1) + a];
            nt = tm[s - 1];
        }
        if (!RAGGED || s >= 0) {
            double xs[4][3];                            // unknowns of knot s (zero at the start)
            if (s >= 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double v = cur[16 + i * 3 + a];
                        if (s <= nk - 1) {              // knot s has a successor among the unknowns
#pragma unroll
                            for (int l = 0; l < 4; ++l) v = fma(-cur[i * 4 + l], xn[l][a], v);
                        }
                        xs[i][a] = v;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int a = 0; a < 3; ++a) xs[i][a] = 0.0;
            }
            double ip[8];
            const double r = 1.0 / T;
            ip[0] = 1.0;
#pragma unroll
            for (int e = 1; e < 8; ++e) ip[e] = ip[e - 1] * r;
            const double x0[3][3] = {{xs[0][0], xs[0][1], xs[0][2]}, {xs[1][0], xs[1][1], xs[1][2]}, {xs[2][0], xs[2][1], xs[2][2]}};
            const double x1[3][3] = {{xn[0][0], xn[0][1], xn[0][2]}, {xn[1][0], xn[1][1], xn[1][2]}, {xn[2][0], xn[2][1], xn[2][2]}};
            double c[8][3];
            segment_coeffs(ip, T, p0, p1, x0, x1, c);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int a = 0; a < 3; ++a) stage[lane * 25 + i * 3 + a] = ok ? c[i][a] : qnan;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < 3; ++a) xn[i][a] = xs[i][a];
#pragma unroll
            for (int a = 0; a < 3; ++a) p1[a] = p0[a];
        }
        lds_wave_fence();
    }
    flush(m_top - 1);
}

}  // namespace

int uavac_launch_solve_bt(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                          int32_t *status, const int64_t *seg_offsets, const int64_t *guard_rows, int64_t guard_capacity,
                          const int32_t *active) {
    // WHICH ELIMINATION ORDER.  1 (default): two-ended, whatever the launch -- a mission's coefficients must not depend on how many
    // other missions share its batch (a rank's shard of a job equals the job's own bits; ragged == uniform; a mission alone == in a
    // batch: all tested bit for bit), and the two orders round differently (5e-14 relative).  0: one-ended.  -1 (opt-in, round 6):
    // by the launch -- one-ended where it is the faster kernel, uniform batches of short missions from three quarters of a chip's
    // worth of lanes on (m <= 8, B >= 48 * SIMDs: 52 against 56-59 us at 65 536 missions of 8 segments, 195 against 213 at 262 144;
    // below that, and for longer missions at any size, the two-ended form wins: profiles/r05_solve_order_time.jsonl) -- for callers
    // who take the last-bits dependence on the batch size for those 4-18 us.  Both orders sit equally close to the dense pivoted
    // solve of the reference (<= 1e-9 on the coefficients against the `solve` goldens, tests/test_gpu_round6.py).
    const bool two_ended = ctx->solve_order == 1 || (ctx->solve_order < 0 && !(m <= 8 && !seg_offsets && (int64_t)B >= (int64_t)48 * ctx->n_simds));
    if (two_ended) return uavac_launch_solve_tw(ctx, wp, times, B, m, coeffs, status, seg_offsets, guard_rows, guard_capacity, active);
    const size_t need = (size_t)(m > 1 ? m - 1 : 1) * 28 * (size_t)B;
    if (need > ctx->ws_cap) {
        if (ctx->d_ws) UAVAC_HIP(ctx, hipFree(ctx->d_ws));
        ctx->d_ws = nullptr;
        ctx->ws_cap = 0;
        UAVAC_HIP(ctx, hipMalloc(&ctx->d_ws, sizeof(double) * need));
        ctx->ws_cap = need;
    }
    // How the solve is launched (never what it computes: tests/test_gpu_planner.py compares the coefficients bit for bit).
    //  * lanes: 64, 32 or 16 lanes of a wave carry a mission.  Below a chip's worth of full waves the kernel is bound by the
    //    latency of its dependent chains at one wave per SIMD, and two half-full waves hide each other's (B = 32 768, m = 8:
    //    53 -> 49 us; B = 4 096: 45 -> 37 us).  Option "solve_lanes".
    //  * parking in LDS (north_star's "LDS-staged" solve): (m - 1) x 28 x lanes doubles per wave instead of the HBM workspace, when
    //    that fits and every wave of the launch is resident at once (a CU holds floor(156 KB / that) of them).  Option "solve_park".
    //  * from a chip's worth of full waves on, the bound is the workspace's traffic (0.71 GB per launch at B = 65 536, m = 12
    //    against 0.18 GB of inputs and coefficients): the first five knots' blocks stay in registers (146 -> 115 us there,
    //    394 -> 278 us at B = 262 144, m = 8).  Uniform batches; option "solve_keep".
    const int waves64 = (B + TB - 1) / TB, cus = ctx->n_simds / 4;
    int lanes = ctx->solve_lanes;
    if (lanes != 64 && lanes != 32 && lanes != 16) lanes = waves64 <= ctx->n_simds ? 32 : 64;
    const int waves = (B + lanes - 1) / lanes;
    const size_t park = (size_t)(m > 1 ? m - 1 : 0) * 28 * lanes * sizeof(double);
    const size_t static_lds = sizeof(double) * TB * 25 + (seg_offsets ? TB * 12 : 12);
    const bool fits = m > 1 && park + static_lds <= (size_t)150 * 1024;
    const int per_cu = fits ? (int)(((size_t)156 * 1024) / (park + static_lds)) : 0;
    const bool lds_park = fits && (ctx->solve_park >= 0 ? ctx->solve_park != 0 : waves <= cus * (per_cu < 8 ? per_cu : 8));
    const bool keep = !seg_offsets && !lds_park && (ctx->solve_keep >= 0 ? ctx->solve_keep != 0 : (ctx->solve_lanes < 0 && waves64 >= ctx->n_simds));
    const dim3 grid(keep ? waves64 : waves);
#define UAVAC_SOLVE_LAUNCH(R, P, N, K)                                                                                             \
    do {                                                                                                                            \
        auto kern = minsnap_solve_bt_kernel<R, P, N, K>;                                                                            \
        if (P && park > 48 * 1024) UAVAC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)park)); \
        hipLaunchKernelGGL(kern, grid, dim3(TB), P ? park : 0, ctx->stream, wp, times, B, m, ctx->d_ws, coeffs, status, ctx->d_flags,   \
                           seg_offsets, guard_rows, guard_capacity, active);                                                        \
    } while (0)
#define UAVAC_SOLVE_LANES(R, P)                                                                                                     \
    do {                                                                                                                            \
        if (lanes == 64) UAVAC_SOLVE_LAUNCH(R, P, 64, 0); else if (lanes == 32) UAVAC_SOLVE_LAUNCH(R, P, 32, 0); else UAVAC_SOLVE_LAUNCH(R, P, 16, 0); \
    } while (0)
    if (keep) {                                           // (the sixth knot's slab in LDS: 28 x 64 doubles of dynamic shared memory)
        auto kern = minsnap_solve_bt_kernel<false, false, 64, 5>;
        hipLaunchKernelGGL(kern, grid, dim3(TB), 28 * TB * sizeof(double), ctx->stream, wp, times, B, m, ctx->d_ws, coeffs, status,
                           ctx->d_flags, seg_offsets, guard_rows, guard_capacity, active);
    }
    else if (lds_park) { if (seg_offsets) UAVAC_SOLVE_LANES(true, true); else UAVAC_SOLVE_LANES(false, true); }
    else { if (seg_offsets) UAVAC_SOLVE_LANES(true, false); else UAVAC_SOLVE_LANES(false, false); }
#undef UAVAC_SOLVE_LANES
#undef UAVAC_SOLVE_LAUNCH
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
