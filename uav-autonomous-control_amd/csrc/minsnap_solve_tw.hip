// Minimum-snap coefficient solve, TWO-ENDED block-Thomas form: two lanes per mission (gfx950).
//
// Same QP, same knot-derivative coordinates and the same 4x4 block-tridiagonal KKT system as minsnap_solve_bt.hip (see there and
// minsnap_solve.hip for the derivation and the reference lines it replaces: uav_ac/planning/minimum_snap.py:138-255).  That kernel
// eliminates the interior knots from the first to the last in ONE lane and substitutes back; what bounds it from a chip's worth of
// waves on is where the forward sweep's [Ut | rt] blocks wait for the backward sweep: a lane holds seven of them on chip (five in
// registers, one in LDS, the last where it was computed), the rest is parked in HBM -- four of eleven at m = 12, twelve of nineteen
// at m = 20 (counter bytes 2.5 x the algorithmic ones).
//
// Here a mission has TWO lanes of one wave: lane q (0..31, the HEAD) eliminates knots 0 .. mid-1 forwards, lane q + 32 (the TAIL)
// eliminates knots nk-1 .. mid+1 backwards; they meet at knot mid, whose 4x4 system collects both Schur complements, and each
// substitutes its own half back and writes its own half of the coefficients.  A lane's chain is half as long (latency-bound batches
// take half the time) and the pair holds fourteen blocks on chip: nothing is parked in HBM up to m = 15, five knots at m = 20.
//
// The tail lane runs the SAME code as the head on the time-reversed mission (waypoints and durations in reverse order): the
// minimum-snap QP is symmetric under t -> T - t, with the knot unknowns (v, a, j, lambda) mapping to J (v, a, j, lambda),
// J = diag(-1, +1, -1, -1) (odd derivatives change sign; the multiplier of "snap_end(prev) - snap_start(next) = 0" changes sign
// because prev and next swap).  So the backward elimination is the forward elimination of the reversed mission, the junction adds
// J H_tail J and J h_tail to the head's Schur complement, and a tail lane's coefficients come from the same `segment_coeffs` with start
// and end swapped back and J applied (sign flips: exact).
//
// Another elimination order than the one-ended kernel's: other rounding (both agree with the dense pivoted solve of the reference
// formulation to ~1e-11 on the sampled trajectories).  Every launch shape of THIS file computes the same bits (lanes per wave, where
// the blocks are parked); which of the two files solves is the ctx option "solve_order" (1 = this one, the default; 0 = one-ended).

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

constexpr int TB = 64;          // one wave: 32 mission slots, lane q the head and lane q + 32 the tail of slot q

// H = C_prev - B_prev^T Ut, h = re_prev - B_prev^T rt: what the segments BEHIND a knot contribute to its 4x4 system (the Schur
// complement of everything eliminated so far).  With `any` false nothing has been eliminated yet: H = C_prev, h = re_prev.
__device__ __forceinline__ void schur_behind(const Seg &prev, bool any, const double Ut[4][4], const double rt[4][3], double H[4][4],
                                             double h[4][3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = prev.C[i][j];
            if (any) {
#pragma unroll
                for (int l = 0; l < 4; ++l) s = fma(-prev.B[l][i], Ut[l][j], s);
            }
            H[i][j] = s;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double s = prev.re[i][a];
            if (any) {
#pragma unroll
                for (int l = 0; l < 4; ++l) s = fma(-prev.B[l][i], rt[l][a], s);
            }
            h[i][a] = s;
        }
    }
}

// sign of unknown i (v, a, j, lambda) under time reversal
__device__ __forceinline__ constexpr double jsign(int i) { return i == 1 ? 1.0 : -1.0; }

// Memory discipline as in minsnap_solve_bt.hip: every load is issued a knot / segment before its use and before the stores of the
// step it is issued in; no workgroup barrier (one wave; LDS operations execute in order: `lds_wave_fence`).
// NM: mission slots of a wave that carry a mission (32, 16, 8): fewer missions per wave = more waves for a batch that does not fill
// the chip (the kernel is bound by the latency of its dependent chains there).
// RAGGED: mission b has m_b = seg_offsets[b + 1] - seg_offsets[b] segments (clamped to 1 .. m_uniform); its two lanes split ITS knots.
// PARK_LDS: the blocks wait in the wave's LDS ([max knots of a lane][28][2 NM] doubles, dynamic) instead of the HBM workspace.
// NREG: the first NREG blocks of a lane stay in registers, block NREG in a [28][64] LDS slab (uniform batches at one wave per SIMD).
template <bool RAGGED, bool PARK_LDS = false, int NM = 32, int NREG = 0>
__global__ void __launch_bounds__(TB) minsnap_solve_tw_kernel(const double *__restrict__ wp, const double *__restrict__ times, int B,
                                                             int m_uniform, double *__restrict__ ws, double *__restrict__ coeffs,
                                                             int32_t *__restrict__ status, int32_t *__restrict__ flags,
                                                             const int64_t *__restrict__ seg_offsets,
                                                             const int64_t *__restrict__ guard_rows, int64_t guard_capacity,
                                                             const int32_t *__restrict__ active) {
    if (guard_rows && *guard_rows > guard_capacity) return;       // a refused planning chain: the coefficients stay what they were
    const int lane = threadIdx.x;
    const int q = lane & 31;                                      // mission slot
    const bool tail = lane >= 32;
    const int b0 = blockIdx.x * NM;
    const int b = b0 + q;
    const bool live = q < NM && b < B;
    if (active) {                                                 // obstacle loop: a wave none of whose missions is active leaves at once
        if (!__any(live && active[b] != 0)) return;
    }
    // one segment's 24 coefficients per LANE (+1 pad) -- and, before the substitution starts, the waiting place of ONE more block per
    // lane ([28][64]): the block of view knot nel - 2 is the first one the substitution asks for (in its step 0, before that step
    // writes the stage), so it spends the end of the elimination here instead of in HBM
    __shared__ double stage[TB * 28];
    extern __shared__ double park_lds[];              // PARK_LDS: [knots of a lane][28][2 NM]; NREG > 0: [28][64]
    __shared__ int64_t seg0_of[RAGGED ? 32 : 1];      // ragged: first segment and segment count of every mission slot
    __shared__ int m_of[RAGGED ? 32 : 1];
    const int bb = live ? b : B - 1;
    const size_t sB = (size_t)B;
    int m = m_uniform;
    const double *w = wp + (size_t)bb * (m_uniform + 1) * 3;
    const double *tm = times + (size_t)bb * m_uniform;
    if (RAGGED) {
        const int64_t s0 = seg_offsets[bb], mb = seg_offsets[bb + 1] - s0;
        m = (int)(mb < 1 ? 1 : (mb > m_uniform ? m_uniform : mb));
        w = wp + ((size_t)s0 + (size_t)bb) * 3;
        tm = times + (size_t)s0;
        if (!tail) { seg0_of[q] = s0; m_of[q] = m; }
        lds_wave_fence();
    }
    // ---- this lane's half of the mission, in VIEW coordinates: the head sees the mission as it is, the tail sees it reversed
    const int nk = m - 1;                              // interior knots
    const bool hasj = nk >= 1;                         // there is a junction knot (mid); m = 1 has no unknowns at all
    const int mid = nk >> 1;
    const int nel = !hasj ? 0 : (tail ? nk - 1 - mid : mid);          // knots this lane eliminates: view knots 0 .. nel - 1
    const int nseg = tail ? (hasj ? nel + 1 : 0) : nel + 1;           // segments this lane writes: view segments nel .. 0
    auto wv = [&](int i) -> const double * { return w + 3 * (tail ? m - i : i); };      // view waypoint i
    auto tv = [&](int s) -> double { return tm[tail ? m - 1 - s : s]; };                 // duration of view segment s
    // where view knot kk's block waits, and how far apart its 28 values are
    const int slot = (tail ? NM : 0) + (q < NM ? q : 0);
    auto in_stage = [&](int kk) -> bool { return !PARK_LDS && kk == nel - 2 && !(NREG > 0 && kk <= NREG); };
    auto park_stride = [&](int kk) -> size_t {
        return ((NREG > 0 && kk == NREG) || in_stage(kk)) ? (size_t)TB : (PARK_LDS ? (size_t)(2 * NM) : sB);
    };
    auto park_at = [&](int kk) -> double * {
        if (NREG > 0 && kk == NREG) return park_lds + lane;
        if (in_stage(kk)) return stage + lane;
        if (PARK_LDS) return park_lds + (size_t)kk * 28 * (2 * NM) + slot;
        return ws + ((size_t)(tail ? nk - 1 - kk : kk) * 28) * sB + bb;       // (row = the knot's index in the mission)
    };
    const bool parks = PARK_LDS ? q < NM : live;
    bool ok = true;
    double kept[NREG > 0 ? NREG : 1][28];

    // ------------------------------------------------------------------ elimination of this lane's knots
    double Ut[4][4], rt[4][3];
    Seg prev;
    {
        Seg cur;
        double p0[3] = {wv(0)[0], wv(0)[1], wv(0)[2]}, p1[3] = {wv(1)[0], wv(1)[1], wv(1)[2]};
        build_segment(prev, tv(0), p0, p1);
        const double *n2 = wv(m >= 2 ? 2 : 1);
        double nw[3] = {n2[0], n2[1], n2[2]}, nt = tv(m > 1 ? 1 : 0);
        for (int kk = 0; kk < nel; ++kk) {
            const int k = kk + 1;                      // view knot kk joins view segments kk (prev) and kk + 1 (cur)
            const double T = nt;
#pragma unroll
            for (int a = 0; a < 3; ++a) { p0[a] = p1[a]; p1[a] = nw[a]; }
            {
                const double *nx = wv((k + 2 <= m) ? k + 2 : m);
#pragma unroll
                for (int a = 0; a < 3; ++a) nw[a] = nx[a];
                nt = tv((k + 1 <= m - 1) ? k + 1 : m - 1);
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
            const bool in_regs = (NREG > 0 && kk < NREG) || kk == nel - 1;      // (this lane's last block is what its substitution starts from)
            const bool to_stage = in_stage(kk);                                 // (a lane's own column of the stage: any lane may use it)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { Ut[i][j] = R[i][j]; if (!in_regs && (parks || to_stage)) o[(size_t)(i * 4 + j) * ost] = R[i][j]; }
#pragma unroll
                for (int a = 0; a < 3; ++a) { rt[i][a] = R[i][4 + a]; if (!in_regs && (parks || to_stage)) o[(size_t)(16 + i * 3 + a) * ost] = R[i][4 + a]; }
            }
            if (NREG > 0 && kk < NREG) {
#pragma unroll
                for (int qq = 0; qq < NREG; ++qq)
                    if (kk == qq) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) kept[qq][i * 4 + j] = R[i][j];
#pragma unroll
                            for (int a = 0; a < 3; ++a) kept[qq][16 + i * 3 + a] = R[i][4 + a];
                        }
                    }
            }
            prev = cur;
        }
    }

    // ------------------------------------------------------------------ the junction knot (view knot nel of BOTH lanes)
    double xn[4][3];                                    // unknowns of the knot at the far end of the segment at hand, view coordinates
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) xn[i][a] = 0.0;
    {
        double H[4][4], h[4][3];
        schur_behind(prev, nel > 0, Ut, rt, H, h);
        // the tail's contribution travels to the head, the solution back (every lane takes part in the exchanges)
        double S[4][4], R[4][7];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double other = __shfl(H[i][j], q + 32);
                S[i][j] = H[i][j] + (jsign(i) * jsign(j)) * other;
                R[i][j] = 0.0;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double other = __shfl(h[i][a], q + 32);
                R[i][4 + a] = h[i][a] + jsign(i) * other;
            }
        }
        const bool okj = solve4(S, R);
        const bool ok_tail = __shfl(ok ? 1 : 0, q + 32) != 0;
        const bool ok_head = ok && (!hasj || okj) && ok_tail;         // (evaluated by every lane; meaningful on the head)
        ok = __shfl(ok_head ? 1 : 0, q) != 0;                          // both lanes of the pair agree
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double xh = __shfl(R[i][4 + a], q);              // the head's solution, to both lanes
                if (hasj) xn[i][a] = tail ? jsign(i) * xh : xh;
            }
    }
    if (live && !tail) {
        if (!ok) atomicOr(&flags[1], 1);
        if (status) status[b] = ok ? 0 : 1;
    }

    // ------------------------------------------------ substitution + coefficients, from the junction outwards
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    int nseg_top = live ? nseg : 0;                    // steps: the longest half of the wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nseg_top = max(nseg_top, __shfl_xor(nseg_top, d));
    // 64 lanes x 24 doubles of one step leave the stage as 192-byte runs: lane e's view segment nel_e - step, i.e. segment
    // (head: that; tail: m_e - 1 - that) of its mission
    auto flush = [&](int step) {
        for (int e = lane; e < TB * 24; e += TB) {
            const int ql = e / 24, j = e - ql * 24;
            const int sl = ql & 31;
            const bool tl = ql >= 32;
            const int mq = RAGGED ? m_of[sl < NM ? sl : 0] : m_uniform;
            const int nkq = mq - 1, midq = nkq >> 1;
            const int nelq = nkq < 1 ? 0 : (tl ? nkq - 1 - midq : midq);
            const int nsegq = tl ? (nkq >= 1 ? nelq + 1 : 0) : nelq + 1;
            const int sv = nelq - step;
            const size_t first = RAGGED ? (size_t)seg0_of[sl < NM ? sl : 0] : (size_t)(b0 + sl) * m_uniform;
            if (sl < NM && b0 + sl < B && step < nsegq) coeffs[(first + (size_t)(tl ? mq - 1 - sv : sv)) * 24 + j] = stage[ql * 25 + j];
        }
    };
    // on their way while the segment before is computed: the block of view knot s - 2, start waypoint and duration of view segment s - 1
    double nxt[28], nw[3], nt;
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[i * 4 + j] = Ut[i][j];
#pragma unroll
            for (int a = 0; a < 3; ++a) nxt[16 + i * 3 + a] = rt[i][a];
        }
        const double *s0 = wv(nel);
#pragma unroll
        for (int a = 0; a < 3; ++a) nw[a] = s0[a];
        nt = tv(nel);
    }
    double p1[3] = {wv(nel + 1)[0], wv(nel + 1)[1], wv(nel + 1)[2]};
    for (int step = 0; step < nseg_top; ++step) {
        const int s = nel - step;                       // this lane's view segment; < 0: its half is finished
        double cur[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) cur[i] = nxt[i];
        const double T = nt;
        const double p0[3] = {nw[0], nw[1], nw[2]};
        if (step > 0) flush(step - 1);                  // reads the stage before this step overwrites it (LDS is in order)
        if (s >= 2) {
            if (NREG > 0 && s - 2 < NREG) {
#pragma unroll
                for (int qq = 0; qq < NREG; ++qq)
                    if (s - 2 == qq) {
#pragma unroll
                        for (int i = 0; i < 28; ++i) nxt[i] = kept[qq][i];
                    }
            } else {
                const double *o = park_at(s - 2);
                const size_t ost = park_stride(s - 2);
#pragma unroll
                for (int i = 0; i < 28; ++i) nxt[i] = o[(size_t)i * ost];
            }
        }
        if (s >= 1) {
            const double *sw = wv(s - 1);
#pragma unroll
            for (int a = 0; a < 3; ++a) nw[a] = sw[a];
            nt = tv(s - 1);
        }
        if (s >= 0 && step < nseg) {
            double xs[4][3];                            // unknowns of view knot s - 1 (the near end of view segment s); zero at the mission's end point
            if (s >= 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double v = cur[16 + i * 3 + a];
#pragma unroll
                        for (int l = 0; l < 4; ++l) v = fma(-cur[i * 4 + l], xn[l][a], v);
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
            // in the mission's own time direction the segment starts at the view's near end for the head and at its far end for the tail
            double x0[3][3], x1[3][3], q0[3], q1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    x0[i][a] = tail ? jsign(i) * xn[i][a] : xs[i][a];
                    x1[i][a] = tail ? jsign(i) * xs[i][a] : xn[i][a];
                }
#pragma unroll
            for (int a = 0; a < 3; ++a) { q0[a] = tail ? p1[a] : p0[a]; q1[a] = tail ? p0[a] : p1[a]; }
            double c[8][3];
            segment_coeffs(ip, T, q0, q1, x0, x1, c);
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
    if (nseg_top > 0) flush(nseg_top - 1);
}

}  // namespace

int uavac_launch_solve_tw(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                          int32_t *status, const int64_t *seg_offsets, const int64_t *guard_rows, int64_t guard_capacity,
                          const int32_t *active) {
    // How the solve is launched (never what it computes: the coefficients are compared bit for bit across these choices).
    //  * missions per wave: 32, 16 or 8 (option "solve_lanes" = 64 / 32 / 16 lanes that carry a mission's half): below a chip's worth
    //    of full waves the kernel is bound by the latency of its dependent chains, and more, emptier waves hide each other's.
    //  * parking in LDS when that fits and every wave of the launch is resident at once (option "solve_park").
    //  * from a chip's worth of full waves on: five blocks per lane in registers, the sixth in an LDS slab (option "solve_keep").
    const int cus = ctx->n_simds / 4;
    const int waves32 = (B + 31) / 32;
    int nm = ctx->solve_lanes == 64 ? 32 : (ctx->solve_lanes == 32 ? 16 : (ctx->solve_lanes == 16 ? 8 : (waves32 <= ctx->n_simds / 2 ? 16 : 32)));
    const int waves = (B + nm - 1) / nm;
    const int lane_knots = m > 1 ? (m - 1) / 2 : 0;                       // most blocks one lane parks
    const size_t park = (size_t)lane_knots * 28 * (2 * nm) * sizeof(double);
    const size_t static_lds = sizeof(double) * TB * 28 + (seg_offsets ? 32 * 12 : 12);
    const bool fits = lane_knots > 0 && park + static_lds <= (size_t)150 * 1024;
    const int per_cu = fits ? (int)(((size_t)156 * 1024) / (park + static_lds)) : 0;
    const bool lds_park = fits && (ctx->solve_park >= 0 ? ctx->solve_park != 0 : waves <= cus * (per_cu < 8 ? per_cu : 8));
    // (a lane with up to three blocks has little to keep -- one waits in the stage, one where it was computed: at m = 8 the 196-register
    // form at two waves per SIMD runs 55.5 against 59.3 us at B = 65 536 and 213 against 227 at 262 144; from four blocks on -- m = 12:
    // 85.8 against 99.2 -- the 484-register form wins)
    const bool keep = !seg_offsets && !lds_park &&
                      (ctx->solve_keep >= 0 ? ctx->solve_keep != 0 : (ctx->solve_lanes < 0 && waves32 >= ctx->n_simds && lane_knots >= 4));
    const dim3 grid(keep ? waves32 : waves);
    // The HBM workspace [m - 1][28][B] (rows = a knot's index in its mission) is only needed by the forms that park there: a launch
    // that parks in LDS never touches it (round-5 advice: 411 MB at B = 262 144, m = 8 were allocated for nothing).
    if (!lds_park) {
        const size_t need = (size_t)(m > 1 ? m - 1 : 1) * 28 * (size_t)B;
        if (need > ctx->ws_cap) {
            if (ctx->d_ws) UAVAC_HIP(ctx, hipFree(ctx->d_ws));
            ctx->d_ws = nullptr;
            ctx->ws_cap = 0;
            UAVAC_HIP(ctx, hipMalloc(&ctx->d_ws, sizeof(double) * need));
            ctx->ws_cap = need;
        }
    }
#define UAVAC_SOLVE_LAUNCH(R, P, N, K)                                                                                             \
    do {                                                                                                                            \
        auto kern = minsnap_solve_tw_kernel<R, P, N, K>;                                                                            \
        if (P && park > 48 * 1024) UAVAC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)park)); \
        hipLaunchKernelGGL(kern, grid, dim3(TB), P ? park : 0, ctx->stream, wp, times, B, m, ctx->d_ws, coeffs, status, ctx->d_flags,   \
                           seg_offsets, guard_rows, guard_capacity, active);                                                        \
    } while (0)
#define UAVAC_SOLVE_LANES(R, P)                                                                                                     \
    do {                                                                                                                            \
        if (nm == 32) UAVAC_SOLVE_LAUNCH(R, P, 32, 0); else if (nm == 16) UAVAC_SOLVE_LAUNCH(R, P, 16, 0); else UAVAC_SOLVE_LAUNCH(R, P, 8, 0); \
    } while (0)
    if (keep) {                                           // (the sixth block's slab in LDS: 28 x 64 doubles of dynamic shared memory)
        auto kern = minsnap_solve_tw_kernel<false, false, 32, 5>;
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
