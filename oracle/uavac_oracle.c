/*
 * uavac_oracle.c -- scalar C restatement of the reference hot path.
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it (through oracle/c_oracle.py).  libuavac.so never links it.
 *
 * One UAV / one mission at a time, in the reference's own formulation (upstream paths):
 *   planner   uav_ac/planning/minimum_snap.py: times :311-321, A/b :171-255 + :293-309,
 *             H :155-169, KKT solve with method="solve" (np.linalg.solve = LU with partial
 *             pivoting) :138-153, sampler :100-119, polynom :257-286, yaw scan :126-136
 *   control   uav_ac/control/controller.py:26-191, uav_ac/quadrotor/quad.py:88-155,189-213,
 *             uav_ac/main.py:37-61
 *   dynamics  rotor wrench uav_ac/simulation/mujoco_sim.py:232-251 + MuJoCo Euler free-joint step
 *             (parity vs MuJoCo itself is UNPINNED, see control_oracle.py)
 *
 * Pinned by tests/test_oracle_c.py against the golden vectors produced by importing the
 * reference (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NC 8
#define PI 3.14159265358979323846

/* ------------------------------------------------------------------------------- planner */
static void polynom(int order, double t, double *row) { /* minimum_snap.py:257-286 */
    for (int i = 0; i < NC; ++i) {
        double poly = 1.0, der = (double)i;
        for (int k = 0; k < order; ++k) {
            poly *= der;
            if (der > 0) der -= 1.0;
        }
        row[i] = poly * pow(t, der);
    }
}

void oracle_times(const double *wp, int m, double velocity, double *times) { /* :311-321 */
    for (int i = 0; i < m; ++i) {
        double dx = wp[3 * (i + 1)] - wp[3 * i], dy = wp[3 * (i + 1) + 1] - wp[3 * i + 1],
               dz = wp[3 * (i + 1) + 2] - wp[3 * i + 2];
        /* np.linalg.norm of a 3-vector = sqrt(x.dot(x)); BLAS ddot accumulates with fused multiply-adds (see
         * rrt_oracle.c).  Only this form reproduces the committed reference times bit for bit. */
        double t = sqrt(fma(dz, dz, fma(dy, dy, dx * dx))) / velocity;
        if (i == 0 || i == m - 1) t *= 1.5;
        times[i] = t;
    }
}

/* dense LU with partial pivoting, nrhs right-hand sides, in place; returns 0 or -1 if singular */
static int lu_solve(double *A, double *b, int n, int nrhs) {
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (!(best > 0.0)) return -1;
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            for (int j = 0; j < nrhs; ++j) { double t = b[k * nrhs + j]; b[k * nrhs + j] = b[p * nrhs + j]; b[p * nrhs + j] = t; }
        }
        for (int i = k + 1; i < n; ++i) {
            double l = A[i * n + k] / A[k * n + k];
            if (l == 0.0) continue;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
            for (int j = 0; j < nrhs; ++j) b[i * nrhs + j] -= l * b[k * nrhs + j];
        }
    }
    for (int i = n - 1; i >= 0; --i)
        for (int j = 0; j < nrhs; ++j) {
            double s = b[i * nrhs + j];
            for (int c = i + 1; c < n; ++c) s -= A[i * n + c] * b[c * nrhs + j];
            b[i * nrhs + j] = s / A[i * n + i];
        }
    return 0;
}

/* coeffs [8m][3]; returns 0 ok, -1 singular, -2 out of memory */
int oracle_solve(const double *wp, int m, double velocity, double *coeffs, double *times) {
    oracle_times(wp, m, velocity, times);
    int nu = NC * m, ncon = 6 * m + 2, n = nu + ncon;
    double *K = (double *)calloc((size_t)n * n, sizeof(double));
    double *rhs = (double *)calloc((size_t)n * 3, sizeof(double));
    if (!K || !rhs) { free(K); free(rhs); return -2; }
    double row[NC], row0[NC];
    int r = 0;
    /* A occupies K[nu + r][c] and its transpose K[c][nu + r] */
#define SETA(rr, cc, v) do { K[(size_t)(nu + (rr)) * n + (cc)] = (v); K[(size_t)(cc) * n + nu + (rr)] = (v); } while (0)
    polynom(0, 0.0, row0);                                      /* positions at t = 0  (:240-245) */
    for (int s = 0; s < m; ++s, ++r) {
        for (int i = 0; i < NC; ++i) SETA(r, s * NC + i, row0[i]);
        for (int j = 0; j < 3; ++j) rhs[(size_t)(nu + r) * 3 + j] = wp[3 * s + j];
    }
    for (int s = 0; s < m; ++s, ++r) {                          /* positions at t = T  (:248-255) */
        polynom(0, times[s], row);
        for (int i = 0; i < NC; ++i) SETA(r, s * NC + i, row[i]);
        for (int j = 0; j < 3; ++j) rhs[(size_t)(nu + r) * 3 + j] = wp[3 * (s + 1) + j];
    }
    for (int k = 1; k <= 3; ++k, ++r) {                         /* start at rest (:214-217) */
        polynom(k, 0.0, row);
        for (int i = 0; i < NC; ++i) SETA(r, i, row[i]);
    }
    for (int k = 1; k <= 3; ++k, ++r) {                         /* goal at rest (:220-223) */
        polynom(k, times[m - 1], row);
        for (int i = 0; i < NC; ++i) SETA(r, (m - 1) * NC + i, row[i]);
    }
    for (int s = 1; s < m; ++s)                                 /* continuity k = 1..4 (:191-198) */
        for (int k = 1; k <= 4; ++k, ++r) {
            polynom(k, times[s - 1], row);
            polynom(k, 0.0, row0);
            for (int i = 0; i < NC; ++i) { SETA(r, (s - 1) * NC + i, row[i]); SETA(r, s * NC + i, -row0[i]); }
        }
    for (int s = 0; s < m; ++s)                                 /* snap cost (:155-169) */
        for (int a = 4; a < NC; ++a)
            for (int c = 4; c < NC; ++c) {
                double fa = a * (a - 1) * (a - 2) * (a - 3), fc = c * (c - 1) * (c - 2) * (c - 3);
                int e = a + c - 7;
                K[(size_t)(s * NC + a) * n + s * NC + c] = fa * fc * pow(times[s], e) / e;
            }
    int rc = lu_solve(K, rhs, n, 3);
    if (rc == 0) memcpy(coeffs, rhs, sizeof(double) * nu * 3);
    free(K); free(rhs);
    return rc;
}

int64_t oracle_row_count(const double *times, int m, double dt) { /* len(np.arange(0, T, dt)) */
    int64_t n = 0;
    for (int s = 0; s < m; ++s) { double q = ceil(times[s] / dt); if (q > 0) n += (int64_t)q; }
    return n;
}

static double floored_mod(double a, double b) { /* Python / NumPy float % for b > 0 */
    double r = fmod(a, b);
    if (r != 0.0 && r < 0.0) r += b;
    return r;
}

/* traj [nrows][11]; returns rows written */
int64_t oracle_sample(const double *coeffs, const double *times, int m, double dt, double *traj) {
    int64_t n = 0;
    double row[NC];
    for (int s = 0; s < m; ++s) {                               /* :100-119 */
        int64_t cnt = (int64_t)ceil(times[s] / dt);
        for (int64_t k = 0; k < cnt; ++k, ++n) {
            double t = (double)k * dt;
            double *o = traj + n * 11;
            for (int ord = 0; ord < 3; ++ord) {
                polynom(ord, t, row);
                for (int j = 0; j < 3; ++j) {
                    double acc = 0.0;
                    for (int i = 0; i < NC; ++i) acc += row[i] * coeffs[(size_t)(s * NC + i) * 3 + j];
                    o[3 * ord + j] = acc;
                }
            }
            o[10] = (double)s;
        }
    }
    /* yaw scan (:126-136): unwrap over the valid subset, hold, back-fill */
    int have = 0;
    double prev_raw = 0.0, cum = 0.0, last = 0.0;
    int64_t first = -1;
    for (int64_t i = 0; i < n; ++i) {
        double vx = traj[i * 11 + 3], vy = traj[i * 11 + 4];
        if (sqrt(vx * vx + vy * vy) >= 1e-3) {
            double a = atan2(vy, vx);
            if (have) {
                double dd = a - prev_raw;
                double ddmod = floored_mod(dd + PI, 2 * PI) - PI;
                if (ddmod == -PI && dd > 0) ddmod = PI;
                double corr = ddmod - dd;
                if (fabs(dd) < PI) corr = 0.0;
                cum += corr;
            } else { first = i; }
            have = 1;
            prev_raw = a;
            last = a + cum;
        }
        traj[i * 11 + 9] = have ? last : 0.0;
    }
    if (first > 0) for (int64_t i = 0; i < first; ++i) traj[i * 11 + 9] = traj[first * 11 + 9];
    return n;
}

/* ------------------------------------------------------------------------------- control */
typedef struct {
    double g, dt, dt_outer, mass, I[3], arm, kf, kappa, min_thrust, max_thrust, tau_rise, tau_fall;
    double max_ascent, max_descent, max_speed_xy, max_horiz_accel, max_tilt;
    double kp_xy, kd_xy, kp_z, kd_z, ki_z, kp_roll, kp_pitch, kp_yaw, kp_p, kp_q, kp_r;
    int32_t F, ground;
    double ground_z, ground_clearance, ground_timeconst;
} oracle_vehicle;   /* same field order as uavac_vehicle so tests can share one ctypes struct */

static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void quat_to_rot(const double *q, double R[3][3]) { /* quad.py:133-155 */
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double a = q[0] / n, b = q[1] / n, c = q[2] / n, d = q[3] / n;
    double S[3][3] = {{0, -d, c}, {d, 0, -b}, {-c, b, 0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double ss = 0.0;
            for (int k = 0; k < 3; ++k) ss += S[i][k] * S[k][j];
            R[i][j] = (i == j ? 1.0 : 0.0) + 2.0 * ss + 2.0 * a * S[i][j];
        }
}

typedef struct {
    double X[13], omega[4], omega_cmd[4], integ, thrust, pqr[3];
    int32_t idx, inner, collided, gbits;
} uav_t;

static void controller_tick(const oracle_vehicle *V, uav_t *u, const double *traj, int64_t nrows) {
    if (u->inner % V->F == 0 && nrows > 0) {                    /* main.py:47-61 */
        const double *tg = traj + (int64_t)u->idx * 11;
        double R[3][3];
        quat_to_rot(u->X + 3, R);
        /* altitude, controller.py:26-56 */
        double zd = clipd(tg[5], -V->max_ascent, V->max_descent);
        double e = tg[2] - u->X[2], ed = zd - u->X[9];
        u->integ = clipd(u->integ + e * V->dt_outer, -10.0, 10.0);
        double acc = V->kp_z * e + V->ki_z * u->integ + V->kd_z * ed + tg[8] - V->g;
        acc = acc / R[2][2];
        double c = clipd(-V->mass * acc, 4 * V->min_thrust, 4 * V->max_thrust);
        u->thrust = c;
        /* lateral, controller.py:58-97 */
        double vdx = tg[3], vdy = tg[4];
        double vm = sqrt(vdx * vdx + vdy * vdy);
        if (vm > V->max_speed_xy) { vdx = vdx / vm * V->max_speed_xy; vdy = vdy / vm * V->max_speed_xy; }
        double ax = V->kp_xy * (tg[0] - u->X[0]) + V->kd_xy * (vdx - u->X[7]) + tg[6];
        double ay = V->kp_xy * (tg[1] - u->X[1]) + V->kd_xy * (vdy - u->X[8]) + tg[7];
        double am = sqrt(ax * ax + ay * ay);
        if (am > V->max_horiz_accel) { ax = ax / am * V->max_horiz_accel; ay = ay / am * V->max_horiz_accel; }
        double az = -c / V->mass;
        double bx = clipd(ax / az, -V->max_tilt, V->max_tilt), by = clipd(ay / az, -V->max_tilt, V->max_tilt);
        /* roll / pitch, controller.py:132-154 */
        double bdx = V->kp_roll * (bx - R[0][2]), bdy = V->kp_pitch * (by - R[1][2]);
        double pc = (R[1][0] / R[2][2]) * bdx + (-R[0][0] / R[2][2]) * bdy;
        double qc = (R[1][1] / R[2][2]) * bdx + (-R[0][1] / R[2][2]) * bdy;
        /* yaw, controller.py:156-168 with quad.py:189-213 on the stored quaternion */
        const double *q = u->X + 3;
        double phi = atan2(2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[1] * q[1] + q[2] * q[2]));
        double theta = asin(clipd(2 * (q[0] * q[2] - q[3] * q[1]), -1.0, 1.0));
        double psi = atan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] * q[2] + q[3] * q[3]));
        double pd = floored_mod(tg[9], 2 * PI);
        double ye = floored_mod(pd - psi + PI, 2 * PI) - PI;
        double rc = (V->kp_yaw * ye * cos(theta) - qc * sin(phi)) / cos(phi);
        u->pqr[0] = pc; u->pqr[1] = qc; u->pqr[2] = rc;
        u->idx = (u->idx + 1 < nrows - 1) ? u->idx + 1 : (int32_t)(nrows - 1);
    }
    /* body rates, controller.py:115-130 */
    const double *w = u->X + 10;
    double kp[3] = {V->kp_p, V->kp_q, V->kp_r}, Iw[3], M[3];
    for (int i = 0; i < 3; ++i) Iw[i] = V->I[i] * w[i];
    M[0] = V->I[0] * kp[0] * (u->pqr[0] - w[0]) + (w[1] * Iw[2] - w[2] * Iw[1]);
    M[1] = V->I[1] * kp[1] * (u->pqr[1] - w[1]) + (w[2] * Iw[0] - w[0] * Iw[2]);
    M[2] = V->I[2] * kp[2] * (u->pqr[2] - w[2]) + (w[0] * Iw[1] - w[1] * Iw[0]);
    /* allocation, quad.py:105-122 */
    double cbar = clipd(u->thrust, 4 * V->min_thrust, 4 * V->max_thrust);
    double pb = M[0] / V->arm, qb = M[1] / V->arm, rb = -M[2] / V->kappa;
    double mf[4] = {(pb + qb + rb) / 4, (-pb + qb - rb) / 4, (-pb - qb + rb) / 4, (pb - qb - rb) / 4};
    double col = cbar / 4, lim = 1e300;
    for (int i = 0; i < 4; ++i) {
        double l = 1.0;
        if (mf[i] > 0) l = (V->max_thrust - col) / mf[i];
        else if (mf[i] < 0) l = (V->min_thrust - col) / mf[i];
        if (l < lim) lim = l;
    }
    double sc = clipd(lim, 0.0, 1.0);
    for (int i = 0; i < 4; ++i) {                               /* quad.py:88-103 */
        double f = clipd(col + sc * mf[i], V->min_thrust, V->max_thrust);
        u->omega_cmd[i] = sqrt(f / V->kf);
        double tau = u->omega_cmd[i] > u->omega[i] ? V->tau_rise : V->tau_fall;
        u->omega[i] += (1 - exp(-V->dt / tau)) * (u->omega_cmd[i] - u->omega[i]);
    }
    u->inner += 1;
}

static void dynamics_step(const oracle_vehicle *V, uav_t *u) {
    double f[4], dt = V->dt;
    for (int i = 0; i < 4; ++i) f[i] = V->kf * u->omega[i] * u->omega[i];
    double T = f[0] + f[1] + f[2] + f[3];
    double tau[3] = {V->arm * (f[0] + f[3] - f[1] - f[2]), V->arm * (f[0] + f[1] - f[2] - f[3]),
                     V->kappa * (-f[0] + f[1] - f[2] + f[3])};
    double R[3][3];
    quat_to_rot(u->X + 3, R);
    double *w = u->X + 10, Iw[3];
    for (int i = 0; i < 3; ++i) Iw[i] = V->I[i] * w[i];
    double cr[3] = {w[1] * Iw[2] - w[2] * Iw[1], w[2] * Iw[0] - w[0] * Iw[2], w[0] * Iw[1] - w[1] * Iw[0]};
    double acc[3] = {-(T / V->mass) * R[0][2], -(T / V->mass) * R[1][2], V->g - (T / V->mass) * R[2][2]};
    double wn2 = 0.0;
    /* BUILD-DEFINED ground contact (MuJoCo's soft-constraint solver cannot run here): while the body's
     * lowest point is below the plane, the vertical velocity update may not exceed the critically damped reference
     * vz + dt (-b vz - k r), b = 2/tc, k = 1/tc^2; the plane only pushes; no friction, no contact torque. */
    double vz_ref = 0.0;
    int touching = 0;
    if (V->ground) {
        double r = u->X[2] - (V->ground_z - V->ground_clearance), tc = V->ground_timeconst;
        if (r > 0.0) { touching = 1; vz_ref = u->X[9] + dt * -(2.0 / tc * u->X[9] + r / (tc * tc)); }
    }
    for (int i = 0; i < 3; ++i) {
        u->X[7 + i] += dt * acc[i];
        if (i == 2 && touching && vz_ref < u->X[9]) u->X[9] = vz_ref;
        w[i] += dt * ((tau[i] - cr[i]) / V->I[i]);
        u->X[i] += dt * u->X[7 + i];
        wn2 += w[i] * w[i];
    }
    double *q = u->X + 3, nq[4] = {q[0], q[1], q[2], q[3]};
    double wn = sqrt(wn2);
    if (wn > 0.0) {
        double h = 0.5 * wn * dt, s = sin(h) / wn, d0 = cos(h), d1 = s * w[0], d2 = s * w[1], d3 = s * w[2];
        nq[0] = q[0] * d0 - q[1] * d1 - q[2] * d2 - q[3] * d3;
        nq[1] = q[0] * d1 + q[1] * d0 + q[2] * d3 - q[3] * d2;
        nq[2] = q[0] * d2 - q[1] * d3 + q[2] * d0 + q[3] * d1;
        nq[3] = q[0] * d3 + q[1] * d2 - q[2] * d1 + q[3] * d0;
    }
    double nn = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int i = 0; i < 4; ++i) q[i] = nq[i] / nn;
    if (V->ground) {                                  /* MujocoSimulation._record_collisions, mujoco_sim.py:220-230 */
        if (V->ground_z - u->X[2] >= 0.1) u->gbits |= 2;                           /* TAKEOFF_HEIGHT reached (sticky) */
        int now = u->X[2] - (V->ground_z - V->ground_clearance) > 0.0;
        u->gbits = now ? (u->gbits | 1) : (u->gbits & ~1);
        if (now && (u->gbits & 2)) u->gbits |= 4;                                  /* contact after take-off (sticky) */
    }
}

/* state [26] / istate [4] in the row order of include/uavac.h; logs [K][13] and [K][12] (or NULL) */
void oracle_rollout(const oracle_vehicle *V, const double *traj, int64_t nrows, double *state, int32_t *istate,
                    int K, double *state_log, double *cmd_log, const double *aabbs, int n_obs) {
    uav_t u;
    memcpy(u.X, state, 13 * sizeof(double));
    memcpy(u.omega, state + 13, 4 * sizeof(double));
    memcpy(u.omega_cmd, state + 17, 4 * sizeof(double));
    u.integ = state[21]; u.thrust = state[22];
    memcpy(u.pqr, state + 23, 3 * sizeof(double));
    u.idx = istate[0]; u.inner = istate[1]; u.collided = istate[2]; u.gbits = istate[3];
    for (int k = 0; k < K; ++k) {
        controller_tick(V, &u, traj, nrows);
        if (cmd_log) {
            double *c = cmd_log + (size_t)k * 12;
            c[0] = u.thrust; memcpy(c + 1, u.pqr, 24); memcpy(c + 4, u.omega_cmd, 32); memcpy(c + 8, u.omega, 32);
        }
        dynamics_step(V, &u);
        for (int o = 0; o < n_obs; ++o) {
            const double *c = aabbs + 6 * o;
            if (u.X[0] >= c[0] && u.X[0] <= c[1] && u.X[1] >= c[2] && u.X[1] <= c[3] && u.X[2] >= c[4] && u.X[2] <= c[5])
                u.collided = 1;
        }
        if (state_log) memcpy(state_log + (size_t)k * 13, u.X, 13 * sizeof(double));
    }
    memcpy(state, u.X, 13 * sizeof(double));
    memcpy(state + 13, u.omega, 4 * sizeof(double));
    memcpy(state + 17, u.omega_cmd, 4 * sizeof(double));
    state[21] = u.integ; state[22] = u.thrust;
    memcpy(state + 23, u.pqr, 3 * sizeof(double));
    istate[0] = u.idx; istate[1] = u.inner; istate[2] = u.collided; istate[3] = u.gbits;
}

/* ------------------------------------------------------------------ all-core timing leg (bench.py cpu_baseline)
 * n_threads POSIX threads, each planning and flying whole missions (mission i of wps[n][m+1][3], i = tid,
 * tid + n_threads, ... recycled) with its own buffers until budget_s of wall time has passed.  Returns the
 * missions completed by all threads; *elapsed_s = wall time from start to the last thread's end. */
#include <pthread.h>
#include <time.h>

typedef struct {
    const oracle_vehicle *V;
    const double *wps;
    int n, m, ticks, tid, n_threads;
    double velocity, dt, t_end;
    int64_t done;
} bench_arg;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *bench_worker(void *p) {
    bench_arg *a = (bench_arg *)p;
    const int m = a->m;
    double *coeffs = (double *)malloc(sizeof(double) * 24 * (size_t)m);
    double *times = (double *)malloc(sizeof(double) * (size_t)m);
    double *slog = (double *)malloc(sizeof(double) * 13 * (size_t)a->ticks);
    double *traj = NULL;
    int64_t cap = 0;
    for (int64_t i = a->tid; now_s() < a->t_end; i += a->n_threads) {
        const double *wp = a->wps + (size_t)(i % a->n) * (size_t)(m + 1) * 3;
        if (oracle_solve(wp, m, a->velocity, coeffs, times) != 0) break;
        const int64_t rows = oracle_row_count(times, m, a->dt);
        if (rows > cap) { free(traj); cap = rows + rows / 4; traj = (double *)malloc(sizeof(double) * 11 * (size_t)cap); }
        oracle_sample(coeffs, times, m, a->dt, traj);
        double state[26] = {0};
        int32_t istate[4] = {0, 0, 0, 0};
        state[0] = traj[0]; state[1] = traj[1]; state[2] = traj[2]; state[3] = 1.0;
        const double hover = sqrt(a->V->mass * a->V->g / (4.0 * a->V->kf));
        for (int r = 13; r < 21; ++r) state[r] = hover;
        oracle_rollout(a->V, traj, rows, state, istate, a->ticks, slog, NULL, NULL, 0);
        ++a->done;
    }
    free(coeffs); free(times); free(slog); free(traj);
    return NULL;
}

int64_t oracle_bench_threads(const oracle_vehicle *V, const double *wps, int n, int m, double velocity, double dt,
                             int ticks, int n_threads, double budget_s, double *elapsed_s) {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    bench_arg *args = (bench_arg *)malloc(sizeof(bench_arg) * (size_t)n_threads);
    const double t0 = now_s();
    for (int t = 0; t < n_threads; ++t) {
        bench_arg a = {V, wps, n, m, ticks, t, n_threads, velocity, dt, t0 + budget_s, 0};
        args[t] = a;
        pthread_create(&th[t], NULL, bench_worker, &args[t]);
    }
    int64_t done = 0;
    for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], NULL); done += args[t].done; }
    *elapsed_s = now_s() - t0;
    free(th); free(args);
    return done;
}
