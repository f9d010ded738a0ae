// C ABI of libuavac.so: context management, argument validation, device-pointer entry points and
// their host-pointer twins (which stage through device scratch).  See include/uavac.h.

#include "uavac_internal.h"

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() { return static_cast<T *>(p); }
};

bool finite_all(const double *p, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(p[i])) return false;
    return true;
}

int check_plan_args(uavac_ctx *ctx, const void *wp, int B, int m) {
    if (!ctx) return UAVAC_EINVAL;
    if (!wp) return uavac_fail(ctx, UAVAC_EINVAL, "null waypoint pointer");
    if (B < 1) return uavac_fail(ctx, UAVAC_EINVAL, "B must be >= 1");
    if (m < 1 || m > UAVAC_MAX_SEGMENTS) return uavac_fail(ctx, UAVAC_EINVAL, "m must be in [1, UAVAC_MAX_SEGMENTS]");
    return UAVAC_OK;
}

int read_flags(uavac_ctx *ctx, int32_t out[4]) {
    UAVAC_HIP(ctx, hipMemcpyAsync(out, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int32_t), ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

}  // namespace

VehK uavac_make_vehk(const uavac_vehicle &V) {
    VehK k{};
    k.g = V.g; k.dt = V.dt; k.dt_outer = V.dt_outer; k.mass = V.mass; k.inv_mass = 1.0 / V.mass;
    for (int i = 0; i < 3; ++i) { k.I[i] = V.inertia[i]; k.inv_I[i] = 1.0 / V.inertia[i]; }
    k.arm = V.arm; k.inv_arm = 1.0 / V.arm; k.kappa = V.kappa; k.inv_kappa = 1.0 / V.kappa;
    k.kf = V.kf; k.inv_kf = 1.0 / V.kf;
    k.min_thrust = V.min_thrust; k.max_thrust = V.max_thrust;
    k.c_min = 4.0 * V.min_thrust; k.c_max = 4.0 * V.max_thrust;
    k.resp_rise = 1.0 - std::exp(-V.dt / V.tau_rise);
    k.resp_fall = 1.0 - std::exp(-V.dt / V.tau_fall);
    k.max_ascent = V.max_ascent; k.max_descent = V.max_descent; k.max_speed_xy = V.max_speed_xy;
    k.max_horiz_accel = V.max_horiz_accel; k.max_tilt = V.max_tilt;
    k.kp_xy = V.kp_xy; k.kd_xy = V.kd_xy; k.kp_z = V.kp_z; k.kd_z = V.kd_z; k.ki_z = V.ki_z;
    k.kp_roll = V.kp_roll; k.kp_pitch = V.kp_pitch; k.kp_yaw = V.kp_yaw;
    k.ikp[0] = V.inertia[0] * V.kp_p; k.ikp[1] = V.inertia[1] * V.kp_q; k.ikp[2] = V.inertia[2] * V.kp_r;
    k.hover_omega = std::sqrt(V.mass * V.g / (4.0 * V.kf));
    k.F = V.inner_per_outer;
    return k;
}

int uavac_check_vehicle(uavac_ctx *ctx, const uavac_vehicle *V) {
    if (!V) return uavac_fail(ctx, UAVAC_EINVAL, "null vehicle");
    const double pos[] = {V->g, V->dt, V->dt_outer, V->mass, V->inertia[0], V->inertia[1], V->inertia[2], V->arm,
                          V->kf, V->kappa, V->max_thrust, V->tau_rise, V->tau_fall};
    for (double v : pos)
        if (!(v > 0.0) || !std::isfinite(v)) return uavac_fail(ctx, UAVAC_EINVAL, "vehicle constant must be finite and > 0");
    if (!(V->min_thrust >= 0.0) || !(V->max_thrust > V->min_thrust))
        return uavac_fail(ctx, UAVAC_EINVAL, "thrust limits must satisfy 0 <= min < max");
    if (V->inner_per_outer < 1) return uavac_fail(ctx, UAVAC_EINVAL, "inner_per_outer must be >= 1");
    return UAVAC_OK;
}

extern "C" {

int uavac_version(void) { return UAVAC_VERSION; }

int uavac_create(uavac_ctx **out, int device_id) {
    if (!out) return UAVAC_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return UAVAC_EHIP;   // no CPU fallback, ever
    uavac_ctx *ctx = new (std::nothrow) uavac_ctx();
    if (!ctx) return UAVAC_ENOMEM;
    if (device_id >= 0) {
        if (device_id >= ndev || hipSetDevice(device_id) != hipSuccess) { delete ctx; return UAVAC_EHIP; }
        ctx->device = device_id;
    } else if (hipGetDevice(&ctx->device) != hipSuccess) { delete ctx; return UAVAC_EHIP; }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return UAVAC_EHIP; }
    ctx->stream = ctx->own_stream;
    if (hipMalloc(&ctx->d_flags, 4 * sizeof(int32_t)) != hipSuccess ||
        hipMemset(ctx->d_flags, 0, 4 * sizeof(int32_t)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return UAVAC_EHIP;
    }
    *out = ctx;
    return UAVAC_OK;
}

void uavac_destroy(uavac_ctx *ctx) {
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->d_totals) (void)hipFree(ctx->d_totals);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *uavac_last_error(const uavac_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int uavac_set_stream(uavac_ctx *ctx, void *hip_stream) {
    if (!ctx) return UAVAC_EINVAL;
    ctx->stream = static_cast<hipStream_t>(hip_stream);      // NULL = HIP's legacy default stream
    return UAVAC_OK;
}

int uavac_reset_stream(uavac_ctx *ctx) {
    if (!ctx) return UAVAC_EINVAL;
    ctx->stream = ctx->own_stream;
    return UAVAC_OK;
}

int uavac_synchronize(uavac_ctx *ctx) {
    if (!ctx) return UAVAC_EINVAL;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

void uavac_vehicle_default(uavac_vehicle *V) {
    if (!V) return;
    std::memset(V, 0, sizeof(*V));
    V->g = 9.81; V->dt = 0.001; V->inner_per_outer = 10; V->dt_outer = V->dt * V->inner_per_outer;
    V->mass = 0.5; V->inertia[0] = 0.0023; V->inertia[1] = 0.0023; V->inertia[2] = 0.0046;
    V->arm = 0.120208; V->kf = 1.0; V->kappa = 0.016;
    V->min_thrust = 0.1; V->max_thrust = 4.5; V->tau_rise = 0.0125; V->tau_fall = 0.025;
    V->max_ascent = 3.0; V->max_descent = 2.0; V->max_speed_xy = 3.0; V->max_horiz_accel = 12.0; V->max_tilt = 0.7;
    // second_order_gains(tau, zeta) = (1/tau^2, 2 zeta/tau): quad.py:124-127 with the constants of :42-51
    V->kp_xy = 1.0 / (0.25 * 0.25); V->kd_xy = 2.0 * 0.875 / 0.25;
    V->kp_z = 1.0 / (0.2 * 0.2);    V->kd_z = 2.0 * 0.8 / 0.2;
    V->ki_z = 0.1;
    V->kp_roll = 1.0 / 0.07; V->kp_pitch = 1.0 / 0.07; V->kp_yaw = 1.0 / 0.25;
    V->kp_p = 1.0 / 0.008; V->kp_q = 1.0 / 0.008; V->kp_r = 1.0 / 0.09;
}

// ------------------------------------------------------------------------------- planning, device
int uavac_minsnap_row_counts_dev(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt,
                                 double *times, int32_t *seg_rows, int64_t *row_offsets) {
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!times || !seg_rows || !row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null output pointer");
    if (!std::isfinite(velocity) || !std::isfinite(dt)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite velocity or dt");
    if (!(velocity > 0.0) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity and dt must be > 0");
    return uavac_launch_row_counts(ctx, wp, B, m, velocity, dt, times, seg_rows, row_offsets);
}

int uavac_minsnap_solve_dev(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                            int32_t *status) {
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!times || !coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    return uavac_launch_solve_bt(ctx, wp, times, B, m, coeffs, status);
}

int uavac_minsnap_solve_banded_dev(uavac_ctx *ctx, const double *wp, const double *times, int B, int m,
                                   double *coeffs, int32_t *status) {
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!times || !coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    return uavac_launch_solve(ctx, wp, times, B, m, coeffs, status);
}

int uavac_minsnap_sample_dev(uavac_ctx *ctx, const double *coeffs, const double *times, const int32_t *seg_rows,
                             const int64_t *row_offsets, int B, int m, double dt, double *traj) {
    (void)times;
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (!seg_rows || !row_offsets || !traj) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj);
}

int uavac_minsnap_sample_yaw_dev(uavac_ctx *ctx, const double *coeffs, const double *times, const int32_t *seg_rows,
                                 const int64_t *row_offsets, int B, int m, double dt, double *traj, double *yaw) {
    (void)times;
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (!seg_rows || !row_offsets || !traj || !yaw) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, nullptr, nullptr, yaw);
}

int uavac_minsnap_sample_hits_dev(uavac_ctx *ctx, const double *coeffs, const double *times, const int32_t *seg_rows,
                                  const int64_t *row_offsets, int B, int m, double dt, double *traj,
                                  const double *aabb, int32_t *hit) {
    (void)times;
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (!seg_rows || !row_offsets || !traj || !aabb || !hit) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, aabb, hit);
}

// --------------------------------------------------------------------------------- planning, host
int uavac_minsnap_row_counts(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt,
                             double *times, int32_t *seg_rows, int64_t *row_offsets) {
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null row_offsets");
    const size_t nwp = (size_t)B * (m + 1) * 3, nseg = (size_t)B * m;
    if (!finite_all(wp, nwp) || !std::isfinite(velocity) || !std::isfinite(dt))
        return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite waypoint, velocity or dt");
    DevBuf dwp, dt_, dsr, dro;
    UAVAC_HIP(ctx, dwp.alloc(nwp * 8));
    UAVAC_HIP(ctx, dt_.alloc(nseg * 8));
    UAVAC_HIP(ctx, dsr.alloc(nseg * 4));
    UAVAC_HIP(ctx, dro.alloc(((size_t)B + 1) * 8));
    UAVAC_HIP(ctx, hipMemcpyAsync(dwp.p, wp, nwp * 8, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_minsnap_row_counts_dev(ctx, dwp.as<double>(), B, m, velocity, dt, dt_.as<double>(),
                                              dsr.as<int32_t>(), dro.as<int64_t>())) return rc;
    if (times) UAVAC_HIP(ctx, hipMemcpyAsync(times, dt_.p, nseg * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (seg_rows) UAVAC_HIP(ctx, hipMemcpyAsync(seg_rows, dsr.p, nseg * 4, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(row_offsets, dro.p, ((size_t)B + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    int32_t fl[4];
    if (int rc = read_flags(ctx, fl)) return rc;
    if (fl[0]) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite segment duration");
    return UAVAC_OK;
}

int uavac_minsnap_solve(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double *coeffs,
                        double *times) {
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null coeffs");
    const size_t nwp = (size_t)B * (m + 1) * 3, nseg = (size_t)B * m, nco = (size_t)B * 24 * m;
    if (!finite_all(wp, nwp) || !std::isfinite(velocity))
        return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite waypoint or velocity");
    if (!(velocity > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity must be > 0");
    DevBuf dwp, dt_, dsr, dro, dco;
    UAVAC_HIP(ctx, dwp.alloc(nwp * 8));
    UAVAC_HIP(ctx, dt_.alloc(nseg * 8));
    UAVAC_HIP(ctx, dsr.alloc(nseg * 4));
    UAVAC_HIP(ctx, dro.alloc(((size_t)B + 1) * 8));
    UAVAC_HIP(ctx, dco.alloc(nco * 8));
    UAVAC_HIP(ctx, hipMemcpyAsync(dwp.p, wp, nwp * 8, hipMemcpyHostToDevice, ctx->stream));
    // dt only shapes the row counts, which this entry point does not return
    if (int rc = uavac_launch_row_counts(ctx, dwp.as<double>(), B, m, velocity, 1.0, dt_.as<double>(),
                                         dsr.as<int32_t>(), dro.as<int64_t>())) return rc;
    if (int rc = uavac_launch_solve_bt(ctx, dwp.as<double>(), dt_.as<double>(), B, m, dco.as<double>(), nullptr)) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(coeffs, dco.p, nco * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (times) UAVAC_HIP(ctx, hipMemcpyAsync(times, dt_.p, nseg * 8, hipMemcpyDeviceToHost, ctx->stream));
    int32_t fl[4];
    if (int rc = read_flags(ctx, fl)) return rc;
    if (fl[0]) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite segment duration");
    if (fl[1]) return uavac_fail(ctx, UAVAC_ESINGULAR, "singular knot system (repeated waypoint?)");
    return UAVAC_OK;
}

int uavac_minsnap_sample(uavac_ctx *ctx, const double *coeffs, const double *times, int B, int m, double dt,
                         const int64_t *row_offsets, double *traj) {
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (!times || !row_offsets || !traj) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    const size_t nseg = (size_t)B * m, nco = (size_t)B * 24 * m;
    // rows per segment, as uavac_minsnap_row_counts computes them; must agree with the caller's offsets
    std::vector<int32_t> sr(nseg);
    for (int b = 0; b < B; ++b) {
        int64_t tot = 0;
        for (int s = 0; s < m; ++s) {
            double q = std::ceil(times[(size_t)b * m + s] / dt);
            int32_t r = (std::isfinite(q) && q > 0.0 && q < 2.0e9) ? (int32_t)q : 0;
            sr[(size_t)b * m + s] = r;
            tot += r;
        }
        if (row_offsets[b + 1] - row_offsets[b] != tot)
            return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets inconsistent with ceil(times/dt)");
    }
    const int64_t total = row_offsets[B] - row_offsets[0];
    if (row_offsets[0] != 0 || total < 0) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must start at 0");
    DevBuf dco, dsr, dro, dtr;
    UAVAC_HIP(ctx, dco.alloc(nco * 8));
    UAVAC_HIP(ctx, dsr.alloc(nseg * 4));
    UAVAC_HIP(ctx, dro.alloc(((size_t)B + 1) * 8));
    UAVAC_HIP(ctx, dtr.alloc((size_t)total * UAVAC_TRAJ_COLS * 8));
    UAVAC_HIP(ctx, hipMemcpyAsync(dco.p, coeffs, nco * 8, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(dsr.p, sr.data(), nseg * 4, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(dro.p, row_offsets, ((size_t)B + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_launch_sample(ctx, dco.as<double>(), dsr.as<int32_t>(), dro.as<int64_t>(), B, m, dt,
                                     dtr.as<double>())) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(traj, dtr.p, (size_t)total * UAVAC_TRAJ_COLS * 8, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

// -------------------------------------------------------------------------------- control, device
int uavac_state_init_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *positions, int B, int hover,
                         double *state, int32_t *istate) {
    if (!ctx) return UAVAC_EINVAL;
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || !state || !istate) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null state");
    return uavac_launch_state_init(ctx, uavac_make_vehk(*V), positions, B, hover, state, istate);
}

int uavac_control_rollout_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                              double *state, int32_t *istate, int B, int K, double *state_log, double *cmd_log,
                              const double *aabbs, int n_obs) {
    if (!ctx) return UAVAC_EINVAL;
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || K < 0 || !traj || !row_offsets || !state || !istate || n_obs < 0)
        return uavac_fail(ctx, UAVAC_EINVAL, "bad size or null pointer");
    if (K == 0) return UAVAC_OK;
    return uavac_launch_rollout(ctx, uavac_make_vehk(*V), traj, row_offsets, state, istate, B, K, state_log, cmd_log,
                                aabbs, n_obs);
}

int uavac_control_rollout_plan_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *coeffs, const int32_t *seg_rows,
                                   const int64_t *row_offsets, const double *yaw, int m, double dt, double *state,
                                   int32_t *istate, int B, int K, double *state_log, double *cmd_log, const double *aabbs,
                                   int n_obs) {
    if (!ctx) return UAVAC_EINVAL;
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (K < 0 || !seg_rows || !row_offsets || !yaw || !state || !istate || n_obs < 0)
        return uavac_fail(ctx, UAVAC_EINVAL, "bad size or null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    if (K == 0) return UAVAC_OK;
    PlanRef plan;
    plan.coeffs = coeffs; plan.seg_rows = seg_rows; plan.yaw = yaw; plan.dt = dt; plan.m = m;
    return uavac_launch_rollout(ctx, uavac_make_vehk(*V), nullptr, row_offsets, state, istate, B, K, state_log, cmd_log,
                                aabbs, n_obs, &plan);
}

int uavac_control_step_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                           double *state, int32_t *istate, int B) {
    return uavac_control_rollout_dev(ctx, V, traj, row_offsets, state, istate, B, 1, nullptr, nullptr, nullptr, 0);
}

// ---------------------------------------------------------------------------------- control, host
int uavac_state_init(uavac_ctx *ctx, const uavac_vehicle *V, const double *positions, int B, int hover,
                     double *state, int32_t *istate) {
    if (!ctx) return UAVAC_EINVAL;
    if (B < 1 || !state || !istate) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null state");
    if (positions && !finite_all(positions, (size_t)B * 3)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite position");
    DevBuf dp, ds, di;
    UAVAC_HIP(ctx, ds.alloc((size_t)B * UAVAC_STATE_ROWS * 8));
    UAVAC_HIP(ctx, di.alloc((size_t)B * UAVAC_ISTATE_ROWS * 4));
    if (positions) {
        UAVAC_HIP(ctx, dp.alloc((size_t)B * 24));
        UAVAC_HIP(ctx, hipMemcpyAsync(dp.p, positions, (size_t)B * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    if (int rc = uavac_state_init_dev(ctx, V, positions ? dp.as<double>() : nullptr, B, hover, ds.as<double>(),
                                      di.as<int32_t>())) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(state, ds.p, (size_t)B * UAVAC_STATE_ROWS * 8, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(istate, di.p, (size_t)B * UAVAC_ISTATE_ROWS * 4, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_control_rollout(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                          double *state, int32_t *istate, int B, int K, double *state_log, double *cmd_log,
                          const double *aabbs, int n_obs) {
    if (!ctx) return UAVAC_EINVAL;
    if (B < 1 || K < 0 || !traj || !row_offsets || !state || !istate || n_obs < 0)
        return uavac_fail(ctx, UAVAC_EINVAL, "bad size or null pointer");
    if (row_offsets[0] != 0) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must start at 0");
    for (int b = 0; b < B; ++b)
        if (row_offsets[b + 1] < row_offsets[b]) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must be non-decreasing");
    const size_t total = (size_t)row_offsets[B];
    if (!finite_all(state, (size_t)B * UAVAC_STATE_ROWS)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite state");
    DevBuf dtr, dro, ds, di, dsl, dcl, dab;
    UAVAC_HIP(ctx, dtr.alloc(total * UAVAC_TRAJ_COLS * 8));
    UAVAC_HIP(ctx, dro.alloc(((size_t)B + 1) * 8));
    UAVAC_HIP(ctx, ds.alloc((size_t)B * UAVAC_STATE_ROWS * 8));
    UAVAC_HIP(ctx, di.alloc((size_t)B * UAVAC_ISTATE_ROWS * 4));
    UAVAC_HIP(ctx, hipMemcpyAsync(dtr.p, traj, total * UAVAC_TRAJ_COLS * 8, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(dro.p, row_offsets, ((size_t)B + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(ds.p, state, (size_t)B * UAVAC_STATE_ROWS * 8, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(di.p, istate, (size_t)B * UAVAC_ISTATE_ROWS * 4, hipMemcpyHostToDevice, ctx->stream));
    if (state_log) UAVAC_HIP(ctx, dsl.alloc((size_t)K * 13 * B * 8));
    if (cmd_log) UAVAC_HIP(ctx, dcl.alloc((size_t)K * UAVAC_CMD_COLS * B * 8));
    if (aabbs && n_obs > 0) {
        UAVAC_HIP(ctx, dab.alloc((size_t)n_obs * 48));
        UAVAC_HIP(ctx, hipMemcpyAsync(dab.p, aabbs, (size_t)n_obs * 48, hipMemcpyHostToDevice, ctx->stream));
    }
    if (int rc = uavac_control_rollout_dev(ctx, V, dtr.as<double>(), dro.as<int64_t>(), ds.as<double>(),
                                           di.as<int32_t>(), B, K, state_log ? dsl.as<double>() : nullptr,
                                           cmd_log ? dcl.as<double>() : nullptr,
                                           (aabbs && n_obs > 0) ? dab.as<double>() : nullptr, n_obs)) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(state, ds.p, (size_t)B * UAVAC_STATE_ROWS * 8, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(istate, di.p, (size_t)B * UAVAC_ISTATE_ROWS * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (state_log) UAVAC_HIP(ctx, hipMemcpyAsync(state_log, dsl.p, (size_t)K * 13 * B * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (cmd_log) UAVAC_HIP(ctx, hipMemcpyAsync(cmd_log, dcl.p, (size_t)K * UAVAC_CMD_COLS * B * 8, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

}  // extern "C"
