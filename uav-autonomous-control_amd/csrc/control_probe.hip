// Per-function probes of the control law (gfx950): the same __device__ functions the fused rollout
// kernel inlines, exposed one stage at a time so that the reference's unit-level known answers
// (tests/unit/control/test_controller.py, tests/unit/quadrotor/test_quad.py upstream) and the
// single-UAV facade (uav_ac.control.controller.CascadedController, uav_ac.quadrotor.quad.Quad)
// run on the HIP path.  Records are array-of-structs; batches are small; not a hot path.

#include "control_law.h"
#include "minsnap_yaw.h"

#include <cstring>
#include <new>

namespace {

using namespace uavac_dev;

// in [B][41]: X(13) | R(9) | target(11) | integ | thrust_in | bxy_in(2) | euler_in(3: phi theta psi) | q_cmd_in
// out[B][21]: R(9) | phi theta psi | thrust | integ' | bxy(2) | pq(2) | pqr(3)
__global__ void probe_outer_kernel(const VehK V, const double *__restrict__ in, int B, int mask,
                                   double *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *r = in + (size_t)b * UAVAC_PROBE_OUTER_IN;
    double *o = out + (size_t)b * UAVAC_PROBE_OUTER_OUT;
    const double px = r[0], py = r[1], pz = r[2], q0 = r[3], q1 = r[4], q2 = r[5], q3 = r[6];
    const double vx = r[7], vy = r[8], vz = r[9];
    const double *tg = r + 22;
    Rot R = quat_to_rot(q0, q1, q2, q3);
    if (mask & UAVAC_PROBE_USE_R) {
        R.r00 = r[13]; R.r01 = r[14]; R.r02 = r[15]; R.r10 = r[16]; R.r11 = r[17]; R.r12 = r[18];
        R.r20 = r[19]; R.r21 = r[20]; R.r22 = r[21];
    }
    o[0] = R.r00; o[1] = R.r01; o[2] = R.r02; o[3] = R.r10; o[4] = R.r11; o[5] = R.r12; o[6] = R.r20; o[7] = R.r21;
    o[8] = R.r22;
    // Euler angles of the stored quaternion (quad.py:189-213)
    double phi = atan2(2.0 * (q0 * q1 + q2 * q3), 1.0 - 2.0 * (q1 * q1 + q2 * q2));
    double theta = asin(clampd(2.0 * (q0 * q2 - q3 * q1), -1.0, 1.0));
    double psi, cth, sphi, cphi;
    euler_trig(q0, q1, q2, q3, psi, cth, sphi, cphi);
    if (mask & UAVAC_PROBE_USE_EULER) {
        phi = r[37]; theta = r[38]; psi = r[39];
        cth = cos(theta); sphi = sin(phi); cphi = cos(phi);
    }
    o[9] = phi; o[10] = theta; o[11] = psi;
    double integ = r[33];
    double thrust = altitude(V, tg[2], tg[5], tg[8], pz, vz, R.r22, integ);
    o[12] = thrust; o[13] = integ;
    if (mask & UAVAC_PROBE_USE_THRUST) thrust = r[34];
    double bxc, byc;
    lateral(V, tg[0], tg[3], tg[6], tg[1], tg[4], tg[7], px, py, vx, vy, thrust, bxc, byc);
    o[14] = bxc; o[15] = byc;
    if (mask & UAVAC_PROBE_USE_BXY) { bxc = r[35]; byc = r[36]; }
    double pc, qc;
    roll_pitch(V, bxc, byc, R, pc, qc);
    o[16] = pc; o[17] = qc;
    const double qcmd = (mask & UAVAC_PROBE_USE_QCMD) ? r[40] : qc;
    o[18] = pc; o[19] = qc;
    o[20] = yaw_rate(V, tg[9], psi, cth, sphi, cphi, qcmd);
}

// in [B][24]: X(13) | pqr_cmd(3) | thrust | omega(4) | moment_in(3)
// out[B][15]: moment(3) | rotor forces(4) | omega_command(4) | omega'(4)
__global__ void probe_inner_kernel(const VehK V, const double *__restrict__ in, int B, int mask,
                                   double *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *r = in + (size_t)b * UAVAC_PROBE_INNER_IN;
    double *o = out + (size_t)b * UAVAC_PROBE_INNER_OUT;
    double Mx, My, Mz;
    body_rate(V, r[13], r[14], r[15], r[10], r[11], r[12], Mx, My, Mz);
    o[0] = Mx; o[1] = My; o[2] = Mz;
    if (mask & UAVAC_PROBE_USE_MOMENT) { Mx = r[21]; My = r[22]; Mz = r[23]; }
    double f[4], om[4] = {r[17], r[18], r[19], r[20]}, omc[4];
    allocate(V, r[16], Mx, My, Mz, f);
    motors(V, f, om, omc);
    for (int i = 0; i < 4; ++i) { o[3 + i] = f[i]; o[7 + i] = omc[i]; o[11 + i] = om[i]; }
}


// Controller half of a tick on SoA state (TrajectoryController.step, main.py:37-61): outer loop every
// F-th call, body-rate loop, allocation and motor lag.  Does NOT integrate the vehicle.
__global__ void controller_tick_kernel(const VehK V, const double *__restrict__ traj,
                                       const int64_t *__restrict__ row_offsets, double *__restrict__ state,
                                       int32_t *__restrict__ istate, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t sB = (size_t)B;
    double X[13], om[4], omc[4];
    for (int i = 0; i < 13; ++i) X[i] = state[i * sB + b];
    for (int i = 0; i < 4; ++i) { om[i] = state[(13 + i) * sB + b]; omc[i] = state[(17 + i) * sB + b]; }
    double integ = state[21 * sB + b], thrust = state[22 * sB + b];
    double pc = state[23 * sB + b], qc = state[24 * sB + b], rc = state[25 * sB + b];
    int idx = istate[b], inner = istate[sB + b];
    const int64_t off = row_offsets[b];
    const int nrows = (int)(row_offsets[b + 1] - off);
    if (inner % V.F == 0 && nrows > 0) {
        const double *tg = traj + (off + min(max(idx, 0), nrows - 1)) * UAVAC_TRAJ_COLS;
        const Rot R = quat_to_rot(X[3], X[4], X[5], X[6]);
        thrust = altitude(V, tg[2], tg[5], tg[8], X[2], X[9], R.r22, integ);
        double bxc, byc;
        lateral(V, tg[0], tg[3], tg[6], tg[1], tg[4], tg[7], X[0], X[1], X[7], X[8], thrust, bxc, byc);
        roll_pitch(V, bxc, byc, R, pc, qc);
        double psi, cth, sphi, cphi;
        euler_trig(X[3], X[4], X[5], X[6], psi, cth, sphi, cphi);
        rc = yaw_rate(V, tg[9], psi, cth, sphi, cphi, qc);
        idx = min(idx + 1, nrows - 1);
    }
    double Mx, My, Mz, f[4];
    body_rate(V, pc, qc, rc, X[10], X[11], X[12], Mx, My, Mz);
    allocate(V, thrust, Mx, My, Mz, f);
    motors(V, f, om, omc);
    for (int i = 0; i < 4; ++i) { state[(13 + i) * sB + b] = om[i]; state[(17 + i) * sB + b] = omc[i]; }
    state[21 * sB + b] = integ; state[22 * sB + b] = thrust;
    state[23 * sB + b] = pc; state[24 * sB + b] = qc; state[25 * sB + b] = rc;
    istate[b] = idx; istate[sB + b] = inner + 1;
}

// Vehicle half of a tick (MujocoSimulation.step in free flight, mujoco_sim.py:144-151): rotor wrench +
// semi-implicit Euler free-body step on X from the current rotor speeds, optional sticky AABB flag.
__global__ void dynamics_step_kernel(const VehK V, double *__restrict__ state, int32_t *__restrict__ istate, int B,
                                     const double *__restrict__ aabbs, int n_obs) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t sB = (size_t)B;
    double X[13], om[4];
    for (int i = 0; i < 13; ++i) X[i] = state[i * sB + b];
    for (int i = 0; i < 4; ++i) om[i] = state[(13 + i) * sB + b];
    const double qn2 = X[3] * X[3] + X[4] * X[4] + X[5] * X[5] + X[6] * X[6];
    const double inv_n2 = (fabs(qn2 - 1.0) < 1.0e-12) ? 1.0 : 1.0 / qn2;
    if (V.ground) free_body_step<true>(V, om, X[0], X[1], X[2], X[3], X[4], X[5], X[6], X[7], X[8], X[9], X[10], X[11], X[12], inv_n2);
    else free_body_step<false>(V, om, X[0], X[1], X[2], X[3], X[4], X[5], X[6], X[7], X[8], X[9], X[10], X[11], X[12], inv_n2);
    for (int i = 0; i < 13; ++i) state[i * sB + b] = X[i];
    if (V.ground && istate) istate[3 * sB + b] = ground_bits(V, X[2], istate[3 * sB + b]);
    if (aabbs && istate) {
        int hit = 0;
        for (int o = 0; o < n_obs; ++o) {
            const double *c = aabbs + 6 * o;
            hit |= (X[0] >= c[0] && X[0] <= c[1] && X[1] >= c[2] && X[1] <= c[3] && X[2] >= c[4] && X[2] <= c[5]);
        }
        if (hit) istate[2 * sB + b] = 1;
    }
}

template <class T> T *take(uavac_ctx *ctx, size_t count) { return static_cast<T *>(uavac_arena_take(ctx, count * sizeof(T))); }

template <class Kern>
int run_probe(uavac_ctx *ctx, const uavac_vehicle *V, Kern kern, const double *in, int nin, int B, int mask,
              double *out, int nout) {
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || !in || !out) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size((size_t)B * nin * 8) + uavac_arena_size((size_t)B * nout * 8))) return rc;
    double *din = take<double>(ctx, (size_t)B * nin), *dout = take<double>(ctx, (size_t)B * nout);
    if (int rc = uavac_h2d(ctx, din, in, (size_t)B * nin * 8)) return rc;
    hipLaunchKernelGGL(kern, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, uavac_make_vehk(*V), din, B, mask, dout);
    UAVAC_HIP(ctx, hipGetLastError());
    if (int rc = uavac_d2h(ctx, out, dout, (size_t)B * nout * 8)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

// the sampler's heading next to the device library's atan2 on the same operands (they must agree bit for bit)
__global__ void probe_heading_kernel(const double *__restrict__ y, const double *__restrict__ x, long long n,
                                     double *__restrict__ heading, double *__restrict__ library) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        heading[i] = uavac_yaw::heading(y[i], x[i]);
        library[i] = atan2(y[i], x[i]);
    }
}

// uavac_create's self-check of that agreement: 2^16 operand pairs from a counter-based generator (magnitudes over the whole
// exponent range, both signs) plus every pairing of the special values; counts the pairs whose bits differ.
__device__ __forceinline__ unsigned long long selfcheck_mix(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ void heading_selfcheck_kernel(int n_random, int32_t *__restrict__ mismatches) {
    const double special[12] = {0.0, -0.0, 1.0, -1.0, __builtin_inf(), -__builtin_inf(), __builtin_nan(""), 4.9406564584124654e-324,
                                -2.2250738585072014e-308, 1.7976931348623157e308, 1e-3, -3.0};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double y, x;
    if (i < n_random) {
        const unsigned long long a = selfcheck_mix(2ull * i), b = selfcheck_mix(2ull * i + 1);
        if (i & 1) {                      // any finite bit pattern
            y = __longlong_as_double((long long)(a & ~(0x7ffull << 52)) | (long long)((a >> 52) % 2047ull) << 52);
            x = __longlong_as_double((long long)(b & ~(0x7ffull << 52)) | (long long)((b >> 52) % 2047ull) << 52);
        } else {                          // velocities as the sampler sees them: a few m/s, either sign
            y = ((double)(a >> 11) * 0x1p-53 - 0.5) * 12.0;
            x = ((double)(b >> 11) * 0x1p-53 - 0.5) * 12.0;
        }
    } else if (i < n_random + 144) {
        y = special[(i - n_random) / 12];
        x = special[(i - n_random) % 12];
    } else {
        return;
    }
    const double mine = uavac_yaw::heading(y, x), lib = atan2(y, x);
    const bool same = __double_as_longlong(mine) == __double_as_longlong(lib) || (mine != mine && lib != lib);
    if (!same) atomicAdd(mismatches, 1);
}

// One wave that stamps shader cycles and real time, sleeps through `ticks_100mhz` of real time and stamps again (uavac_clock_probe_dev).
__global__ void __launch_bounds__(64) clock_probe_kernel(long long ticks_100mhz, long long *__restrict__ stamps) {
    const long long c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    long long r1 = r0;
    // (bounded: a sleep is ~2 k shader cycles, about a microsecond -- at most four naps per 100 MHz tick asked for, so that a
    // real-time counter that stalls ends the probe with a short window instead of hanging the queue)
    const long long max_naps = 4 * ticks_100mhz + 1024;
    for (long long nap = 0; r1 - r0 < ticks_100mhz && nap < max_naps; ++nap) {
        __builtin_amdgcn_s_sleep(32);                      // ~2 k cycles without an instruction issued
        r1 = (long long)__builtin_amdgcn_s_memrealtime();
    }
    const long long c1 = (long long)__builtin_amdgcn_s_memtime();
    r1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[0] = c0; stamps[1] = r0; stamps[2] = c1; stamps[3] = r1; }
}

}  // namespace

int uavac_heading_selfcheck(uavac_ctx *ctx, int *mismatches) {
    constexpr int kRandom = 1 << 16;
    int32_t *d = nullptr;
    if (hipMalloc(&d, sizeof(int32_t)) != hipSuccess) return UAVAC_EHIP;
    int32_t h = -1;
    bool ok = hipMemsetAsync(d, 0, sizeof(int32_t), ctx->stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(heading_selfcheck_kernel, dim3((kRandom + 144 + 255) / 256), dim3(256), 0, ctx->stream, kRandom, d);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&h, d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    (void)hipFree(d);
    if (!ok) return UAVAC_EHIP;
    *mismatches = h;
    return UAVAC_OK;
}

extern "C" {

int uavac_clock_probe_dev(uavac_ctx *ctx, int window_us, int64_t *stamps) {
    UAVAC_ENTER(ctx);
    if (!stamps || window_us < 1 || window_us > 1000000) return uavac_fail(ctx, UAVAC_EINVAL, "null stamps or window outside 1 us .. 1 s");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, (long long)window_us * 100, (long long *)stamps);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_probe_heading_dev(uavac_ctx *ctx, const double *y, const double *x, int64_t n, double *heading, double *library) {
    UAVAC_ENTER(ctx);
    if (n < 1 || !y || !x || !heading || !library) return uavac_fail(ctx, UAVAC_EINVAL, "bad n or null pointer");
    hipLaunchKernelGGL(probe_heading_kernel, dim3(1024), dim3(256), 0, ctx->stream, y, x, (long long)n, heading, library);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_probe_outer(uavac_ctx *ctx, const uavac_vehicle *V, const double *in, int B, int mask, double *out) {
    return run_probe(ctx, V, probe_outer_kernel, in, UAVAC_PROBE_OUTER_IN, B, mask, out, UAVAC_PROBE_OUTER_OUT);
}

int uavac_probe_inner(uavac_ctx *ctx, const uavac_vehicle *V, const double *in, int B, int mask, double *out) {
    return run_probe(ctx, V, probe_inner_kernel, in, UAVAC_PROBE_INNER_IN, B, mask, out, UAVAC_PROBE_INNER_OUT);
}

int uavac_controller_tick_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                              double *state, int32_t *istate, int B) {
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || !traj || !row_offsets || !state || !istate) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    hipLaunchKernelGGL(controller_tick_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, uavac_make_vehk(*V), traj,
                       row_offsets, state, istate, B);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_dynamics_step_dev(uavac_ctx *ctx, const uavac_vehicle *V, double *state, int32_t *istate, int B,
                            const double *aabbs, int n_obs) {
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || !state || n_obs < 0) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    hipLaunchKernelGGL(dynamics_step_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, uavac_make_vehk(*V), state,
                       istate, B, n_obs > 0 ? aabbs : nullptr, n_obs);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_controller_tick(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                          double *state, int32_t *istate, int B) {
    UAVAC_ENTER(ctx);
    if (B < 1 || !traj || !row_offsets || !state || !istate) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    if (row_offsets[0] != 0) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must start at 0");
    const size_t ntr = (size_t)row_offsets[B] * UAVAC_TRAJ_COLS, ns = (size_t)B * UAVAC_STATE_ROWS, ni = (size_t)B * UAVAC_ISTATE_ROWS;
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(ntr * 8) + uavac_arena_size(((size_t)B + 1) * 8) +
                                              uavac_arena_size(ns * 8) + uavac_arena_size(ni * 4))) return rc;
    double *dtr = take<double>(ctx, ntr);
    int64_t *dro = take<int64_t>(ctx, (size_t)B + 1);
    double *ds = take<double>(ctx, ns);
    int32_t *di = take<int32_t>(ctx, ni);
    if (int rc = uavac_h2d(ctx, dtr, traj, ntr * 8)) return rc;
    if (int rc = uavac_h2d(ctx, dro, row_offsets, ((size_t)B + 1) * 8)) return rc;
    if (int rc = uavac_h2d(ctx, ds, state, ns * 8)) return rc;
    if (int rc = uavac_h2d(ctx, di, istate, ni * 4)) return rc;
    if (int rc = uavac_controller_tick_dev(ctx, V, dtr, dro, ds, di, B)) return rc;
    if (int rc = uavac_d2h(ctx, state, ds, ns * 8)) return rc;
    if (int rc = uavac_d2h(ctx, istate, di, ni * 4)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_dynamics_step(uavac_ctx *ctx, const uavac_vehicle *V, double *state, int32_t *istate, int B,
                        const double *aabbs, int n_obs) {
    UAVAC_ENTER(ctx);
    if (B < 1 || !state || n_obs < 0 || !V) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    const bool obs = aabbs && n_obs > 0;
    const bool flag = istate && (obs || V->ground);          // istate carries the obstacle flag and the ground bits
    const size_t ns = (size_t)B * UAVAC_STATE_ROWS, ni = (size_t)B * UAVAC_ISTATE_ROWS;
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(ns * 8) + uavac_arena_size(ni * 4) +
                                              uavac_arena_size((size_t)n_obs * 48))) return rc;
    double *ds = take<double>(ctx, ns);
    int32_t *di = flag ? take<int32_t>(ctx, ni) : nullptr;
    double *dab = obs ? take<double>(ctx, (size_t)n_obs * 6) : nullptr;
    if (int rc = uavac_h2d(ctx, ds, state, ns * 8)) return rc;
    if (flag) if (int rc = uavac_h2d(ctx, di, istate, ni * 4)) return rc;
    if (obs) if (int rc = uavac_h2d(ctx, dab, aabbs, (size_t)n_obs * 48)) return rc;
    if (int rc = uavac_dynamics_step_dev(ctx, V, ds, di, B, dab, obs ? n_obs : 0)) return rc;
    if (int rc = uavac_d2h(ctx, state, ds, ns * 8)) return rc;
    if (flag) if (int rc = uavac_d2h(ctx, istate, di, ni * 4)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Resident tick-by-tick session for callers that own the loop the way uav_ac/main.py does (tc.step(); sim.step() per
// inner tick, with Python reading and writing quad.X / quad.omega in between).  The trajectory rows are uploaded ONCE;
// the (small) state lives in pinned, device-mapped host memory that the kernels read and write in place, so a tick is
// one kernel launch + one stream synchronisation -- no staging copies, no allocation.
struct uavac_pilot {
    uavac_ctx *ctx = nullptr;
    int B = 0;
    double *d_traj = nullptr;
    int64_t *d_offsets = nullptr;
    double *d_aabbs = nullptr;
    int n_obs = 0;
    double *h_state = nullptr;       // pinned + mapped: [UAVAC_STATE_ROWS][B]
    int32_t *h_istate = nullptr;     // pinned + mapped: [4][B]
};

int uavac_pilot_create(uavac_ctx *ctx, const double *traj, const int64_t *row_offsets, int B, uavac_pilot **out) {
    UAVAC_ENTER(ctx);
    if (!out) return uavac_fail(ctx, UAVAC_EINVAL, "null output pointer");
    *out = nullptr;
    if (B < 1 || !traj || !row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    if (row_offsets[0] != 0) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must start at 0");
    for (int b = 0; b < B; ++b)
        if (row_offsets[b + 1] < row_offsets[b]) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must be non-decreasing");
    uavac_pilot *p = new (std::nothrow) uavac_pilot();
    if (!p) return UAVAC_ENOMEM;
    p->ctx = ctx;
    p->B = B;
    const size_t ntr = (size_t)row_offsets[B] * UAVAC_TRAJ_COLS;
    auto fail = [&](int rc) { uavac_pilot_destroy(p); return rc; };
    if (hipMalloc(&p->d_traj, ntr * 8 + 8) != hipSuccess || hipMalloc(&p->d_offsets, ((size_t)B + 1) * 8) != hipSuccess ||
        hipHostMalloc(&p->h_state, (size_t)B * UAVAC_STATE_ROWS * 8, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc(&p->h_istate, (size_t)B * UAVAC_ISTATE_ROWS * 4, hipHostMallocMapped) != hipSuccess)
        return fail(uavac_fail(ctx, UAVAC_EHIP, "pilot allocation failed"));
    std::memset(p->h_state, 0, (size_t)B * UAVAC_STATE_ROWS * 8);
    std::memset(p->h_istate, 0, (size_t)B * UAVAC_ISTATE_ROWS * 4);
    if (int rc = uavac_h2d(ctx, p->d_traj, traj, ntr * 8)) return fail(rc);
    if (int rc = uavac_h2d(ctx, p->d_offsets, row_offsets, ((size_t)B + 1) * 8)) return fail(rc);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(uavac_fail(ctx, UAVAC_EHIP, "pilot upload failed"));
    *out = p;
    return UAVAC_OK;
}

void uavac_pilot_destroy(uavac_pilot *p) {
    if (!p) return;
    uavac_device_guard guard(p->ctx);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->d_traj) (void)hipFree(p->d_traj);
    if (p->d_offsets) (void)hipFree(p->d_offsets);
    if (p->d_aabbs) (void)hipFree(p->d_aabbs);
    if (p->h_state) (void)hipHostFree(p->h_state);
    if (p->h_istate) (void)hipHostFree(p->h_istate);
    delete p;
}

double *uavac_pilot_state(uavac_pilot *p) { return p ? p->h_state : nullptr; }
int32_t *uavac_pilot_istate(uavac_pilot *p) { return p ? p->h_istate : nullptr; }

int uavac_pilot_set_obstacles(uavac_pilot *p, const double *aabbs, int n_obs) {
    if (!p) return UAVAC_EINVAL;
    uavac_ctx *ctx = p->ctx;
    UAVAC_ENTER(ctx);
    if (n_obs < 0 || (n_obs > 0 && !aabbs)) return uavac_fail(ctx, UAVAC_EINVAL, "bad obstacle list");
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (p->d_aabbs) UAVAC_HIP(ctx, hipFree(p->d_aabbs));
    p->d_aabbs = nullptr;
    p->n_obs = 0;
    if (n_obs > 0) {
        UAVAC_HIP(ctx, hipMalloc(&p->d_aabbs, (size_t)n_obs * 48));
        if (int rc = uavac_h2d(ctx, p->d_aabbs, aabbs, (size_t)n_obs * 48)) return rc;
        UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        p->n_obs = n_obs;
    }
    return UAVAC_OK;
}

int uavac_pilot_tick(uavac_pilot *p, const uavac_vehicle *V, int what) {
    if (!p) return UAVAC_EINVAL;
    uavac_ctx *ctx = p->ctx;
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (!(what & (UAVAC_PILOT_CONTROLLER | UAVAC_PILOT_DYNAMICS))) return uavac_fail(ctx, UAVAC_EINVAL, "nothing to do");
    const VehK K = uavac_make_vehk(*V);
    const dim3 grid((p->B + 63) / 64), block(64);
    if (what & UAVAC_PILOT_CONTROLLER)
        hipLaunchKernelGGL(controller_tick_kernel, grid, block, 0, ctx->stream, K, p->d_traj, p->d_offsets, p->h_state,
                           p->h_istate, p->B);
    if (what & UAVAC_PILOT_DYNAMICS)
        hipLaunchKernelGGL(dynamics_step_kernel, grid, block, 0, ctx->stream, K, p->h_state, p->h_istate, p->B,
                           p->n_obs > 0 ? p->d_aabbs : nullptr, p->n_obs);
    UAVAC_HIP(ctx, hipGetLastError());
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the caller reads the pinned state next
    return UAVAC_OK;
}

}  // extern "C"
