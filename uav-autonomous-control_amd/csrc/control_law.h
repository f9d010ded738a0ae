// Device-side control law, actuation and free-body step shared by the fused rollout kernel and the
// per-function probe kernels (gfx950).  Every function names the upstream code it restates.
#pragma once

#include "uavac_internal.h"

// No implicit FMA contraction in the control law: every fused multiply-add is written as fma().  The
// arithmetic is then the same instruction sequence in every kernel that inlines these functions (fused
// rollout with or without logs, single tick, probes), which is what makes chunked / single-step / fused
// runs bit-identical (tests/test_gpu_control.py).
#pragma clang fp contract(off)

namespace uavac_dev {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kIntegralLimit = 10.0;     // CascadedController.INTEGRAL_ERROR_LIMIT, controller.py:10

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// A constant that lives in a vector register.  (Call it ONCE, ahead of the loop that uses the value: the asm ties its input
// to its output, so executed inside a loop it costs a v_mov_b64 per execution -- the literals of the per-tick path therefore
// travel in VehK (lit_*) and are laundered in the rollout's prologue, not where they are used.)  fp64 literals that are not inline constants cannot be encoded in a vector
// instruction: the compiler builds them in a scalar register pair (two s_mov_b32), and in the rollout's tick loop -- whose
// ~50 vehicle constants already overflow the scalar file, at the price of v_readlane / v_writelane spills inside the loop --
// it rebuilds them on every tick.  Laundered through an empty (non-volatile, hoistable) asm they are loaded once, ahead of the
// loop, and cost nothing per tick.  Same value, same arithmetic: results cannot change.
__device__ __forceinline__ double vk(double x) {
    asm("" : "+v"(x));
    return x;
}

// fma(a, b, k) with a register-resident constant k as the addend, as ONE three-address v_fma_f64: the two-address form
// (v_fmac_f64, destination = addend) would have to copy a constant that must survive first, and a v_mov_b64 is as
// expensive on this part as the FMA itself (tools/fp64_operand_probe.hip).
__device__ __forceinline__ double fmak(double a, double b, double k) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(k));
    return d;
}

// Python's float `%` for a positive divisor (controller.py:173,178): fmod, then + b when the remainder is negative.
// fmod is exact, so in the two ranges the yaw law lives in its value is known without the (long, iterative) fp64 fmod:
// |a| < b -> a itself; b <= a < 2b -> a - b, which is exact too (Sterbenz: b/2 <= a <= 2b).  Anything else takes fmod.
__device__ __forceinline__ double floored_mod(double a, double b) {
    double r;
    if (fabs(a) < b) r = a;
    else if (a >= b && a < 2.0 * b) r = a - b;
    else r = fmod(a, b);
    if (r != 0.0 && r < 0.0) r += b;
    return r;
}


// ---- reduced-cost fp64 primitives for the per-tick path ------------------------------------------
// v_rcp_f64 / v_rsq_f64 are ~2^-26 seeds; two Newton steps reach fp64 (<= 1-2 ulp).  No denormal /
// special-case scaling: arguments on the per-tick path are normal, finite and positive by construction.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    double e = fma(-(hx * y), y, 0.5);
    y = fma(y, e, y);
    e = fma(-(hx * y), y, 0.5);
    return fma(y, e, y);
}
// sqrt(x) for x >= 0: Goldschmidt step on the rsq seed + one residual correction.  x is floored at the
// smallest normal so that x == 0 needs no special case (returns 1.5e-154 instead of 0).
__device__ __forceinline__ double fast_sqrt(double x, double smallest_normal) {
    x = fmax(x, smallest_normal);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return g;
}

struct Rot { double r00, r01, r02, r10, r11, r12, r20, r21, r22; };

// Quad.quat_to_rot (quad.py:133-155): normalise, R = I + 2 S S + 2 q0 S (body -> world)
__device__ __forceinline__ Rot quat_to_rot(double q0, double q1, double q2, double q3) {
    const double inv_n = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    const double a0 = q0 * inv_n, a1 = q1 * inv_n, a2 = q2 * inv_n, a3 = q3 * inv_n;
    Rot R;
    R.r00 = 1.0 - 2.0 * (a2 * a2 + a3 * a3);
    R.r01 = 2.0 * (a1 * a2 - a0 * a3);
    R.r02 = 2.0 * (a1 * a3 + a0 * a2);
    R.r10 = 2.0 * (a1 * a2 + a0 * a3);
    R.r11 = 1.0 - 2.0 * (a1 * a1 + a3 * a3);
    R.r12 = 2.0 * (a2 * a3 - a0 * a1);
    R.r20 = 2.0 * (a1 * a3 - a0 * a2);
    R.r21 = 2.0 * (a2 * a3 + a0 * a1);
    R.r22 = 1.0 - 2.0 * (a1 * a1 + a2 * a2);
    return R;
}

// CascadedController.altitude (controller.py:26-56).  tz, tzd, tzdd = target z, z', z''.
__device__ __forceinline__ double altitude(const VehK &V, double tz, double tzd, double tzdd, double pz, double vz,
                                           double R22, double &integ) {
    const double zd_des = clampd(tzd, -V.max_ascent, V.max_descent);
    const double ez = tz - pz;
    const double ezd = zd_des - vz;
    integ = clampd(integ + ez * V.dt_outer, -kIntegralLimit, kIntegralLimit);     // updated before use
    double acc_z = V.kp_z * ez + V.ki_z * integ + V.kd_z * ezd + tzdd - V.g;
    acc_z = acc_z / R22;
    return clampd(-V.mass * acc_z, V.c_min, V.c_max);
}

// CascadedController.lateral (controller.py:58-97)
__device__ __forceinline__ void lateral(const VehK &V, double tx, double txd, double txdd, double ty, double tyd,
                                        double tydd, double px, double py, double vx, double vy, double thrust,
                                        double &bxc, double &byc) {
    double vdx = txd, vdy = tyd;
    const double vmag = sqrt(vdx * vdx + vdy * vdy);
    if (vmag > V.max_speed_xy) { const double sc = V.max_speed_xy / vmag; vdx *= sc; vdy *= sc; }
    double acx = V.kp_xy * (tx - px) + V.kd_xy * (vdx - vx) + txdd;
    double acy = V.kp_xy * (ty - py) + V.kd_xy * (vdy - vy) + tydd;
    const double amag = sqrt(acx * acx + acy * acy);
    if (amag > V.max_horiz_accel) { const double sc = V.max_horiz_accel / amag; acx *= sc; acy *= sc; }
    const double inv_accz = -V.mass / thrust;                      // 1 / (-c/m)
    bxc = clampd(acx * inv_accz, -V.max_tilt, V.max_tilt);
    byc = clampd(acy * inv_accz, -V.max_tilt, V.max_tilt);
}

// CascadedController.roll_pitch_controller (controller.py:132-154)
__device__ __forceinline__ void roll_pitch(const VehK &V, double bxc, double byc, const Rot &R, double &pc,
                                           double &qc) {
    const double bdx = V.kp_roll * (bxc - R.r02);
    const double bdy = V.kp_pitch * (byc - R.r12);
    const double inv = 1.0 / R.r22;
    pc = (R.r10 * bdx - R.r00 * bdy) * inv;
    qc = (R.r11 * bdx - R.r01 * bdy) * inv;
}

// CascadedController.yaw_controller (controller.py:156-168) given psi and the trig of phi, theta
__device__ __forceinline__ double yaw_rate(const VehK &V, double psi_des, double psi, double cos_theta,
                                           double sin_phi, double cos_phi, double q_cmd) {
    const double pd = floored_mod(psi_des, kTwoPi);
    const double yaw_err = floored_mod(pd - psi + kPi, kTwoPi) - kPi;
    return (V.kp_yaw * yaw_err * cos_theta - q_cmd * sin_phi) / cos_phi;
}

// Euler angles of the STORED (un-normalised) quaternion, quad.py:189-213, as psi and the trig the
// yaw controller needs: sin/cos(phi) and cos(theta) come straight from the atan2 / asin arguments.
__device__ __forceinline__ void euler_trig(double q0, double q1, double q2, double q3, double &psi,
                                           double &cos_theta, double &sin_phi, double &cos_phi) {
    const double sn = 2.0 * (q0 * q1 + q2 * q3), cn = 1.0 - 2.0 * (q1 * q1 + q2 * q2);
    const double h = sqrt(sn * sn + cn * cn);
    if (h > 0.0) { sin_phi = sn / h; cos_phi = cn / h; } else { sin_phi = 0.0; cos_phi = 1.0; }   // atan2(0,0) = 0
    const double st = clampd(2.0 * (q0 * q2 - q3 * q1), -1.0, 1.0);
    cos_theta = sqrt(fmax(1.0 - st * st, 0.0));
    psi = atan2(2.0 * (q0 * q3 + q1 * q2), 1.0 - 2.0 * (q2 * q2 + q3 * q3));
}

// CascadedController.body_rate_controller (controller.py:115-130): I kp (cmd - w) + w x (I w)
__device__ __forceinline__ void body_rate(const VehK &V, double pc, double qc, double rc, double wp, double wq,
                                          double wr, double &Mx, double &My, double &Mz) {
    const double Iwx = V.I[0] * wp, Iwy = V.I[1] * wq, Iwz = V.I[2] * wr;
    Mx = fma(V.ikp[0], pc - wp, fma(wq, Iwz, -(wr * Iwy)));
    My = fma(V.ikp[1], qc - wq, fma(wr, Iwx, -(wp * Iwz)));
    Mz = fma(V.ikp[2], rc - wr, fma(wp, Iwy, -(wq * Iwx)));
}

// Quad._allocate_rotor_forces (quad.py:105-122); rotor order FL, FR, RR, RL (quad.py:157-166).
// The part that depends on the collective thrust command alone -- col = clip(c, 4 min, 4 max) / 4 and the head-room to the
// rotor limits on either side -- changes only when an OUTER tick writes a new command: the rollout computes it there and
// keeps it across the inner ticks in between (same operations on the same values: same bits as computing it every tick).
struct Collective { double col, up, dn; };
__device__ __forceinline__ Collective collective_of(const VehK &V, double thrust) {
    Collective c;
    c.col = clampd(thrust, V.c_min, V.c_max) * 0.25;
    c.up = V.max_thrust - c.col;                                               // both >= 0
    c.dn = c.col - V.min_thrust;
    return c;
}
__device__ __forceinline__ void allocate(const VehK &V, const Collective &c, double Mx, double My, double Mz, double f[4]) {
    const double col = c.col;
    const double pb = Mx * V.inv_arm, qb = My * V.inv_arm, rb = -Mz * V.inv_kappa;
    double mf[4];
    mf[0] = (pb + qb + rb) * 0.25;
    mf[1] = (-pb + qb - rb) * 0.25;
    mf[2] = (-pb - qb + rb) * 0.25;
    mf[3] = (pb - qb - rb) * 0.25;
    // moment_scale = clip(min_i limit_i, 0, 1), limit_i = (max-col)/mf_i for mf_i > 0, (min-col)/mf_i for
    // mf_i < 0, 1 for mf_i == 0.  Both numerators are rotor independent, so the smallest limit on each side
    // belongs to the largest |mf_i| of that sign; the two candidates are compared by cross-multiplication
    // and only the winner is divided.  (The mf_i sum to zero: either both signs occur or all are zero.)
    const double up = c.up, dn = c.dn;
    const double mpos = fmax(fmax(mf[0], mf[1]), fmax(mf[2], mf[3]));          // >= 0
    const double mneg = -fmin(fmin(mf[0], mf[1]), fmin(mf[2], mf[3]));         // >= 0
    const bool pos_wins = up * mneg < dn * mpos;                               // up/mpos < dn/mneg
    const double na = pos_wins ? up : dn, nb = pos_wins ? mpos : mneg;
    const double sc = (na < nb) ? na * fast_rcp(nb) : 1.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = clampd(fma(sc, mf[i], col), V.min_thrust, V.max_thrust);
}
__device__ __forceinline__ void allocate(const VehK &V, double thrust, double Mx, double My, double Mz,
                                         double f[4]) {
    allocate(V, collective_of(V, thrust), Mx, My, Mz, f);
}

// Quad.set_propeller_speed (quad.py:88-103): omega_cmd = sqrt(f/kf), first-order lag (rise / fall)
__device__ __forceinline__ void motors(const VehK &V, const double f[4], double om[4], double omc[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        omc[i] = fast_sqrt(f[i] * V.inv_kf, V.lit_tiny);
        const double resp = (omc[i] > om[i]) ? V.resp_rise : V.resp_fall;
        om[i] = fma(resp, omc[i] - om[i], om[i]);
    }
}

// Rotor wrench (mujoco_sim.py:232-251 in FRD) + semi-implicit Euler free-body step (MuJoCo mj_step,
// Euler integrator, free joint):
//   v' = g e3 - (T/m) R e3 ; w' = I^-1 (tau - w x I w) ; v += dt v' ; w += dt w' ; p += dt v_new ;
//   q <- normalise(q (x) exp(dt w_new))
// inv_n2 = 1/|q|^2 of the incoming attitude (exactly 1 when this function produced it).
//
// GROUND (uavac_vehicle.ground): a horizontal plane at z = V.ground_z the body rests on and takes off from
// (lab_course.xml:34,98-101; the reference starts on the ground with stopped rotors, test_mujoco_sim.py:150-174).  MuJoCo
// resolves that contact with its soft-constraint solver, which cannot run here; this is a BUILD-DEFINED stand-in of the
// same character: while the body's lowest point is below the plane (r = pz - ground_zc > 0), the vertical velocity update
// may not exceed the critically damped reference  vz + dt (-b vz - k r)  (b = 2/tc, k = 1/tc^2, tc = MuJoCo's default
// solref time constant) -- the plane only pushes, there is no friction and no contact torque.  Out of contact the
// arithmetic is the free-flight one, bit for bit.
template <bool GROUND = false>
__device__ __forceinline__ void free_body_step(const VehK &V, const double om[4], double &px, double &py,
                                               double &pz, double &q0, double &q1, double &q2, double &q3,
                                               double &vx, double &vy, double &vz, double &wp, double &wq,
                                               double &wr, double inv_n2) {
    const double f0 = V.kf * om[0] * om[0], f1 = V.kf * om[1] * om[1];
    const double f2 = V.kf * om[2] * om[2], f3 = V.kf * om[3] * om[3];
    const double T = f0 + f1 + f2 + f3;
    const double tx = V.arm * (f0 + f3 - f1 - f2);
    const double ty = V.arm * (f0 + f1 - f2 - f3);
    const double tz = V.kappa * (-f0 + f1 - f2 + f3);
    {
        const double s2 = 2.0 * inv_n2;                               // R e3 of the normalised attitude
        const double bzx = fma(q1, q3, q0 * q2) * s2;
        const double bzy = fma(q2, q3, -(q0 * q1)) * s2;
        const double bzz = fma(-s2, fma(q1, q1, q2 * q2), 1.0);
        const double tm = T * V.inv_mass * V.dt;
        vx = fma(-tm, bzx, vx);
        vy = fma(-tm, bzy, vy);
        const double vz_free = fma(V.dt, V.g, fma(-tm, bzz, vz));
        if (GROUND) {
            const double r = pz - V.ground_zc;
            const double vz_ref = fma(V.dt, -fma(V.ground_b, vz, V.ground_k * r), vz);
            vz = (r > 0.0 && vz_ref < vz_free) ? vz_ref : vz_free;
        } else {
            vz = vz_free;
        }
    }
    {
        const double Jx = V.I[0] * wp, Jy = V.I[1] * wq, Jz = V.I[2] * wr;
        const double cx = fma(wq, Jz, -(wr * Jy)), cy = fma(wr, Jx, -(wp * Jz)), cz = fma(wp, Jy, -(wq * Jx));
        wp = fma(V.dt, (tx - cx) * V.inv_I[0], wp);
        wq = fma(V.dt, (ty - cy) * V.inv_I[1], wq);
        wr = fma(V.dt, (tz - cz) * V.inv_I[2], wr);
    }
    px = fma(V.dt, vx, px); py = fma(V.dt, vy, py); pz = fma(V.dt, vz, pz);
    // dq = [cos h, sin(h) w/|w|], h = |w| dt / 2
    const double w2 = fma(wp, wp, fma(wq, wq, wr * wr));
    const double h2 = 0.25 * V.dt * V.dt * w2;
    double ch, sh_over;                           // cos(h), sin(h)/|w| = (dt/2) sinc(h)
    if (h2 < V.lit_h2_small) {
        // |h| < 0.0316 (|w| < 63 rad/s at dt = 1 ms): Taylor series through h^8, truncation < 3e-22
        ch = fma(h2, fma(h2, fmak(h2, fmak(h2, V.lit_c8, V.lit_c6), V.lit_c4), -0.5), 1.0);
        const double sinc = fma(h2, fmak(h2, fmak(h2, fmak(h2, V.lit_s9, V.lit_s7), V.lit_s5), V.lit_s3), 1.0);
        sh_over = 0.5 * V.dt * sinc;
    } else {
        const double wn = sqrt(w2), h = 0.5 * V.dt * wn;
        ch = cos(h);
        sh_over = sin(h) / wn;
    }
    const double d1 = sh_over * wp, d2 = sh_over * wq, d3 = sh_over * wr;
    const double n0 = fma(q0, ch, -fma(q1, d1, fma(q2, d2, q3 * d3)));
    const double n1 = fma(q0, d1, fma(q1, ch, fma(q2, d3, -(q3 * d2))));
    const double n2 = fma(q0, d2, fma(q2, ch, fma(q3, d1, -(q1 * d3))));
    const double n3 = fma(q0, d3, fma(q3, ch, fma(q1, d2, -(q2 * d1))));
    // |q (x) dq|^2 = |q|^2 (dq is unit to rounding), so e = n - 1 is ~1e-16 for a unit q and
    // 1/sqrt(1+e) = 1 - e/2 + 3e^2/8 is exact to rounding; a non-unit q (first tick of a caller-supplied
    // state) takes the general path.
    const double e = fma(n0, n0, fma(n1, n1, fma(n2, n2, n3 * n3))) - 1.0;
    double inv = fma(e, fma(e, V.lit_375, -0.5), 1.0);
    if (fabs(e) > V.lit_e_small) inv = fast_rsqrt(e + 1.0);
    q0 = n0 * inv; q1 = n1 * inv; q2 = n2 * inv; q3 = n3 * inv;
}

// Ground bookkeeping after a tick (MujocoSimulation._record_collisions, mujoco_sim.py:220-230): take-off is reached at
// TAKEOFF_HEIGHT above the plane; touching the plane before that is the start, touching it afterwards is a collision.
__device__ __forceinline__ int ground_bits(const VehK &V, double pz, int bits) {
    if (V.ground_z - pz >= UAVAC_TAKEOFF_HEIGHT) bits |= UAVAC_GROUND_TAKEN_OFF;
    const bool touching = pz - V.ground_zc > 0.0;
    bits = touching ? (bits | UAVAC_GROUND_IN_CONTACT) : (bits & ~UAVAC_GROUND_IN_CONTACT);
    if (touching && (bits & UAVAC_GROUND_TAKEN_OFF)) bits |= UAVAC_GROUND_HIT_AFTER_TAKEOFF;
    return bits;
}

}  // namespace uavac_dev
