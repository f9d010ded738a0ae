// Device-side control law, actuation and free-body step shared by the fused rollout kernel and the
// per-function probe kernels (gfx950).  Every function names the upstream code it restates.
#pragma once

#include "uavac_internal.h"

namespace uavac_dev {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kIntegralLimit = 10.0;     // CascadedController.INTEGRAL_ERROR_LIMIT, controller.py:10

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// Python's float `%` for a positive divisor (controller.py:173,178)
__device__ __forceinline__ double floored_mod(double a, double b) {
    double r = fmod(a, b);
    if (r != 0.0 && r < 0.0) r += b;
    return r;
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
    Mx = V.ikp[0] * (pc - wp) + (wq * Iwz - wr * Iwy);
    My = V.ikp[1] * (qc - wq) + (wr * Iwx - wp * Iwz);
    Mz = V.ikp[2] * (rc - wr) + (wp * Iwy - wq * Iwx);
}

// Quad._allocate_rotor_forces (quad.py:105-122); rotor order FL, FR, RR, RL (quad.py:157-166)
__device__ __forceinline__ void allocate(const VehK &V, double thrust, double Mx, double My, double Mz,
                                         double f[4]) {
    const double col = clampd(thrust, V.c_min, V.c_max) * 0.25;
    const double pb = Mx * V.inv_arm, qb = My * V.inv_arm, rb = -Mz * V.inv_kappa;
    double mf[4];
    mf[0] = (pb + qb + rb) * 0.25;
    mf[1] = (-pb + qb - rb) * 0.25;
    mf[2] = (-pb - qb + rb) * 0.25;
    mf[3] = (pb - qb - rb) * 0.25;
    double lim = 1.0e300;
    const double up = V.max_thrust - col, dn = V.min_thrust - col;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double num = (mf[i] > 0.0) ? up : dn;
        const double l = (mf[i] != 0.0) ? num / mf[i] : 1.0;
        lim = fmin(lim, l);
    }
    const double sc = clampd(lim, 0.0, 1.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = clampd(col + sc * mf[i], V.min_thrust, V.max_thrust);
}

// Quad.set_propeller_speed (quad.py:88-103): omega_cmd = sqrt(f/kf), first-order lag (rise / fall)
__device__ __forceinline__ void motors(const VehK &V, const double f[4], double om[4], double omc[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        omc[i] = sqrt(f[i] * V.inv_kf);
        const double resp = (omc[i] > om[i]) ? V.resp_rise : V.resp_fall;
        om[i] += resp * (omc[i] - om[i]);
    }
}

// Rotor wrench (mujoco_sim.py:232-251 in FRD) + semi-implicit Euler free-body step (MuJoCo mj_step,
// Euler integrator, free joint):
//   v' = g e3 - (T/m) R e3 ; w' = I^-1 (tau - w x I w) ; v += dt v' ; w += dt w' ; p += dt v_new ;
//   q <- normalise(q (x) exp(dt w_new))
__device__ __forceinline__ void free_body_step(const VehK &V, const double om[4], double &px, double &py,
                                               double &pz, double &q0, double &q1, double &q2, double &q3,
                                               double &vx, double &vy, double &vz, double &wp, double &wq,
                                               double &wr) {
    const double f0 = V.kf * om[0] * om[0], f1 = V.kf * om[1] * om[1];
    const double f2 = V.kf * om[2] * om[2], f3 = V.kf * om[3] * om[3];
    const double T = f0 + f1 + f2 + f3;
    const double tx = V.arm * (f0 + f3 - f1 - f2);
    const double ty = V.arm * (f0 + f1 - f2 - f3);
    const double tz = V.kappa * (-f0 + f1 - f2 + f3);
    {
        const double inv_n2 = 1.0 / (q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);      // R e3 of the normalised q
        const double bzx = 2.0 * (q1 * q3 + q0 * q2) * inv_n2;
        const double bzy = 2.0 * (q2 * q3 - q0 * q1) * inv_n2;
        const double bzz = 1.0 - 2.0 * (q1 * q1 + q2 * q2) * inv_n2;
        const double tm = T * V.inv_mass;
        vx += V.dt * (-tm * bzx);
        vy += V.dt * (-tm * bzy);
        vz += V.dt * (V.g - tm * bzz);
    }
    {
        const double Jx = V.I[0] * wp, Jy = V.I[1] * wq, Jz = V.I[2] * wr;
        const double cx = wq * Jz - wr * Jy, cy = wr * Jx - wp * Jz, cz = wp * Jy - wq * Jx;
        wp += V.dt * ((tx - cx) * V.inv_I[0]);
        wq += V.dt * ((ty - cy) * V.inv_I[1]);
        wr += V.dt * ((tz - cz) * V.inv_I[2]);
    }
    px += V.dt * vx; py += V.dt * vy; pz += V.dt * vz;
    // dq = [cos h, sin(h) w/|w|], h = |w| dt / 2
    const double w2 = wp * wp + wq * wq + wr * wr;
    const double h2 = 0.25 * V.dt * V.dt * w2;
    double ch, sh_over;                           // cos(h), sin(h)/|w| = (dt/2) sinc(h)
    if (h2 < 0.0625) {
        // |h| < 0.25: Taylor series through h^14, truncation < 1e-19 relative
        ch = 1.0 + h2 * (-1.0 / 2 + h2 * (1.0 / 24 + h2 * (-1.0 / 720 + h2 * (1.0 / 40320 + h2 * (-1.0 / 3628800 +
             h2 * (1.0 / 479001600 + h2 * (-1.0 / 87178291200.0)))))));
        const double sinc = 1.0 + h2 * (-1.0 / 6 + h2 * (1.0 / 120 + h2 * (-1.0 / 5040 + h2 * (1.0 / 362880 +
             h2 * (-1.0 / 39916800 + h2 * (1.0 / 6227020800.0 + h2 * (-1.0 / 1307674368000.0)))))));
        sh_over = 0.5 * V.dt * sinc;
    } else {
        const double wn = sqrt(w2), h = 0.5 * V.dt * wn;
        ch = cos(h);
        sh_over = sin(h) / wn;
    }
    const double d1 = sh_over * wp, d2 = sh_over * wq, d3 = sh_over * wr;
    const double n0 = q0 * ch - q1 * d1 - q2 * d2 - q3 * d3;
    const double n1 = q0 * d1 + q1 * ch + q2 * d3 - q3 * d2;
    const double n2 = q0 * d2 - q1 * d3 + q2 * ch + q3 * d1;
    const double n3 = q0 * d3 + q1 * d2 - q2 * d1 + q3 * ch;
    const double inv = 1.0 / sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
    q0 = n0 * inv; q1 = n1 * inv; q2 = n2 * inv; q3 = n3 * inv;
}

}  // namespace uavac_dev
