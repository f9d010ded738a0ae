// The per-row pieces of MinimumSnap._calculate_yaws (minimum_snap.py:126-136) shared by the sampler, which scans a mission's
// yaw 64 rows at a time, and by the rollout, which can scan it one row per outer tick instead of reading it (YAWSCAN):
// both must produce the same bits, so both take every operation from here.
#pragma once

namespace uavac_yaw {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kTwoPi = 2.0 * kPi;

// floored modulo of NumPy's float `%` for a positive divisor
__device__ __forceinline__ double floored_mod(double a, double b) {
    double r = fmod(a, b);
    if (r != 0.0) { if (r < 0.0) r += b; } else { r = 0.0; }
    return r;
}

// np.unwrap's per-step correction for dd = p[i] - p[i-1]:
//     ddmod = mod(dd + pi, 2 pi) - pi ; ddmod[(ddmod == -pi) & (dd > 0)] = pi ; corr = ddmod - dd ; corr[|dd| < pi] = 0
// Headings of consecutive samples rarely jump by pi or more, and NumPy discards the modulo's result whenever they do
// not: the (long) fp64 fmod runs only for the lanes that need it, i.e. for almost no wave.
__device__ __forceinline__ double unwrap_correction(double dd) {
    if (fabs(dd) < kPi) return 0.0;
    double ddmod = floored_mod(dd + kPi, kTwoPi) - kPi;
    if (ddmod == -kPi && dd > 0.0) ddmod = kPi;
    return ddmod - dd;
}

// |v_xy| >= MIN_HORIZONTAL_SPEED_FOR_YAW as NumPy evaluates it (np.linalg.norm(..., axis=1) = sqrt(add.reduce(x * x)):
// two rounded products, one rounded sum -- no fused multiply-add), without the square root: sqrt is correctly rounded and
// monotonic, so sqrt(s) >= 1e-3 holds exactly for s >= s*, s* the smallest double whose root rounds to >= 1e-3.  That is
// 0x1.0c6f7a0b5ed8dp-20 (= the double nearest 1e-6; its predecessor's root is below 1e-3 -- checked with exact
// rationals against ((1e-3 + pred(1e-3)) / 2)^2).  Infinities pass and NaNs fail either way.
constexpr double kMinSpeedSquared = 0x1.0c6f7a0b5ed8dp-20;
__device__ __forceinline__ bool has_heading(double vx, double vy) {
    return __dadd_rn(__dmul_rn(vx, vx), __dmul_rn(vy, vy)) >= kMinSpeedSquared;
}

}  // namespace uavac_yaw
