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
// (contraction is switched off IN the function: hipcc contracts by default and HIP's __dmul_rn / __dadd_rn are plain operators,
// so without the pragma this compiled to v_mul_f64 + v_fmac_f64 in every translation unit that had not switched it off itself)
__device__ __forceinline__ bool has_heading(double vx, double vy) {
#pragma clang fp contract(off)
    const double xx = vx * vx, yy = vy * vy;
    return xx + yy >= kMinSpeedSquared;
}

// The heading atan2(vy, vx) of a sample, for the sampler's inner loop: the SAME operations in the same order as the device
// library's atan2 (ROCm ocml atan2 f64: q = min(|x|,|y|) / max(|x|,|y|) correctly rounded, a = q + q * (q^2 * P(q^2)) with P
// of degree 19 in Horner form, pi/2 - a when |y| > |x|, pi - a when x carries a sign bit, the sign of y copied on) -- hence the
// same bits for every finite input; lanes with an infinity, a NaN or two zeros take the library call itself.  What it saves is
// instruction issue, 82 -> 47 vector instructions per 64 samples: the compiler turns the library's Horner chain into
// v_fmac_f64 (accumulating INTO the coefficient register), which costs a v_mov_b64 of every loop-invariant coefficient per
// step; here every step is one three-source v_fma_f64, and the tests for infinities, NaNs and zeros are gone (zeros take the
// general path to the library's own results: q = 0, a = 0, then the same quadrant fix-ups).
// Checked bit for bit against atan2 on the device over 2^24 random and all special operand pairs (uavac_probe_heading,
// tests/test_gpu_planner.py).
__device__ __forceinline__ double horner_step(double t, double p, double c) {      // t * p + c, never v_fmac
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(t), "v"(p), "v"(c));
    return r;
}
__device__ __forceinline__ double heading(double y, double x) {
    const double ay = fabs(y), ax = fabs(x);
    // |x| + |y| positive and finite (class mask: +denormal | +normal) <=> both finite and not both zero; the others (and the
    // pairs whose sum overflows) take the library call, which is right for every operand
    if (!__builtin_amdgcn_class(ax + ay, 0x180)) return atan2(y, x);
    const double u = fmax(ax, ay), v = fmin(ax, ay);
    const double q = v / u;
    const double t = q * q;
    double p = horner_step(t, 0x1.ba404b5e68a13p-17, -0x1.3e260bd3237f4p-13);
    p = horner_step(t, p, 0x1.b2bb069efb384p-11);
    p = horner_step(t, p, -0x1.7952daf56de9bp-9);
    p = horner_step(t, p, 0x1.d6d43a595c56fp-8);
    p = horner_step(t, p, -0x1.c6ea4a57d9582p-7);
    p = horner_step(t, p, 0x1.67e295f08b19fp-6);
    p = horner_step(t, p, -0x1.e9ae6fc27006ap-6);
    p = horner_step(t, p, 0x1.2c15b5711927ap-5);
    p = horner_step(t, p, -0x1.59976e82d3ff0p-5);
    p = horner_step(t, p, 0x1.82d5d6ef28734p-5);
    p = horner_step(t, p, -0x1.ae5ce6a214619p-5);
    p = horner_step(t, p, 0x1.e1bb48427b883p-5);
    p = horner_step(t, p, -0x1.110e48b207f05p-4);
    p = horner_step(t, p, 0x1.3b13657b87036p-4);
    p = horner_step(t, p, -0x1.745d119378e4fp-4);
    p = horner_step(t, p, 0x1.c71c717e1913cp-4);
    p = horner_step(t, p, -0x1.2492492376b7dp-3);
    p = horner_step(t, p, 0x1.99999999952ccp-3);
    p = horner_step(t, p, -0x1.5555555555523p-2);
    double a = fma(q, t * p, q);
    a = ay > ax ? 0x1.921fb54442d18p+0 - a : a;
    a = __double2hiint(x) < 0 ? 0x1.921fb54442d18p+1 - a : a;
    return copysign(a, y);
}

}  // namespace uavac_yaw
