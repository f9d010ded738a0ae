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
// same bits for every input (the library's fix-ups for infinities, NaNs and zeros sit behind one class test).  What it saves is
// instruction issue, 82 -> 47 vector instructions per 64 samples: the compiler turns the library's Horner chain into
// v_fmac_f64 (accumulating INTO the coefficient register), which costs a v_mov_b64 of every loop-invariant coefficient per
// step; here every step is one three-source v_fma_f64, and the tests for infinities, NaNs and zeros are skipped by every
// ordinary operand pair.
// Checked bit for bit against atan2 on the device over 2^24 random and all special operand pairs (uavac_probe_heading,
// tests/test_gpu_planner.py).
__device__ __forceinline__ double horner_step(double t, double p, double c) {      // t * p + c, never v_fmac
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(t), "v"(p), "v"(c));
    return r;
}
// The 20 coefficients of P, highest power first (ROCm ocml atanred, f64).
constexpr int kHeadingCoefficients = 20;
__device__ constexpr double kHeadingPoly[kHeadingCoefficients] = {
    0x1.ba404b5e68a13p-17, -0x1.3e260bd3237f4p-13, 0x1.b2bb069efb384p-11, -0x1.7952daf56de9bp-9, 0x1.d6d43a595c56fp-8,
    -0x1.c6ea4a57d9582p-7, 0x1.67e295f08b19fp-6, -0x1.e9ae6fc27006ap-6, 0x1.2c15b5711927ap-5, -0x1.59976e82d3ff0p-5,
    0x1.82d5d6ef28734p-5, -0x1.ae5ce6a214619p-5, 0x1.e1bb48427b883p-5, -0x1.110e48b207f05p-4, 0x1.3b13657b87036p-4,
    -0x1.745d119378e4fp-4, 0x1.c71c717e1913cp-4, -0x1.2492492376b7dp-3, 0x1.99999999952ccp-3, -0x1.5555555555523p-2};

// Where the coefficients come from is the caller's choice: literals (40 registers that live as long as the loop around the
// call does), or registers filled from LDS just before (HeadingFromLds: the streaming sampler, whose occupancy those 40
// registers would cost).  `coef.begin()` is called before the division, `coef.ready()` before the first use.
struct HeadingLiterals {
    __device__ __forceinline__ void begin() const {}
    __device__ __forceinline__ void ready() const {}
    __device__ __forceinline__ double operator()(int k) const { return kHeadingPoly[k]; }
};

template <class Coef>
__device__ __forceinline__ double heading_with(double y, double x, Coef &coef) {
    const double ay = fabs(y), ax = fabs(x);
    coef.begin();
    const double u = fmax(ax, ay), v = fmin(ax, ay);
    const double q = v / u;
    const double t = q * q;
    coef.ready();
    double p = horner_step(t, coef(0), coef(1));
#pragma unroll
    for (int k = 2; k < kHeadingCoefficients; ++k) p = horner_step(t, p, coef(k));
    double a = fma(q, t * p, q);
    a = ay > ax ? 0x1.921fb54442d18p+0 - a : a;
    a = __double2hiint(x) < 0 ? 0x1.921fb54442d18p+1 - a : a;
    // the library's three fix-ups, in its order, for operands that are not both finite with one of them non-zero (|x| + |y|
    // positive and finite: class mask +denormal | +normal; a sum that overflows comes here too and leaves unchanged)
    if (!__builtin_amdgcn_class(ax + ay, 0x180)) {
        const bool xneg = __double2hiint(x) < 0;
        if (y == 0.0) a = xneg ? 0x1.921fb54442d18p+1 : 0.0;
        if (isinf(x) && isinf(y)) a = xneg ? 0x1.2d97c7f3321d2p+1 : 0x1.921fb54442d18p-1;
        if (isnan(x) || isnan(y)) a = __builtin_nan("");
    }
    return copysign(a, y);
}

__device__ __forceinline__ double heading(double y, double x) {
    HeadingLiterals lit;
    return heading_with(y, x, lit);
}

// (The loads' operands are output-only and the compiler does not know the data to be in flight: uav_ac/_buildcheck.py
// check_heading_prefetch() verifies in the disassembly of every sampler variant that nothing touches the destination registers
// between the ten loads and the wait -- the rule of the rollout's row prefetch.)
// The coefficients as ten 16-byte LDS reads into registers that live for the length of one call (lds = LDS byte address of a
// copy of kHeadingPoly, 16-byte aligned).  The reads are issued in begin() and waited for in ready(), the division in between.
struct HeadingFromLds {
    typedef double pair __attribute__((ext_vector_type(2)));
    unsigned lds;
    pair c[10];
    __device__ __forceinline__ void begin() {
        asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:16\n\tds_read_b128 %2, %10 offset:32\n\t"
                     "ds_read_b128 %3, %10 offset:48\n\tds_read_b128 %4, %10 offset:64\n\tds_read_b128 %5, %10 offset:80\n\t"
                     "ds_read_b128 %6, %10 offset:96\n\tds_read_b128 %7, %10 offset:112\n\tds_read_b128 %8, %10 offset:128\n\t"
                     "ds_read_b128 %9, %10 offset:144"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7]),
                       "=&v"(c[8]), "=&v"(c[9])
                     : "v"(lds) : "memory");
    }
    __device__ __forceinline__ void ready() {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]),
                                               "+v"(c[7]), "+v"(c[8]), "+v"(c[9]) :: "memory");
    }
    __device__ __forceinline__ double operator()(int k) const { return c[k >> 1][k & 1]; }
};

}  // namespace uavac_yaw
