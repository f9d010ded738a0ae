// One trajectory sample from the 24 coefficients of its segment: position, velocity, acceleration on the three
// axes at local time t (minimum_snap.py:100-119, polynom :257-286).  Shared by the sampler and by the rollout
// variant that evaluates rows itself, so that both produce the SAME bits: every multiply-add is an explicit fma.
//
// Horner with running derivatives: p, p' and p''/2 cost three FMAs per power and axis, no i * c_i products.
// c[(3 i + axis) * STRIDE] = coefficient of t^i (STRIDE 1: the mission's (8,3) block; 64: a [24][64] LDS tile).
#pragma once

// STRIDE 0: the PAIRED tile of the plan-fed rollout, [12][64][2] doubles -- coefficient k of lane l at ((k >> 1) * 64 + l) * 2 +
// (k & 1), c pointing at the lane's first pair: the layout in which twelve 16-byte LDS-DMA loads per lane deposit a segment's
// 24 coefficients (control_rollout.hip).  Same coefficients, same operations, same bits.
template <int STRIDE>
__device__ __forceinline__ constexpr int minsnap_coeff_index(int k) { return STRIDE == 0 ? (k >> 1) * 128 + (k & 1) : k * STRIDE; }

template <int STRIDE>
__device__ __forceinline__ void minsnap_eval_row(const double *c, double t, double &px, double &py, double &pz,
                                                 double &vx, double &vy, double &vz, double &ax, double &ay, double &az) {
    double d1x = 0, d1y = 0, d1z = 0, d2x = 0, d2y = 0, d2z = 0;
    px = c[minsnap_coeff_index<STRIDE>(21)]; py = c[minsnap_coeff_index<STRIDE>(22)]; pz = c[minsnap_coeff_index<STRIDE>(23)];
#pragma unroll
    for (int i = 6; i >= 0; --i) {
        d2x = fma(d2x, t, d1x); d2y = fma(d2y, t, d1y); d2z = fma(d2z, t, d1z);
        d1x = fma(d1x, t, px);  d1y = fma(d1y, t, py);  d1z = fma(d1z, t, pz);
        px = fma(px, t, c[minsnap_coeff_index<STRIDE>(3 * i)]); py = fma(py, t, c[minsnap_coeff_index<STRIDE>(3 * i + 1)]);
        pz = fma(pz, t, c[minsnap_coeff_index<STRIDE>(3 * i + 2)]);
    }
    vx = d1x; vy = d1y; vz = d1z;
    ax = 2.0 * d2x; ay = 2.0 * d2y; az = 2.0 * d2z;
}

// One axis of the same sample (axis a = 0, 1, 2): the identical fma sequence as minsnap_eval_row, hence the same bits; the
// sampler evaluates axis after axis and stages each at once, which keeps a third of the coefficients and sums in registers.
template <int STRIDE>
__device__ __forceinline__ void minsnap_eval_axis(const double *c, int a, double t, double &p, double &v, double &acc) {
    double d1 = 0, d2 = 0;
    p = c[minsnap_coeff_index<STRIDE>(21 + a)];
#pragma unroll
    for (int i = 6; i >= 0; --i) {
        d2 = fma(d2, t, d1);
        d1 = fma(d1, t, p);
        p = fma(p, t, c[minsnap_coeff_index<STRIDE>(3 * i + a)]);
    }
    v = d1;
    acc = 2.0 * d2;
}

// Third and fourth derivative of the same polynomials (polynom(8, 3, t) @ coeffs and polynom(8, 4, t) @ coeffs,
// minimum_snap.py:111-112): Horner with running derivatives carried two levels further, d_k = p^(k) / k!.
template <int STRIDE>
__device__ __forceinline__ void minsnap_eval_jerk_snap(const double *c, double t, double jerk[3], double snap[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double p = c[(21 + a) * STRIDE], d1 = 0, d2 = 0, d3 = 0, d4 = 0;
#pragma unroll
        for (int i = 6; i >= 0; --i) {
            d4 = fma(d4, t, d3);
            d3 = fma(d3, t, d2);
            d2 = fma(d2, t, d1);
            d1 = fma(d1, t, p);
            p = fma(p, t, c[(3 * i + a) * STRIDE]);
        }
        jerk[a] = 6.0 * d3;
        snap[a] = 24.0 * d4;
    }
}
