// Minimum-snap sampler + yaw scan (gfx950): coefficients -> rows [p(3) v(3) a(3) yaw spline_id].
//
// Replaces uav_ac/planning/minimum_snap.py (upstream paths):
//   _generate_trajectory sampling loop   :100-119  (t = k*dt for k < ceil(T/dt); polynom(8,k,t) @ coeffs)
//   _calculate_yaws                      :126-136  (atan2 on samples with |v_xy| >= 1e-3, np.unwrap over the
//                                                   valid subset, hold last valid, back-fill leading rows)
//   np.hstack row assembly               :122-123
//
// One 256-thread workgroup per mission walks its rows in chunks of 256.  The kernel is a pure
// HBM write stream (88 B per row): rows are evaluated one per lane (Horner, coefficients broadcast
// from LDS), the yaw hold/unwrap is a wave-ballot + shuffle scan carried across chunks, and each
// chunk is staged in LDS so that the row-major (N,11) output leaves as contiguous 16-byte stores.

#include "uavac_internal.h"

namespace {

constexpr int SB = 256;                 // rows per chunk == threads per workgroup
constexpr int NW = SB / 64;
constexpr double kMinSpeedForYaw = 1e-3;   // MinimumSnap.MIN_HORIZONTAL_SPEED_FOR_YAW, minimum_snap.py:11
constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kTwoPi = 2.0 * kPi;

// floored modulo of NumPy's float `%` for a positive divisor
__device__ __forceinline__ double floored_mod(double a, double b) {
    double r = fmod(a, b);
    if (r != 0.0) { if (r < 0.0) r += b; } else { r = 0.0; }
    return r;
}

// np.unwrap's per-step correction for dd = p[i] - p[i-1]
__device__ __forceinline__ double unwrap_correction(double dd) {
    double ddmod = floored_mod(dd + kPi, kTwoPi) - kPi;
    if (ddmod == -kPi && dd > 0.0) ddmod = kPi;
    double corr = ddmod - dd;
    if (fabs(dd) < kPi) corr = 0.0;
    return corr;
}

__device__ __forceinline__ int segment_of(const int *__restrict__ pre, int m, int r, int s) {
    while (s + 1 < m && r >= pre[s + 1]) ++s;
    return s;
}

__global__ void __launch_bounds__(SB) minsnap_sample_kernel(const double *__restrict__ coeffs,
                                                           const int32_t *__restrict__ seg_rows,
                                                           const int64_t *__restrict__ row_offsets, int B, int m,
                                                           double dt, double *__restrict__ traj) {
    extern __shared__ double lds[];
    double *stage = lds;                         // [SB*11]
    double *cl = stage + SB * UAVAC_TRAJ_COLS;   // [24*m] coefficients of this mission
    int *pre = reinterpret_cast<int *>(cl + 24 * m);   // [m+1] exclusive prefix of seg_rows
    __shared__ double w_ang[NW], w_sum[NW];
    __shared__ int w_has[NW];
    __shared__ double s_first_yaw;
    __shared__ int s_first_row;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    const int64_t row0 = row_offsets[b];
    const int N = (int)(row_offsets[b + 1] - row0);

    for (int i = tid; i < 24 * m; i += SB) cl[i] = coeffs[(size_t)b * 24 * m + i];
    if (tid == 0) {
        int acc = 0;
        for (int s = 0; s < m; ++s) { pre[s] = acc; acc += seg_rows[(size_t)b * m + s]; }
        pre[m] = acc;
        s_first_row = N;
        s_first_yaw = 0.0;
    }
    __syncthreads();

    // ---- pre-pass: first row whose horizontal speed is usable, and its heading (back-fill value) ----
    {
        int s = 0;
        for (int c0 = 0; c0 < N; c0 += SB) {
            int r = c0 + tid;
            bool valid = false;
            double vx = 0.0, vy = 0.0;
            if (r < N) {
                s = segment_of(pre, m, r, s);
                double t = (double)(r - pre[s]) * dt;
                const double *c = cl + s * 24;
                vx = 7.0 * c[21]; vy = 7.0 * c[22];
#pragma unroll
                for (int i = 6; i >= 1; --i) { vx = vx * t + (double)i * c[3 * i]; vy = vy * t + (double)i * c[3 * i + 1]; }
                valid = sqrt(vx * vx + vy * vy) >= kMinSpeedForYaw;
            }
            if (valid) atomicMin(&s_first_row, r);
            __syncthreads();
            int fr = s_first_row;
            __syncthreads();
            if (fr < N) {
                if (r == fr) s_first_yaw = atan2(vy, vx);
                break;
            }
        }
        __syncthreads();
    }
    const int first_row = s_first_row;
    const double first_yaw = s_first_yaw;

    // ---- main pass ---------------------------------------------------------------------------------
    bool carry_has = false;
    double carry_ang = 0.0, carry_sum = 0.0;
    int s = 0;
    for (int c0 = 0; c0 < N; c0 += SB) {
        const int r = c0 + tid;
        const bool active = r < N;
        double px = 0, py = 0, pz = 0, vx = 0, vy = 0, vz = 0, ax = 0, ay = 0, az = 0;
        if (active) {
            s = segment_of(pre, m, r, s);
            const double t = (double)(r - pre[s]) * dt;
            const double *c = cl + s * 24;
            px = c[21]; py = c[22]; pz = c[23];
            vx = 7.0 * c[21]; vy = 7.0 * c[22]; vz = 7.0 * c[23];
            ax = 42.0 * c[21]; ay = 42.0 * c[22]; az = 42.0 * c[23];
#pragma unroll
            for (int i = 6; i >= 0; --i) {
                const double c0x = c[3 * i], c0y = c[3 * i + 1], c0z = c[3 * i + 2];
                px = px * t + c0x; py = py * t + c0y; pz = pz * t + c0z;
                if (i >= 1) {
                    const double f = (double)i;
                    vx = vx * t + f * c0x; vy = vy * t + f * c0y; vz = vz * t + f * c0z;
                }
                if (i >= 2) {
                    const double f = (double)(i * (i - 1));
                    ax = ax * t + f * c0x; ay = ay * t + f * c0y; az = az * t + f * c0z;
                }
            }
        }
        // rows before first_row are never valid and first_row always is (decided once, in the pre-pass)
        const bool valid = active && (r == first_row || (r > first_row && sqrt(vx * vx + vy * vy) >= kMinSpeedForYaw));
        const double ang = valid ? atan2(vy, vx) : 0.0;

        // last valid heading strictly before this row: in-wave via ballot, then earlier waves, then the carry
        const unsigned long long mask = __ballot(valid);
        const unsigned long long lower = mask & ((1ull << lane) - 1ull);
        bool prev_has = lower != 0ull;
        double prev_ang = __shfl(ang, prev_has ? 63 - __clzll((long long)lower) : 0);
        const double wlast = __shfl(ang, mask ? 63 - __clzll((long long)mask) : 0);
        if (lane == 0) { w_has[wv] = mask != 0ull; w_ang[wv] = wlast; }
        __syncthreads();
        if (!prev_has) {
            for (int w = wv - 1; w >= 0 && !prev_has; --w)
                if (w_has[w]) { prev_has = true; prev_ang = w_ang[w]; }
            if (!prev_has && carry_has) { prev_has = true; prev_ang = carry_ang; }
        }
        double corr = (valid && prev_has) ? unwrap_correction(ang - prev_ang) : 0.0;
        // inclusive prefix sum of the corrections (np.cumsum) across the chunk, on top of the carry
        double incl = corr;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) w_sum[wv] = incl;
        __syncthreads();
        double base = carry_sum;
        for (int w = 0; w < wv; ++w) base += w_sum[w];
        const double cum = base + incl;
        double yaw;
        if (r < first_row) yaw = first_yaw;                  // also the all-invalid mission: zeros
        else yaw = (valid ? ang : prev_ang) + cum;

        // carries for the next chunk (uniform across the workgroup)
        double tot = carry_sum;
        bool any = false;
        double lastang = carry_ang;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            tot += w_sum[w];
            if (w_has[w]) { any = true; lastang = w_ang[w]; }
        }

        if (active) {
            double *o = stage + tid * UAVAC_TRAJ_COLS;
            o[0] = px; o[1] = py; o[2] = pz; o[3] = vx; o[4] = vy; o[5] = vz;
            o[6] = ax; o[7] = ay; o[8] = az; o[9] = yaw; o[10] = (double)s;
        }
        __syncthreads();
        carry_sum = tot;
        carry_has = carry_has || any;
        carry_ang = lastang;

        // coalesced write-out of the staged chunk: 16-byte stores from an even element index
        const int nrows = min(SB, N - c0);
        const int nel = nrows * UAVAC_TRAJ_COLS;
        double *dst = traj + (row0 + c0) * UAVAC_TRAJ_COLS;
        const int head = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
        if (head && tid == 0) dst[0] = stage[0];
        const int npairs = (nel - head) >> 1;
        for (int p = tid; p < npairs; p += SB) {
            double2 v;
            v.x = stage[head + 2 * p];
            v.y = stage[head + 2 * p + 1];
            *reinterpret_cast<double2 *>(dst + head + 2 * p) = v;
        }
        if (((nel - head) & 1) && tid == 64) dst[nel - 1] = stage[nel - 1];
        __syncthreads();
    }
}

}  // namespace

int uavac_launch_sample(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets,
                        int B, int m, double dt, double *traj) {
    size_t lds = sizeof(double) * ((size_t)SB * UAVAC_TRAJ_COLS + (size_t)24 * m) + sizeof(int) * (size_t)(m + 2);
    hipLaunchKernelGGL(minsnap_sample_kernel, dim3(B), dim3(SB), lds, ctx->stream, coeffs, seg_rows, row_offsets, B,
                       m, dt, traj);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
