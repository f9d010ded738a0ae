// The heading a mission's leading rows take (`first_yaw`), computed WITHOUT writing the mission's rows (gfx950).
//
// Replaces, for a caller that flies plan-fed and never reads the sampled rows, the one value the rollout needs from
// uav_ac/planning/minimum_snap.py _calculate_yaws (:126-136, upstream path): `valid = |v_xy| >= 1e-3`, the heading of the
// FIRST valid sample (np.arctan2 on it; the rows before it take that heading through searchsorted(...) - 1 clipped to 0), and
// 0 when no sample is valid (:128-129).  The rollout scans everything after that row itself (state rows 26-29).
//
// Until round 6 only the sampler produced it -- as a by-product of writing 88 bytes per row.  A rank of a multi-GPU job that
// ships its PLAN to the root (uavac_gather_plan_dev) and flies plan-fed wrote ~10 KB of rows per spline for that one double:
// the same rows the root samples again from the gathered plan.  This kernel walks a mission's rows 64 at a time from row 0 with
// the sampler's own arithmetic -- the segment of row r from the prefix sums of the row counts, t = (r - first row of the
// segment) * dt, minsnap_eval_axis on x and y, has_heading, heading -- and stops at the first item that holds a valid sample:
// nearly always the first (a mission starts at rest and reaches 1 mm/s within a few samples); a vertical climb walks on.  One
// wavefront per mission, no barrier, 192 bytes of coefficients read per segment touched, 8 bytes written per mission.
// Bit-identical to the sampler's first_yaw: same functions, same operand order (tests/test_gpu_round6.py).

#include "uavac_internal.h"
#include "minsnap_eval.h"
#include "minsnap_yaw.h"

namespace {

using namespace uavac_yaw;
constexpr int kWaves = 4;                                   // missions (wavefronts) per workgroup

__global__ void __launch_bounds__(64 * kWaves) minsnap_first_yaw_kernel(
    const double *__restrict__ coeffs, const int32_t *__restrict__ seg_rows, const int64_t *__restrict__ seg_offsets, int B, int m,
    double dt, double *__restrict__ first_yaw) {
    __shared__ int pre_all[kWaves][UAVAC_MAX_SEGMENTS + 2];     // per wave: exclusive prefix of the row counts, then the total
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * kWaves + w;
    if (b >= B) return;                                      // (no workgroup barrier below: a wave may leave on its own)
    long long sb = (long long)b * m;                         // first segment of the mission in the batch
    int mb = m;
    if (seg_offsets) {                                       // ragged batch: clamped like the sampler clamps it
        sb = seg_offsets[b];
        const long long n = seg_offsets[b + 1] - sb;
        mb = (int)(n < 1 ? 1 : (n > m ? m : n));
    }
    int *pre = pre_all[w];
    const int v = lane < mb ? seg_rows[sb + lane] : 0;       // lane s holds segment s (m <= 64)
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane < mb) pre[lane] = inc - v;
    if (lane == mb - 1) pre[mb] = inc;
    const int N = __builtin_amdgcn_readlane(inc, 63);        // rows of the mission (lanes past mb add nothing)
    lds_wave_fence();

    double out = 0.0;                                        // no row has a heading: zeros (minimum_snap.py:128-129)
    const double *cm = coeffs + (size_t)sb * 24;
    for (int r0 = 0; r0 < N; r0 += 64) {
        const int r = r0 + lane;
        const bool active = r < N;
        double vx = 0.0, vy = 0.0;
        if (active) {
            int s = 0;
            while (s + 1 < mb && r >= pre[s + 1]) ++s;       // first segment whose rows reach past r (the sampler's choice)
            const double t = (double)(r - pre[s]) * dt;
            const double *c = cm + s * 24;
            double p, a;
            minsnap_eval_axis<1>(c, 0, t, p, vx, a);
            minsnap_eval_axis<1>(c, 1, t, p, vy, a);
        }
        const bool valid = active && has_heading(vx, vy);
        const unsigned long long mask = __ballot(valid);
        if (mask != 0ull) {
            const double ang = valid ? heading(vy, vx) : 0.0;
            const int l = __builtin_ctzll(mask);
            const int lo = __builtin_amdgcn_readlane(__double2loint(ang), l), hi = __builtin_amdgcn_readlane(__double2hiint(ang), l);
            out = __hiloint2double(hi, lo);
            break;
        }
    }
    if (lane == 0) first_yaw[b] = out;
}

}  // namespace

int uavac_launch_first_yaw(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *seg_offsets, int B, int m,
                           double dt, double *first_yaw) {
    hipLaunchKernelGGL(minsnap_first_yaw_kernel, dim3((B + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, ctx->stream, coeffs,
                       seg_rows, seg_offsets, B, m, dt, first_yaw);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
