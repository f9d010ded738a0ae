// The heading a mission's leading rows take (`first_yaw`), computed WITHOUT writing the mission's rows (gfx950).
//
// Replaces, for a caller that flies plan-fed and never reads the sampled rows, the one value the rollout needs from
// uav_ac/planning/minimum_snap.py _calculate_yaws (:126-136, upstream path): `valid = |v_xy| >= 1e-3`, the heading of the
// FIRST valid sample (np.arctan2 on it; the rows before it take that heading through searchsorted(...) - 1 clipped to 0), and
// 0 when no sample is valid (:128-129).  The rollout scans everything after that row itself (state rows 26-29).
//
// Until round 6 only the sampler produced it -- as a by-product of writing 88 bytes per row.  A rank of a multi-GPU job that
// ships its PLAN to the root (uavac_gather_plan_dev) and flies plan-fed wrote ~10 KB of rows per spline for that one double:
// the same rows the root samples again from the gathered plan.  This kernel walks a mission's rows SIXTEEN at a time from row 0
// with the sampler's own arithmetic -- the segment of row r from the row counts, t = (r - first row of the segment) * dt,
// minsnap_eval_axis on x and y, has_heading, heading -- and stops at the first step that holds a valid sample: nearly always the
// first (a mission starts at rest and reaches 1 mm/s within a few samples); a vertical climb walks on.  Sixteen lanes per
// mission, four missions per wavefront, no LDS, no barrier; 192 bytes of coefficients read per segment touched, 8 bytes written
// per mission.  Bit-identical to the sampler's first_yaw: same functions, same operand order (tests/test_gpu_round6.py).

#include "uavac_internal.h"
#include "minsnap_eval.h"
#include "minsnap_yaw.h"

namespace {

using namespace uavac_yaw;
constexpr int kWaves = 4;                                   // wavefronts per workgroup
constexpr int kLanes = 16;                                  // lanes (= rows per step) per mission
constexpr int kPerWave = 64 / kLanes;                       // missions per wavefront

// SIXTEEN lanes per mission, four missions per wavefront: a mission starts at rest and reaches 1 mm/s within a few samples, so
// the first sixteen rows nearly always hold the first valid heading; a quarter of the waves of the one-wave-per-mission form this
// kernel started as (37 450 missions of 8 segments: 19.8 -> ~7 us, profiles/r06_rows_free_chain_kernel_stats.csv).  A lane finds
// the segment of its row by walking the mission's row counts from where it stood the step before (rows only grow): no prefix
// sums, no LDS, and a vertical climb of thousands of rows still costs one pass over the counts.
__global__ void __launch_bounds__(64 * kWaves) minsnap_first_yaw_kernel(
    const double *__restrict__ coeffs, const int32_t *__restrict__ seg_rows, const int64_t *__restrict__ seg_offsets, int B, int m,
    double dt, double *__restrict__ first_yaw) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = lane / kLanes, l = lane % kLanes;
    const int b = (blockIdx.x * kWaves + w) * kPerWave + g;
    const bool live = b < B;
    const int bb = live ? b : B - 1;                         // (dead groups shadow the last mission and write nothing)
    long long sb = (long long)bb * m;                        // first segment of the mission in the batch
    int mb = m;
    if (seg_offsets) {                                       // ragged batch: clamped like the sampler clamps it
        sb = seg_offsets[bb];
        const long long n = seg_offsets[bb + 1] - sb;
        mb = (int)(n < 1 ? 1 : (n > m ? m : n));
    }
    const int32_t *rows_of = seg_rows + sb;
    const double *cm = coeffs + (size_t)sb * 24;
    const unsigned long long field = 0xffffull << (kLanes * g);   // this mission's lanes in a ballot

    double out = 0.0;                                        // no row has a heading: zeros (minimum_snap.py:128-129)
    bool done = !live;
    int s = 0, base = 0, cnt = rows_of[0];                   // this lane's segment, its first row, its row count
    for (int r = l; __ballot(!done) != 0ull; r += kLanes) {
        bool active = false;
        double vx = 0.0, vy = 0.0;
        if (!done) {
            while (s + 1 < mb && r >= base + cnt) { base += cnt; ++s; cnt = rows_of[s]; }     // first segment whose rows reach past r
            active = r < base + cnt;                         // (past the last segment's rows: past the mission's end)
            if (active) {
                const double t = (double)(r - base) * dt;
                const double *c = cm + s * 24;
                double p, a;
                minsnap_eval_axis<1>(c, 0, t, p, vx, a);
                minsnap_eval_axis<1>(c, 1, t, p, vy, a);
            }
        }
        const bool valid = active && has_heading(vx, vy);
        const unsigned long long vmask = __ballot(valid) & field, amask = __ballot(active) & field;
        if (!done) {
            if (vmask != 0ull) {                             // the mission's first valid sample is in this step
                const double ang = valid ? heading(vy, vx) : 0.0;
                out = __shfl(ang, __builtin_ctzll(vmask));
                done = true;
            } else if (amask != field) {                     // the mission's rows ended in (or before) this step
                done = true;
            }
        }
    }
    if (live && l == 0) first_yaw[b] = out;
}

}  // namespace

int uavac_launch_first_yaw(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *seg_offsets, int B, int m,
                           double dt, double *first_yaw) {
    const int per_wg = kWaves * kPerWave;
    hipLaunchKernelGGL(minsnap_first_yaw_kernel, dim3((B + per_wg - 1) / per_wg), dim3(64 * kWaves), 0, ctx->stream, coeffs,
                       seg_rows, seg_offsets, B, m, dt, first_yaw);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
