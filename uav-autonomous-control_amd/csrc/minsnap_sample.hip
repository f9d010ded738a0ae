// Minimum-snap sampler + yaw scan (gfx950): coefficients -> rows [p(3) v(3) a(3) yaw spline_id].
//
// Replaces uav_ac/planning/minimum_snap.py (upstream paths):
//   _generate_trajectory sampling loop   :100-119  (t = k*dt for k < ceil(T/dt); polynom(8,k,t) @ coeffs)
//   _calculate_yaws                      :126-136  (atan2 on samples with |v_xy| >= 1e-3, np.unwrap over the
//                                                   valid subset, hold last valid, back-fill leading rows)
//   np.hstack row assembly               :122-123
//
// One wavefront (64-thread workgroup) per mission walks its rows in chunks of 64, in a single pass and
// without workgroup barriers.  The kernel is a pure HBM write stream (88 B per row): rows are evaluated one
// per lane (Horner, coefficients broadcast from LDS), the yaw hold / unwrap / back-fill is a wave ballot +
// shuffle scan carried across chunks in scalar registers, and each chunk is staged in LDS so that the
// row-major (N,11) output leaves as contiguous 16-byte stores (5.6 KB per chunk).

#include "uavac_internal.h"
#include "minsnap_eval.h"
#include "minsnap_yaw.h"

namespace {

constexpr int SB = 64;                  // rows per chunk == threads per workgroup == one wavefront
constexpr int kCarryMax = 16;           // doubles of a chunk that may wait for the next one (less than one 128-byte line)
constexpr int kStageDoubles = kCarryMax + SB * UAVAC_TRAJ_COLS;
using namespace uavac_yaw;

// double held by lane `l` (wave-uniform index): two v_readlane instead of two LDS-pipe bpermutes
__device__ __forceinline__ double lane_value(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int segment_of(const int *__restrict__ pre, int m, int r, int s) {
    while (s + 1 < m && r >= pre[s + 1]) ++s;
    return s;
}

// State of _calculate_yaws carried from one 64-row chunk to the next (wave-uniform): has a usable heading been seen,
// its raw angle, and the running sum of np.unwrap's corrections.
struct YawCarry {
    bool has = false;
    double ang = 0.0, sum = 0.0;
};

// One chunk of the yaw scan (minimum_snap.py:126-136): lane i holds row c0 + i (valid: its heading `ang` is usable).
// Returns the row's yaw: the unwrapped heading of the last usable row at or before it; rows before the sequence's first
// usable heading get that heading when it lies in this chunk (first_here / first_yaw; whole chunks of such rows that
// came earlier hold the placeholder 0 and are patched by the caller) and 0 otherwise.
__device__ __forceinline__ double yaw_chunk(bool valid, double ang, int lane, YawCarry &carry, bool &first_here,
                                            double &first_yaw) {
    // last valid heading strictly before this row: inside the chunk via ballot, else the carry
    const unsigned long long mask = __ballot(valid);
    const unsigned long long lower = mask & ((1ull << lane) - 1ull);
    bool prev_has = lower != 0ull;
    double prev_ang = __shfl(ang, prev_has ? 63 - __clzll((long long)lower) : 0);
    if (!prev_has && carry.has) { prev_has = true; prev_ang = carry.ang; }
    const double corr = (valid && prev_has) ? unwrap_correction(ang - prev_ang) : 0.0;
    // np.cumsum of the corrections, in NumPy's own order: left to right.  Headings rarely wrap, so the running sum is
    // advanced lane by lane over the few lanes that hold a non-zero correction (adding the zeros of the others would not
    // change a bit); every row then takes the sum up to and including itself.  The rollout, which visits the rows one by
    // one, reproduces exactly this sequence when it scans the yaw itself (control_rollout.hip, YAWSCAN).
    double cum = carry.sum;
    unsigned long long wraps = __ballot(corr != 0.0);
    if (wraps != 0ull) {
        double run = carry.sum;
        while (wraps != 0ull) {
            const int j = __builtin_ctzll(wraps);
            wraps &= wraps - 1ull;
            run = run + lane_value(corr, j);
            if (lane >= j) cum = run;
        }
        carry.sum = run;
    }
    // rows before the mission's first valid heading take that heading (np.searchsorted(...)-1 clipped to 0)
    first_here = !carry.has && mask != 0ull;
    const int first_lane = first_here ? __builtin_ctzll(mask) : 0;
    first_yaw = lane_value(ang, first_lane);       // its unwrap sum is 0 by construction
    double yaw;
    if (valid || prev_has) yaw = (valid ? ang : prev_ang) + cum;
    else yaw = first_here ? first_yaw : 0.0;                   // 0 = placeholder, patched by the caller if needed
    if (mask != 0ull) { carry.has = true; carry.ang = lane_value(ang, 63 - __clzll((long long)mask)); }
    return yaw;
}

// HITS: also flag the splines whose samples enter the cuboid aabb (collision scan of the obstacle re-plan loop).
// DERIVS: also write jerk / snap, [N][3] each -- the two outputs the reference computes in comments only
// (minimum_snap.py:111-112,118-119: polynom(8, 3 | 4, t) @ coeffs); separate arrays, never extra row columns.
// capacity_rows >= 0: the row buffer holds that many rows; a plan that needs more is refused as a whole (flag 2).
// YG: chunks whose dense-column yaws leave together (LDS buffer of YG * 64 doubles per wave).
// RAGGED: mission b has seg_offsets[b + 1] - seg_offsets[b] segments (clamped to 1 .. m, m = the batch's maximum, which sizes
// the LDS); coefficients, row counts and hit flags of the batch lie back to back.
template <bool HITS, bool DERIVS, int kYawGroup, bool RAGGED = false>
__global__ void __launch_bounds__(SB) minsnap_sample_kernel(const double *__restrict__ coeffs,
                                                           const int32_t *__restrict__ seg_rows,
                                                           const int64_t *__restrict__ row_offsets, int B, int m,
                                                           double dt, double *__restrict__ traj,
                                                           const double *__restrict__ aabb, int32_t *__restrict__ hit,
                                                           double *__restrict__ yaw_dense, double *__restrict__ jerk,
                                                           double *__restrict__ snap, int64_t capacity_rows,
                                                           int32_t *__restrict__ flags, double *__restrict__ first_yaw_out,
                                                           const int64_t *__restrict__ seg_offsets) {
    extern __shared__ double lds[];
    double *stage = lds;                         // [kCarryMax + SB*11]: what is left of the previous chunk, then this chunk
    double *cl = stage + kStageDoubles;          // [24*m] coefficients of this mission
    int *pre = reinterpret_cast<int *>(cl + 24 * m);   // [m+1] exclusive prefix of seg_rows
    // yaw column on its own: collected over kYawGroup chunks and written as one contiguous piece, so that the row
    // stream is interrupted a quarter as often (the dense column is 9 % of the bytes; written per chunk it cost up to 21 %)
    double *ybuf = reinterpret_cast<double *>(pre + ((m + 2 + 1) & ~1));          // [kYawGroup * SB]

    const int lane = threadIdx.x;
    const int b = xcd_contiguous(blockIdx.x, gridDim.x);      // consecutive missions (consecutive rows in HBM) per XCD
    const int64_t row0 = row_offsets[b];
    const int N = (int)(row_offsets[b + 1] - row0);
    if (capacity_rows >= 0 && row_offsets[B] > capacity_rows) {       // uniform over the launch: nobody writes
        if (blockIdx.x == 0 && lane == 0) atomicOr(&flags[2], 1);
        return;
    }

    int mb = m;                                   // segments of this mission, and where they start in the batch
    size_t seg0 = (size_t)b * m;
    if (RAGGED) {
        seg0 = (size_t)seg_offsets[b];
        const int64_t n = seg_offsets[b + 1] - seg_offsets[b];
        mb = (int)(n < 1 ? 1 : (n > m ? m : n));
    }
    for (int i = lane; i < 24 * mb; i += SB) cl[i] = coeffs[seg0 * 24 + i];
    {   // exclusive prefix of the per-segment row counts: lane s holds segment s (m <= 64)
        int v = (lane < mb) ? seg_rows[seg0 + lane] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane < mb) pre[lane] = inc - v;
        if (lane == mb - 1) pre[mb] = inc;
    }
    __syncthreads();

    // carried across chunks (wave-uniform): has a valid heading been seen, its raw angle, the running
    // unwrap sum (np.cumsum of np.unwrap's corrections), and the heading used for the back-fill
    YawCarry carry;
    double mission_first_yaw = 0.0;          // heading of the first row that has one (what rows before it take); 0 if none
    // The rows leave in pieces that END on a 128-byte line of the output (a mission's rows start at a multiple of 88 bytes,
    // so a 64-row chunk -- 44 lines' worth -- straddles lines at both ends): the doubles behind the last line boundary wait
    // in LDS (`held` of them, at the front of `stage`) and go out with the next chunk.  Every line inside a mission is then
    // written once, whole, by one run of stores; only a mission's first and last line are shared with its neighbours.
    int held = 0;
    double *next_out = traj + row0 * UAVAC_TRAJ_COLS;        // first element not yet stored
    // the cuboid's bounds, read once (inside the chunk loop, written with &&, they were up to six dependent scalar-cache
    // round trips per chunk: the atomic on `hit` keeps the compiler from hoisting them itself)
    double box[6] = {0, 0, 0, 0, 0, 0};
    if (HITS) {
#pragma unroll
        for (int j = 0; j < 6; ++j) box[j] = aabb[j];
    }
    int s = 0;
    for (int c0 = 0; c0 < N; c0 += SB) {
        const int r = c0 + lane;
        const bool active = r < N;
        double px = 0, py = 0, pz = 0, vx = 0, vy = 0, vz = 0, ax = 0, ay = 0, az = 0;
        if (active) {
            s = segment_of(pre, mb, r, s);
            const double t = (double)(r - pre[s]) * dt;
            const double *c = cl + s * 24;
            minsnap_eval_row<1>(c, t, px, py, pz, vx, vy, vz, ax, ay, az);
            if (DERIVS) {
                double j[3], q[3];
                minsnap_eval_jerk_snap<1>(c, t, j, q);
                const size_t o = (size_t)(row0 + r) * 3;
                if (jerk) { jerk[o] = j[0]; jerk[o + 1] = j[1]; jerk[o + 2] = j[2]; }
                if (snap) { snap[o] = q[0]; snap[o + 1] = q[1]; snap[o + 2] = q[2]; }
            }
        }
        if (HITS) {
            // inclusive AABB test on the sampled position (minimum_snap.py:327-357); flags the row's spline
            const bool in = active & (px >= box[0]) & (px <= box[1]) & (py >= box[2]) & (py <= box[3]) &
                            (pz >= box[4]) & (pz <= box[5]);
            if (in) atomicOr(&hit[seg0 + s], 1);
        }
        const bool valid = active && has_heading(vx, vy);
        const double ang = valid ? heading(vy, vx) : 0.0;
        bool first_here;
        double first_yaw;
        const double yaw = yaw_chunk(valid, ang, lane, carry, first_here, first_yaw);
        if (first_here) mission_first_yaw = first_yaw;
        if (first_here && c0 > 0) {
            // the first usable heading arrived after whole chunks of placeholders: patch their yaw column
            // (same wave, same addresses, program order => the later store wins)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // those rows' own stores have landed (once per mission at most)
            for (int i = lane; i < c0; i += SB) traj[(row0 + i) * UAVAC_TRAJ_COLS + 9] = first_yaw;
            if (lane < 2 && c0 - 1 - lane >= 0) {             // ... and in what is still held back in LDS (at most two rows reach into it)
                const long rel = (traj + (row0 + c0 - 1 - lane) * UAVAC_TRAJ_COLS + 9) - next_out;
                if (rel >= 0 && rel < held) stage[rel] = first_yaw;
            }
            if (yaw_dense) {
                const int flushed = (c0 / (kYawGroup * SB)) * (kYawGroup * SB);      // rows already written from ybuf
                for (int i = lane; i < flushed; i += SB) yaw_dense[row0 + i] = first_yaw;
                for (int i = flushed + lane; i < c0; i += SB) ybuf[i - flushed] = first_yaw;
            }
        }
        if (yaw_dense) {
            const int g0 = (c0 / (kYawGroup * SB)) * (kYawGroup * SB);          // first row of this group of chunks
            if (active) ybuf[r - g0] = yaw;
            const bool last_of_group = (c0 + SB - g0 == kYawGroup * SB) || (c0 + SB >= N);
            if (last_of_group) {
                lds_wave_fence();
                const int n_group = min(N, c0 + SB) - g0;
                // 16-byte stores from an even element (like the rows below); at most one 8-byte head and tail
                double *yd = yaw_dense + row0 + g0;
                const int yh = (int)((reinterpret_cast<uintptr_t>(yd) >> 3) & 1);
                if (yh && lane == 0) yd[0] = ybuf[0];
                const int ypairs = (n_group - yh) >> 1;
                for (int p = lane; p < ypairs; p += SB) {
                    double2 v;
                    v.x = ybuf[yh + 2 * p];
                    v.y = ybuf[yh + 2 * p + 1];
                    *reinterpret_cast<double2 *>(yd + yh + 2 * p) = v;
                }
                if (((n_group - yh) & 1) && lane == 63) yd[n_group - 1] = ybuf[n_group - 1];
            }
        }
        if (active) {
            double *o = stage + held + lane * UAVAC_TRAJ_COLS;
            o[0] = px; o[1] = py; o[2] = pz; o[3] = vx; o[4] = vy; o[5] = vz;
            o[6] = ax; o[7] = ay; o[8] = az; o[9] = yaw; o[10] = (double)s;
        }
        lds_wave_fence();                         // single wave: orders the LDS writes before the reads below

        // coalesced write-out: what was held back + this chunk, up to the last 128-byte line boundary (everything at the end
        // of the mission), as 16-byte stores from an even element index
        const int nrows = min(SB, N - c0);
        const int have = held + nrows * UAVAC_TRAJ_COLS;
        double *dst = next_out;
        const int beyond = (int)((reinterpret_cast<uintptr_t>(dst + have) >> 3) & 15);      // doubles past the last line boundary
        const int nel = (c0 + SB >= N || beyond >= have) ? have : have - beyond;
        const int head = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
        if (head && lane == 0) dst[0] = stage[0];
        const int npairs = (nel - head) >> 1;
        for (int p = lane; p < npairs; p += SB) {
            double2 v;
            v.x = stage[head + 2 * p];
            v.y = stage[head + 2 * p + 1];
            *reinterpret_cast<double2 *>(dst + head + 2 * p) = v;
        }
        if (((nel - head) & 1) && lane == 63) dst[nel - 1] = stage[nel - 1];
        const int rest = have - nel;                                 // < 16: moves to the front of the stage
        const double keep = (lane < rest) ? stage[nel + lane] : 0.0;
        lds_wave_fence();                         // the staged chunk is in registers / on its way; its stores stay in flight
        if (lane < rest) stage[lane] = keep;
        held = rest;
        next_out = dst + nel;
    }
    if (first_yaw_out && lane == 0) first_yaw_out[b] = mission_first_yaw;
}

// _calculate_yaws (minimum_snap.py:126-136) on its own: B independent velocity sequences, sequence b = rows
// [offsets[b], offsets[b+1]) of velocities[.][3]; one wavefront per sequence, 64 rows per step.
__global__ void __launch_bounds__(SB) yaw_scan_kernel(const double *__restrict__ vel, const int64_t *__restrict__ offsets,
                                                     double *__restrict__ yaws) {
    const int lane = threadIdx.x;
    const int64_t row0 = offsets[blockIdx.x];
    const int64_t N = offsets[blockIdx.x + 1] - row0;
    YawCarry carry;
    for (int64_t c0 = 0; c0 < N; c0 += SB) {
        const int64_t r = c0 + lane;
        const bool active = r < N;
        const double vx = active ? vel[(row0 + r) * 3] : 0.0, vy = active ? vel[(row0 + r) * 3 + 1] : 0.0;
        const bool valid = active && has_heading(vx, vy);
        const double ang = valid ? heading(vy, vx) : 0.0;
        bool first_here;
        double first_yaw;
        const double yaw = yaw_chunk(valid, ang, lane, carry, first_here, first_yaw);
        if (first_here && c0 > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the placeholders are down before they are replaced
            for (int64_t i = lane; i < c0; i += SB) yaws[row0 + i] = first_yaw;
        }
        if (active) yaws[row0 + r] = yaw;
    }
}

}  // namespace

int uavac_launch_yaw_scan(uavac_ctx *ctx, const double *velocities, const int64_t *offsets, int B, double *yaws) {
    hipLaunchKernelGGL(yaw_scan_kernel, dim3(B), dim3(SB), 0, ctx->stream, velocities, offsets, yaws);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_launch_sample(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets,
                        int B, int m, double dt, double *traj, const SampleExtras &x) {
    const bool plain = !(x.aabb && x.hit) && !(x.jerk || x.snap);
    // Dense yaw column: 8 chunks (4 KB) leave together.  Measured for the bench's plan on a typical box of the pool, rows +
    // column / rows only: 1 chunk 1.84 / 1.52 ms, 4 chunks 1.77, 8 chunks 1.66, 16 chunks 1.68 (LDS then costs occupancy).
    const int yg = (plain && (ctx->yaw_group == 1 || ctx->yaw_group == 4 || ctx->yaw_group == 16)) ? ctx->yaw_group : 8;
    size_t lds = sizeof(double) * ((size_t)kStageDoubles + (size_t)24 * m) + sizeof(int) * (size_t)((m + 2 + 1) & ~1) +
                 (x.yaw_dense ? sizeof(double) * yg * SB : 0);
    const bool hits = x.aabb && x.hit, derivs = x.jerk || x.snap;
    const bool ragged = x.seg_offsets != nullptr;
    if (ragged && (derivs || x.yaw_dense || x.total_segments < 0))
        return uavac_fail(ctx, UAVAC_EINVAL, "ragged sampling: rows (+ hit flags, first yaws) only, and the segment total");
    if (hits)
        UAVAC_HIP(ctx, hipMemsetAsync(x.hit, 0, sizeof(int32_t) * (ragged ? (size_t)x.total_segments : (size_t)B * m), ctx->stream));
    // default: the chunk-streaming kernel (minsnap_sample_stream.hip); sampler_waves == 1 keeps the one-wave-per-mission form
    if (ctx->sampler_waves > 1)
        return uavac_launch_sample_stream(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, ctx->sampler_waves, ctx->sampler_group);
#define UAVAC_SAMPLE(H, D, Y)                                                                                          \
    hipLaunchKernelGGL((minsnap_sample_kernel<H, D, Y>), dim3(B), dim3(SB), lds, ctx->stream, coeffs, seg_rows, row_offsets, \
                       B, m, dt, traj, x.aabb, x.hit, x.yaw_dense, x.jerk, x.snap, x.capacity_rows, ctx->d_flags, x.first_yaw, \
                       x.seg_offsets)
    if (ragged) {
        if (hits) hipLaunchKernelGGL((minsnap_sample_kernel<true, false, 8, true>), dim3(B), dim3(SB), lds, ctx->stream, coeffs,
                                     seg_rows, row_offsets, B, m, dt, traj, x.aabb, x.hit, x.yaw_dense, x.jerk, x.snap,
                                     x.capacity_rows, ctx->d_flags, x.first_yaw, x.seg_offsets);
        else hipLaunchKernelGGL((minsnap_sample_kernel<false, false, 8, true>), dim3(B), dim3(SB), lds, ctx->stream, coeffs,
                                seg_rows, row_offsets, B, m, dt, traj, x.aabb, x.hit, x.yaw_dense, x.jerk, x.snap,
                                x.capacity_rows, ctx->d_flags, x.first_yaw, x.seg_offsets);
    }
    else if (hits) { if (derivs) UAVAC_SAMPLE(true, true, 8); else UAVAC_SAMPLE(true, false, 8); }
    else if (derivs) UAVAC_SAMPLE(false, true, 8);
    else if (yg == 1) UAVAC_SAMPLE(false, false, 1);
    else if (yg == 4) UAVAC_SAMPLE(false, false, 4);
    else if (yg == 16) UAVAC_SAMPLE(false, false, 16);
    else UAVAC_SAMPLE(false, false, 8);
#undef UAVAC_SAMPLE
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
