// Minimum-snap sampler, W wavefronts per mission (gfx950): coefficients -> rows [p(3) v(3) a(3) yaw spline_id].
//
// Same arithmetic, same bits as minsnap_sample.hip (uav_ac/planning/minimum_snap.py:100-136 upstream: the sampling loop, the
// yaw scan of _calculate_yaws, the hstack row assembly); what differs is WHO writes WHAT WHEN.  There one wavefront walks a
// mission 64 rows at a time, so ~4 600 wavefronts hold ~4 600 write heads ~114 KB apart, each advancing 5.6 KB at a time.
// How fast HBM takes that depends on where the row buffer's pages lie (tools/buffer_placement_probe.py: 5.1 or 6.0 TB/s for
// the life of the buffer).  Store-only probes of other shapes on the same buffers (tools/sampler_shape_probe.hip, round 3,
// twelve buffers: slow / medium / fast ones): one wave per mission 1.40 / 1.28 / 1.14 ms, SIXTEEN waves per mission -- a
// workgroup writes 90 KB of consecutive addresses per round, a mission in two rounds -- 1.28 / 1.18 / 1.10 ms, the same as
// a compact address-ordered front.  Here: one workgroup of W = 16 wavefronts per mission, a round = 1 024 consecutive rows.
//
// The yaw column is a scan over the rows (heading of the last row that has one, np.unwrap'ped, back-filled).  Inside a round
// every wave first reduces its 64 rows to a summary that needs no history (has a heading / first / last heading / the
// unwrap corrections between its own headings); after one barrier every wave combines the summaries of the waves before it
// with the carry of the rounds before -- in parallel over the waves when no heading wraps in the round (almost always),
// else by the same left-to-right additions np.cumsum makes -- and finishes its rows.  The rows of a round are staged in
// LDS and leave as 16-byte stores up to the last 128-byte line boundary; the doubles behind it wait for the next round.
// Three LDS-only barriers per round; global stores stay in flight across them.

#include "uavac_internal.h"
#include "minsnap_eval.h"
#include "minsnap_yaw.h"

namespace {

constexpr int kCarryMax = 16;           // doubles of a round that may wait for the next one (less than one 128-byte line)
using namespace uavac_yaw;

__device__ __forceinline__ double lane_value(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int segment_of(const int *__restrict__ pre, int m, int r, int s) {
    while (s + 1 < m && r >= pre[s + 1]) ++s;
    return s;
}

template <int W>
__global__ void __launch_bounds__(64 * W) minsnap_sample_wide_kernel(const double *__restrict__ coeffs,
                                                                    const int32_t *__restrict__ seg_rows,
                                                                    const int64_t *__restrict__ row_offsets, int B, int m,
                                                                    double dt, double *__restrict__ traj,
                                                                    int64_t capacity_rows, int32_t *__restrict__ flags,
                                                                    double *__restrict__ first_yaw_out) {
    constexpr int NT = 64 * W;                   // threads per workgroup = rows per round
    extern __shared__ double lds[];
    double *stage = lds;                                        // [kCarryMax + NT * 11]: what the last round left, then this round
    double *cl = stage + kCarryMax + NT * UAVAC_TRAJ_COLS;      // [24 m] coefficients of this mission
    int *pre = reinterpret_cast<int *>(cl + 24 * m);            // [m + 1] exclusive prefix of seg_rows
    double *sum_first = reinterpret_cast<double *>(pre + ((m + 2 + 1) & ~1));     // [W] first heading of wave w's rows
    double *sum_last = sum_first + W;                           // [W] last heading
    double *corr_list = sum_last + W;                           // [W][64] non-zero unwrap corrections between a wave's own headings
    int *sum_has = reinterpret_cast<int *>(corr_list + W * 64); // [W] any heading among the wave's rows
    int *sum_nint = sum_has + W;                                // [W] entries of corr_list[w]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = xcd_contiguous(blockIdx.x, gridDim.x);        // consecutive missions (consecutive rows in HBM) per XCD
    const int64_t row0 = row_offsets[b];
    const int N = (int)(row_offsets[b + 1] - row0);
    if (capacity_rows >= 0 && row_offsets[B] > capacity_rows) {         // uniform over the launch: nobody writes
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[2], 1);
        return;
    }
    const size_t seg0 = (size_t)b * m;
    for (int i = tid; i < 24 * m; i += NT) cl[i] = coeffs[seg0 * 24 + i];
    if (w == 0) {   // exclusive prefix of the per-segment row counts: lane s holds segment s (m <= 64)
        const int v = (lane < m) ? seg_rows[seg0 + lane] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane < m) pre[lane] = inc - v;
        if (lane == m - 1) pre[m] = inc;
    }
    __syncthreads();

    // carry of the yaw scan into the round (uniform over the workgroup: every wave derives it from the same summaries)
    bool c_has = false;
    double c_ang = 0.0, c_sum = 0.0;
    double mission_first_yaw = 0.0;          // heading of the first row that has one (what rows before it take); 0 if none
    int held = 0;                            // doubles waiting at the front of `stage`
    double *next_out = traj + row0 * UAVAC_TRAJ_COLS;          // first element not yet stored
    int s = 0;
    for (int c0 = 0; c0 < N; c0 += NT) {
        const int r = c0 + tid;
        const bool active = r < N;
        double px = 0, py = 0, pz = 0, vx = 0, vy = 0, vz = 0, ax = 0, ay = 0, az = 0;
        if (active) {
            s = segment_of(pre, m, r, s);
            minsnap_eval_row<1>(cl + s * 24, (double)(r - pre[s]) * dt, px, py, pz, vx, vy, vz, ax, ay, az);
        }
        // ---- this wave's 64 rows on their own: headings, and the unwrap corrections between them
        const bool valid = active && has_heading(vx, vy);
        const double ang = valid ? atan2(vy, vx) : 0.0;
        const unsigned long long mask = __ballot(valid);
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long lower = mask & below;
        const bool prev_in = lower != 0ull;                                      // a heading earlier in this wave's rows
        const double prev_ang_in = __shfl(ang, prev_in ? 63 - __clzll((long long)lower) : 0);
        const double corr_in = (valid && prev_in) ? unwrap_correction(ang - prev_ang_in) : 0.0;
        const unsigned long long wraps_in = __ballot(corr_in != 0.0);
        if (corr_in != 0.0) corr_list[w * 64 + __popcll(wraps_in & below)] = corr_in;
        if (lane == 0) {
            sum_has[w] = mask != 0ull;
            sum_nint[w] = __popcll(wraps_in);
        }
        {
            const double f = lane_value(ang, mask ? __builtin_ctzll(mask) : 0);
            const double l = lane_value(ang, mask ? 63 - __clzll((long long)mask) : 0);
            if (lane == 0) { sum_first[w] = f; sum_last[w] = l; }
        }
        lds_barrier();                                                           // B: every wave's summary is in LDS

        // ---- the waves before this one (lane j looks at wave j)
        const int h_j = (lane < W) ? sum_has[lane] : 0;
        const double f_j = (lane < W) ? sum_first[lane] : 0.0, l_j = (lane < W) ? sum_last[lane] : 0.0;
        const int n_j = (lane < W) ? sum_nint[lane] : 0;
        const unsigned long long hasmask = __ballot(h_j != 0);
        const unsigned long long lowerh = hasmask & below;
        const bool pj_has = lowerh != 0ull || c_has;
        const double pl = __shfl(l_j, lowerh ? 63 - __clzll((long long)lowerh) : 0);
        const double pj_ang = lowerh ? pl : c_ang;
        // correction at wave j's first heading against the last heading before it (np.unwrap's step across the wave boundary)
        const double cb_j = (h_j && pj_has) ? unwrap_correction(f_j - pj_ang) : 0.0;
        const bool any_wrap = __ballot(cb_j != 0.0 || n_j != 0) != 0ull;
        // what reaches this wave: last heading and running sum of the corrections
        const unsigned long long lowerw = hasmask & ((1ull << w) - 1ull);
        const bool in_has = lowerw != 0ull || c_has;
        const double in_ang = lowerw ? lane_value(l_j, 63 - __clzll((long long)lowerw)) : c_ang;
        double in_sum = c_sum, out_sum = c_sum;
        if (any_wrap) {
            // a heading wraps somewhere in this round: np.cumsum's own order, wave after wave, left to right (uniform; rare)
            double run = c_sum;
            for (int j = 0; j < W; ++j) {
                if (j == w) in_sum = run;
                const double cbj = lane_value(cb_j, j);
                if (cbj != 0.0) run = run + cbj;
                const int nj = __builtin_amdgcn_readlane(n_j, j);
                for (int i = 0; i < nj; ++i) run = run + corr_list[j * 64 + i];
            }
            out_sum = run;
        }
        const bool first_round = !c_has && hasmask != 0ull;                      // the mission's first heading lies in this round
        if (first_round) mission_first_yaw = lane_value(f_j, __builtin_ctzll(hasmask));

        // ---- this wave's rows, with the history in hand (minimum_snap.py:126-136)
        double cum = in_sum;
        const double cb_w = lane_value(cb_j, w);
        if (cb_w != 0.0 || wraps_in != 0ull) {
            double run = in_sum;
            if (cb_w != 0.0) {
                run = run + cb_w;
                if (lane >= __builtin_ctzll(mask)) cum = run;
            }
            unsigned long long wr = wraps_in;
            while (wr != 0ull) {
                const int j = __builtin_ctzll(wr);
                wr &= wr - 1ull;
                run = run + lane_value(corr_in, j);
                if (lane >= j) cum = run;
            }
        }
        const bool prev_has = prev_in || in_has;
        const double prev_ang = prev_in ? prev_ang_in : in_ang;
        double yaw;
        if (valid || prev_has) yaw = (valid ? ang : prev_ang) + cum;
        else yaw = (hasmask != 0ull) ? mission_first_yaw : 0.0;    // before the first heading: that heading, or a placeholder
        if (first_round && c0 > 0) {
            // the first heading arrived after whole rounds of placeholders: patch their yaw column.  Those rows were stored
            // by other waves of this workgroup: everybody's stores have completed before anybody patches.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            for (int i = tid; i < c0; i += NT) traj[(row0 + i) * UAVAC_TRAJ_COLS + 9] = mission_first_yaw;
            if (tid < 2 && c0 - 1 - tid >= 0) {               // ... and in what is still held back in LDS (at most two rows reach into it)
                const long rel = (traj + (row0 + c0 - 1 - tid) * UAVAC_TRAJ_COLS + 9) - next_out;
                if (rel >= 0 && rel < held) stage[rel] = mission_first_yaw;
            }
        }
        if (active) {
            double *o = stage + held + tid * UAVAC_TRAJ_COLS;
            o[0] = px; o[1] = py; o[2] = pz; o[3] = vx; o[4] = vy; o[5] = vz;
            o[6] = ax; o[7] = ay; o[8] = az; o[9] = yaw; o[10] = (double)s;
        }
        // carry into the next round
        if (hasmask != 0ull) { c_has = true; c_ang = lane_value(l_j, 63 - __clzll((long long)hasmask)); }
        c_sum = out_sum;
        lds_barrier();                                                           // C: the round is staged

        // ---- write-out: what was held back + this round, up to the last 128-byte line boundary (everything at the end of
        // the mission), as 16-byte stores from an even element index; consecutive threads, consecutive addresses
        const int nrows = min(NT, N - c0);
        const int have = held + nrows * UAVAC_TRAJ_COLS;
        double *dst = next_out;
        const int beyond = (int)((reinterpret_cast<uintptr_t>(dst + have) >> 3) & 15);      // doubles past the last line boundary
        const int nel = (c0 + NT >= N || beyond >= have) ? have : have - beyond;
        const int head = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
        if (head && tid == 0) dst[0] = stage[0];
        const int npairs = (nel - head) >> 1;
        for (int p = tid; p < npairs; p += NT) {
            double2 v;
            v.x = stage[head + 2 * p];
            v.y = stage[head + 2 * p + 1];
            *reinterpret_cast<double2 *>(dst + head + 2 * p) = v;
        }
        if (((nel - head) & 1) && tid == NT - 1) dst[nel - 1] = stage[nel - 1];
        const int rest = have - nel;                                 // < 16: moves to the front of the stage
        const double keep = (tid < rest) ? stage[nel + tid] : 0.0;
        lds_barrier();                                                           // D: the stage has been read
        if (tid < rest) stage[tid] = keep;
        held = rest;
        next_out = dst + nel;
    }
    if (first_yaw_out && tid == 0) first_yaw_out[b] = mission_first_yaw;
}

template <int W>
int launch_wide(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets, int B, int m,
                double dt, double *traj, int64_t capacity_rows, double *first_yaw) {
    const size_t lds = sizeof(double) * ((size_t)kCarryMax + (size_t)64 * W * UAVAC_TRAJ_COLS + (size_t)24 * m) +
                       sizeof(int) * (size_t)((m + 2 + 1) & ~1) + sizeof(double) * ((size_t)2 * W + (size_t)W * 64) +
                       sizeof(int) * (size_t)2 * W;
    auto kern = minsnap_sample_wide_kernel<W>;
    if (lds > 64 * 1024) UAVAC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(B), dim3(64 * W), lds, ctx->stream, coeffs, seg_rows, row_offsets, B, m, dt, traj,
                       capacity_rows, ctx->d_flags, first_yaw);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

}  // namespace

int uavac_launch_sample_wide(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets, int B,
                             int m, double dt, double *traj, int64_t capacity_rows, double *first_yaw, int waves) {
    if (waves == 8) return launch_wide<8>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, capacity_rows, first_yaw);
    if (waves == 4) return launch_wide<4>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, capacity_rows, first_yaw);
    return launch_wide<16>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, capacity_rows, first_yaw);
}
