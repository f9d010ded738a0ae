// The obstacle loop of MinimumSnap._generate_collision_free_trajectory (uav_ac/planning/minimum_snap.py:63-95 upstream) on the
// device, for B ragged missions against one cuboid: per round the active missions are planned (times, solve: existing
// kernels), their splines are scanned for samples inside the cuboid (:81-87, is_collision_cuboid :327-357) WITHOUT storing
// rows -- inside the loop only the hit flags matter, the trajectories are sampled once from the final waypoints --, and a
// midpoint is inserted before the end waypoint of every hit spline (insert_midpoints_at_indexes :359-391) into the next
// round's waypoint arrays.  Round 2 did the insertion on the host: one round trip of hit flags and waypoint lists per round,
// ~260 rounds when a few missions run the bounded loop to its end.

#include "uavac_internal.h"

namespace {

// Collision scan: one wavefront per active mission, lanes over its rows.  Positions by the sampler's own Horner chain
// (minsnap_eval.h: px = fma(px, t, c_i), i = 6 .. 0 -- the position does not depend on the derivative chains it is
// interleaved with there), so the samples tested are bit for bit the rows the sampler would store.
__global__ void __launch_bounds__(64) obstacle_hits_kernel(const double *__restrict__ coeffs, const int32_t *__restrict__ seg_rows,
                                                          const int64_t *__restrict__ seg_offsets, int B, int max_m, double dt,
                                                          const double *__restrict__ aabb, const int32_t *__restrict__ active,
                                                          int32_t *__restrict__ hit) {
    extern __shared__ double lds[];
    double *cl = lds;                                     // [24 * max_m]
    int *pre = reinterpret_cast<int *>(cl + 24 * max_m);  // [max_m + 1]
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (!active[b]) return;
    const size_t seg0 = (size_t)seg_offsets[b];
    const int64_t n = seg_offsets[b + 1] - seg_offsets[b];
    const int mb = (int)(n < 1 ? 1 : (n > max_m ? max_m : n));
    for (int i = lane; i < 24 * mb; i += 64) cl[i] = coeffs[seg0 * 24 + i];
    {
        const int v = (lane < mb) ? seg_rows[seg0 + lane] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane < mb) { pre[lane] = inc - v; hit[seg0 + lane] = 0; }
        if (lane == mb - 1) pre[mb] = inc;
    }
    __syncthreads();
    const int N = pre[mb];
    const double x0 = aabb[0], x1 = aabb[1], y0 = aabb[2], y1 = aabb[3], z0 = aabb[4], z1 = aabb[5];
    int s = 0;
    for (int r = lane; r < N; r += 64) {
        while (s + 1 < mb && r >= pre[s + 1]) ++s;
        const double t = (double)(r - pre[s]) * dt;
        const double *c = cl + s * 24;
        double px = c[21], py = c[22], pz = c[23];
#pragma unroll
        for (int i = 6; i >= 0; --i) { px = fma(px, t, c[3 * i]); py = fma(py, t, c[3 * i + 1]); pz = fma(pz, t, c[3 * i + 2]); }
        const bool in = (px >= x0) & (px <= x1) & (py >= y0) & (py <= y1) & (pz >= z0) & (pz <= z1);    // inclusive
        if (in) hit[seg0 + s] = 1;                        // after the zeroing above in program order (one wave)
    }
}

__device__ __forceinline__ int64_t block_scan_256(int64_t v, int64_t *wsum) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    int64_t base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    return v + base;
}

// Per mission: how many segments will it have after this round?  Active missions without a hit are clean for this cuboid
// and leave the loop; a mission that would outgrow max_m is flagged and leaves it as it is.
// counters[0] += missions that stay active, counters[1] += missions that outgrew max_m, counters[2] = max segments of any mission
// (counters[3] = segments of the whole batch after the round: insert_scatter_kernel).
__global__ void __launch_bounds__(256) insert_count_kernel(const int64_t *__restrict__ seg_offsets, const int32_t *__restrict__ hit,
                                                          int B, int max_m, int32_t *__restrict__ active,
                                                          int32_t *__restrict__ overflow, int32_t *__restrict__ touched,
                                                          int32_t *__restrict__ totals, int64_t *__restrict__ tile_sum,
                                                          int32_t *__restrict__ counters) {
    __shared__ int64_t wsum[4];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int64_t new_m = 0;
    if (b < B) {
        const int64_t s0 = seg_offsets[b];
        const int m = (int)(seg_offsets[b + 1] - s0);
        new_m = m;
        if (active[b]) {
            int n_hit = 0;
            for (int s = 0; s < m; ++s) n_hit += hit[s0 + s] != 0;
            if (n_hit == 0) {
                active[b] = 0;                            // no sample inside the cuboid: done with this obstacle
            } else if (m + n_hit > max_m) {
                active[b] = 0;
                overflow[b] = 1;
                atomicAdd(&counters[1], 1);
            } else {
                new_m = m + n_hit;
                active[b] = 2;                            // 2 = receives midpoints in this round (insert_scatter_kernel)
                touched[b] = 1;
                atomicAdd(&counters[0], 1);
            }
        }
        totals[b] = (int32_t)new_m;
        atomicMax(&counters[2], (int32_t)new_m);
    }
    const int64_t inc = block_scan_256(new_m, wsum);
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = inc;
}

// insert_midpoints_at_indexes (minimum_snap.py:359-391): before the end waypoint of every hit spline goes the midpoint
// (wp[i-1] + wp[i]) / 2 of its two waypoints; everything else is copied.  One thread per mission (<= 65 waypoints).
__global__ void __launch_bounds__(256) insert_scatter_kernel(const double *__restrict__ wp, const int64_t *__restrict__ seg_offsets,
                                                            const int32_t *__restrict__ hit, int B, int32_t *__restrict__ active,
                                                            double *__restrict__ wp_out,
                                                            const int64_t *__restrict__ seg_offsets_out,
                                                            int32_t *__restrict__ counters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (b == 0) counters[3] = (int32_t)(seg_offsets_out[B] > 2147483647LL ? 2147483647LL : seg_offsets_out[B]);
    const int64_t s0 = seg_offsets[b];
    const int m = (int)(seg_offsets[b + 1] - s0);
    const double *w = wp + (s0 + b) * 3;
    double *o = wp_out + (seg_offsets_out[b] + b) * 3;
    const bool ins = active[b] == 2;
    if (ins) active[b] = 1;
    double ax = w[0], ay = w[1], az = w[2];
    o[0] = ax; o[1] = ay; o[2] = az;
    int k = 1;
    for (int s = 0; s < m; ++s) {
        const double bx = w[3 * s + 3], by = w[3 * s + 4], bz = w[3 * s + 5];
        if (ins && hit[s0 + s] != 0) {
            o[3 * k] = (ax + bx) / 2.0; o[3 * k + 1] = (ay + by) / 2.0; o[3 * k + 2] = (az + bz) / 2.0;
            ++k;
        }
        o[3 * k] = bx; o[3 * k + 1] = by; o[3 * k + 2] = bz;
        ++k;
        ax = bx; ay = by; az = bz;
    }
}

}  // namespace

int uavac_launch_obstacle_scan_and_insert(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, const double *coeffs,
                                          const int32_t *seg_rows, int B, int max_m, double dt, const double *aabb,
                                          int32_t *active, int32_t *overflow, int32_t *touched, int32_t *hit, double *wp_out,
                                          int64_t *seg_offsets_out, int32_t *counters) {
    const size_t lds = sizeof(double) * (size_t)24 * max_m + sizeof(int) * (size_t)(max_m + 2);
    hipLaunchKernelGGL(obstacle_hits_kernel, dim3(B), dim3(64), lds, ctx->stream, coeffs, seg_rows, seg_offsets, B, max_m, dt,
                       aabb, active, hit);
    int32_t *totals = nullptr;
    int64_t *tiles = nullptr;
    if (int rc = uavac_ensure_totals(ctx, B, &totals, &tiles)) return rc;
    const int n_tiles = (B + 255) / 256;
    // (the limit a mission may grow to is the library's, not this round's batch maximum)
    hipLaunchKernelGGL(insert_count_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, seg_offsets, hit, B, (int)UAVAC_MAX_SEGMENTS,
                       active, overflow, touched, totals, tiles, counters);
    if (int rc = uavac_launch_totals_scan(ctx, B, seg_offsets_out)) return rc;
    hipLaunchKernelGGL(insert_scatter_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, wp, seg_offsets, hit, B, active, wp_out,
                       seg_offsets_out, counters);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
