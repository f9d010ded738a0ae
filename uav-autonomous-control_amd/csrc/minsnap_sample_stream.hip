// Minimum-snap sampler + yaw scan, chunk-streaming form (gfx950): coefficients -> rows [p(3) v(3) a(3) yaw spline_id].
//
// Replaces uav_ac/planning/minimum_snap.py (upstream paths):
//   _generate_trajectory sampling loop   :100-119  (t = k*dt for k < ceil(T/dt); polynom(8,k,t) @ coeffs)
//   _calculate_yaws                      :126-136  (atan2 on samples with |v_xy| >= 1e-3, np.unwrap over the
//                                                   valid subset, hold last valid, back-fill leading rows)
//   np.hstack row assembly               :122-123
//
// The row buffer (N,11) is cut into CHUNKS of 64 rows whose first byte lies on a 128-byte line: a chunk is 5 632 bytes =
// 44 whole lines, whatever mission its rows belong to (rows are 88 bytes and missions start at multiples of 88 bytes, so the
// cut is made in the row index of the whole batch, shifted by a phase that depends on the buffer's address only).  A
// workgroup of W wavefronts owns G consecutive missions and walks their chunks in address order, wave w taking items w,
// w + W, w + 2W, ... (an item = the rows of ONE mission inside one chunk; a chunk that holds the end of one mission and the
// start of the next is two items): at any moment a workgroup writes W consecutive chunks, a CU a few such runs, an XCD one
// compact stretch of the buffer -- instead of one write head per mission 114 KB apart (the one-wave-per-mission kernel this
// replaces: ~4 600 heads on the chip; store-only probes of both shapes: tools/sampler_shape_probe.hip).
//
// The yaw column is a scan along a mission's rows.  Every wave first evaluates its 64 rows and reduces their headings to
// what needs no history; then it takes the scan's carry (heading seen / last heading / running sum of np.unwrap's
// corrections / the mission's first heading) from the wave that holds the chunk before -- an LDS mailbox per wave with a
// sequence number, polled; no workgroup barrier anywhere in the loop -- adds its own part in NumPy's left-to-right order,
// hands the carry on and only then finishes and stages its rows.  The hop on the chain is a dozen instructions; waves of a
// workgroup never wait for a LATER item, so the chain cannot dead-lock.  Rows leave through a per-wave LDS stage as 16-byte
// stores of whole lines; only a mission's first and last item are partial.
//
// A mission whose first usable heading arrives after whole chunks of rows (a vertical climb) needs those rows' yaw patched
// (np.searchsorted(...) - 1 clipped to 0: rows before the first heading take it).  Other waves wrote them: the patch is noted
// in LDS and applied after the workgroup's ONE barrier at its end, when every wave's stores have completed.

#include "uavac_internal.h"
#include "minsnap_eval.h"
#include "minsnap_yaw.h"

namespace {

constexpr int kChunkRows = 64;
constexpr int kChunkDoubles = kChunkRows * UAVAC_TRAJ_COLS;       // 704 doubles = 5 632 bytes = 44 lines
using namespace uavac_yaw;

__device__ __forceinline__ double lane_value(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double uniform_double(double v) { return lane_value(v, 0); }
__device__ __forceinline__ long long uniform64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (long long)lo;
}

// The scan's carry out of an item (wave-uniform) as it travels through LDS.  seq = 1 + the workgroup-local index of the item
// it was computed from; written last, read first.
struct Mail {
    int seq, has;
    double ang, sum, first;
};

// LDS accesses of the mailboxes as explicit DS instructions that wait for LDS alone.  (volatile C++ accesses to LDS are
// compiled to FLAT loads / stores followed by s_waitcnt vmcnt(0): every poll would wait for the wave's row stores to land.)
__device__ __forceinline__ unsigned lds_address(const void *p) { return (unsigned)(uintptr_t)p; }   // low half of a generic LDS pointer
// {seq, has} of a mailbox: one 8-byte read
__device__ __forceinline__ void mail_poll(unsigned a, int &seq, int &has) {
    long long v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    // wave-uniform, and said so: with per-lane values the compiler builds the poll loop out of exec masks (unrolled six times)
    seq = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    has = __builtin_amdgcn_readfirstlane((int)(v >> 32));
}
__device__ __forceinline__ void mail_read(unsigned a, double &ang, double &sum, double &first) {
    asm volatile("ds_read_b64 %0, %3 offset:8\n\tds_read_b64 %1, %3 offset:16\n\tds_read_b64 %2, %3 offset:24\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ang), "=&v"(sum), "=&v"(first) : "v"(a) : "memory");
}
// values first, {seq, has} last: DS operations of one wave execute in order
__device__ __forceinline__ void mail_write(unsigned a, int seq, int has, double ang, double sum, double first) {
    const long long v = ((long long)has << 32) | (long long)(unsigned)seq;
    asm volatile("ds_write_b64 %0, %1 offset:8\n\tds_write_b64 %0, %2 offset:16\n\tds_write_b64 %0, %3 offset:24\n\tds_write_b64 %0, %4"
                 :: "v"(a), "v"(ang), "v"(sum), "v"(first), "v"(v) : "memory");
}
__device__ __forceinline__ int padded_pre(int m) { return (m + 2 + 1) & ~1; }

// HITS / DERIVS / RAGGED as in minsnap_sample.hip.  G = missions per workgroup; phase = rows by which the 64-row grid is
// shifted so that chunk starts are 128-byte aligned in `traj` (0 .. 63, from the buffer's address).
template <int W, bool HITS, bool DERIVS, bool RAGGED>
__global__ void __launch_bounds__(64 * W, (W == 16 || DERIVS ? 4 : 6)) minsnap_sample_stream_kernel(
    const double *__restrict__ coeffs, const int32_t *__restrict__ seg_rows, const int64_t *__restrict__ row_offsets, int B,
    int m, double dt, double *__restrict__ traj, const double *__restrict__ aabb, int32_t *__restrict__ hit,
    double *__restrict__ yaw_dense, double *__restrict__ jerk, double *__restrict__ snap, int64_t capacity_rows,
    int32_t *__restrict__ flags, double *__restrict__ first_yaw_out, const int64_t *__restrict__ seg_offsets, int G, int phase) {
    constexpr int NT = 64 * W;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mp = padded_pre(m);
    double *stage = lds + (size_t)w * kChunkDoubles;                        // this wave's chunk, row-major
    double *cl = lds + (size_t)W * kChunkDoubles;                           // [G][24 m] coefficients
    int *pre = reinterpret_cast<int *>(cl + (size_t)G * 24 * m);            // [G][mp] exclusive prefix of seg_rows (+ total)
    long long *rowoff = reinterpret_cast<long long *>(pre + (size_t)G * mp);   // [G + 1] first row of each mission in the batch
    long long *segbase = rowoff + G + 1;                                    // [G] first segment of each mission in the batch
    double *patch_yaw = reinterpret_cast<double *>(segbase + G);            // [G] heading for the rows before the first heading
    int *segn = reinterpret_cast<int *>(patch_yaw + G);                     // [G] segments of each mission
    // the mailboxes are a STATIC LDS array: volatile accesses through a pointer derived from the dynamic array stay generic
    // (flat) loads and stores, and a flat access waits for the wave's global stores as well (s_waitcnt vmcnt(0) on every poll)
    __shared__ Mail mail[W];
    __shared__ __attribute__((aligned(16))) double heading_poly[kHeadingCoefficients];      // see HeadingFromLds
    int *patch_rows = segn + G;                                             // [G] how many leading rows to patch (0: none)

    if (capacity_rows >= 0 && row_offsets[B] > capacity_rows) {             // uniform over the launch: nobody writes
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[2], 1);
        return;
    }
    const int wg = xcd_contiguous(blockIdx.x, gridDim.x);                   // consecutive missions (consecutive rows) per XCD
    const int b0 = wg * G;
    const int Gn = min(G, B - b0);

    // ---- tables of this workgroup's missions
    if (tid <= Gn) rowoff[tid] = row_offsets[b0 + tid];
    if (tid < W) { mail[tid].seq = 0; }
    if (tid < kHeadingCoefficients) heading_poly[tid] = kHeadingPoly[tid];
    for (int j = 0; j < Gn; ++j) {
        long long sb = (long long)(b0 + j) * m;                             // uniform: scalar loads
        int mb = m;
        if (RAGGED) {
            sb = seg_offsets[b0 + j];
            const long long n = seg_offsets[b0 + j + 1] - sb;
            mb = (int)(n < 1 ? 1 : (n > m ? m : n));
        }
        if (w == j % W) {                            // exclusive prefix of the row counts per segment: lane s holds segment s (m <= 64)
            const int v = (lane < mb) ? seg_rows[sb + lane] : 0;
            int inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            int *pj = pre + (size_t)j * mp;
            if (lane < mb) pj[lane] = inc - v;
            if (lane == mb - 1) pj[mb] = inc;
            if (lane == 0) { segn[j] = mb; segbase[j] = sb; patch_rows[j] = 0; }
        }
        double *cj = cl + (size_t)j * 24 * m;
        for (int i = tid; i < 24 * mb; i += NT) cj[i] = coeffs[(size_t)sb * 24 + i];
    }
    __syncthreads();

    double box[6] = {0, 0, 0, 0, 0, 0};
    if (HITS) {
#pragma unroll
        for (int j = 0; j < 6; ++j) box[j] = aabb[j];
    }

    // ---- items in address order; this wave takes those congruent to w modulo W.  Everything that steers the loops is
    // wave-uniform and kept in scalar registers (readfirstlane: values read from LDS are not uniform to the compiler).
    int item0 = 0;                                   // workgroup-local index of mission j's first item
    for (int j = 0; j < Gn; ++j) {
        const long long Rj = uniform64(rowoff[j]), Rj1 = uniform64(rowoff[j + 1]);
        const int Nj = (int)(Rj1 - Rj);
        if (Nj <= 0) {
            if (first_yaw_out && tid == 0) first_yaw_out[b0 + j] = 0.0;
            continue;
        }
        const long long kfirst = (Rj + phase) >> 6, klast = (Rj1 - 1 + phase) >> 6;
        const int nitems = (int)(klast - kfirst + 1);
        const int mb = __builtin_amdgcn_readfirstlane(segn[j]);
        const double *clj = cl + (size_t)j * 24 * m;
        const int *prej = pre + (size_t)j * mp;
        const int endv = (lane < mb) ? prej[lane + 1] : 0x7fffffff;       // lane s: first row past segment s
        // (the compiler keeps this loop's counter in a vector register -- it takes the loop for divergent -- and with it every
        // address derived from it: the body works on copies made scalar by hand)
        const int item0s = __builtin_amdgcn_readfirstlane(item0);
        for (int iv = item0s + ((w - item0s) % W + W) % W; iv < item0s + nitems; iv += W) {   // first item >= item0 congruent to w, ...
            const int i = __builtin_amdgcn_readfirstlane(iv);
            // (made scalar by hand: the compiler takes them for per-lane values and does the write-out's address arithmetic in vector instructions)
            const long long g0 = uniform64(((kfirst + (i - item0s)) << 6) - phase);   // row (of the batch) in lane 0; may lie before the buffer
            const int rel0 = __builtin_amdgcn_readfirstlane((int)(g0 - Rj));         // the same, counted from the mission's first row
            const int r = rel0 + lane;
            const bool active = r >= 0 && r < Nj;
            // ---- the rows on their own: evaluated and staged at once (the yaw column follows when the history is known)
            double vx = 0, vy = 0;
            int s = __popcll(__ballot(endv <= (rel0 > 0 ? rel0 : 0)));     // segments that end at or before the item's first row
            s = s < mb - 1 ? s : mb - 1;
            if (active) {
                while (s + 1 < mb && r >= prej[s + 1]) ++s;
                const double t = (double)(r - prej[s]) * dt;
                const double *c = clj + s * 24;
                double *o = stage + lane * UAVAC_TRAJ_COLS;
                double px, py, pz, vz, ax, ay, az;
                minsnap_eval_axis<1>(c, 0, t, px, vx, ax);
                o[0] = px; o[3] = vx; o[6] = ax;
                minsnap_eval_axis<1>(c, 1, t, py, vy, ay);
                o[1] = py; o[4] = vy; o[7] = ay;
                minsnap_eval_axis<1>(c, 2, t, pz, vz, az);
                o[2] = pz; o[5] = vz; o[8] = az; o[10] = (double)s;
                if (DERIVS) {
                    double jk[3], q[3];
                    minsnap_eval_jerk_snap<1>(c, t, jk, q);
                    const size_t od = (size_t)(g0 + lane) * 3;
                    if (jerk) { jerk[od] = jk[0]; jerk[od + 1] = jk[1]; jerk[od + 2] = jk[2]; }
                    if (snap) { snap[od] = q[0]; snap[od + 1] = q[1]; snap[od + 2] = q[2]; }
                }
                if (HITS) {
                    // inclusive AABB test on the sampled position (minimum_snap.py:327-357); flags the row's spline
                    const bool in = (px >= box[0]) & (px <= box[1]) & (py >= box[2]) & (py <= box[3]) & (pz >= box[4]) &
                                    (pz <= box[5]);
                    if (in) atomicOr(&hit[segbase[j] + s], 1);
                }
            }
            const bool valid = active && has_heading(vx, vy);
            double ang = 0.0;
            if (valid) {
                HeadingFromLds hc;
                hc.lds = lds_address(heading_poly);
                ang = heading_with(vy, vx, hc);
            }
            const unsigned long long mask = __ballot(valid);
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long lower = mask & below;
            const bool prev_in = lower != 0ull;                             // a heading earlier in this item
            const double prev_ang_in = __shfl(ang, prev_in ? 63 - __clzll((long long)lower) : 0);
            double corr = (valid && prev_in) ? unwrap_correction(ang - prev_ang_in) : 0.0;
            const int first_lane = mask ? __builtin_ctzll(mask) : 0;
            const double first_ang = lane_value(ang, first_lane);
            const double last_ang = lane_value(ang, mask ? 63 - __clzll((long long)mask) : 0);

            const unsigned long long wraps_own = __ballot(corr != 0.0);    // corrections inside the item (known before the carry)
            // ---- the carry: wait for the item before (ordering of the mailboxes; its values only inside a mission)
            bool c_has = false;
            double c_ang = 0.0, c_sum = 0.0, c_first = 0.0;
            // (the hand-over is the one serial piece of a workgroup: it runs ahead of the other waves of its SIMD)
            __builtin_amdgcn_s_setprio(2);
            if (i > 0) {
                const unsigned src = lds_address(&mail[w == 0 ? W - 1 : w - 1]);
                int seq, has;
                mail_poll(src, seq, has);
                // (a wait is a few hundred cycles; a chain that stays broken for ~1 s aborts the launch instead of hanging the GPU)
                for (unsigned spins = 0; seq != i; ++spins) {
                    if (spins > (1u << 24)) __builtin_trap();
                    __builtin_amdgcn_s_sleep(1);
                    mail_poll(src, seq, has);
                }
                if (i > item0s) {
                    double ca, cs, cf;
                    mail_read(src, ca, cs, cf);
                    c_has = __builtin_amdgcn_readfirstlane(has) != 0;
                    c_ang = uniform_double(ca); c_sum = uniform_double(cs); c_first = uniform_double(cf);
                }
            }
            // np.unwrap's step from the last heading before this item to its first one, then np.cumsum's order, left to right
            // (an item without corrections -- nearly every one -- hands the sum on as it came: the test is all that stands between
            // the carry's arrival and its publication)
            const double cb = (mask != 0ull && c_has) ? unwrap_correction(first_ang - c_ang) : 0.0;
            double cum = c_sum, run = c_sum;
            if (wraps_own != 0ull || __ballot(cb != 0.0) != 0ull) {
                if (lane == first_lane && valid) corr = cb;                 // (0 unless c_has: that lane has no predecessor)
                unsigned long long wraps = __ballot(corr != 0.0);
                while (wraps != 0ull) {
                    const int l = __builtin_ctzll(wraps);
                    wraps &= wraps - 1ull;
                    run = run + lane_value(corr, l);
                    if (lane >= l) cum = run;
                }
            }
            const bool first_here = !c_has && mask != 0ull;
            const double m_first = c_has ? c_first : (mask != 0ull ? first_ang : 0.0);
            if (lane == 0)
                mail_write(lds_address(&mail[w]), i + 1, (c_has || mask != 0ull) ? 1 : 0, mask != 0ull ? last_ang : c_ang, run, m_first);
            __builtin_amdgcn_s_setprio(0);
            // ---- the rows with their history
            const bool prev_has = prev_in || c_has;
            const double prev_ang = prev_in ? prev_ang_in : c_ang;
            double yaw;
            if (valid || prev_has) yaw = (valid ? ang : prev_ang) + cum;
            else yaw = first_here ? first_ang : 0.0;                        // 0 = placeholder, patched at the end if needed
            if (first_here && rel0 > 0) {                                   // rows of this mission in earlier chunks wait for this heading
                if (lane == 0) { patch_yaw[j] = first_ang; patch_rows[j] = rel0; }
            }
            if (i == item0s + nitems - 1 && first_yaw_out && lane == 0) first_yaw_out[b0 + j] = m_first;
            if (active) {
                if (yaw_dense) yaw_dense[g0 + lane] = yaw;
                stage[lane * UAVAC_TRAJ_COLS + 9] = yaw;
            }
            lds_wave_fence();
            // ---- write-out: the item's doubles [e0, e1) of the chunk, 16-byte stores from even elements, whole lines
            double *dst = traj + g0 * UAVAC_TRAJ_COLS;                      // 128-byte aligned (never dereferenced before e0)
            if (rel0 >= 0 && rel0 + 64 <= Nj) {
                // a whole chunk (all but a mission's first and last item): 5 1/2 stores of 1 KB, scalar base + lane offset + immediate
                // (the second base keeps every immediate below 4 KB: beyond that the compiler falls back to per-lane 64-bit addresses)
                const double2 *sp = reinterpret_cast<const double2 *>(stage) + lane;
                double *dst2 = dst + 512;
                asm("" : "+s"(dst2));                                        // (opaque: otherwise folded back into dst + 4096)
                double2 *dp = reinterpret_cast<double2 *>(dst) + lane, *dq = reinterpret_cast<double2 *>(dst2) + lane;
                double2 v0 = sp[0], v1 = sp[64], v2 = sp[128], v3 = sp[192], v4 = sp[256];
                dp[0] = v0; dp[64] = v1; dp[128] = v2; dp[192] = v3; dq[0] = v4;
                if (lane < 32) dq[64] = sp[320];
            } else {
                const int l0 = rel0 < 0 ? -rel0 : 0, l1 = Nj - rel0 < 64 ? Nj - rel0 : 64;
                const int e0 = l0 * UAVAC_TRAJ_COLS, e1 = l1 * UAVAC_TRAJ_COLS;
                if ((e0 & 1) && lane == 0) dst[e0] = stage[e0];
                if ((e1 & 1) && lane == 63) dst[e1 - 1] = stage[e1 - 1];
                const int p0 = (e0 + 1) >> 1, p1 = e1 >> 1;
                for (int p = p0 + lane; p < p1; p += 64) {
                    double2 v;
                    v.x = stage[2 * p];
                    v.y = stage[2 * p + 1];
                    *reinterpret_cast<double2 *>(dst + 2 * p) = v;
                }
            }
            lds_wave_fence();                     // the staged chunk is in registers / on its way; its stores stay in flight
        }
        item0 += nitems;
    }

    // ---- leading rows that were written before their mission's first heading was known
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < Gn; ++j) {
        const int n = patch_rows[j];
        if (n == 0) continue;
        const double y = patch_yaw[j];
        const long long Rj = rowoff[j];
        for (int q = tid; q < n; q += NT) {
            traj[(Rj + q) * UAVAC_TRAJ_COLS + 9] = y;
            if (yaw_dense) yaw_dense[Rj + q] = y;
        }
    }
}

size_t stream_lds_bytes(int W, int G, int m) {
    return sizeof(double) * ((size_t)W * kChunkDoubles + (size_t)G * 24 * m) + sizeof(int) * (size_t)G * ((m + 2 + 1) & ~1) +
           sizeof(long long) * (size_t)(2 * G + 1) + sizeof(double) * (size_t)G + sizeof(int) * (size_t)2 * G;
}

template <int W, bool HITS, bool DERIVS, bool RAGGED>
int launch_stream(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets, int B, int m,
                  double dt, double *traj, const SampleExtras &x, int G) {
    size_t lds = stream_lds_bytes(W, G, m);
    while (G > 1 && lds > 150 * 1024) lds = stream_lds_bytes(W, --G, m);
    auto kern = minsnap_sample_stream_kernel<W, HITS, DERIVS, RAGGED>;
    if (lds > 64 * 1024) UAVAC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // chunk starts on 128-byte lines: (address / 8 + 11 * rho) % 16 == 0 for chunk starts at rows rho (mod 64) -> rho = 13 a % 16
    const int a = (int)((reinterpret_cast<uintptr_t>(traj) >> 3) & 15);
    const int rho = (13 * a) & 15;
    const int phase = (64 - rho) & 63;
    const int groups = (B + G - 1) / G;
    hipLaunchKernelGGL(kern, dim3(groups), dim3(64 * W), lds, ctx->stream, coeffs, seg_rows, row_offsets, B, m, dt, traj, x.aabb,
                       x.hit, x.yaw_dense, x.jerk, x.snap, x.capacity_rows, ctx->d_flags, x.first_yaw, x.seg_offsets, G, phase);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

template <int W>
int launch_stream_w(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets, int B, int m,
                    double dt, double *traj, const SampleExtras &x, int G) {
    const bool hits = x.aabb && x.hit, derivs = x.jerk || x.snap, ragged = x.seg_offsets != nullptr;
    if (ragged) {
        if (hits) return launch_stream<W, true, false, true>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, G);
        return launch_stream<W, false, false, true>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, G);
    }
    if (hits && derivs) return launch_stream<W, true, true, false>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, G);
    if (hits) return launch_stream<W, true, false, false>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, G);
    if (derivs) return launch_stream<W, false, true, false>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, G);
    return launch_stream<W, false, false, false>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, G);
}

}  // namespace

int uavac_launch_sample_stream(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets, int B,
                               int m, double dt, double *traj, const SampleExtras &x, int waves, int group) {
    if (group < 1) group = 1;
    if (waves == 2) return launch_stream_w<2>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, group);
    if (waves == 4) return launch_stream_w<4>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, group);
    if (waves == 16) return launch_stream_w<16>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, group);
    return launch_stream_w<8>(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x, group);
}
