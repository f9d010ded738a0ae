// Internal declarations shared by the libuavac.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "uavac.h"

struct uavac_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;    // == own_stream unless borrowed
    int32_t *d_flags = nullptr;      // [4] device-side error flags (0: non-finite input, 1: singular system)
    int32_t *d_totals = nullptr;     // scratch for the row-count scan
    size_t totals_cap = 0;
    double *d_ws = nullptr;          // block-Thomas workspace [m-1][28][B]
    size_t ws_cap = 0;               // in doubles
    char *d_plan = nullptr;          // uavac_minsnap_plan_dev: times / seg_rows / row_offsets before they are committed
    size_t plan_cap = 0;             // in bytes
    // Device scratch of the host-pointer twins and small internal temporaries: one arena, grown on demand and kept,
    // handed out by bump allocation inside one entry point (uavac_arena_reserve, then uavac_arena_take).  Reuse across
    // calls is ordered by the ctx stream.
    char *d_arena = nullptr;
    size_t arena_cap = 0, arena_top = 0;
    // Pinned host staging (hipHostMalloc) of the host-pointer twins: two halves used as a ping-pong pipeline.
    char *h_pin = nullptr;
    size_t pin_cap = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    std::string err;
    int sampler_waves = 4;           // tuning: wavefronts per workgroup of the sampler: 2, 4, 8, 16 (minsnap_sample_stream.hip); 1 = one wave per mission (minsnap_sample.hip)
    int sampler_group = 1;           // tuning: consecutive missions per workgroup of the streaming sampler
    int yaw_group = 8;               // tuning: chunks of the sampler's dense yaw column that leave together (1, 4, 8, 16)
    int rollout_align = 1;           // tuning: launch the 2-wave aligner kernel before a logged launch of shape 1
    int late_handover = -1;          // tuning: -1 = the launcher picks per launch; 0 / 1 = slab handed over at the end of the tick / a third of a tick later
    int coeff_dma = -1;              // tuning: -1 = the launcher picks per launch; 0 / 1 / 2 = the plan-fed rollout's PMODE (control_rollout.hip)
    int solve_order = 1;             // which elimination order solves: 1 = two-ended, two lanes per mission (minsnap_solve_tw.hip; the default), 0 = one-ended (minsnap_solve_bt.hip: other rounding, kept as the cross-check)
    int solve_park = -1;             // tuning: where the block-Thomas solve parks its forward sweep: -1 = the launcher picks, 0 = HBM workspace, 1 = LDS (when it fits)
    int solve_lanes = -1;            // tuning: lanes per wave of the block-Thomas solve that carry a mission (64 / 32 / 16; -1 = the launcher picks)
    int solve_keep = -1;             // tuning: the solve of a uniform batch keeps its first five knots' parked blocks in registers: -1 = the launcher picks, 0 / 1
    int idle_waves = -1;             // tuning: placeholder wave between compute and store wave (0 / 1); -1 = the launcher picks
    int cu_balance = 1;              // tuning: a logged rollout below a full chip asks for as much LDS per workgroup as keeps a CU from taking more workgroups than its even share (0: off)
    int lds_pad = 0;                 // tuning: extra dynamic LDS per rollout workgroup (bytes): caps the workgroups a CU takes
    int64_t log_pitch = 0;           // doubles per row of the rollout's logs; 0 = B (option "log_pitch")
    int n_simds = 1024;              // SIMDs of the device (4 per CU): the logged rollout launches one workgroup per SIMD at most
    std::string last_rollout;        // name and template arguments of the rollout kernel launched last (diagnostics)
    int last_rollout_vgprs = 0;      // ... and its vector registers per lane (hipFuncGetAttributes); 0 = unknown
    int launch_rc = UAVAC_OK;        // set by a rollout launcher that had to give up before the launch (text in err)
};

// Every entry point that launches, allocates or copies runs with the ctx's device current and puts the caller's
// device back on return: a ctx created for GPU 1 must work while GPU 0 is the thread's current device.
struct uavac_device_guard {
    int prev = -1;
    bool switched = false;
    explicit uavac_device_guard(const uavac_ctx *ctx) {
        if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~uavac_device_guard() {
        if (switched) (void)hipSetDevice(prev);
    }
    uavac_device_guard(const uavac_device_guard &) = delete;
    uavac_device_guard &operator=(const uavac_device_guard &) = delete;
};
#define UAVAC_ENTER(ctx)                \
    if (!(ctx)) return UAVAC_EINVAL;    \
    uavac_device_guard uavac_guard_(ctx)

#define UAVAC_HIP(ctx, call)                                                                     \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return UAVAC_EHIP;                                                                   \
        }                                                                                        \
    } while (0)

static inline int uavac_fail(uavac_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// blockIdx -> tile such that the blocks of one XCD own a contiguous range of tiles; a bijection on [0, n) for every
// n.  Workgroups go to the chip's 8 XCDs round-robin (blockIdx % 8); when consecutive tiles are consecutive in memory,
// this gives every XCD's L2 one contiguous span to stream to HBM instead of every eighth piece.  Store-only probes:
// the sampler's write pattern 5.6 -> 6.7 TB/s, the rollout's log 5.4 -> 6.3 TB/s (tools/sampler_store_probe.hip,
// tools/log_layout_probe.hip).
#ifdef __HIPCC__
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a release/acquire fence for GLOBAL memory:
// the compiler puts `s_waitcnt vmcnt(0)` in front of it, so a wave that has stores (or a prefetch) in flight waits
// for their completion -- a full HBM round trip -- at every barrier.  The kernels here hand data over through LDS
// and never read back what they store to HBM, so they wait for LDS (lgkmcnt) alone and leave vector-memory
// operations in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Same for a workgroup that is a single wavefront: LDS operations of one wave execute in order, so only the
// compiler needs to be told not to move memory accesses across this point.
__device__ __forceinline__ void lds_wave_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ int xcd_contiguous(int block, int n) {
    const int x = block & 7, q = n >> 3, r = n & 7;
    return x * q + (x < r ? x : r) + (block >> 3);
}
#endif

// uavac_create's self-check (control_probe.hip): the sampler's heading() against the device library's atan2, which the rollout's
// yaw scan uses, on 2^16 + 144 operand pairs; *mismatches = pairs whose bits differ.
int uavac_heading_selfcheck(uavac_ctx *ctx, int *mismatches);

// Device scratch arena (uavac_api.hip).  reserve() makes room for `bytes` in total (synchronises the stream and
// reallocates when it has to grow) and rewinds the arena; take() hands out 256-byte aligned pieces of it.
int uavac_arena_reserve(uavac_ctx *ctx, size_t bytes);
void *uavac_arena_take(uavac_ctx *ctx, size_t bytes);
static inline size_t uavac_arena_size(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
// One small temporary: reserve + take.
int uavac_scratch(uavac_ctx *ctx, size_t bytes, void **out);
// Pinned staging: at least `bytes` of page-locked host memory in ctx->h_pin.
int uavac_pin_reserve(uavac_ctx *ctx, size_t bytes);
// Pageable host buffer <-> device through the pinned ping-pong buffer, ordered on the ctx stream (uavac_api.hip).
// h2d returns once the source has been read; d2h once the destination holds the data.
int uavac_h2d(uavac_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int uavac_d2h(uavac_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

// Kernel-side view of uavac_vehicle with the per-call constants hoisted on the host.
struct VehK {
    double g, dt, dt_outer, mass, inv_mass;
    double I[3], inv_I[3];
    double arm, inv_arm, kappa, inv_kappa, kf, inv_kf;
    double min_thrust, max_thrust, c_min, c_max;
    double resp_rise, resp_fall;     // 1 - exp(-dt/tau): quad.py:102
    double max_ascent, max_descent, max_speed_xy, max_horiz_accel, max_tilt;
    double kp_xy, kd_xy, kp_z, kd_z, ki_z, kp_roll, kp_pitch, kp_yaw;
    double ikp[3];                   // I * kp_pqr: controller.py:128
    double hover_omega;
    double ground_zc, ground_k, ground_b, ground_z;   // contact starts at pz > ground_zc = ground_z - clearance; 1/tc^2, 2/tc
    // the non-inline fp64 literals of the per-tick path (control_law.h): smallest normal, the large-angle threshold on
    // (|w| dt / 2)^2, the Taylor coefficients of cos and sinc, 3/8 and the threshold of the renormalisation series.  They
    // travel with the vehicle constants so that the rollout can put them into vector registers ONCE, ahead of its tick loop
    double lit_tiny, lit_h2_small, lit_c8, lit_c6, lit_c4, lit_s9, lit_s7, lit_s5, lit_s3, lit_375, lit_e_small;
    int F;
    int ground;
};

VehK uavac_make_vehk(const uavac_vehicle &V);
int uavac_check_vehicle(uavac_ctx *ctx, const uavac_vehicle *V);

// launchers (one per .hip file)
// seg_offsets (device, [B+1]) != NULL: ragged batch -- mission b has seg_offsets[b+1] - seg_offsets[b] segments (1 .. m,
// m = the batch's maximum); waypoints, times, row counts, coefficients and hit flags lie back to back
int uavac_launch_row_counts(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt,
                            double *times, int32_t *seg_rows, int64_t *row_offsets, const int64_t *seg_offsets = nullptr);
// guard_rows (device, may be NULL): the launch does nothing when *guard_rows > guard_capacity (a refused planning chain)
int uavac_launch_solve_bt(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                          int32_t *status, const int64_t *seg_offsets = nullptr, const int64_t *guard_rows = nullptr,
                          int64_t guard_capacity = 0, const int32_t *active = nullptr);
// the two-ended form (minsnap_solve_tw.hip): two lanes per mission; uavac_launch_solve_bt hands over to it unless ctx->solve_order == 0
int uavac_launch_solve_tw(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                          int32_t *status, const int64_t *seg_offsets, const int64_t *guard_rows, int64_t guard_capacity,
                          const int32_t *active);
// One round of the obstacle loop on the device (minsnap_obstacles.hip): collision scan of the active missions' splines
// (no rows stored) + midpoint insertion into the next waypoint arrays
int uavac_launch_obstacle_scan_and_insert(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, const double *coeffs,
                                          const int32_t *seg_rows, int B, int max_m, double dt, const double *aabb,
                                          int32_t *active, int32_t *overflow, int32_t *touched, int32_t *hit, double *wp_out,
                                          int64_t *seg_offsets_out, int32_t *counters);
// row_offsets [B+1] from per-segment row counts that exist already (minsnap_solve.hip)
int uavac_launch_row_offsets(uavac_ctx *ctx, const int32_t *seg_rows, int B, int m, int64_t *row_offsets,
                             const int64_t *seg_offsets = nullptr);
// exclusive prefix sum of ctx->d_totals [B] (filled by the caller's kernel together with the tile sums) -> out [B+1] i64
int uavac_launch_totals_scan(uavac_ctx *ctx, int B, int64_t *out);
int uavac_ensure_totals(uavac_ctx *ctx, int B, int32_t **totals, int64_t **tile_sums);
// times / seg_rows / row_offsets from scratch into the caller's arrays unless row_offsets_s[B] > capacity_rows
int uavac_launch_plan_commit(uavac_ctx *ctx, const double *times_s, const int32_t *seg_rows_s, const int64_t *row_offsets_s,
                             int B, int m, int64_t capacity_rows, double *times, int32_t *seg_rows, int64_t *row_offsets);
int uavac_launch_solve(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                       int32_t *status);
// Optional outputs / inputs of the sampler (minsnap_sample.hip)
struct SampleExtras {
    const double *aabb = nullptr;    // [6] cuboid of the collision scan, with
    int32_t *hit = nullptr;          // [B][m] hit flags
    double *yaw_dense = nullptr;     // [rows] the yaw column on its own
    double *first_yaw = nullptr;     // [B] heading of each mission's first row that has one (0 when none has)
    double *jerk = nullptr;          // [rows][3]
    double *snap = nullptr;          // [rows][3]
    int64_t capacity_rows = -1;      // rows the trajectory buffer holds; < 0: not checked
    const int64_t *seg_offsets = nullptr;   // [B+1] ragged batch (see uavac_launch_row_counts); m is then the maximum
    int64_t total_segments = -1;            // ... and its number of segments (sizes the hit flags)
};
int uavac_launch_sample(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets,
                        int B, int m, double dt, double *traj, const SampleExtras &x);
// the same rows from workgroups of `waves` (4, 8, 16) wavefronts that stream the 64-row chunks of `group` consecutive missions
// in address order (minsnap_sample_stream.hip)
int uavac_launch_sample_stream(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets, int B,
                               int m, double dt, double *traj, const SampleExtras &x, int waves, int group);
int uavac_launch_yaw_scan(uavac_ctx *ctx, const double *velocities, const int64_t *offsets, int B, double *yaws);
// first_yaw [B] without the rows (minsnap_first_yaw.hip): what the sampler writes there, from the coefficients and row counts alone
int uavac_launch_first_yaw(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *seg_offsets, int B, int m,
                           double dt, double *first_yaw);
int uavac_launch_state_init(uavac_ctx *ctx, const VehK &V, const double *positions, int B, int hover, double *state,
                            int32_t *istate);
// What the rollout needs to evaluate target rows itself instead of reading them (control_rollout.hip, POLY)
struct PlanRef {
    const double *coeffs = nullptr;      // [B][8m][3]
    const int32_t *seg_rows = nullptr;   // [B][m]
    const double *yaw = nullptr;         // [row_offsets[B]] dense yaw column of the sampler, or NULL: the rollout scans the yaw
    const double *first_yaw = nullptr;   // [B] (with yaw == NULL) heading the rows before a mission's first heading take
    double dt = 0.0;
    int m = 0;
    const int64_t *seg_offsets = nullptr;   // [B+1] ragged batch: coeffs [S][8][3], seg_rows [S] back to back, m = the maximum
};
int uavac_launch_rollout(uavac_ctx *ctx, const VehK &V, const double *traj, const int64_t *row_offsets, double *state,
                         int32_t *istate, int B, int K, double *state_log, double *cmd_log, const double *aabbs,
                         int n_obs, const PlanRef *plan = nullptr);
