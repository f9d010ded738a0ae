/*
 * uavac.h -- C ABI of libuavac.so: batched minimum-snap planning and cascaded
 * control + 6-DoF rollout for fleets of independent quadrotors on AMD MI355X
 * (gfx950).  Hand-written HIP kernels behind plain pointers and sizes.
 *
 * The reference (Mdhvince/UAV-Autonomous-control) has no FFI layer: its boundary
 * is a set of Python classes.  Each entry point below names the reference
 * interface it replaces (paths relative to the upstream repository root); the
 * Python binding a maintainer adds on the reference side is in INTEGRATION.md.
 *
 * Conventions
 *   - all floating point is IEEE fp64; NED world frame, FRD body frame;
 *   - every function returns UAVAC_OK (0) or a negative UAVAC_E* code; the text
 *     of the last failure is available from uavac_last_error(ctx);
 *   - a ctx owns one HIP stream (or borrows the caller's, see
 *     uavac_set_stream) and is not thread-safe; distinct ctxs are independent;
 *   - functions with the _dev suffix take DEVICE pointers, enqueue on the ctx
 *     stream and return without synchronising; the un-suffixed twins take HOST
 *     pointers, stage through device scratch and are synchronous on return;
 *   - nothing here ever falls back to a CPU path: without a usable GPU
 *     uavac_create fails with UAVAC_EHIP.
 */
#ifndef UAVAC_H
#define UAVAC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVAC_VERSION 310 /* 0.3.1: rows-free planning chain (uavac_minsnap_plan_dev with traj = NULL), uavac_minsnap_first_yaw_dev;
                             0.3.0: plan gather, log pitch, row offsets from row counts; 0.2.0: ground-plane fields, istate has 4 rows */

#define UAVAC_OK 0
#define UAVAC_EINVAL (-1)    /* bad shape / size / null pointer                    */
#define UAVAC_ENONFINITE (-2)/* non-finite waypoint, velocity or dt                */
#define UAVAC_EHIP (-3)      /* HIP runtime error (text in uavac_last_error)       */
#define UAVAC_ESINGULAR (-4) /* a mission's knot system is singular (e.g. repeated waypoint) */
#define UAVAC_ENOMEM (-5)
#define UAVAC_ECOMM (-6)     /* RCCL error (text in uavac_last_error)              */
#define UAVAC_ETOOLCHAIN (-7)/* uavac_create's self-check (first context of a process): the atan2 of the device math library this build
                                linked no longer has the bits its sampler's heading() reproduces -- heading() (csrc/minsnap_yaw.h) must
                                be re-derived for that library; rebuilding alone reproduces the failure (text on stderr) */

#define UAVAC_MAX_SEGMENTS 64   /* m, segments per mission                          */
#define UAVAC_TRAJ_COLS 11      /* x y z vx vy vz ax ay az yaw spline_id: minimum_snap.py:122-123 */
#define UAVAC_STATE_ROWS 30     /* see uavac_control_* below                        */
#define UAVAC_ISTATE_ROWS 4
#define UAVAC_CMD_COLS 12

typedef struct uavac_ctx uavac_ctx;

/* Vehicle constants, limits and controller gains.
 * Replaces the attributes of uav_ac/quadrotor/quad.py:11-86 (Quad.__init__) that the
 * controller reads, with the values models/lab_course.xml:3,9-13,100,116 provides. */
typedef struct uavac_vehicle {
    double g;               /* gravity [m/s^2]                      quad.py:39  */
    double dt;              /* inner (dynamics / motor) step [s]    quad.py:40  */
    double dt_outer;        /* CascadedController.dt = dt * inner_per_outer  main.py:97-98 */
    double mass;
    double inertia[3];      /* diagonal body inertia                */
    double arm;             /* roll/pitch lever arm of each rotor   */
    double kf;              /* rotor speed^2 -> thrust coefficient  */
    double kappa;           /* rotor reaction torque / thrust       */
    double min_thrust, max_thrust;      /* per rotor [N]             */
    double tau_rise, tau_fall;          /* motor time constants [s]  */
    double max_ascent, max_descent, max_speed_xy, max_horiz_accel, max_tilt; /* flight_limits */
    double kp_xy, kd_xy, kp_z, kd_z, ki_z;                /* quad.py:65-67 */
    double kp_roll, kp_pitch, kp_yaw, kp_p, kp_q, kp_r;   /* quad.py:68-73 */
    int32_t inner_per_outer;            /* config.ini:2 `frequency`  */
    int32_t ground;                     /* 0: free flight (default).  1: a horizontal ground plane the body can rest on
                                         * and take off from -- the on-ground start of the reference's scene
                                         * (lab_course.xml:34,98: plane at z = 0, body box of half height 0.02 m
                                         * starting 1 mm above it).  BUILD-DEFINED contact, not MuJoCo's solver: a
                                         * normal acceleration toward the critically damped reference
                                         * -2 vz / tc - r / tc^2 (r = penetration of the body's lowest point) whenever
                                         * that pushes up harder than free flight does; no friction, no contact
                                         * torque.  Contact FORCES are therefore not comparable with MuJoCo's. */
    double ground_z;                    /* NED z of the plane (0 = the reference scene)                      */
    double ground_clearance;            /* body centre above its lowest point (half height of geom "body") */
    double ground_timeconst;            /* tc: MuJoCo's default solref time constant, 0.02 s               */
} uavac_vehicle;
/* Ground bookkeeping bits in istate row 3 (only touched when V->ground != 0), after
 * MujocoSimulation._record_collisions (mujoco_sim.py:220-230): */
#define UAVAC_GROUND_IN_CONTACT 1     /* the body touches the plane after this tick (`has_collision`)      */
#define UAVAC_GROUND_TAKEN_OFF 2      /* sticky: height >= UAVAC_TAKEOFF_HEIGHT has been reached            */
#define UAVAC_GROUND_HIT_AFTER_TAKEOFF 4 /* sticky: contact after take-off (`collision_detected`)            */
#define UAVAC_TAKEOFF_HEIGHT 0.1      /* TAKEOFF_HEIGHT, mujoco_sim.py:17                                   */

/* ---- context ------------------------------------------------------------------ */
int uavac_version(void);
/* device_id < 0: use the current HIP device. */
int uavac_create(uavac_ctx **out, int device_id);
void uavac_destroy(uavac_ctx *ctx);
const char *uavac_last_error(const uavac_ctx *ctx);
/* Borrow a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  NULL is a
 * valid handle: HIP's legacy default stream (what torch uses unless told otherwise).
 * uavac_reset_stream goes back to the ctx-owned (non-blocking) stream. */
int uavac_set_stream(uavac_ctx *ctx, void *hip_stream);
int uavac_reset_stream(uavac_ctx *ctx);
int uavac_synchronize(uavac_ctx *ctx);
/* The HIP device the ctx was created for.  Every entry point makes that device current for its own
 * duration and restores the caller's: a ctx for GPU 1 works while GPU 0 is the thread's device. */
int uavac_device(const uavac_ctx *ctx);
/* Name and template arguments <compute waves, store waves, state log, command log, obstacle test,
 * plan-fed> of the rollout kernel the ctx launched last ("" before the first): what a profile of
 * the same call will show.  Diagnostics for benchmarks; the string lives in the ctx. */
const char *uavac_last_rollout_kernel(const uavac_ctx *ctx);
/* Vector registers per lane of that kernel in the loaded code object (hipFuncGetAttributes; 0 = unknown).  The rollout
 * kernels must stay within 256: two wavefronts of the kernel share a SIMD (compute + store wave), and at 258 a launch ran
 * 1.96 ms instead of 1.27 ms.  __graft_entry__.build() refuses a build that crosses the line; this is the same number seen
 * from the running process. */
int uavac_last_rollout_vgprs(const uavac_ctx *ctx);
/* "libuavac <version>; gfx950; HIP <x.y.z>; <compiler version>" of the build (static string). */
const char *uavac_build_info(void);
/* Which physical GPU the ctx runs on: "uuid=<32 hex digits>;pci=<domain:bus:device.function>;name=<gcnArchName>" into buf
 * (NUL-terminated, truncated to n bytes).  The multi-GPU bench gathers one per rank: eight ranks must name eight devices. */
int uavac_device_identity(uavac_ctx *ctx, char *buf, int n);
/* The shader clock the chip holds while other work runs: enqueues ONE wavefront on the ctx's stream that stamps
 * s_memtime (shader cycles) and s_memrealtime (100 MHz) , sleeps until `window_us` of real time have passed and stamps again;
 * stamps [4] (DEVICE memory, int64) = {cycles0, real0, cycles1, real1}.  Clock = (cycles1 - cycles0) / (real1 - real0) x 100 MHz
 * (MI355X_MICROARCH.md, DVFS give-back (6)).  Bind the ctx to a side stream (uavac_set_stream) to run it BESIDE the kernels of
 * interest; it holds one wave slot and issues nothing but s_sleep.  1 <= window_us <= 1 000 000. */
int uavac_clock_probe_dev(uavac_ctx *ctx, int window_us, int64_t *stamps);
/* Tuning knobs; results never depend on them (tested bit for bit).  "rollout_align": 1 (default) =
 * precede a rollout launch that writes a log by an empty kernel of the same workgroup shape (one
 * compute + one store wave), which makes the hardware place one wave of each kind on every SIMD
 * whatever ran before; 0 = do not.  "yaw_group": 1, 4, 8 (default) or 16 = how
 * many 64-row chunks of the sampler's dense yaw column are written together.  "late_handover": -1
 * (default: chosen per launch), 0, 1 = when the compute wave hands a tick's log values to the store
 * wave.  "lds_pad": extra LDS bytes per rollout workgroup (caps the workgroups a CU takes; 0).
 * "cu_balance": 1 (default) = a logged rollout that needs two or three workgroups on every CU sizes their LDS so
 * that no CU takes more of them than its even share (the dispatcher otherwise gives some CUs three and
 * some one where two each would do), 0 = off.
 * "solve_order" is NOT a tuning knob -- it selects the elimination order of the coefficient solve and with
 * it the rounding: 1 (default) = two-ended, two lanes per mission that meet at the middle knot
 * (csrc/minsnap_solve_tw.hip); 0 = one-ended (csrc/minsnap_solve_bt.hip, rounds 1-4), kept as the
 * cross-check; the two agree to ~5e-14 relative on the coefficients.  -1 (opt-in) = the launcher picks the faster of the two by
 * (m, B) -- one-ended for uniform batches of m <= 8 from 48 missions per SIMD on -- at the price that a mission's last bits then
 * depend on the size of the batch it is planned in (the default keeps a shard's coefficients equal to the whole job's).
 * "idle_waves": -1 (default: chosen per launch), 0, 1 = a placeholder wave between the compute and the
 * store wave of every rollout workgroup, which lets two workgroups on a CU occupy all four SIMDs
 * (16 385 .. 32 768 UAVs).  "coeff_dma": -1 (default: chosen per launch), 0, 1, 2 = the plan-fed rollout's mode: 0 the compute
 * wave evaluates target rows and reloads a segment's coefficients through registers on the spot; 1 the same with the
 * coefficients arriving by LDS-DMA an outer tick ahead; 2 the second (store) wave owns the cursor and evaluates the rows in
 * its idle time (kernels that have one; inner_per_outer >= 7).  Same bits in every mode.  "solve_park": -1 (default: chosen per launch), 0, 1 = the coefficient solve parks its forward sweep in the HBM
 * workspace / in LDS (when (m - 1) x 14 KB fit; same bits); "solve_lanes": -1 (default: chosen per launch), 64, 32, 16 = lanes
 * of a wavefront of the solve that carry (half of) a mission -- the two-ended solve gives a mission two lanes: 32, 16, 8 missions
 * per wavefront -- (fewer = more wavefronts for the same batch; same bits); "solve_keep": -1
 * (default: chosen per launch), 0, 1 = the solve of a uniform batch keeps the first five knots' blocks of its forward sweep in
 * registers instead of the workspace (same bits).  "sampler_waves": 4 (default), 2, 8, 16 = wavefronts per workgroup of the
 * chunk-streaming sampler, "sampler_group": 1 (default) .. 64 = consecutive missions per workgroup;
 * "sampler_waves" 1 = the one-wave-per-mission sampler (same rows bit for bit; faster into some row
 * buffers, slower into most: DESIGN K2).
 * Defaults from the environment (UAVAC_ROLLOUT_ALIGN, UAVAC_YAW_GROUP, UAVAC_SAMPLER_WAVES, UAVAC_SAMPLER_GROUP) at uavac_create.
 * ONE option is not a tuning knob but part of the log layout: "log_pitch" = P doubles per log row,
 * 0 (default) = B.  With P >= B the rollouts write state_log [K][13][P] and cmd_log [K][12][P]
 * (columns B .. P-1 are never touched).  Rows of a multiple of 16 doubles start on 128-byte lines
 * whatever B is: B = 65 534 with P = 65 536 streams like B = 65 536, with P = B at half that rate. */
int uavac_set_option(uavac_ctx *ctx, const char *name, int value);
/* The _dev planning entry points report data-dependent failures through sticky device-side flags
 * instead of synchronising: flags[0] non-finite segment duration, flags[1] singular knot system,
 * flags[2] trajectory buffer too small (uavac_minsnap_plan_dev), flags[3] a mission with more than
 * 2^31-1 rows.  This call synchronises the stream, returns them and clears them.  The host-pointer
 * twins clear the flags on entry and turn them into UAVAC_E* return codes themselves. */
int uavac_take_flags(uavac_ctx *ctx, int32_t flags[4]);
/* Fill *V with the laboratory vehicle (lab_course.xml) and the gains of quad.py:42-73. */
void uavac_vehicle_default(uavac_vehicle *V);

/* ---- planning -----------------------------------------------------------------
 * Batched drop-in for uav_ac/planning/minimum_snap.py MinimumSnap._generate_trajectory
 * (:97-124) on B independent missions of m segments each (obstacles=None path).
 *
 *   wp          [B][m+1][3]  waypoints
 *   times       [B][m]       segment durations            (_generate_time_per_spline :311-321)
 *   seg_rows    [B][m] i32   rows sampled per segment = len(np.arange(0, T, dt))    (:104)
 *   row_offsets [B+1]  i64   exclusive prefix sum of the per-mission row totals
 *   coeffs      [B][8m][3]   polynomial coefficients, ascending powers, per spline  (:153)
 *   traj        [row_offsets[B]][11]  rows of all missions back to back (:122-123)
 */
int uavac_minsnap_row_counts_dev(uavac_ctx *ctx, const double *wp, int B, int m, double velocity,
                                 double dt, double *times, int32_t *seg_rows, int64_t *row_offsets);
/* Solves the joint minimum-snap QP of _compute_spline_parameters (:138-153) per mission.
 * status [B] i32 (may be NULL): 0 ok, 1 singular system. */
int uavac_minsnap_solve_dev(uavac_ctx *ctx, const double *wp, const double *times, int B, int m,
                            double *coeffs, int32_t *status);
/* The same QP through the other device solver: wave-per-mission banded LU with partial pivoting in LDS
 * (uavac_minsnap_solve_dev is the lane-per-mission block-Thomas recurrence).  Kept as an independent,
 * pivoted cross-check; ~50x slower. */
int uavac_minsnap_solve_banded_dev(uavac_ctx *ctx, const double *wp, const double *times, int B, int m,
                                   double *coeffs, int32_t *status);
/* Sampler (:100-119) + yaw scan (_calculate_yaws :126-136). */
int uavac_minsnap_sample_dev(uavac_ctx *ctx, const double *coeffs, const double *times,
                             const int32_t *seg_rows, const int64_t *row_offsets, int B, int m,
                             double dt, double *traj);
/* uavac_minsnap_sample_dev that also writes the yaw column on its own: yaw[row_offsets[B]]
 * (yaw[i] == traj[i][9]); input of uavac_control_rollout_plan_dev. */
int uavac_minsnap_sample_yaw_dev(uavac_ctx *ctx, const double *coeffs, const double *times,
                                 const int32_t *seg_rows, const int64_t *row_offsets, int B, int m,
                                 double dt, double *traj, double *yaw);
/* Sampler that also reports, per mission and spline, whether any sampled position lies inside the cuboid
 * aabb[6] = xmin xmax ymin ymax zmin zmax (device pointer; inclusive test of is_collision_cuboid :327-357):
 * hit [B][m] i32 (0/1).  This is the collision scan of _generate_collision_free_trajectory (:81-87), fused
 * into the sampling pass; the midpoint insertion that follows (:91-92) is host logic. */
int uavac_minsnap_sample_hits_dev(uavac_ctx *ctx, const double *coeffs, const double *times,
                                  const int32_t *seg_rows, const int64_t *row_offsets, int B, int m,
                                  double dt, double *traj, const double *aabb, int32_t *hit);

/* The sampler with every optional output: yaw [rows] (or NULL) as above; first_yaw [B] (or NULL) =
 * the heading of each mission's first row that has one (what the rows before it take; 0 when no
 * row has one) -- all a plan-fed rollout needs to scan the yaw itself; jerk / snap [rows][3] (or
 * NULL) = polynom(8, 3, t) @ coeffs and polynom(8, 4, t) @ coeffs, the two samples the reference
 * evaluates in comments only (minimum_snap.py:111-112,118-119).  They are separate arrays: the
 * (N, 11) row layout of get_trajectory() never changes. */
int uavac_minsnap_sample_derivs_dev(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows,
                                    const int64_t *row_offsets, int B, int m, double dt, double *traj,
                                    double *yaw, double *first_yaw, double *jerk, double *snap);
/* The whole planning chain of MinimumSnap.get_trajectory() (obstacles=None; minimum_snap.py:59-61,
 * 97-124) enqueued by ONE call: times + row counts, row offsets, coefficient solve, sampler (+ yaw
 * column when yaw != NULL, + the missions' first headings when first_yaw != NULL) -- four kernel
 * launches back to back, no host code in between.  The row
 * buffer must have been sized by the caller: traj holds traj_capacity_rows rows (yaw as many
 * values); when the plan needs more it is refused AS A WHOLE and flag 2 is raised (uavac_take_flags):
 * times, seg_rows, row_offsets, coeffs, status, traj, yaw and first_yaw all keep what they held, so the
 * previous plan stays consistent and flyable (times / row counts / offsets are computed into ctx scratch
 * and copied into the caller's arrays by a device-side commit only when the rows fit).
 * Typical use: size the buffers once with uavac_minsnap_row_counts_dev, then re-plan in place.
 *
 * ROWS-FREE form: traj == NULL (then yaw must be NULL and traj_capacity_rows is ignored).  The chain is times + row counts,
 * row offsets, coefficient solve and -- when first_yaw != NULL -- uavac_minsnap_first_yaw_dev: everything a plan-fed rollout
 * (uavac_control_rollout_plan_dev with yaw == NULL) and uavac_gather_plan_dev need, and not one of the 88-byte rows.  For ranks
 * of a multi-GPU job whose trajectories are sampled where they are wanted (the root of the final gather re-samples them from
 * the gathered plan, bit-identical): sampling them on the peer as well was the same work done twice.  Nothing can be refused,
 * so times / seg_rows / row_offsets are written in place. */
int uavac_minsnap_plan_dev(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt,
                           double *times, int32_t *seg_rows, int64_t *row_offsets, double *coeffs,
                           int32_t *status, double *traj, int64_t traj_capacity_rows, double *yaw,
                           double *first_yaw);

/* first_yaw [B] of a solved plan WITHOUT sampling its rows: the heading of each mission's first sample with |v_xy| >= 1e-3
 * (MinimumSnap._calculate_yaws, minimum_snap.py:126-136: the rows before it take that heading; 0 when no sample has one) --
 * bit for bit what the sampler writes into first_yaw, from coeffs [B][8m][3] and seg_rows [B][m] alone.  One wavefront per
 * mission walks the rows 64 at a time from row 0 and stops at the first valid one.  seg_offsets [B+1] (device) != NULL: a
 * ragged batch (coeffs [S][8][3], seg_rows [S] back to back, m = the largest segment count); NULL: uniform. */
int uavac_minsnap_first_yaw_dev(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *seg_offsets,
                                int B, int m, double dt, double *first_yaw);

/* row_offsets [B+1] from per-segment row counts that exist already -- the second half of
 * uavac_minsnap_row_counts_dev on its own, for a plan whose seg_rows [B][m] came from elsewhere (the
 * peers' plans after uavac_gather_plan_dev): exclusive prefix sum of the per-mission totals
 * (sum of len(np.arange(0, T, dt)) over the mission's splines, minimum_snap.py:104). */
int uavac_minsnap_row_offsets_dev(uavac_ctx *ctx, const int32_t *seg_rows, int B, int m, int64_t *row_offsets);
/* The same for a ragged batch (seg_rows [S] back to back, mission b's at seg_offsets[b] .. seg_offsets[b+1]). */
int uavac_minsnap_row_offsets_ragged_dev(uavac_ctx *ctx, const int32_t *seg_rows, const int64_t *seg_offsets, int B,
                                         int max_m, int64_t *row_offsets);

/* Ragged batches: missions with different numbers of waypoints in one call -- what a fleet of MinimumSnap objects with
 * paths of different lengths is (minimum_snap.py:13-57 takes any path), and what the obstacle loop (:63-95) produces as
 * soon as one mission has received a midpoint.  Mission b has m_b = seg_offsets[b+1] - seg_offsets[b] segments
 * (1 <= m_b <= max_m <= UAVAC_MAX_SEGMENTS; seg_offsets [B+1] i64, device, seg_offsets[0] = 0) and m_b + 1 waypoints.
 * Everything per-segment lies back to back in mission order: wp [S + B][3] (mission b starts at waypoint
 * seg_offsets[b] + b), times / seg_rows / hit [S], coeffs [S][8][3], S = seg_offsets[B] = total_segments.
 * Same kernels, same arithmetic: mission b's outputs equal those of a uniform call on it alone, bit for bit.
 * A segment count outside 1 .. max_m raises sticky flag 0 (uavac_take_flags) and is clamped.  The sampler takes the
 * capacity of the row buffer like uavac_minsnap_plan_dev (flag 2 and nothing written when it is too small; < 0: not
 * checked), an optional cuboid + hit flags (both or neither) and optional first_yaw [B]. */
/* The same on HOST buffers, the whole chain in one call (seg_offsets on the host; validated: 1 .. UAVAC_MAX_SEGMENTS
 * segments each): times [S] and coeffs [S][8][3] optional (NULL), row_offsets [B+1] always; traj NULL (or
 * traj_capacity_rows < row_offsets[B]: UAVAC_EINVAL after everything else was produced) skips the rows -- call once with
 * traj = NULL to learn row_offsets[B], allocate, call again. */
int uavac_minsnap_plan_ragged(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B, double velocity,
                              double dt, double *times, int64_t *row_offsets, double *coeffs, double *traj,
                              int64_t traj_capacity_rows);
int uavac_minsnap_row_counts_ragged_dev(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B,
                                        int max_m, double velocity, double dt, double *times,
                                        int32_t *seg_rows, int64_t *row_offsets);
int uavac_minsnap_solve_ragged_dev(uavac_ctx *ctx, const double *wp, const double *times,
                                   const int64_t *seg_offsets, int B, int max_m, double *coeffs,
                                   int32_t *status);
int uavac_minsnap_sample_ragged_dev(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows,
                                    const int64_t *seg_offsets, const int64_t *row_offsets, int B, int max_m,
                                    int64_t total_segments, double dt, double *traj,
                                    int64_t traj_capacity_rows, const double *aabb, int32_t *hit,
                                    double *first_yaw);
/* ONE ROUND of the obstacle loop of MinimumSnap._generate_collision_free_trajectory (minimum_snap.py:81-93) for B ragged
 * missions against one cuboid aabb[6], entirely on the device (no rows are stored: inside the loop only the hit flags
 * matter; sample the final waypoints once with uavac_minsnap_*_ragged_dev afterwards):
 *   - the missions with active[b] != 0 are planned (durations, row counts, coefficient solve) -- the others are left alone;
 *   - every spline of an active mission with a sample inside the cuboid (inclusive test, is_collision_cuboid :327-357,
 *     on the very positions the sampler would store) gets the midpoint of its two waypoints inserted before its end
 *     waypoint (insert_midpoints_at_indexes :359-391): wp_out / seg_offsets_out are the waypoint arrays of the next round
 *     (all B missions; untouched ones are copied), laid out like wp / seg_offsets;
 *   - an active mission without a hit is clean for this cuboid: active[b] = 0; one that would outgrow UAVAC_MAX_SEGMENTS is
 *     left as it is: active[b] = 0, overflow[b] = 1; a mission that received midpoints: touched[b] = 1, stays active;
 *   - counters [4] i32 (device): missions still active, missions that outgrew UAVAC_MAX_SEGMENTS in this round, the largest segment
 *     count after the round (a valid max_m for the next one), the segment total of the batch after the round.
 * max_m >= every mission's segment count, <= UAVAC_MAX_SEGMENTS.  SCRATCH (device, caller-owned, sizes TRUSTED -- the call
 * cannot check them; S_cap >= the segment total before the round): times [S_cap], seg_rows [S_cap] i32, row_offsets [B+1],
 * coeffs [S_cap][8][3], hit [S_cap] i32.  They are scratch in the strict sense: what they hold after the call is valid only
 * for the missions that were active IN THIS ROUND (waves without an active mission skip the solve while the segment offsets
 * move from round to round): do not sample from them -- plan the final waypoints with uavac_minsnap_*_ragged_dev.  wp_out
 * holds S_cap_next + B waypoints with S_cap_next >= the segment total after the round (at most twice the one before, and
 * never more than B * UAVAC_MAX_SEGMENTS).  A singular knot system (repeated waypoint) raises sticky flag 1; its mission's
 * positions are NaN, inside no cuboid: it leaves the loop as if collision-free -- check uavac_take_flags. */
int uavac_minsnap_obstacle_round_dev(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B, int max_m,
                                     double velocity, double dt, const double *aabb, int32_t *active, int32_t *overflow,
                                     int32_t *touched, double *wp_out, int64_t *seg_offsets_out, int32_t *counters,
                                     double *times, int32_t *seg_rows, int64_t *row_offsets, double *coeffs, int32_t *hit);

/* The whole obstacle loop on HOST buffers: MinimumSnap(path_b, obstacles, velocity, dt) up to the final waypoint list, for B
 * ragged missions (wp [S + B][3], seg_offsets [B+1] as above) against n_cuboids cuboids [n_cuboids][6] visited in order
 * (minimum_snap.py:72-93: earlier ones are not re-checked).  Unlike the reference's, the loop is bounded: at most
 * max_iterations + 1 rounds per obstacle; a mission that runs out, or would outgrow UAVAC_MAX_SEGMENTS, keeps the waypoints
 * it has and is reported in converged [B] (0; may be NULL).  recheck_passes > 0 sweeps the missions that received midpoints
 * over the obstacle list again (beyond the reference).  Output: the final waypoints wp_out [seg_offsets_out[B] + B][3]
 * (wp_capacity rows available; B * (UAVAC_MAX_SEGMENTS + 1) always suffice; UAVAC_EINVAL with seg_offsets_out filled in when
 * it is too small) -- sample them with uavac_minsnap_plan_ragged to get get_trajectory()'s rows.  UAVAC_ESINGULAR (outputs
 * complete): some mission has a repeated waypoint; its collision scan was void. */
int uavac_minsnap_obstacle_waypoints(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B, double velocity,
                                     double dt, const double *cuboids, int n_cuboids, int max_iterations, int recheck_passes,
                                     double *wp_out, int64_t wp_capacity, int64_t *seg_offsets_out, int32_t *converged);

/* MinimumSnap._calculate_yaws (minimum_snap.py:126-136) on its own, for B independent velocity
 * sequences of any length: sequence b = rows [offsets[b], offsets[b+1]) of velocities[.][3] (only
 * vx, vy are read); yaws[offsets[B]].  Headings of rows with |v_xy| >= 1e-3, np.unwrap over those,
 * hold-last-valid, leading rows take the first valid heading, all zeros when none is valid. */
int uavac_yaw_scan_dev(uavac_ctx *ctx, const double *velocities, const int64_t *offsets, int B,
                       double *yaws);
/* Host twin for one sequence: velocities [n][3] -> yaws [n]. */
int uavac_yaw_scan(uavac_ctx *ctx, const double *velocities, int64_t n, double *yaws);

/* Host-pointer twins (synchronous).  They stage through device scratch and a pinned ping-pong
 * buffer that the ctx keeps between calls (no allocation per call once warm). */
int uavac_minsnap_row_counts(uavac_ctx *ctx, const double *wp, int B, int m, double velocity,
                             double dt, double *times, int32_t *seg_rows, int64_t *row_offsets);
int uavac_minsnap_solve(uavac_ctx *ctx, const double *wp, int B, int m, double velocity,
                        double *coeffs, double *times);
int uavac_minsnap_sample(uavac_ctx *ctx, const double *coeffs, const double *times, int B, int m,
                         double dt, const int64_t *row_offsets, double *traj);

/* ---- control ------------------------------------------------------------------
 * Batched drop-in for one tick of uav_ac/main.py TrajectoryController.step (:37-61)
 * [CascadedController: uav_ac/control/controller.py:26-168; rotor allocation and motor
 * lag: uav_ac/quadrotor/quad.py:88-122] followed by MujocoSimulation.step
 * (uav_ac/simulation/mujoco_sim.py:144-151) restricted to free flight (rotor wrench
 * :232-251 + semi-implicit Euler free-body step; no contacts).
 *
 * State is struct-of-arrays over the batch (lane b = UAV b):
 *   state  [30][B] f64: rows 0-12  X = x y z | q0 q1 q2 q3 | vx vy vz | p q r   (quad.py:75-80)
 *                       rows 13-16 omega, rows 17-20 omega_command               (quad.py:83-86)
 *                       row  21    altitude integral error                       (controller.py:20)
 *                       row  22    thrust_cmd, rows 23-25 pqr_cmd                (main.py:26-27)
 *                       rows 26-29 the yaw scan a plan-fed rollout carries when it is given no dense yaw column:
 *                                  the row it stands before, heading seen (0/1), last heading, unwrap sum
 *                                  (minimum_snap.py:126-136; zero after uavac_state_init, maintained by the kernel)
 *   istate [4][B]  i32: trajectory_index, inner_step (main.py:24-25), collided (sticky obstacle flag),
 *                       ground bookkeeping bits (UAVAC_GROUND_*; stays 0 in free flight)
 *   traj / row_offsets: as produced by uavac_minsnap_sample (UAV b follows mission b).
 */
/* X = [position, identity attitude, rest]; rotors at hover speed if hover != 0, else 0;
 * controller memory cleared (TrajectoryController.reset, main.py:29-35).
 * positions [B][3] may be NULL (origin). */
int uavac_state_init_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *positions, int B,
                         int hover, double *state, int32_t *istate);
/* K ticks fused in one launch.  state_log [K][13][B] (X after every tick) or NULL;
 * cmd_log [K][12][B] (thrust_cmd, pqr_cmd, omega_command, omega after the controller part
 * of every tick) or NULL; aabbs [n_obs][6] = xmin xmax ymin ymax zmin zmax (inclusive test of
 * minimum_snap.py:327-357, evaluated on the position after every tick) or NULL.
 * Any B is accepted; the logs stream at full rate when their rows start on 128-byte lines: B a
 * multiple of 16, or uavac_set_option(ctx, "log_pitch", P) with P >= B a multiple of 16 -- the logs
 * are then [K][13][P] and [K][12][P]. */
int uavac_control_rollout_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj,
                              const int64_t *row_offsets, double *state, int32_t *istate, int B,
                              int K, double *state_log, double *cmd_log, const double *aabbs,
                              int n_obs);
/* The same rollout fed by the plan instead of the sampled rows: the target row of every outer tick is
 * evaluated inside the kernel from the coefficients of the UAV's current segment (bit-identical to
 * the sampler's rows).  The yaw -- the one column that is a scan over all earlier rows -- comes
 * either from the dense yaw column uavac_minsnap_sample_yaw_dev writes (yaw [row_offsets[B]]), or,
 * with yaw == NULL, from the scan the vehicle carries itself in state rows 26-29 (it visits its rows
 * in order), seeded with first_yaw [B] from the sampler; bit-identical either way, and in the second
 * form no yaw byte is read or written.  coeffs [B][8m][3], seg_rows [B][m], row_offsets [B+1], dt
 * as given to the sampler.  Same results as uavac_control_rollout_dev on the sampled trajectory,
 * without its HBM read traffic (80 B per UAV and outer tick, fetched as 128-byte lines).  A cursor
 * (istate row 0) that the caller moved is honoured: the carried scan is rebuilt from row 0 at launch. */
int uavac_control_rollout_plan_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *coeffs,
                                   const int32_t *seg_rows, const int64_t *row_offsets,
                                   const double *yaw, const double *first_yaw, int m, double dt,
                                   double *state, int32_t *istate, int B, int K, double *state_log,
                                   double *cmd_log, const double *aabbs, int n_obs);

/* The same for a ragged batch (uavac_minsnap_*_ragged_dev): coeffs [S][8][3] and seg_rows [S] back to back, mission b's
 * segments at seg_offsets[b] .. seg_offsets[b+1]; the vehicles scan the yaw themselves from first_yaw [B]. */
int uavac_control_rollout_plan_ragged_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *coeffs,
                                          const int32_t *seg_rows, const int64_t *seg_offsets,
                                          const int64_t *row_offsets, const double *first_yaw, int max_m,
                                          double dt, double *state, int32_t *istate, int B, int K,
                                          double *state_log, double *cmd_log, const double *aabbs, int n_obs);
/* One tick (K = 1, no logs): the literal drop-in of tc.step() + simulation.step(). */
int uavac_control_step_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj,
                           const int64_t *row_offsets, double *state, int32_t *istate, int B);

/* Host-pointer twins (synchronous). */
int uavac_state_init(uavac_ctx *ctx, const uavac_vehicle *V, const double *positions, int B,
                     int hover, double *state, int32_t *istate);
int uavac_control_rollout(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj,
                          const int64_t *row_offsets, double *state, int32_t *istate, int B, int K,
                          double *state_log, double *cmd_log, const double *aabbs, int n_obs);

/* The two halves of a tick on their own, for callers that own the simulation loop the way
 * uav_ac/main.py does (controller callback + external physics):
 *   uavac_controller_tick = TrajectoryController.step (main.py:37-61): outer loop every
 *       inner_per_outer-th call, body-rate loop, allocation, motor lag; X is read, not advanced;
 *   uavac_dynamics_step   = MujocoSimulation.step in free flight (mujoco_sim.py:144-151,232-251):
 *       advances X from the current rotor speeds; aabbs/istate optional (sticky flag in istate row 2). */
int uavac_controller_tick_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj,
                              const int64_t *row_offsets, double *state, int32_t *istate, int B);
int uavac_dynamics_step_dev(uavac_ctx *ctx, const uavac_vehicle *V, double *state, int32_t *istate,
                            int B, const double *aabbs, int n_obs);
int uavac_controller_tick(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj,
                          const int64_t *row_offsets, double *state, int32_t *istate, int B);
int uavac_dynamics_step(uavac_ctx *ctx, const uavac_vehicle *V, double *state, int32_t *istate, int B,
                        const double *aabbs, int n_obs);

/* ---- resident tick-by-tick session ----------------------------------------------
 * For callers that own the loop like uav_ac/main.py:113-118 does -- `tc.step(); simulation.step()`
 * once per inner tick, with host code reading and writing quad.X / quad.omega in between.  The
 * trajectory rows of the B UAVs are uploaded once at creation; the state lives in pinned host memory
 * that is mapped into the device and that the tick kernels read and write IN PLACE:
 * uavac_pilot_state() -> [30][B] f64, uavac_pilot_istate() -> [4][B] i32 (layouts above), valid until
 * uavac_pilot_destroy.  uavac_pilot_tick runs the controller half (uavac_controller_tick), the
 * vehicle half (uavac_dynamics_step) or both on that state and returns when the results are
 * visible to the host: one kernel launch (or two) and one stream synchronisation per call, no
 * copies, no allocation.  The host may edit the state between calls (that is what the reference's
 * tests do to quad.X). */
typedef struct uavac_pilot uavac_pilot;
#define UAVAC_PILOT_CONTROLLER 1
#define UAVAC_PILOT_DYNAMICS 2
int uavac_pilot_create(uavac_ctx *ctx, const double *traj, const int64_t *row_offsets, int B,
                       uavac_pilot **out);                 /* traj / row_offsets: HOST pointers */
void uavac_pilot_destroy(uavac_pilot *pilot);
double *uavac_pilot_state(uavac_pilot *pilot);
int32_t *uavac_pilot_istate(uavac_pilot *pilot);
int uavac_pilot_set_obstacles(uavac_pilot *pilot, const double *aabbs, int n_obs);   /* HOST [n_obs][6] */
int uavac_pilot_tick(uavac_pilot *pilot, const uavac_vehicle *V, int what);

/* ---- per-function probes ------------------------------------------------------
 * One stage of the control law at a time on small array-of-struct batches (host pointers,
 * synchronous).  They run the very __device__ functions the fused rollout inlines, so the
 * reference's unit-level known answers (tests/unit/control/test_controller.py,
 * tests/unit/quadrotor/test_quad.py) can be replayed against the HIP path, and they back the
 * single-UAV facade classes.  mask selects which optional inputs override computed values.
 *
 * outer: in [B][41] = X(13) | R(9) | target row(11) | integral | thrust_in | bxy_in(2) |
 *                     euler_in(phi,theta,psi) | q_cmd_in
 *        out[B][21] = R(9) [Quad.R, quad.py:129-155] | phi theta psi [quad.py:189-213] |
 *                     thrust, integral' [altitude, controller.py:26-56] | bxy(2) [lateral :58-97] |
 *                     p_c q_c [roll_pitch_controller :132-154] | pqr_cmd(3) [reduced_attitude :99-113]
 * inner: in [B][24] = X(13) | pqr_cmd(3) | thrust_cmd | omega(4) | moment_in(3)
 *        out[B][15] = moment(3) [body_rate_controller :115-130] | rotor forces(4)
 *                     [_allocate_rotor_forces, quad.py:105-122] | omega_command(4), omega'(4)
 *                     [set_propeller_speed, quad.py:88-103] */
#define UAVAC_PROBE_OUTER_IN 41
#define UAVAC_PROBE_OUTER_OUT 21
#define UAVAC_PROBE_INNER_IN 24
#define UAVAC_PROBE_INNER_OUT 15
#define UAVAC_PROBE_USE_R 1       /* altitude / roll_pitch take the given rot_mat                */
#define UAVAC_PROBE_USE_THRUST 2  /* lateral takes thrust_in instead of altitude's result        */
#define UAVAC_PROBE_USE_BXY 4     /* roll_pitch takes bxy_in instead of lateral's result         */
#define UAVAC_PROBE_USE_EULER 8   /* yaw_controller takes euler_in (duck-typed quad)             */
#define UAVAC_PROBE_USE_QCMD 16   /* yaw_controller takes q_cmd_in instead of roll_pitch's q_c   */
#define UAVAC_PROBE_USE_MOMENT 1  /* (inner) allocation takes moment_in instead of body_rate's   */
int uavac_probe_outer(uavac_ctx *ctx, const uavac_vehicle *V, const double *in, int B, int mask, double *out);
int uavac_probe_inner(uavac_ctx *ctx, const uavac_vehicle *V, const double *in, int B, int mask, double *out);
/* The sampler's heading of a velocity sample (csrc/minsnap_yaw.h: the device library's atan2 with the instruction count cut)
 * beside the library's own atan2(y, x) on the same operands; DEVICE pointers, n values each, asynchronous.  They must agree bit
 * for bit (np.arctan2 in MinimumSnap._calculate_yaws, minimum_snap.py:131, is matched to <= 1e-5 by either). */
int uavac_probe_heading_dev(uavac_ctx *ctx, const double *y, const double *x, int64_t n, double *heading, double *library);

/* ---------------------------------------------------------------------------------------------
 * RRT* planner -- replaces uav_ac/planning/rrt.py.
 *
 * B independent planning problems, one wavefront each, all sharing step (= max_distance),
 * max_iter and the obstacle list.  Random numbers stay on the host: samples[B][max_iter][3] holds,
 * per iteration, the node RRTStar._generate_random_node (rrt.py:118-127) returned (NumPy's legacy
 * global generator, one uniform() for the goal bias then three for the coordinates).
 * With cap = max_iter + 1, per problem:
 *   nodes       [cap][3]  `all_nodes` in insertion order (entry 0 = round(start, 2)); rows past
 *                          counts[0] are zero
 *   canon       [cap]     first entry with bit-identical coordinates = the dict key of the entry
 *   parent      [cap]     indexed by key: key of tree[key] at the end of run(), -1 = no such key
 *   best_parent [cap]     the same for `best_tree` (stored at the last improvement)
 *   best_path   [cap][3]  `best_path`, start -> goal, counts[4] rows
 *   counts      [6]       n_nodes, iterations begun, status, entries when best_tree was stored,
 *                          best_path rows, dynamic_it_counter
 *   best_cost             path_cost(best_path) (rrt.py:84-91); +inf without a path
 * status: UAVAC_RRT_OK, or what the reference raises -- UAVAC_RRT_NO_PATH ("No path found",
 * rrt.py:72-73), UAVAC_RRT_COST_INCREASED (:55-56), UAVAC_RRT_KEY_ERROR (a dict lookup failed).
 * Per-problem failures are reported in counts, not in the return value. */
#define UAVAC_RRT_OK 0
#define UAVAC_RRT_NO_PATH 1
#define UAVAC_RRT_COST_INCREASED 2
#define UAVAC_RRT_KEY_ERROR 3
int uavac_rrt_star_dev(uavac_ctx *ctx, const double *start, const double *goal, int B, double step,
                       int max_iter, const double *samples, const double *cuboids, int n_obs,
                       double *nodes, int32_t *canon, int32_t *parent, int32_t *best_parent,
                       double *best_path, int32_t *counts, double *best_cost);
int uavac_rrt_star(uavac_ctx *ctx, const double *start, const double *goal, int B, double step,
                   int max_iter, const double *samples, const double *cuboids, int n_obs,
                   double *nodes, int32_t *canon, int32_t *parent, int32_t *best_parent,
                   double *best_path, int32_t *counts, double *best_cost);
/* E candidate edges p0[e] -> p1[e] against n_obs cuboids [xmin xmax ymin ymax zmin zmax]:
 * hit[e] = 1 when the segment crosses any of them (RRTStar._is_valid_connection is False),
 * slab test of RRTStar._segment_intersects_cuboid (rrt.py:231-274). */
int uavac_rrt_segment_hits_dev(uavac_ctx *ctx, const double *p0, const double *p1, int E,
                               const double *cuboids, int n_obs, int32_t *hit);
int uavac_rrt_segment_hits(uavac_ctx *ctx, const double *p0, const double *p1, int E,
                           const double *cuboids, int n_obs, int32_t *hit);
/* out[e] = np.linalg.norm(p1[e] - p0[e]) as _find_nearest_node / _find_valid_neighbors /
 * _cost_to_come / path_cost take it (rrt.py:84-91,129-173).  p1 is [E][3], or one point [3] for
 * every edge when p1_is_single != 0. */
int uavac_rrt_edge_lengths_dev(uavac_ctx *ctx, const double *p0, const double *p1,
                               int p1_is_single, int E, double *out);
int uavac_rrt_edge_lengths(uavac_ctx *ctx, const double *p0, const double *p1, int p1_is_single,
                           int E, double *out);
/* The nodes RRTStar._generate_random_node (rrt.py:118-127) returns in n consecutive calls after
 * np.random.seed(seeds[b]) -- NumPy's legacy MT19937 stream reproduced on the GPU, bit for bit --
 * for B problems: samples[B][n][3] (what uavac_rrt_star takes); consumed[B][n] (or NULL) = doubles
 * of the stream used up to and including draw i.  limits_lw / limits_up are HOST pointers to 3
 * doubles (the space limits shared by the batch); goals[B][3] must already be rounded to 0.01. */
int uavac_rrt_draw_nodes_dev(uavac_ctx *ctx, const uint32_t *seeds, const double *goals, int B, int n,
                             const double *limits_lw, const double *limits_up, double epsilon,
                             double *samples, int64_t *consumed);
/* RRTStar.simplify_path (rrt.py:93-116) for B paths at once: paths[B][cap][3] with lens[B] waypoints
 * each (e.g. best_path / counts[4] of uavac_rrt_star) -> out_paths[B][cap][3], out_lens[B]: from
 * each kept waypoint the farthest one with a clear direct connection is kept next. */
int uavac_rrt_simplify_dev(uavac_ctx *ctx, const double *paths, const int32_t *lens, int B, int cap,
                           const double *cuboids, int n_obs, double *out_paths, int32_t *out_lens);
int uavac_rrt_simplify(uavac_ctx *ctx, const double *paths, const int32_t *lens, int B, int cap,
                       const double *cuboids, int n_obs, double *out_paths, int32_t *out_lens);
/* RRTStar.path_cost (rrt.py:84-91): the edge lengths of the polyline path[n][3] summed in path
 * order (also the accumulation of _cost_to_come, :163-173, on the node -> start chain). */
int uavac_rrt_path_cost_dev(uavac_ctx *ctx, const double *path, int n, double *cost);
int uavac_rrt_path_cost(uavac_ctx *ctx, const double *path, int n, double *cost);
/* RRTStar._adapt_random_node_position (rrt.py:140-148) for E (sample, nearest node) pairs:
 * out[e] = sample[e] when within step of nearest[e], else the rounded point at step from it. */
int uavac_rrt_steer_dev(uavac_ctx *ctx, const double *sample, const double *nearest, int E,
                        double step, double *out);
int uavac_rrt_steer(uavac_ctx *ctx, const double *sample, const double *nearest, int E, double step,
                    double *out);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU.  One process per GPU, one ctx per process; missions are sharded by
 * contiguous index blocks and every entry point above works on its own shard -- nothing is
 * exchanged while planning or flying.  The ONE exchange of the path is the final gather of the
 * ragged row blocks (trajectories, or any [n][row_elems] f64 block) to a root rank over RCCL:
 * ncclGroupStart + ncclRecv per peer on the root / ncclSend on the peers + ncclGroupEnd
 * (rccl.h:700,722,923), i.e. every peer uses its own direct xGMI link into the root at once.
 * (No reference counterpart: upstream plans and flies one mission per process, main.py:87-120.)
 *
 * nccl_comm is an ncclComm_t (passed as void* so that this header does not need rccl.h): either the
 * caller's own communicator or one made by uavac_comm_init_rank.  Bootstrap: rank 0 calls
 * uavac_comm_unique_id and hands the 128 bytes to the other ranks by any side channel (environment,
 * file, torch.distributed store); then every rank calls uavac_comm_init_rank (collective). */
#define UAVAC_COMM_ID_BYTES 128
int uavac_comm_unique_id(uavac_ctx *ctx, char id[UAVAC_COMM_ID_BYTES]);
int uavac_comm_init_rank(uavac_ctx *ctx, const char id[UAVAC_COMM_ID_BYTES], int world, int rank,
                         void **nccl_comm);
int uavac_comm_destroy(uavac_ctx *ctx, void *nccl_comm);
int uavac_comm_abort(uavac_ctx *ctx, void *nccl_comm);      /* after a failure or a timeout */
int uavac_comm_shape(uavac_ctx *ctx, void *nccl_comm, int *world, int *rank);
/* Every rank contributes its row count; counts [world] (HOST) receives all of them
 * (ncclAllGather of one int64 per rank; synchronous). */
int uavac_gather_counts(uavac_ctx *ctx, void *nccl_comm, int64_t n_rows, int64_t *counts);
/* Gather: rank r's rows [counts[r]][row_elems] (device) land at row offset sum(counts[:r]) of out
 * (device, root only: [sum(counts)][row_elems]; ignored elsewhere).  counts is the HOST array of
 * uavac_gather_counts.  Enqueued on the ctx stream; uavac_comm_finish synchronises the stream and
 * reports asynchronous RCCL errors. */
int uavac_gather_rows_dev(uavac_ctx *ctx, void *nccl_comm, const double *rows, int64_t n_rows,
                          int row_elems, const int64_t *counts, int root, double *out);
/* Gather of the PLAN instead of the rows.  The rows of a mission are a deterministic, bit-reproducible function of its
 * coefficients and per-segment row counts (uavac_minsnap_sample_dev), 204 B per segment against ~10 KB of rows: the peers
 * send coeffs [n_segments][8][3], times [n_segments] (optional: NULL on every rank or on none) and seg_rows [n_segments]
 * to the root in ONE grouped launch (three ncclSend per peer / three ncclRecv per peer on the root), the root then calls
 * uavac_minsnap_row_offsets_dev + uavac_minsnap_sample_dev on the gathered plan and holds the very rows the peers hold,
 * written at its own HBM rate instead of arriving at the rate of its xGMI links (BASELINE config 4: 0.4 GB instead of
 * 20.8 GB through the root's seven links).  seg_counts is the HOST array of uavac_gather_counts(n_segments); rank r's block
 * lands at segment offset sum(seg_counts[:r]) of the *_out arrays (device, root only).  Enqueued on the ctx stream;
 * uavac_comm_finish synchronises.  (No reference counterpart, like the row gather.) */
int uavac_gather_plan_dev(uavac_ctx *ctx, void *nccl_comm, const double *coeffs, const double *times,
                          const int32_t *seg_rows, int64_t n_segments, const int64_t *seg_counts, int root,
                          double *coeffs_out, double *times_out, int32_t *seg_rows_out);
/* One PART of uavac_gather_plan_dev, for a gather that is PIPELINED with the root's re-sampling: segments
 * [part_first[r], part_first[r] + part_counts[r]) of rank r's block travel and land on the root where the gather of the whole
 * blocks puts them (segment offset sum(seg_counts[:r]) + part_first[r] of the *_out arrays).  The pointers are those of the rank's
 * WHOLE block / of the root's whole output arrays; an array travels when its pointer is non-NULL, on every rank alike (the root:
 * when its *_out pointer is non-NULL) -- e.g. first times + seg_rows of the whole blocks (12 B per segment: the root can then lay
 * out every mission's rows), then the coefficients (192 B per segment) in a few parts, each of which the root samples while the
 * next one arrives.  seg_counts / part_first / part_counts [world] are HOST arrays, the same on every rank.  Enqueued on the ctx
 * stream like the other gathers; RCCL executes the operations of one communicator in the order they were issued, whatever
 * streams they were issued on.  (No reference counterpart.) */
int uavac_gather_plan_part_dev(uavac_ctx *ctx, void *nccl_comm, const double *coeffs, const double *times,
                               const int32_t *seg_rows, const int64_t *seg_counts, const int64_t *part_first,
                               const int64_t *part_counts, int root, double *coeffs_out, double *times_out,
                               int32_t *seg_rows_out);
/* NCCL_VERSION_CODE of the rccl.h this library was built with, and ncclGetVersion() of the RCCL the process has
 * mapped (in a Python process: the one bundled with torch).  uavac_comm_init_rank refuses a different major version
 * or a runtime older than 2.10; only entry points stable since then are used, so the minor versions may differ. */
int uavac_comm_versions(int *built_with, int *runtime);
int uavac_comm_finish(uavac_ctx *ctx, void *nccl_comm);
/* Self-test of the transport on a single GPU: src [n] -> dst [n] through ncclSend + ncclRecv with
 * this rank as its own peer, grouped exactly like the gather (enqueued; then uavac_comm_finish). */
int uavac_comm_loopback_dev(uavac_ctx *ctx, void *nccl_comm, const double *src, double *dst, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* UAVAC_H */
