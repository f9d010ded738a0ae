// Fused cascaded controller + rotor allocation + motor lag + NED 6-DoF free-body step (gfx950).
//
// One lane per UAV, K ticks per launch, all controller / vehicle state in registers.
// Per tick (upstream paths):
//   every F-th tick: TrajectoryController._update_outer_loop      uav_ac/main.py:47-61
//       Quad.R / quat_to_rot                                      uav_ac/quadrotor/quad.py:129-155
//       CascadedController.altitude                               uav_ac/control/controller.py:26-56
//       CascadedController.lateral                                :58-97
//       CascadedController.roll_pitch_controller / yaw_controller :132-168 (Euler angles quad.py:189-213)
//   every tick:  CascadedController.body_rate_controller          controller.py:115-130
//                Quad._allocate_rotor_forces / set_propeller_speed quad.py:88-122
//                MujocoSimulation._apply_rotor_forces + mj_step    uav_ac/simulation/mujoco_sim.py:144-151,232-251
//                (free joint, Euler integrator, no contacts; restated in NED/FRD)
//
// HBM traffic per UAV tick: 104 B of state log (13 f64, coalesced) + one 88 B trajectory row every F ticks.
//
// Workgroup = one compute wave + (when a log is requested) one STORE wave.
//   Measured on MI355X at B = 65 536, per 1 000 ticks: arithmetic alone 0.76 ms; the 6.8 GB log stream alone,
//   in this [K][13][B] pattern, 1.25-1.4 ms (tools/store_wave_probe.hip; a contiguous fill reaches 6.5 TB/s).
//   A wave that issues a vector store stalls until the CU's store path has taken it, so with the stores in
//   the compute waves a tick costs arithmetic + store time (1.74-1.8 ms), and neither start-time stagger,
//   nor spreading the stores through the tick, nor removing the compiler's per-tick vmcnt(0) changes that
//   much: the storing wave itself is what blocks.  Here the compute wave never stores: each tick it drops
//   its 13 (+12) log values into an LDS slab (ds_write, non-blocking), one workgroup barrier hands the slab
//   to the store wave, and that wave streams it to HBM while the compute wave is already in the next tick
//   (two slabs, ping-pong): 1.66 ms, i.e. the store stream (slowed ~25 % by the concurrent arithmetic) now
//   sets the pace and the arithmetic hides under it.

#include "control_law.h"
#include "minsnap_eval.h"
#include "minsnap_yaw.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <string>

namespace {

using namespace uavac_dev;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- trajectory-row prefetch with a hand-placed wait ------------------------------------------------
// On gfx950 loads and stores share one in-order counter (vmcnt).  A row load the compiler knows to be in
// flight across ticks makes it put `s_waitcnt vmcnt(0)` at every join of the tick loop (its registers are
// loop-carried).  Issued from inline asm the load is invisible to that pass; the registers are touched by
// nothing until row_wait(), which every later read is data-dependent on.
// The operands are output-only ("=&v").  What makes that right is not the constraint but the BUILD CHECK: uav_ac/_buildcheck.py
// (run by __graft_entry__.build() and by the CPU tests) disassembles every row-fed variant and fails the build unless both issue
// sites load into the same twenty registers and no instruction reads or writes one of them between the loads and an
// `s_waitcnt vmcnt(0)` -- i.e. unless the loads land in the very registers row_wait() hands on.  (An output-only operand MAY be
// given a register of its own and copied into the carried one before the data has arrived; it happened once, NOTES R4-4, and
// the check is what catches it.  The "+v" form, which ties the destination to the carried register by construction, was measured
// in round 5: +1.5 % on every row-fed tick -- the twenty registers then stay live through the outer block -- and +2.7 % with an
// empty block in front to end their live range, tools/r05_asm_ab.sh; so the check carries the guarantee instead.)
struct RowRegs { u32x4 q[5]; };          // columns 0..9 of a row (x y z vx vy vz ax ay az yaw): 80 bytes

__device__ __forceinline__ void row_issue(RowRegs &r, const double *p) {
    asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                 "global_load_dwordx4 %1, %5, off offset:16\n\t"
                 "global_load_dwordx4 %2, %5, off offset:32\n\t"
                 "global_load_dwordx4 %3, %5, off offset:48\n\t"
                 "global_load_dwordx4 %4, %5, off offset:64"
                 : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.q[3]), "=&v"(r.q[4])
                 : "v"(p)
                 : "memory");
}
__device__ __forceinline__ void row_wait(RowRegs &r) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.q[3]), "+v"(r.q[4])::"memory");
}
template <class T> __device__ __forceinline__ void settle(T &x) { asm volatile("" : "+v"(x)); }

// Plan-fed rollout: the 24 coefficients of a lane's NEXT segment travel from HBM straight into the lane's column of the wave's
// LDS tile (LDS-DMA, global_load_lds_dwordx4: no registers, nothing to wait for here) -- issued when the cursor enters the
// segment, needed one outer tick (F ticks) later.  Loaded through registers on the spot they cost the whole load latency at
// every segment change of any lane of the wave (44 % of the outer ticks), and that latency grows with the chip's load: 1.1 k
// cycles at 8 192 UAVs, 2.3 k at 32 768, 3 k at 49 152 (NOTES R4-3, profiles/r04_tick_stamps_*.jsonl) -- it was the half-full chip's longer tick.
// Tile layout [12][64][2] doubles (minsnap_eval.h, STRIDE 0): instruction p moves doubles 2p, 2p + 1 of every active lane; lane l
// lands at M0 + offset + 16 l, and the instruction offset counts for the GLOBAL address and for the LDS address alike, so M0
// advances by 1024 - 16 per instruction.  Masked-off lanes keep their column (tools/scratch/ldsdma_gather_probe.hip).
// Like the row prefetch the loads are invisible to the compiler's vmcnt bookkeeping; `plan_loads_wait` covers them.
__device__ __forceinline__ void coeffs_dma(const double *src, unsigned tile_lds_bytes) {
    // M0 is a register the compiler reserves for itself: it takes no clobber for it ("inline asm clobber list contains reserved
    // registers: m0"), so the block saves it in a scalar register of its own and puts it back -- whatever the compiler keeps
    // there survives.  (The DMA instructions read M0 when they issue; the value may change behind them.)
    unsigned saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:16\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:32\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:48\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:64\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:80\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:96\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:112\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:128\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:144\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:160\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off offset:176\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "v"(src), "s"(tile_lds_bytes) : "memory", "scc");
}
// the row count of the segment AFTER the one just entered: needed a whole segment later
// ("+v": the load lands IN the loop-carried register; an output-only operand may get a register of its own and be copied
// into the carried one right away, before the data has arrived)
__device__ __forceinline__ void seg_rows_issue(int &r, const int32_t *p) {
    asm volatile("global_load_dword %0, %1, off" : "+v"(r) : "v"(p) : "memory");
}
__device__ __forceinline__ void plan_loads_wait(int &r) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(r)::"memory"); }
template <int N> __device__ __forceinline__ void store_wave_loads_wait(int &r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory"); }

__device__ __forceinline__ double row_col(const RowRegs &r, int c) {      // c is a compile-time constant
    const u32x4 q = r.q[c >> 1];
    u32x2 h;
    if (c & 1) { h.x = q.z; h.y = q.w; } else { h.x = q.x; h.y = q.y; }
    return __builtin_bit_cast(double, h);
}

// A store whose address is a wave-uniform 64-bit base in SGPRs plus a 32-bit byte offset per lane.  The store wave issues
// 13 (+12) of these per tick; with the lane folded into a 64-bit VGPR pointer every one of them needed a 64-bit vector
// add first, and those adds queue behind the compute wave's fp64 instructions on the SIMD's single vector issue port
// (each waits up to one fp64 instruction, ~7 cycles).  In this form the row addresses advance with scalar adds and the
// store wave issues no vector-ALU instruction per store.
__device__ __forceinline__ void store_uniform_base(double *base, unsigned lane_bytes, double v) {
    asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(lane_bytes), "v"(v), "s"(base) : "memory");
}

// The constants only the outer loop needs (gains, flight limits) are re-read from the kernel-argument
// segment inside the outer block instead of living in SGPRs across the whole tick loop: the per-tick path
// alone needs ~50 SGPRs of constants, both sets together overflow the 102 available and the overflow
// would be paid in v_readlane on every tick.  VehK is the kernel's first argument => offset 0.
__device__ __forceinline__ VehK outer_constants() {
    typedef const __attribute__((address_space(4))) VehK *kptr;
    kptr vp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(vp));                 // opaque: the scalar loads below stay inside the outer block
    VehK O;
    O.g = vp->g; O.dt_outer = vp->dt_outer; O.mass = vp->mass; O.c_min = vp->c_min; O.c_max = vp->c_max;
    O.max_ascent = vp->max_ascent; O.max_descent = vp->max_descent; O.max_speed_xy = vp->max_speed_xy;
    O.max_horiz_accel = vp->max_horiz_accel; O.max_tilt = vp->max_tilt;
    O.kp_xy = vp->kp_xy; O.kd_xy = vp->kd_xy; O.kp_z = vp->kp_z; O.kd_z = vp->kd_z; O.ki_z = vp->ki_z;
    O.kp_roll = vp->kp_roll; O.kp_pitch = vp->kp_pitch; O.kp_yaw = vp->kp_yaw;
    return O;
}

// CW: compute waves per workgroup (UAVs per workgroup = 64 CW).  LDS slab layout: [2][NR][64 CW] doubles,
// NR = 13 state rows (+ 12 command rows).
// POLY: the target row of an outer tick is not read from the sampled trajectory but evaluated from the 24
// coefficients of the UAV's current segment (the sampler's own function, bit for bit: minsnap_eval.h), with the
// yaw -- the one column that is a scan over all earlier rows -- taken from the sampler's dense yaw array 16 rows at
// a time.  Per UAV and outer tick that is 8 B of yaw plus 192 B of coefficients per ~110 rows instead of an 80-B
// row whose 128-B lines the log stream has evicted from L2 by the next outer tick: HBM read traffic per launch
// drops from 1.3 GB to 0.1 GB at B = 65 536.  Coefficients [24][64] and yaws [16][64] of a compute wave live in LDS.
constexpr int poly_tile_doubles(bool yawscan) { return (24 + (yawscan ? 0 : 16)) * 64; }    // no yaw slots when the kernel scans the yaw

// YAWSCAN (with POLY): the yaw of a target row -- the one column that is a scan over all earlier rows -- is not read from a
// dense column either: the vehicle visits its rows in order, one per outer tick, so it carries the scan itself (has a
// heading been seen, the last one, the running sum of np.unwrap's corrections: state rows 27-29, with row 26 saying
// which row they stand before) and evaluates minimum_snap.py:126-136 for the row at hand with the sampler's own
// functions (minsnap_yaw.h; the sampler sums the corrections in the same left-to-right order).  Rows before a mission's
// first heading take P.first_yaw[b].  A cursor that does not match the carried scan (a caller moved it, another kernel
// advanced it) is caught at launch and the scan is rebuilt from row 0.  No yaw bytes are read or written at all.
// (The heading here is the device library's atan2, the sampler's is uavac_yaw::heading(): the same operations in the same order by
// construction, with the library's version inlined by the compiler.  Both written as heading() the plan-fed kernels need 257-260
// vector registers -- one wave per SIMD -- so the agreement is CHECKED instead: uavac_create() runs both over 2^16 operand pairs
// and every special value and refuses to hand out a context (UAVAC_ETOOLCHAIN) when a single bit differs.)
template <int CW, int SW, bool LOG_STATE, bool LOG_CMD, bool AABB, bool POLY, bool GROUND, bool YAWSCAN, int PMODE = 0>
__global__ void __launch_bounds__((LOG_STATE || LOG_CMD || AABB) ? 64 * CW + 128 : 64 * CW)       // compute [+ placeholder] + store
control_rollout_kernel(const VehK V, const double *__restrict__ traj, const int64_t *__restrict__ row_offsets,
                       double *__restrict__ state, int32_t *__restrict__ istate, int B, int K,
                       double *__restrict__ state_log, double *__restrict__ cmd_log,
                       const double *__restrict__ aabbs, int n_obs, int n_tiles, size_t log_pitch, const PlanRef P,
                       int late_handover, int n_idle) {
    // WATCH: obstacles but no log at all -- the second wave exists all the same and only WATCHES: it takes the three
    // position values of every tick through the slab and tests them against obstacle bounds held in its registers.  In the
    // compute wave the same test cost 0.4 us per tick for four obstacles (a scalar-cache round trip per obstacle on the
    // one dependent instruction stream; bounds in lanes + v_readlane were slower still, and it has no registers to hold them).
    constexpr bool WATCH = AABB && !LOG_STATE && !LOG_CMD && CW == SW;
    constexpr bool LOGGING = LOG_STATE || LOG_CMD || WATCH;            // "a second wave takes a slab per tick"
    // PMODE (plan-fed kernels): who evaluates a target row and how a segment's coefficients reach the LDS tile.
    //   0  the compute wave; coefficients through registers on the spot (what a full chip without a second wave uses)
    //   1  the compute wave; coefficients by LDS-DMA an outer tick ahead (see coeffs_dma)
    //   2  the SECOND wave (TGW): it idles four fifths of every tick at the barrier, so it owns the trajectory cursor, evaluates
    //      the next target row (Horner, atan2, yaw scan: ~1 200 cycles per outer tick, 6-10 % of the compute wave's tick) while
    //      the compute wave flies the inner ticks, and hands the row over through a [10][64] LDS tile.
    constexpr bool ADMA = PMODE == 1;
    constexpr bool TGW = PMODE == 2;
    static_assert(!TGW || (POLY && LOGGING), "target rows by the second wave: plan-fed kernels that have one");
    constexpr int NU = 64 * CW;                                        // UAVs per workgroup
    constexpr int NR = (LOG_STATE ? 13 : 0) + (LOG_CMD ? UAVAC_CMD_COLS : 0) + (WATCH ? 3 : 0);
    constexpr int CMD0 = LOG_STATE ? 13 : 0;                           // first command row in a slab
    extern __shared__ double slab[];                                   // [2][NR][NU]
    const size_t sB = (size_t)B;
    // The logs are [K][13 | 12][log_pitch]: rows of log_pitch >= B doubles.  With log_pitch a multiple of 16 every row starts
    // on a 128-byte line whatever B is (B = 65 534 with pitch B ran at half the rate of 65 536: every 512-byte wave store
    // straddled two partially written lines).
    const size_t sP = log_pitch;
    // PERSISTENT TILES.  A tile = NU consecutive UAVs flown for the launch's K ticks.  The launch has at most one workgroup
    // per SIMD (the launcher caps the grid when a log is written); a batch with more tiles than that is walked by the same
    // workgroups, pass after pass, each workgroup moving on to its next tile as soon as its own K ticks are done -- no launch
    // boundary between the passes at which every SIMD would wait for the slowest one (round 2 issued one launch per 65 536
    // columns: 262 144 UAVs ran at 45 G steps/s against 52 G at 65 536).  Results do not depend on the split (tested).
    // XCD-aware tile order inside a pass (uavac_internal.h): every XCD owns one contiguous span of the pass's columns.
    const int grid = (int)gridDim.x;
    int kk = 0;                                    // ticks of earlier tiles: the slab ping-pong keeps alternating across tiles

    // PLACEHOLDER WAVES.  A CU deals the waves of a workgroup round its SIMDs in the order s, s+2, s+1, s+3 and starts the
    // NEXT workgroup one position later in that sequence (tools/census_detail.py).  Four [compute, store] workgroups on a CU
    // therefore end up one compute + one store wave per SIMD -- but TWO of them (B <= 32 768) put the second compute wave on
    // the SIMD of the first store wave and leave one SIMD idle.  With a wave between the two that ends at once --
    // [compute, placeholder, store] -- two workgroups occupy all four SIMDs with one wave each.  A wave that has ended no
    // longer takes part in the workgroup's barriers.
    if (LOGGING && n_idle > 0 && threadIdx.x >= NU && threadIdx.x < NU + 64 * n_idle) return;

    if (LOGGING && threadIdx.x >= NU) {
        // ------------------------------------------------------------------------------ store wave(s)
        static_assert(CW == 1 && SW == 1, "one compute + one store wave per 64-UAV tile");
        constexpr int QPL = 1;
        // (Measured and not kept, all bit-identical, tools/rollout_shapes.py, profiles/r03_rollout_shapes_*.jsonl: two or three
        // store waves per tile sharing a tick's 13 log rows -- 0.93 -> 0.92 ms per 1 000 ticks at B = 32 768, slower from
        // 40 960 up; two compute + two store waves per 128-UAV workgroup, which a CU deals one per SIMD -- slower at every
        // size, 0.87 against 0.77 ms even at B = 16 384: the per-tick barrier then couples two compute waves.)
        const int lane = threadIdx.x & 63;
        __builtin_amdgcn_s_setprio(3);            // few instructions, all on the critical store stream: issue first
        const unsigned lane_bytes = (unsigned)lane * 8u;
        // With a state log the per-tick obstacle test runs HERE, on the positions this wave is about to store, after
        // its stores have been issued: the compute wave's tick stays as short as without obstacles (the two stages
        // couple through one barrier per tick; lengthening the compute stage to the length of the store stage cost
        // 40 % at config 5), and the comparisons fill time in which this wave would wait for the store path anyway.
        constexpr bool AABB_HERE = AABB && (LOG_STATE || WATCH) && QPL == 1;
        // The first kBoxRegs obstacles live in vector registers for the whole launch (this wave has ~200 to spare: the
        // kernel's allocation is sized by the compute wave).  Fetched through uniform addresses they would be scalar
        // loads -- one s_load + s_waitcnt round trip per obstacle per TICK on the critical store stream.
        constexpr int kBoxRegs = 8;
        double box[kBoxRegs][6];
        if (AABB_HERE) {
            int zero = 0;
            asm volatile("" : "+v"(zero));            // a per-lane offset the compiler cannot see through: vector loads
#pragma unroll
            for (int o = 0; o < kBoxRegs; ++o)
#pragma unroll
                for (int j = 0; j < 6; ++j) box[o][j] = o < n_obs ? aabbs[6 * o + j + zero] : 0.0;
            // the loads have landed before the tick loop starts: a load the compiler still sees in flight at the loop
            // head costs an s_waitcnt vmcnt(0) at the first use in EVERY iteration, i.e. a wait for this wave's own stores
#pragma unroll
            for (int o = 0; o < kBoxRegs; ++o)
#pragma unroll
                for (int j = 0; j < 6; ++j) settle(box[o][j]);
        }
      for (int tile0 = 0; tile0 < n_tiles; tile0 += grid, kk += K) {
        const int n_here = min(grid, n_tiles - tile0);
        if ((int)blockIdx.x >= n_here) break;
        const int col0 = (tile0 + xcd_contiguous(blockIdx.x, n_here)) * NU;
        const bool full = col0 + NU <= B;          // every column of this workgroup exists: no per-store mask
        const bool mine = col0 + lane < B;
        int coll = (AABB_HERE && mine) ? istate[2 * sB + col0 + lane] : 0;
        if (AABB_HERE) settle(coll);               // landed before the tick loop (see above)
        // ---- TGW: this wave owns the trajectory cursor of its 64 UAVs (the same arithmetic, in the same order, as the compute
        // wave's in the other modes: E = evaluate the row under the cursor, A = advance; E0 A0 E1 A1 ... -- here E runs one
        // outer tick AHEAD and its update of the yaw scan stays pending until the compute wave has consumed the row, so that
        // what a launch saves is exactly what the other modes save)
        double *tgt = slab + (size_t)2 * NR * NU + (size_t)CW * poly_tile_doubles(YAWSCAN) + lane;        // tgt[j * 64], j = 0 .. 9
        double *tile2 = slab + (size_t)2 * NR * NU;                                                       // the coefficient tile, [12][64][2]
        double *cf2 = tile2 + 2 * lane, *yw2 = tile2 + 24 * 64 + lane;
        const unsigned tile2_lds = TGW ? (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)tile2) : 0u;
        const int bb2 = mine ? col0 + lane : B - 1;
        int t_idx = 0, t_phase = 0, t_nrows = 0, t_pm = 1, t_seg = 0, t_rin = 0, t_srows = 0, t_srows_nx = 0, t_ybase = 0;
        int t_yhas = 0, t_yhas_n = 0, t_asked = 0;
        double t_yprev = 0.0, t_ysum = 0.0, t_yprev_n = 0.0, t_ysum_n = 0.0, t_first_yaw = 0.0;
        const int32_t *t_seg_rows = nullptr;
        const double *t_coeffs = nullptr, *t_yaws = nullptr;
        auto t_load_coeffs = [&](int s_) {           // through registers, on the spot: launch start, the scan's rebuild, empty segments
            const double *src = t_coeffs + 24 * s_;
#pragma unroll
            for (int j = 0; j < 24; ++j) cf2[minsnap_coeff_index<0>(j)] = src[j];
        };
        // E: the row under the cursor -> the target tile, in four pieces (one axis each, then the yaw) spread over four ticks: in
        // one piece it is ~1 300 cycles of a 2 000-cycle tick, and a second wave that is late at the barrier stalls the compute
        // wave (1.30 against 1.26 ms per 1 000 ticks at 65 536 UAVs before the split); the scan's update stays pending
        double t_px = 0, t_py = 0, t_pz = 0, t_vx = 0, t_vy = 0, t_vz = 0, t_ax = 0, t_ay = 0, t_az = 0;
        auto t_eval_axis = [&](int a_) {
            const double t_ = (double)t_rin * P.dt;
            if (a_ == 0) minsnap_eval_axis<0>(cf2, 0, t_, t_px, t_vx, t_ax);
            else if (a_ == 1) minsnap_eval_axis<0>(cf2, 1, t_, t_py, t_vy, t_ay);
            else minsnap_eval_axis<0>(cf2, 2, t_, t_pz, t_vz, t_az);
        };
        auto t_eval_yaw_and_hand_over = [&]() {
            double yaw_;
            t_yhas_n = t_yhas; t_yprev_n = t_yprev; t_ysum_n = t_ysum;
            if (YAWSCAN) {
                const bool yvalid = uavac_yaw::has_heading(t_vx, t_vy);
                const double yang = yvalid ? atan2(t_vy, t_vx) : 0.0;
                const double ycum = (yvalid && t_yhas) ? t_ysum + uavac_yaw::unwrap_correction(yang - t_yprev) : t_ysum;
                yaw_ = yvalid ? yang + ycum : (t_yhas ? t_yprev + t_ysum : t_first_yaw);
                if (t_idx + 1 < t_nrows) {
                    if (yvalid) { t_yhas_n = 1; t_yprev_n = yang; }
                    t_ysum_n = ycum;
                }
            } else {
                yaw_ = yw2[(t_idx - t_ybase) * 64];
            }
            tgt[0] = t_px; tgt[64] = t_py; tgt[128] = t_pz; tgt[192] = t_vx; tgt[256] = t_vy; tgt[320] = t_vz;
            tgt[384] = t_ax; tgt[448] = t_ay; tgt[512] = t_az; tgt[576] = yaw_;
        };
        auto t_eval = [&]() { t_eval_axis(0); t_eval_axis(1); t_eval_axis(2); t_eval_yaw_and_hand_over(); };      // launch start: all at once
        if (TGW) {
            const int64_t off2 = row_offsets[bb2];
            t_nrows = (int)(row_offsets[bb2 + 1] - off2);
            t_idx = istate[0 * sB + bb2];
            t_phase = istate[1 * sB + bb2] % V.F;
            t_pm = P.m;
            size_t seg0 = (size_t)bb2 * P.m;
            if (P.seg_offsets) {
                seg0 = (size_t)P.seg_offsets[bb2];
                const int64_t n_ = P.seg_offsets[bb2 + 1] - P.seg_offsets[bb2];
                t_pm = (int)(n_ < 1 ? 1 : (n_ > P.m ? P.m : n_));
            }
            t_seg_rows = P.seg_rows + seg0;
            t_coeffs = P.coeffs + seg0 * 24;
            t_yaws = YAWSCAN ? nullptr : P.yaw + off2;
            if (t_nrows > 0) {
                t_idx = min(max(t_idx, 0), t_nrows - 1);
                if (YAWSCAN) {
                    t_first_yaw = P.first_yaw[bb2];
                    const double scan_row = state[26 * sB + bb2];
                    if (scan_row == (double)t_idx) {
                        t_yhas = state[27 * sB + bb2] != 0.0;
                        t_yprev = state[28 * sB + bb2];
                        t_ysum = state[29 * sB + bb2];
                    } else {
                        // the cursor is not where the carried scan stands: rebuild it from the mission's first row (rare)
                        int s_ = 0, r_ = 0, n_ = t_seg_rows[0];
                        t_load_coeffs(0);
                        for (int row = 0; row < t_idx; ++row) {
                            while (r_ >= n_ && s_ + 1 < t_pm) { r_ -= n_; ++s_; n_ = t_seg_rows[s_]; t_load_coeffs(s_); }
                            double x_, y_, z_, vx_, vy_, vz_, ax_, ay_, az_;
                            minsnap_eval_row<0>(cf2, (double)r_ * P.dt, x_, y_, z_, vx_, vy_, vz_, ax_, ay_, az_);
                            if (uavac_yaw::has_heading(vx_, vy_)) {
                                const double a_ = atan2(vy_, vx_);
                                if (t_yhas) t_ysum = t_ysum + uavac_yaw::unwrap_correction(a_ - t_yprev);
                                t_yhas = 1;
                                t_yprev = a_;
                            }
                            ++r_;
                        }
                    }
                }
                t_rin = t_idx;
                t_srows = t_seg_rows[0];
                while (t_seg + 1 < t_pm && t_rin >= t_srows) { t_rin -= t_srows; ++t_seg; t_srows = t_seg_rows[t_seg]; }
                t_load_coeffs(t_seg);
                t_srows_nx = t_seg_rows[min(t_seg + 1, t_pm - 1)];
                if (!YAWSCAN) {
                    t_ybase = t_idx;
#pragma unroll
                    for (int j = 0; j < 16; ++j) yw2[j * 64] = t_yaws[min(t_ybase + j, t_nrows - 1)];
                }
                t_eval();
            }
            settle(t_srows_nx); settle(t_idx); settle(t_phase); settle(t_yprev); settle(t_ysum); settle(t_first_yaw);
            lds_barrier();                         // the first target row is in its tile (the compute wave waits here too)
        }
        for (int k = 0; k < K; ++k) {
            lds_barrier();                                             // slab k&1 is complete (stores of earlier ticks stay in flight)
            const double *src = slab + (size_t)((kk + k) & 1) * NR * NU + lane;
            // one log (13 or 12 rows) at a time: every LDS read first, then every store, so that neither the
            // LDS latency nor the store path's acceptance time is paid per element
            if (LOG_STATE) {
                double v[13];
#pragma unroll
                for (int r = 0; r < 13; ++r) v[r] = src[r * NU];
                double *dst = state_log + (size_t)k * 13 * sP + col0;               // wave-uniform: lives in SGPRs
#pragma unroll
                for (int r = 0; r < 13; ++r) {
                    if (full || mine) store_uniform_base(dst + r * sP, lane_bytes, v[r]);      // 512-B coalesced wave store
                    // one obstacle between two stores: the comparisons issue while the store path takes the store
                    if (AABB_HERE && r < kBoxRegs && r < n_obs) {
                        const double x = v[0], y = v[1], z = v[2];
                        const bool hit = (x >= box[r][0]) && (x <= box[r][1]) && (y >= box[r][2]) && (y <= box[r][3]) &&
                                         (z >= box[r][4]) && (z <= box[r][5]);      // inclusive, minimum_snap.py:352-357
                        coll |= hit ? 1 : 0;
                    }
                }
                if (AABB_HERE) {
                    const double x = v[0], y = v[1], z = v[2];
                    for (int o = kBoxRegs; o < n_obs; ++o) {      // more obstacles than registers hold: the slow way
                        const double *c = aabbs + 6 * o;          // uniform address: scalar loads
                        const bool hit = (x >= c[0]) && (x <= c[1]) && (y >= c[2]) && (y <= c[3]) && (z >= c[4]) &&
                                         (z <= c[5]);            // inclusive, minimum_snap.py:352-357
                        coll |= hit ? 1 : 0;
                    }
                }
            }
            if (WATCH) {
                const double x = src[0], y = src[NU], z = src[2 * NU];
#pragma unroll
                for (int o = 0; o < kBoxRegs; ++o)
                    if (o < n_obs) {
                        const bool hit = (x >= box[o][0]) & (x <= box[o][1]) & (y >= box[o][2]) & (y <= box[o][3]) &
                                         (z >= box[o][4]) & (z <= box[o][5]);       // inclusive, minimum_snap.py:352-357
                        coll |= hit ? 1 : 0;
                    }
                for (int o = kBoxRegs; o < n_obs; ++o) {
                    const double *c = aabbs + 6 * o;
                    const double x0 = c[0], x1 = c[1], y0 = c[2], y1 = c[3], z0 = c[4], z1 = c[5];
                    coll |= ((x >= x0) & (x <= x1) & (y >= y0) & (y <= y1) & (z >= z0) & (z <= z1)) ? 1 : 0;
                }
            }
            if (LOG_CMD) {
                double v[UAVAC_CMD_COLS];
#pragma unroll
                for (int r = 0; r < UAVAC_CMD_COLS; ++r) v[r] = src[(CMD0 + r) * NU];
                double *dst = cmd_log + (size_t)k * UAVAC_CMD_COLS * sP + col0;
#pragma unroll
                for (int r = 0; r < UAVAC_CMD_COLS; ++r)
                    if (full || mine) store_uniform_base(dst + r * sP, lane_bytes, v[r]);
            }
            if (TGW) {
                constexpr int kStoresPerTick = (LOG_STATE ? 13 : 0) + (LOG_CMD ? UAVAC_CMD_COLS : 0);
                if (t_phase == 0 && t_nrows > 0) {
                    // the compute wave consumed the tile's row in the tick whose slab has just left: commit its share of the yaw
                    // scan and advance the cursor (main.py:61).  A cursor that enters a new segment asks for its coefficients
                    // (LDS-DMA) and for the row count of the segment after (in flight for a whole segment).  This wave has log
                    // stores in flight all the time and vmcnt counts loads and stores in issue order: waiting for "all but the
                    // youngest tick's stores" covers a load that is older than that without waiting for the newest stores.
                    t_yhas = t_yhas_n; t_yprev = t_yprev_n; t_ysum = t_ysum_n;
                    if (t_idx + 1 < t_nrows) {
                        ++t_idx;
                        if (++t_rin >= t_srows && t_seg + 1 < t_pm) {      // next segment (skipping empty ones, like the sampler's segment_of)
                            store_wave_loads_wait<kStoresPerTick>(t_srows_nx);
                            t_rin -= t_srows; ++t_seg; t_srows = t_srows_nx;
                            while (t_rin >= t_srows && t_seg + 1 < t_pm) { t_rin -= t_srows; ++t_seg; t_srows = t_seg_rows[t_seg]; }
                            coeffs_dma(t_coeffs + 24 * t_seg, tile2_lds);
                            seg_rows_issue(t_srows_nx, t_seg_rows + min(t_seg + 1, t_pm - 1));
                            t_asked = 1;
                        }
                        if (!YAWSCAN && t_idx - t_ybase == 16) {
                            t_ybase = t_idx;
#pragma unroll
                            for (int j = 0; j < 16; ++j) yw2[j * 64] = t_yaws[min(t_ybase + j, t_nrows - 1)];
                        }
                    }
                } else if (t_phase >= 1 && t_phase <= 4 && t_nrows > 0) {
                    // the row of the NEXT outer tick, a piece per tick over the four ticks that follow -- at the priority of a
                    // background job, a compute wave may share this SIMD.  (The coefficients asked for a tick ago are older than
                    // this tick's stores.)  The hand-over in the iteration of phase 4 is ordered before the compute wave's read at
                    // the start of its next phase-0 tick by a barrier for F >= 7 in either hand-over mode: with the late
                    // hand-over the compute wave calls barrier j in the middle of tick j + 1, so this wave's iteration k runs
                    // between the middle of tick k + 1 and the middle of tick k + 2.
                    __builtin_amdgcn_s_setprio(0);
                    if (t_phase == 1) {
                        // (only when some lane of the wave did ask for coefficients a tick ago -- 44 % of the outer ticks: the wait is
                        // for the previous tick's STORES as well, and at two workgroups per CU those are not always down yet)
                        if (__any(t_asked)) store_wave_loads_wait<kStoresPerTick>(t_srows_nx);
                        t_asked = 0;
                        t_eval_axis(0);
                    }
                    else if (t_phase == 2) t_eval_axis(1);
                    else if (t_phase == 3) t_eval_axis(2);
                    else t_eval_yaw_and_hand_over();
                    __builtin_amdgcn_s_setprio(3);
                }
                t_phase = (t_phase + 1 == V.F) ? 0 : t_phase + 1;
            }
        }
        if (TGW && mine) {                         // what the other modes' compute wave saves: cursor and the scan as it stands before it
            istate[0 * sB + col0 + lane] = t_idx;
            if (YAWSCAN) {
                state[26 * sB + col0 + lane] = (double)t_idx;
                state[27 * sB + col0 + lane] = t_yhas ? 1.0 : 0.0;
                state[28 * sB + col0 + lane] = t_yprev;
                state[29 * sB + col0 + lane] = t_ysum;
            }
        }
        if (TGW) store_wave_loads_wait<0>(t_srows_nx);       // nothing stays in flight into the tile the next pass reloads
        if (AABB_HERE && mine) istate[2 * sB + col0 + lane] = coll;
      }
        return;
    }

    // ---------------------------------------------------------------------------------- compute waves
    const int tid = threadIdx.x;
    // The constants of the per-tick path, in vector registers (see vk() in control_law.h): with all of VehK in scalar
    // registers the tick loop spilled SGPRs to VGPR lanes (v_readlane / v_writelane, 22 per tick) and rebuilt its fp64
    // literals on every tick (60 s_mov_b32).  The outer block keeps reading its own constants from the kernel arguments.
    VehK L = V;
#pragma unroll
    for (int i = 0; i < 3; ++i) { L.I[i] = vk(V.I[i]); L.inv_I[i] = vk(V.inv_I[i]); L.ikp[i] = vk(V.ikp[i]); }
    L.inv_mass = vk(V.inv_mass);     // (shares its 8-dword kernel-argument group with I[]: left in scalar registers, the whole
                                     // group was spilled and restored with eight v_readlane on every tick)
    L.arm = vk(V.arm); L.inv_arm = vk(V.inv_arm); L.kappa = vk(V.kappa); L.inv_kappa = vk(V.inv_kappa);
    L.kf = vk(V.kf); L.inv_kf = vk(V.inv_kf);
    L.lit_tiny = vk(V.lit_tiny); L.lit_h2_small = vk(V.lit_h2_small); L.lit_c8 = vk(V.lit_c8); L.lit_c6 = vk(V.lit_c6);
    L.lit_c4 = vk(V.lit_c4); L.lit_s9 = vk(V.lit_s9); L.lit_s7 = vk(V.lit_s7); L.lit_s5 = vk(V.lit_s5); L.lit_s3 = vk(V.lit_s3);
    L.lit_375 = vk(V.lit_375); L.lit_e_small = vk(V.lit_e_small);
    // The plan's sample period, used once per OUTER tick: in a vector register too in the kernels that log.  Left in scalar
    // registers it costs four v_readlane per outer tick -- and the bench launch shape 2.6 % at B = 4 096 (0.853 against 0.831
    // ms per 1 000 ticks, A/B in one process, tools/rollout_ab.py).  The kernels without a second wave keep it scalar: they
    // sit at the 256-register limit of two waves per SIMD (tests/test_abi_and_host.py checks that budget).
    const double plan_dt = !POLY ? 0.0 : (LOGGING ? vk(P.dt) : P.dt);
  for (int tile0 = 0; tile0 < n_tiles; tile0 += grid, kk += K) {
    const int n_here = min(grid, n_tiles - tile0);
    if ((int)blockIdx.x >= n_here) break;
    const int col0 = (tile0 + xcd_contiguous(blockIdx.x, n_here)) * NU;
    const int b = col0 + tid;
    const bool live = b < B;
    const int bb = live ? b : B - 1;                                   // dead lanes shadow the last UAV, store nothing

    double px = state[0 * sB + bb], py = state[1 * sB + bb], pz = state[2 * sB + bb];
    double q0 = state[3 * sB + bb], q1 = state[4 * sB + bb], q2 = state[5 * sB + bb], q3 = state[6 * sB + bb];
    double vx = state[7 * sB + bb], vy = state[8 * sB + bb], vz = state[9 * sB + bb];
    double wp = state[10 * sB + bb], wq = state[11 * sB + bb], wr = state[12 * sB + bb];
    double om[4], omc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { om[i] = state[(13 + i) * sB + bb]; omc[i] = state[(17 + i) * sB + bb]; }
    double integ = state[21 * sB + bb];
    double thrust_cmd = state[22 * sB + bb];
    double pc = state[23 * sB + bb], qc = state[24 * sB + bb], rc = state[25 * sB + bb];
    int idx = istate[0 * sB + bb];
    int inner = istate[1 * sB + bb];
    int collided = istate[2 * sB + bb];
    int gbits = GROUND ? istate[3 * sB + bb] : 0;

    const int64_t off = row_offsets[bb];
    const int nrows = (int)(row_offsets[bb + 1] - off);
    const double *rows = traj + off * UAVAC_TRAJ_COLS;
    int phase = inner % V.F;

    // Make every load above land before the tick loop: a load still pending at the loop header would be
    // waited for with vmcnt at its first use inside the loop on EVERY iteration, and those waits would also
    // expose the latency of the (asm-issued, compiler-invisible) row prefetch.
    settle(px); settle(py); settle(pz); settle(q0); settle(q1); settle(q2); settle(q3); settle(vx); settle(vy);
    settle(vz); settle(wp); settle(wq); settle(wr); settle(integ); settle(thrust_cmd); settle(pc); settle(qc);
    settle(rc);
#pragma unroll
    for (int i = 0; i < 4; ++i) { settle(om[i]); settle(omc[i]); }
    settle(idx); settle(phase); settle(collided);

    constexpr bool BOX_HERE = AABB && !((LOG_STATE || WATCH) && CW == SW);

    RowRegs nxt;
    if (!POLY && nrows > 0) row_issue(nxt, rows + (size_t)min(max(idx, 0), nrows - 1) * UAVAC_TRAJ_COLS);

    // POLY: segment / row-in-segment of the cursor, the segment's coefficients and the next 16 yaws, in LDS
    // this wave's coefficient tile: [24][64] doubles, or -- filled by LDS-DMA -- [12][64][2] (minsnap_eval.h, STRIDE 0)
    constexpr int CST = ADMA ? 0 : 64;
    double *tile = slab + (size_t)2 * NR * NU + (size_t)(tid >> 6) * poly_tile_doubles(YAWSCAN);
    double *cf = tile + (ADMA ? 2 : 1) * (tid & 63);                                                    // this lane's first coefficient
    double *yw = tile + 24 * 64 + (tid & 63);                                                           // yw[j * 64]
    const unsigned tile_lds = ADMA ? (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)tile) : 0u;   // LDS byte address (low half of the generic pointer)
    // segments of this lane's mission: P.m of them at bb * P.m, or -- ragged batch -- seg_offsets[bb + 1] - seg_offsets[bb]
    // of them at seg_offsets[bb] (clamped to 1 .. P.m, the batch's maximum)
    int pm = P.m;
    size_t seg0 = (size_t)bb * P.m;
    if (POLY && P.seg_offsets) {
        seg0 = (size_t)P.seg_offsets[bb];
        const int64_t n_ = P.seg_offsets[bb + 1] - P.seg_offsets[bb];
        pm = (int)(n_ < 1 ? 1 : (n_ > P.m ? P.m : n_));
    }
    const int32_t *seg_rows = POLY ? P.seg_rows + seg0 : nullptr;
    const double *mission_coeffs = POLY ? P.coeffs + seg0 * 24 : nullptr;
    const double *yaws = (POLY && !YAWSCAN) ? P.yaw + off : nullptr;
    int seg = 0, rin = 0, srows = 0, ybase = 0;
    int srows_nx = 0;                              // rows of segment seg + 1 (in flight from the moment seg is entered: plan_loads_wait)
    auto load_coeffs = [&](int s_) {               // through registers, on the spot: launch start and the scan's rebuild only
        const double *src = mission_coeffs + 24 * s_;
#pragma unroll
        for (int j = 0; j < 24; ++j) cf[minsnap_coeff_index<CST>(j)] = src[j];
    };
    auto load_yaws = [&](int base_) {
#pragma unroll
        for (int j = 0; j < 16; ++j) yw[j * 64] = yaws[min(base_ + j, nrows - 1)];
    };
    // YAWSCAN: the carried scan (state rows 26-29)
    int yhas = 0;
    double yprev = 0.0, ysum = 0.0, first_yaw = 0.0;
    if (POLY && !TGW && nrows > 0) {
        idx = min(max(idx, 0), nrows - 1);
        if (YAWSCAN) {
            first_yaw = P.first_yaw[bb];
            const double scan_row = state[26 * sB + bb];
            if (scan_row == (double)idx) {
                yhas = state[27 * sB + bb] != 0.0;
                yprev = state[28 * sB + bb];
                ysum = state[29 * sB + bb];
            } else {
                // the cursor is not where the carried scan stands: rebuild it from the mission's first row (rare: a
                // caller moved the cursor, or a launch without YAWSCAN advanced it)
                int s_ = 0, r_ = 0, n_ = seg_rows[0];
                load_coeffs(0);
                for (int row = 0; row < idx; ++row) {
                    while (r_ >= n_ && s_ + 1 < pm) { r_ -= n_; ++s_; n_ = seg_rows[s_]; load_coeffs(s_); }
                    double x_, y_, z_, vx_, vy_, vz_, ax_, ay_, az_;
                    minsnap_eval_row<CST>(cf, (double)r_ * plan_dt, x_, y_, z_, vx_, vy_, vz_, ax_, ay_, az_);
                    if (uavac_yaw::has_heading(vx_, vy_)) {
                        const double a_ = atan2(vy_, vx_);
                        if (yhas) ysum = ysum + uavac_yaw::unwrap_correction(a_ - yprev);
                        yhas = 1;
                        yprev = a_;
                    }
                    ++r_;
                }
            }
        }
        rin = idx;
        srows = seg_rows[0];
        while (seg + 1 < pm && rin >= srows) { rin -= srows; ++seg; srows = seg_rows[seg]; }
        load_coeffs(seg);
        if (ADMA) srows_nx = seg_rows[min(seg + 1, pm - 1)];
        if (!YAWSCAN) {
            ybase = idx;
            load_yaws(ybase);
        }
    }
    if (POLY && ADMA) settle(srows_nx);

    // 1/|q|^2 of the caller-supplied attitude; the free-body step leaves q unit, so 1 from then on
    // (a state this kernel wrote earlier is unit to rounding: take exactly 1 so that splitting a rollout over
    // launches is bit-identical to one launch)
    const double qn2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    double inv_n2 = (fabs(qn2 - 1.0) < 1.0e-12) ? 1.0 : 1.0 / qn2;

    // TGW: the target row of the next outer tick, put there by the second wave (which also owns the cursor)
    const double *tgt = slab + (size_t)2 * NR * NU + (size_t)CW * poly_tile_doubles(YAWSCAN) + (tid & 63);
    if (TGW) lds_barrier();                        // the first row is in the tile
    for (int k = 0; k < K; ++k) {
        if (phase == 0 && nrows > 0) {
            // ------------------------------------------------------------- outer loop (main.py:47-61)
            double tg_x, tg_y, tg_z, tg_vx, tg_vy, tg_vz, tg_ax, tg_ay, tg_az, tg_yaw;
            if (TGW) {
                tg_x = tgt[0]; tg_y = tgt[64]; tg_z = tgt[128]; tg_vx = tgt[192]; tg_vy = tgt[256]; tg_vz = tgt[320];
                tg_ax = tgt[384]; tg_ay = tgt[448]; tg_az = tgt[512]; tg_yaw = tgt[576];
            } else if (POLY) {
                if (ADMA) plan_loads_wait(srows_nx);            // coefficients (and row count) asked for an outer tick ago have landed
                minsnap_eval_row<CST>(cf, (double)rin * plan_dt, tg_x, tg_y, tg_z, tg_vx, tg_vy, tg_vz, tg_ax, tg_ay, tg_az);
                if (YAWSCAN) {
                    // this row's yaw from the carried scan (minimum_snap.py:126-136); committed below only if the cursor moves on
                    const bool yvalid = uavac_yaw::has_heading(tg_vx, tg_vy);
                    const double yang = yvalid ? atan2(tg_vy, tg_vx) : 0.0;
                    const double ycum = (yvalid && yhas) ? ysum + uavac_yaw::unwrap_correction(yang - yprev) : ysum;
                    tg_yaw = yvalid ? yang + ycum : (yhas ? yprev + ysum : first_yaw);
                    if (idx + 1 < nrows) {
                        if (yvalid) { yhas = 1; yprev = yang; }
                        ysum = ycum;
                    }
                } else {
                    tg_yaw = yw[(idx - ybase) * 64];
                }
            } else {
                row_wait(nxt);
                tg_x = row_col(nxt, 0); tg_y = row_col(nxt, 1); tg_z = row_col(nxt, 2);
                tg_vx = row_col(nxt, 3); tg_vy = row_col(nxt, 4); tg_vz = row_col(nxt, 5);
                tg_ax = row_col(nxt, 6); tg_ay = row_col(nxt, 7); tg_az = row_col(nxt, 8);
                tg_yaw = row_col(nxt, 9);
            }

            const VehK O = outer_constants();
            const Rot R = quat_to_rot(q0, q1, q2, q3);                 // shared by altitude and attitude
            thrust_cmd = altitude(O, tg_z, tg_vz, tg_az, pz, vz, R.r22, integ);
            double bxc, byc;
            lateral(O, tg_x, tg_vx, tg_ax, tg_y, tg_vy, tg_ay, px, py, vx, vy, thrust_cmd, bxc, byc);
            roll_pitch(O, bxc, byc, R, pc, qc);
            double psi, cth, sphi, cphi;
            euler_trig(q0, q1, q2, q3, psi, cth, sphi, cphi);
            rc = yaw_rate(O, tg_yaw, psi, cth, sphi, cphi, qc);
            // next row (main.py:61), consumed F ticks from now; issued last so that nothing in this block
            // still reads the registers it overwrites
            if (TGW) {
                // (the second wave advances the cursor)
            } else if (POLY) {
                if (idx + 1 < nrows) {                    // main.py:61: the cursor stops on the last row
                    ++idx;
                    if (++rin >= srows) {                 // next segment (skipping empty ones, like the sampler's segment_of)
                        if (!ADMA) {                      // full chip: through registers, on the spot (see the launcher)
                            while (rin >= srows && seg + 1 < pm) { rin -= srows; ++seg; srows = seg_rows[seg]; }
                            load_coeffs(seg);
                        } else if (seg + 1 < pm) {
                            rin -= srows; ++seg; srows = srows_nx;             // (its row count came with the segment before)
                            while (rin >= srows && seg + 1 < pm) { rin -= srows; ++seg; srows = seg_rows[seg]; }    // empty segments: rare, on the spot
                            coeffs_dma(mission_coeffs + 24 * seg, tile_lds);   // lands in this lane's column while the inner ticks run
                            seg_rows_issue(srows_nx, seg_rows + min(seg + 1, pm - 1));
                        }
                    }
                    if (!YAWSCAN && idx - ybase == 16) { ybase = idx; load_yaws(ybase); }
                }
            } else {
                idx = min(idx + 1, nrows - 1);
                row_issue(nxt, rows + (size_t)idx * UAVAC_TRAJ_COLS);
            }
        }

        // ----------------------------------------------------------------- inner loop, every tick
        double Mx, My, Mz, f[4];
        body_rate(L, pc, qc, rc, wp, wq, wr, Mx, My, Mz);
        allocate(L, thrust_cmd, Mx, My, Mz, f);
        motors(L, f, om, omc);
        // late hand-over: slab k-1 goes to the store wave HERE, a third of a tick after it was written -- the barrier's wait for
        // this wave's LDS writes then finds nothing outstanding (the launcher says when that pays)
        if (LOGGING && late_handover && k > 0) lds_barrier();

        double *my = LOGGING ? slab + (size_t)((kk + k) & 1) * NR * NU + tid : nullptr;
        if (LOG_CMD) {
            double *c = my + CMD0 * NU;
            c[0] = thrust_cmd; c[1 * NU] = pc; c[2 * NU] = qc; c[3 * NU] = rc;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[(4 + i) * NU] = omc[i]; c[(8 + i) * NU] = om[i]; }
        }

        free_body_step<GROUND>(L, om, px, py, pz, q0, q1, q2, q3, vx, vy, vz, wp, wq, wr, inv_n2);
        inv_n2 = 1.0;
        if (GROUND) gbits = ground_bits(L, pz, gbits);

        if (WATCH) { my[0] = px; my[1 * NU] = py; my[2 * NU] = pz; }
        if (BOX_HERE) {                                   // only with a command log alone; otherwise the second wave tests
            for (int o = 0; o < n_obs; ++o) {
                const double *c = aabbs + 6 * o;          // uniform address: scalar loads
                // all six bounds first, then six comparisons combined without short-circuit: one scalar-cache round
                // trip per obstacle (written with && it was one per BOUND: load, wait, compare, branch, six times)
                const double x0 = c[0], x1 = c[1], y0 = c[2], y1 = c[3], z0 = c[4], z1 = c[5];
                const bool hit = (px >= x0) & (px <= x1) & (py >= y0) & (py <= y1) & (pz >= z0) &
                                 (pz <= z1);              // inclusive, minimum_snap.py:352-357
                collided |= hit ? 1 : 0;
            }
        }

        if (LOG_STATE) {
            my[0] = px; my[1 * NU] = py; my[2 * NU] = pz;
            my[3 * NU] = q0; my[4 * NU] = q1; my[5 * NU] = q2; my[6 * NU] = q3;
            my[7 * NU] = vx; my[8 * NU] = vy; my[9 * NU] = vz;
            my[10 * NU] = wp; my[11 * NU] = wq; my[12 * NU] = wr;
        }
        if (LOGGING && !late_handover) lds_barrier();        // hand slab (kk+k)&1 to the store wave; it was drained two ticks ago
        ++inner;
        phase = (phase + 1 == V.F) ? 0 : phase + 1;
    }
    if (LOGGING && late_handover && K > 0) lds_barrier();      // the last slab

    // nothing may stay in flight into these registers.  UNCONDITIONAL (not `if (nrows > 0)`, the mask the loads were issued under):
    // the build check follows the control-flow graph and cannot know that a skipped wait belongs to a skipped issue -- every path
    // from an issue site to the next pass or to the end of the kernel must cross a wait (a wave without rows waits for nothing)
    if (!POLY) row_wait(nxt);
    if (POLY && ADMA) plan_loads_wait(srows_nx);    // ... nor into this wave's coefficient tile (the next pass, or nobody, owns it)
    if (live) {
    state[0 * sB + b] = px; state[1 * sB + b] = py; state[2 * sB + b] = pz;
    state[3 * sB + b] = q0; state[4 * sB + b] = q1; state[5 * sB + b] = q2; state[6 * sB + b] = q3;
    state[7 * sB + b] = vx; state[8 * sB + b] = vy; state[9 * sB + b] = vz;
    state[10 * sB + b] = wp; state[11 * sB + b] = wq; state[12 * sB + b] = wr;
#pragma unroll
    for (int i = 0; i < 4; ++i) { state[(13 + i) * sB + b] = om[i]; state[(17 + i) * sB + b] = omc[i]; }
    state[21 * sB + b] = integ;
    state[22 * sB + b] = thrust_cmd;
    state[23 * sB + b] = pc; state[24 * sB + b] = qc; state[25 * sB + b] = rc;
    if (!TGW) istate[0 * sB + b] = idx;            // (TGW: the second wave owns and saves the cursor)
    istate[1 * sB + b] = inner;
    if (!AABB || BOX_HERE) istate[2 * sB + b] = collided;
    if (GROUND) istate[3 * sB + b] = gbits;
    if (POLY && YAWSCAN && !TGW) {
        state[26 * sB + b] = (double)idx;
        state[27 * sB + b] = yhas ? 1.0 : 0.0;
        state[28 * sB + b] = yprev;
        state[29 * sB + b] = ysum;
    }
    }
  }
}

__global__ void state_init_kernel(const VehK V, const double *__restrict__ positions, int B, int hover,
                                  double *__restrict__ state, int32_t *__restrict__ istate) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t sB = (size_t)B;
    for (int r = 0; r < UAVAC_STATE_ROWS; ++r) state[r * sB + b] = 0.0;
    if (positions) {
        state[0 * sB + b] = positions[3 * (size_t)b + 0];
        state[1 * sB + b] = positions[3 * (size_t)b + 1];
        state[2 * sB + b] = positions[3 * (size_t)b + 2];
    }
    state[3 * sB + b] = 1.0;
    if (hover) {
        for (int i = 0; i < 4; ++i) { state[(13 + i) * sB + b] = V.hover_omega; state[(17 + i) * sB + b] = V.hover_omega; }
    }
    for (int r = 0; r < UAVAC_ISTATE_ROWS; ++r) istate[r * sB + b] = 0;
}

// Empty kernel with the workgroup shape of the logged rollout (two waves).  Where the dispatcher puts the waves of a
// 2-wave workgroup depends on the shape of the kernel that ran before: after the planning kernels (1-wave
// workgroups) 5-15 % of the SIMDs receive two compute waves and others two store waves, and the launch runs 25 %
// slower (tools/first_launch_bisect.py, profiles/r02_placement_census.txt; it heals by itself over the next two
// launches).  After ANY kernel of 2-wave workgroups the placement is one compute + one store wave on every SIMD.
__global__ void __launch_bounds__(256) rollout_align_kernel() {}

// One compute + one store wave per 64 UAVs, one hand-over per tick.  (Two ticks per hand-over -- the compute wave fills two
// slabs before the barrier, the store wave drains two after it; 2 x 2 slabs + the coefficient tile still fit four workgroups
// per CU once the yaw tile is gone -- is bit-identical and changes nothing: 14.46-14.49 ms per bench step against
// 14.49-14.54 (a throw-away probe on the round-2 tree).  The rendezvous is not what couples the stages.)  (Four compute + four store waves per 256 UAVs -- one workgroup per CU, whose
// 8 waves the dispatcher always deals round the 4 SIMDs evenly, so that no aligner is needed -- was measured at 1.65 ms
// per 1 000 ticks against 1.33 ms: the per-tick barrier then couples eight waves; round 3 measured two + two waves per 128 UAVs
// slower at every batch size as well, profiles/r03_rollout_shapes_wide_workgroup.jsonl.  The kernel keeps its CW / SW
// parameters; only <1, 1> is instantiated.)
template <bool LS, bool LC, bool AB, bool POLY, bool GR, bool YS, int PMODE = 0>
void launch_shape(uavac_ctx *ctx, const VehK &V, const double *traj, const int64_t *row_offsets, double *state,
                  int32_t *istate, int B, int K, double *state_log, double *cmd_log, const double *aabbs, int n_obs,
                  const PlanRef &P) {
    constexpr int CW = 1, SW = 1;
    constexpr bool WATCH = AB && !LS && !LC;           // obstacles without a log: the second wave only watches
    constexpr bool LOGGING = LS || LC || WATCH;
    constexpr int NU = 64 * CW;
    constexpr int NR = (LS ? 13 : 0) + (LC ? UAVAC_CMD_COLS : 0) + (WATCH ? 3 : 0);
    const int n_tiles = (B + NU - 1) / NU;
    // a placeholder wave between compute and store wave where two workgroups share a CU (more than a quarter, at most half
    // as many tiles as the chip has SIMDs: 257 .. 512 on an MI355X) -- see the kernel, PLACEHOLDER WAVES
    int n_idle = 0;
    if (LOGGING) {
        if (ctx->idle_waves >= 0) n_idle = ctx->idle_waves > 1 ? 1 : ctx->idle_waves;
        else n_idle = (4 * n_tiles > ctx->n_simds && 2 * n_tiles <= ctx->n_simds) ? 1 : 0;
    }
    const int threads = NU + (LOGGING ? 64 * (1 + n_idle) : 0);
    const size_t lds = sizeof(double) * (2 * NR * NU + (POLY ? CW * poly_tile_doubles(YS) : 0) + (PMODE == 2 ? 10 * NU : 0));
    auto kern = control_rollout_kernel<CW, SW, LS, LC, AB, POLY, GR, YS, PMODE>;
    // With a second wave per workgroup the launch holds at most one workgroup per SIMD; a batch with more 64-UAV tiles than
    // the chip has SIMDs is walked by those workgroups pass after pass inside ONE launch (persistent tiles, see the kernel).
    // Round 2 issued one launch per 65 536 columns instead (a single launch with two workgroups per SIMD had been measured
    // at 4.04 ms per 1 000 ticks for B = 131 072 against 3.15 ms for two launches): every launch boundary made all SIMDs
    // wait for the slowest one.  Without a second wave one launch of all tiles is kept (two compute waves per SIMD hide
    // each other's latency).  Results do not depend on the split.
    const int max_wgs = ctx->n_simds / CW;                     // one compute wave per SIMD at most
    const int grid = (LOGGING && n_tiles > max_wgs) ? max_wgs : n_tiles;
    const int cols = grid * NU < B ? grid * NU : B;            // columns in flight at a time
    // Hand the slab over at the end of the tick, or a third of a tick later (after the next tick's motor model)?  Results are the
    // same bit for bit.  Round 5 (tools/half_chip_options.py, profiles/r05_launcher_sweep_m*.jsonl: every size from 12 288 to
    // 65 536 UAVs x hand-over point x PMODE, interleaved): the late hand-over wins or ties at EVERY size with the round-4
    // kernels -- 0.857 against 0.890 ms per 1 000 logged ticks at 32 768 UAVs, 0.852 / 0.883 at 24 576, 1.092 / 1.105 at 57 344 --
    // so it is the default (round 3 had found a window from 20 480 to 40 960 UAVs where the end of the tick was better; that was
    // before the target rows left the compute wave's outer tick).  Option "late_handover" still forces either.
    const int late = ctx->late_handover >= 0 ? ctx->late_handover : 1;
    (void)cols;
    // WORKGROUPS PER CU.  Below a full chip nothing but LDS limits how many of these workgroups a CU takes (registers allow four),
    // and the dispatcher does not deal them evenly: at 512 workgroups on 256 CUs some CUs get three and some one, and the launch ends
    // with its slowest CU.  The workgroup therefore asks for as much LDS as makes k + 1 of them NOT fit a CU, k = the number every CU
    // must take, for k = 2 and 3 (round 5, profiles/r05_cu_balance.jsonl: 32 768 UAVs 0.878 -> 0.849 and 0.869 -> 0.854 ms per 1 000 ticks on two
    // boxes; no effect where k workgroups per CU is what happens anyway).  Option "cu_balance" = 0 switches it off; "lds_pad" > 0
    // overrides it.  (A full chip needs nothing: four workgroups per CU is all its registers hold.)
    size_t pad = (size_t)ctx->lds_pad;
    if (LOGGING && pad == 0 && ctx->cu_balance != 0) {
        const int cus = ctx->n_simds / 4;
        const int k = (grid + cus - 1) / cus;
        // (k = 1 is left alone: measured no gain there, and a workgroup that claims half a CU's LDS could starve beside another
        // kernel on a second stream -- the root of config 4 re-samples its peers' rows while it flies)
        if (k >= 2 && k <= 3) {
            const size_t lds_cu = (size_t)160 * 1024;
            const size_t want = ((lds_cu / (size_t)(k + 1) + 1024) + 1023) & ~(size_t)1023;
            if (want * (size_t)k <= lds_cu && want > lds) pad = want - lds;
        }
    }
    const size_t pitch = (LS || LC) ? (ctx->log_pitch > 0 ? (size_t)ctx->log_pitch : (size_t)B) : (size_t)B;
    if (LOGGING && ctx->rollout_align)
        hipLaunchKernelGGL(rollout_align_kernel, dim3(grid), dim3(threads), 0, ctx->stream);
    if (lds + pad > 64 * 1024) {
        // (an "lds_pad" beyond what a workgroup may have is refused HERE, with the runtime's words, not by a launch that never runs)
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + pad));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ctx->err = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize = ") + std::to_string(lds + pad) + "): " + hipGetErrorString(e);
            ctx->launch_rc = UAVAC_EHIP;
            return;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds + pad, ctx->stream, V, traj, row_offsets, state, istate,
                       B, K, state_log, cmd_log, aabbs, n_obs, n_tiles, pitch, P, late, n_idle);
    auto tf = [](bool v) { return v ? "true" : "false"; };
    char name[176];
    snprintf(name, sizeof name, "control_rollout_kernel<%d, %d, %s, %s, %s, %s, %s, %s, %d>", CW, SW, tf(LS), tf(LC), tf(AB), tf(POLY),
             tf(GR), tf(YS), PMODE);                       // the name rocprofv3 prints, argument for argument
    ctx->last_rollout = name;
    // vector registers of that kernel as the loaded code object has them (once per variant): above 256 a SIMD holds ONE
    // wave of it and the launch runs at 0.65x -- a toolchain that crosses the line shows up here and in bench.py's line
    // (one value per kernel variant, the same from every ctx: an atomic, since distinct ctxs may launch from distinct threads)
    static std::atomic<int> vgprs{-1};
    int v = vgprs.load(std::memory_order_relaxed);
    if (v < 0) {
        hipFuncAttributes fa;
        v = hipFuncGetAttributes(&fa, (const void *)kern) == hipSuccess ? fa.numRegs : 0;
        vgprs.store(v, std::memory_order_relaxed);
    }
    ctx->last_rollout_vgprs = v;
}

template <bool LS, bool LC, bool AB>
void launch_variant(uavac_ctx *ctx, const VehK &V, const double *traj, const int64_t *row_offsets, double *state,
                    int32_t *istate, int B, int K, double *state_log, double *cmd_log, const double *aabbs, int n_obs,
                    const PlanRef *plan) {
#define UAVAC_SHAPE_ARGS ctx, V, traj, row_offsets, state, istate, B, K, state_log, cmd_log, aabbs, n_obs
    // Plan-fed kernels, PMODE (see the kernel): who evaluates the target rows, how coefficients reach the LDS tile.  Per 1 000
    // logged ticks, same process and buffers (tools/rollout_ab.py, m = 12), mode 2 / 1 / 0 (= round 3):
    //     16 384 UAVs 0.792 / 0.834 / 0.852 ms     32 768  0.879 / 0.891 / 0.984     49 152  1.011 / 0.997 / 1.055-1.078
    //     65 536      1.297 / 1.27 / 1.25-1.26
    // * up to two workgroups per CU every working wave has a SIMD of its own and the second wave's idle time is free: it
    //   evaluates the rows (2; needs a second wave, and F >= 7 inner ticks per outer tick: the row is evaluated in four pieces,
    //   one to four ticks after the cursor moved);
    // * with three or four workgroups per CU a second wave shares its SIMD with a compute wave: the compute wave evaluates, its
    //   coefficients arrive by LDS-DMA (1) -- also the form of the kernels without a second wave while the chip is not full;
    // * a full chip is bound by its log stream and the twelve DMA instructions per segment change are only in the way: through
    //   registers, on the spot (0).
    // Same bits in every mode.  (Template parameters, not branches: with two forms in one kernel the plan-fed variants need
    // 257-260 vector registers.)  Option "coeff_dma": -1 = as above, 0 / 1 / 2 = that mode where the kernel has it.
    constexpr bool LOGGING = LS || LC || AB;
    const int tiles = (B + 63) / 64, in_flight = (LOGGING && tiles > ctx->n_simds) ? ctx->n_simds : tiles;
    int mode = 0;
    if (plan) {
        if (ctx->coeff_dma >= 0) mode = ctx->coeff_dma > 2 ? 2 : ctx->coeff_dma;
        // (round 5, same sweep: the second wave's rows pay up to 26 624 UAVs -- 0.840 against 0.852 ms at 24 576 -- and lose from
        // 28 672 on -- 0.866 against 0.857 at 32 768: the crossover lies below "two workgroups on every CU")
        else mode = in_flight * 64 >= 60 * ctx->n_simds ? 0 : (in_flight * 64 <= 26 * ctx->n_simds ? 2 : 1);
        if (mode == 2 && (!LOGGING || V.F < 7)) mode = 1;
    }
#define UAVAC_LAUNCH_MODE(GR_, YS_)                                                                     \
    do {                                                                                                \
        if (mode == 1) launch_shape<LS, LC, AB, true, GR_, YS_, 1>(UAVAC_SHAPE_ARGS, *plan);            \
        else if (mode == 2) {                                                                           \
            if constexpr (LOGGING) launch_shape<LS, LC, AB, true, GR_, YS_, 2>(UAVAC_SHAPE_ARGS, *plan); \
        } else launch_shape<LS, LC, AB, true, GR_, YS_, 0>(UAVAC_SHAPE_ARGS, *plan);                    \
    } while (0)
    if (plan && !plan->yaw) {                       // the rollout scans the yaw itself
        if (V.ground) UAVAC_LAUNCH_MODE(true, true); else UAVAC_LAUNCH_MODE(false, true);
    } else if (plan) {
        if (V.ground) UAVAC_LAUNCH_MODE(true, false); else UAVAC_LAUNCH_MODE(false, false);
#undef UAVAC_LAUNCH_MODE
    } else {
        if (V.ground) launch_shape<LS, LC, AB, false, true, false>(UAVAC_SHAPE_ARGS, PlanRef{});
        else launch_shape<LS, LC, AB, false, false, false>(UAVAC_SHAPE_ARGS, PlanRef{});
    }
#undef UAVAC_SHAPE_ARGS
}

void launch_flags(uavac_ctx *ctx, const VehK &V, const double *traj, const int64_t *row_offsets, double *state,
                  int32_t *istate, int B, int K, double *state_log, double *cmd_log, const double *aabbs, int n_obs,
                  const PlanRef *plan) {
    const bool ls = state_log != nullptr, lc = cmd_log != nullptr, ab = (aabbs != nullptr && n_obs > 0);
#define UAVAC_ARGS ctx, V, traj, row_offsets, state, istate, B, K, state_log, cmd_log, aabbs, n_obs, plan
    if (ls) {
        if (lc) { if (ab) launch_variant<true, true, true>(UAVAC_ARGS); else launch_variant<true, true, false>(UAVAC_ARGS); }
        else    { if (ab) launch_variant<true, false, true>(UAVAC_ARGS); else launch_variant<true, false, false>(UAVAC_ARGS); }
    } else {
        if (lc) { if (ab) launch_variant<false, true, true>(UAVAC_ARGS); else launch_variant<false, true, false>(UAVAC_ARGS); }
        else    { if (ab) launch_variant<false, false, true>(UAVAC_ARGS); else launch_variant<false, false, false>(UAVAC_ARGS); }
    }
#undef UAVAC_ARGS
}

}  // namespace

int uavac_launch_state_init(uavac_ctx *ctx, const VehK &V, const double *positions, int B, int hover, double *state,
                            int32_t *istate) {
    hipLaunchKernelGGL(state_init_kernel, dim3((B + 255) / 256), dim3(256), 0, ctx->stream, V, positions, B, hover,
                       state, istate);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_launch_rollout(uavac_ctx *ctx, const VehK &V, const double *traj, const int64_t *row_offsets, double *state,
                         int32_t *istate, int B, int K, double *state_log, double *cmd_log, const double *aabbs,
                         int n_obs, const PlanRef *plan) {
    // One compute wave + one store wave per workgroup.  (Four compute waves sharing one store wave were
    // measured 15 % slower at B = 65 536: a single wave cannot issue a CU's 52 stores per tick fast enough.
    // Two compute waves with the store wave moving 16 B per lane -- 13 x 1 KB wave stores per tick instead of
    // 26 x 512 B -- were 17 % slower too, 1.47 vs 1.25 ms on the same box: the per-tick barrier then couples
    // two compute waves.  Replacing the per-tick barrier by a ring of 4 slabs with produced / consumed counters in
    // LDS, so that the compute wave may run 4 ticks ahead of a stalled store wave, was slower as well: 1.35 vs
    // 1.26 ms -- the hand-over is not what limits the kernel.)  The log rows want B to be a multiple of 16 (128-B lines): B = 65 534 runs at half
    // the rate of B = 65 536 because every 512-B wave store then straddles two partially written lines.
    ctx->launch_rc = UAVAC_OK;
    launch_flags(ctx, V, traj, row_offsets, state, istate, B, K, state_log, cmd_log, aabbs, n_obs, plan);
    if (ctx->launch_rc != UAVAC_OK) return ctx->launch_rc;       // (the text is in ctx->err)
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
