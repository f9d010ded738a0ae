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
// HBM traffic per UAV tick: 104 B of state log (13 f64, coalesced across lanes) + one 88 B trajectory
// row every F ticks (prefetched one outer period ahead) -- everything else stays in VGPRs.

#include "control_law.h"

#include <cmath>

namespace {

using namespace uavac_dev;

struct Row { double v[UAVAC_TRAJ_COLS]; };

__device__ __forceinline__ void load_row(Row &r, const double *__restrict__ p) {
#pragma unroll
    for (int i = 0; i < UAVAC_TRAJ_COLS; ++i) r.v[i] = p[i];
}

template <bool LOG_STATE, bool LOG_CMD, bool AABB>
__global__ void __launch_bounds__(64) control_rollout_kernel(const VehK V, const double *__restrict__ traj,
                                                            const int64_t *__restrict__ row_offsets,
                                                            double *__restrict__ state, int32_t *__restrict__ istate,
                                                            int B, int K, double *__restrict__ state_log,
                                                            double *__restrict__ cmd_log,
                                                            const double *__restrict__ aabbs, int n_obs) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const size_t sB = (size_t)B;

    double px = state[0 * sB + b], py = state[1 * sB + b], pz = state[2 * sB + b];
    double q0 = state[3 * sB + b], q1 = state[4 * sB + b], q2 = state[5 * sB + b], q3 = state[6 * sB + b];
    double vx = state[7 * sB + b], vy = state[8 * sB + b], vz = state[9 * sB + b];
    double wp = state[10 * sB + b], wq = state[11 * sB + b], wr = state[12 * sB + b];
    double om[4], omc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { om[i] = state[(13 + i) * sB + b]; omc[i] = state[(17 + i) * sB + b]; }
    double integ = state[21 * sB + b];
    double thrust_cmd = state[22 * sB + b];
    double pc = state[23 * sB + b], qc = state[24 * sB + b], rc = state[25 * sB + b];
    int idx = istate[0 * sB + b];
    int inner = istate[1 * sB + b];
    int collided = istate[2 * sB + b];

    const int64_t off = row_offsets[b];
    const int nrows = (int)(row_offsets[b + 1] - off);
    const double *rows = traj + off * UAVAC_TRAJ_COLS;
    int phase = inner % V.F;

    Row nxt;
#pragma unroll
    for (int i = 0; i < UAVAC_TRAJ_COLS; ++i) nxt.v[i] = 0.0;
    if (nrows > 0) load_row(nxt, rows + (size_t)min(max(idx, 0), nrows - 1) * UAVAC_TRAJ_COLS);

    for (int k = 0; k < K; ++k) {
        if (phase == 0 && nrows > 0) {
            // ------------------------------------------------------------- outer loop (main.py:47-61)
            const Row tg = nxt;
            idx = min(idx + 1, nrows - 1);
            load_row(nxt, rows + (size_t)idx * UAVAC_TRAJ_COLS);    // consumed F ticks from now

            const Rot R = quat_to_rot(q0, q1, q2, q3);                 // shared by altitude and attitude
            thrust_cmd = altitude(V, tg.v[2], tg.v[5], tg.v[8], pz, vz, R.r22, integ);
            double bxc, byc;
            lateral(V, tg.v[0], tg.v[3], tg.v[6], tg.v[1], tg.v[4], tg.v[7], px, py, vx, vy, thrust_cmd, bxc, byc);
            roll_pitch(V, bxc, byc, R, pc, qc);
            double psi, cth, sphi, cphi;
            euler_trig(q0, q1, q2, q3, psi, cth, sphi, cphi);
            rc = yaw_rate(V, tg.v[9], psi, cth, sphi, cphi, qc);
        }

        // ----------------------------------------------------------------- inner loop, every tick
        double Mx, My, Mz, f[4];
        body_rate(V, pc, qc, rc, wp, wq, wr, Mx, My, Mz);
        allocate(V, thrust_cmd, Mx, My, Mz, f);
        motors(V, f, om, omc);

        if (LOG_CMD) {
            double *c = cmd_log + (size_t)k * UAVAC_CMD_COLS * sB + b;
            c[0] = thrust_cmd; c[1 * sB] = pc; c[2 * sB] = qc; c[3 * sB] = rc;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[(4 + i) * sB] = omc[i]; c[(8 + i) * sB] = om[i]; }
        }

        free_body_step(V, om, px, py, pz, q0, q1, q2, q3, vx, vy, vz, wp, wq, wr);

        if (AABB) {
            for (int o = 0; o < n_obs; ++o) {
                const double *c = aabbs + 6 * o;          // uniform address: scalar loads
                const bool hit = (px >= c[0]) && (px <= c[1]) && (py >= c[2]) && (py <= c[3]) && (pz >= c[4]) &&
                                 (pz <= c[5]);            // inclusive, minimum_snap.py:352-357
                collided |= hit ? 1 : 0;
            }
        }

        if (LOG_STATE) {
            double *s = state_log + (size_t)k * 13 * sB + b;
            s[0] = px; s[1 * sB] = py; s[2 * sB] = pz;
            s[3 * sB] = q0; s[4 * sB] = q1; s[5 * sB] = q2; s[6 * sB] = q3;
            s[7 * sB] = vx; s[8 * sB] = vy; s[9 * sB] = vz;
            s[10 * sB] = wp; s[11 * sB] = wq; s[12 * sB] = wr;
        }
        ++inner;
        phase = (phase + 1 == V.F) ? 0 : phase + 1;
    }

    state[0 * sB + b] = px; state[1 * sB + b] = py; state[2 * sB + b] = pz;
    state[3 * sB + b] = q0; state[4 * sB + b] = q1; state[5 * sB + b] = q2; state[6 * sB + b] = q3;
    state[7 * sB + b] = vx; state[8 * sB + b] = vy; state[9 * sB + b] = vz;
    state[10 * sB + b] = wp; state[11 * sB + b] = wq; state[12 * sB + b] = wr;
#pragma unroll
    for (int i = 0; i < 4; ++i) { state[(13 + i) * sB + b] = om[i]; state[(17 + i) * sB + b] = omc[i]; }
    state[21 * sB + b] = integ;
    state[22 * sB + b] = thrust_cmd;
    state[23 * sB + b] = pc; state[24 * sB + b] = qc; state[25 * sB + b] = rc;
    istate[0 * sB + b] = idx;
    istate[1 * sB + b] = inner;
    istate[2 * sB + b] = collided;
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

template <bool LS, bool LC, bool AB>
void launch_variant(uavac_ctx *ctx, const VehK &V, const double *traj, const int64_t *row_offsets, double *state,
                    int32_t *istate, int B, int K, double *state_log, double *cmd_log, const double *aabbs, int n_obs) {
    hipLaunchKernelGGL((control_rollout_kernel<LS, LC, AB>), dim3((B + 63) / 64), dim3(64), 0, ctx->stream, V, traj,
                       row_offsets, state, istate, B, K, state_log, cmd_log, aabbs, n_obs);
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
                         int n_obs) {
    const bool ls = state_log != nullptr, lc = cmd_log != nullptr, ab = (aabbs != nullptr && n_obs > 0);
#define UAVAC_ARGS ctx, V, traj, row_offsets, state, istate, B, K, state_log, cmd_log, aabbs, n_obs
    if (ls) {
        if (lc) { if (ab) launch_variant<true, true, true>(UAVAC_ARGS); else launch_variant<true, true, false>(UAVAC_ARGS); }
        else    { if (ab) launch_variant<true, false, true>(UAVAC_ARGS); else launch_variant<true, false, false>(UAVAC_ARGS); }
    } else {
        if (lc) { if (ab) launch_variant<false, true, true>(UAVAC_ARGS); else launch_variant<false, true, false>(UAVAC_ARGS); }
        else    { if (ab) launch_variant<false, false, true>(UAVAC_ARGS); else launch_variant<false, false, false>(UAVAC_ARGS); }
    }
#undef UAVAC_ARGS
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}
