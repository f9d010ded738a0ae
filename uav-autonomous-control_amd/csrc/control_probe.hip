// Per-function probes of the control law (gfx950): the same __device__ functions the fused rollout
// kernel inlines, exposed one stage at a time so that the reference's unit-level known answers
// (tests/unit/control/test_controller.py, tests/unit/quadrotor/test_quad.py upstream) and the
// single-UAV facade (uav_ac.control.controller.CascadedController, uav_ac.quadrotor.quad.Quad)
// run on the HIP path.  Records are array-of-structs; batches are small; not a hot path.

#include "control_law.h"

namespace {

using namespace uavac_dev;

// in [B][41]: X(13) | R(9) | target(11) | integ | thrust_in | bxy_in(2) | euler_in(3: phi theta psi) | q_cmd_in
// out[B][21]: R(9) | phi theta psi | thrust | integ' | bxy(2) | pq(2) | pqr(3)
__global__ void probe_outer_kernel(const VehK V, const double *__restrict__ in, int B, int mask,
                                   double *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *r = in + (size_t)b * UAVAC_PROBE_OUTER_IN;
    double *o = out + (size_t)b * UAVAC_PROBE_OUTER_OUT;
    const double px = r[0], py = r[1], pz = r[2], q0 = r[3], q1 = r[4], q2 = r[5], q3 = r[6];
    const double vx = r[7], vy = r[8], vz = r[9];
    const double *tg = r + 22;
    Rot R = quat_to_rot(q0, q1, q2, q3);
    if (mask & UAVAC_PROBE_USE_R) {
        R.r00 = r[13]; R.r01 = r[14]; R.r02 = r[15]; R.r10 = r[16]; R.r11 = r[17]; R.r12 = r[18];
        R.r20 = r[19]; R.r21 = r[20]; R.r22 = r[21];
    }
    o[0] = R.r00; o[1] = R.r01; o[2] = R.r02; o[3] = R.r10; o[4] = R.r11; o[5] = R.r12; o[6] = R.r20; o[7] = R.r21;
    o[8] = R.r22;
    // Euler angles of the stored quaternion (quad.py:189-213)
    double phi = atan2(2.0 * (q0 * q1 + q2 * q3), 1.0 - 2.0 * (q1 * q1 + q2 * q2));
    double theta = asin(clampd(2.0 * (q0 * q2 - q3 * q1), -1.0, 1.0));
    double psi, cth, sphi, cphi;
    euler_trig(q0, q1, q2, q3, psi, cth, sphi, cphi);
    if (mask & UAVAC_PROBE_USE_EULER) {
        phi = r[37]; theta = r[38]; psi = r[39];
        cth = cos(theta); sphi = sin(phi); cphi = cos(phi);
    }
    o[9] = phi; o[10] = theta; o[11] = psi;
    double integ = r[33];
    double thrust = altitude(V, tg[2], tg[5], tg[8], pz, vz, R.r22, integ);
    o[12] = thrust; o[13] = integ;
    if (mask & UAVAC_PROBE_USE_THRUST) thrust = r[34];
    double bxc, byc;
    lateral(V, tg[0], tg[3], tg[6], tg[1], tg[4], tg[7], px, py, vx, vy, thrust, bxc, byc);
    o[14] = bxc; o[15] = byc;
    if (mask & UAVAC_PROBE_USE_BXY) { bxc = r[35]; byc = r[36]; }
    double pc, qc;
    roll_pitch(V, bxc, byc, R, pc, qc);
    o[16] = pc; o[17] = qc;
    const double qcmd = (mask & UAVAC_PROBE_USE_QCMD) ? r[40] : qc;
    o[18] = pc; o[19] = qc;
    o[20] = yaw_rate(V, tg[9], psi, cth, sphi, cphi, qcmd);
}

// in [B][24]: X(13) | pqr_cmd(3) | thrust | omega(4) | moment_in(3)
// out[B][15]: moment(3) | rotor forces(4) | omega_command(4) | omega'(4)
__global__ void probe_inner_kernel(const VehK V, const double *__restrict__ in, int B, int mask,
                                   double *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *r = in + (size_t)b * UAVAC_PROBE_INNER_IN;
    double *o = out + (size_t)b * UAVAC_PROBE_INNER_OUT;
    double Mx, My, Mz;
    body_rate(V, r[13], r[14], r[15], r[10], r[11], r[12], Mx, My, Mz);
    o[0] = Mx; o[1] = My; o[2] = Mz;
    if (mask & UAVAC_PROBE_USE_MOMENT) { Mx = r[21]; My = r[22]; Mz = r[23]; }
    double f[4], om[4] = {r[17], r[18], r[19], r[20]}, omc[4];
    allocate(V, r[16], Mx, My, Mz, f);
    motors(V, f, om, omc);
    for (int i = 0; i < 4; ++i) { o[3 + i] = f[i]; o[7 + i] = omc[i]; o[11 + i] = om[i]; }
}

struct Scratch {
    void *p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
};

template <class Kern>
int run_probe(uavac_ctx *ctx, const uavac_vehicle *V, Kern kern, const double *in, int nin, int B, int mask,
              double *out, int nout) {
    if (!ctx) return UAVAC_EINVAL;
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || !in || !out) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    Scratch din, dout;
    UAVAC_HIP(ctx, hipMalloc(&din.p, (size_t)B * nin * 8));
    UAVAC_HIP(ctx, hipMalloc(&dout.p, (size_t)B * nout * 8));
    UAVAC_HIP(ctx, hipMemcpyAsync(din.p, in, (size_t)B * nin * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(kern, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, uavac_make_vehk(*V),
                       static_cast<const double *>(din.p), B, mask, static_cast<double *>(dout.p));
    UAVAC_HIP(ctx, hipGetLastError());
    UAVAC_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)B * nout * 8, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

}  // namespace

extern "C" {

int uavac_probe_outer(uavac_ctx *ctx, const uavac_vehicle *V, const double *in, int B, int mask, double *out) {
    return run_probe(ctx, V, probe_outer_kernel, in, UAVAC_PROBE_OUTER_IN, B, mask, out, UAVAC_PROBE_OUTER_OUT);
}

int uavac_probe_inner(uavac_ctx *ctx, const uavac_vehicle *V, const double *in, int B, int mask, double *out) {
    return run_probe(ctx, V, probe_inner_kernel, in, UAVAC_PROBE_INNER_IN, B, mask, out, UAVAC_PROBE_INNER_OUT);
}

}  // extern "C"
