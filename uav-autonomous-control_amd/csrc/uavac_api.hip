// C ABI of libuavac.so: context management, argument validation, device-pointer entry points and
// their host-pointer twins (which stage through device scratch).  See include/uavac.h.

#include "uavac_internal.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

constexpr size_t kStageChunk = (size_t)256 << 10;     // bytes per half of the pinned ping-pong buffer
// Transfers up to this size go through the ctx's own pinned buffer (no pinning work inside the runtime, no allocation), up
// to four pieces with the DMA of one overlapping the CPU copy of its neighbour; larger ones are handed to hipMemcpyAsync as
// they are: the runtime's pageable path moves them at 50 GB/s aggregate (H2D + D2H of a 4 096-UAV rollout,
// tools/host_path_rate.py), a single-threaded copy through a staging buffer at 27.
constexpr size_t kStageLimit = (size_t)1 << 20;
static_assert(kStageLimit >= 2 * kStageChunk, "the ping-pong needs transfers of more than one piece to overlap anything");

// Host <-> device copies of the host-pointer twins.  The caller's buffers are pageable; they travel through the ctx's
// pinned staging buffer in kStageChunk pieces, two halves in flight: the DMA of one piece overlaps the CPU copy of the
// next (h2d) or previous (d2h) one.  Ordered on the ctx stream like everything else.
int h2d_staged(uavac_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) { return uavac_h2d(ctx, dst_dev, src_host, bytes); }
int d2h_staged(uavac_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) { return uavac_d2h(ctx, dst_host, src_dev, bytes); }
}  // namespace

int uavac_h2d(uavac_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!bytes) return UAVAC_OK;
    if (bytes > kStageLimit) {
        UAVAC_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return UAVAC_OK;                  // pageable source: the runtime has read it when the call returns
    }
    if (int rc = uavac_pin_reserve(ctx, 2 * kStageChunk)) return rc;
    const char *src = static_cast<const char *>(src_host);
    char *dst = static_cast<char *>(dst_dev);
    size_t i = 0;
    for (size_t off = 0; off < bytes; off += kStageChunk, ++i) {
        const size_t n = (bytes - off < kStageChunk) ? bytes - off : kStageChunk;
        char *half = ctx->h_pin + (i & 1) * kStageChunk;
        if (i >= 2) UAVAC_HIP(ctx, hipEventSynchronize(ctx->pin_ev[i & 1]));     // the DMA that last read this half
        std::memcpy(half, src + off, n);
        UAVAC_HIP(ctx, hipMemcpyAsync(dst + off, half, n, hipMemcpyHostToDevice, ctx->stream));
        UAVAC_HIP(ctx, hipEventRecord(ctx->pin_ev[i & 1], ctx->stream));
    }
    // the halves must not be rewritten by a later call before these DMAs have read them
    UAVAC_HIP(ctx, hipEventSynchronize(ctx->pin_ev[0]));
    if (i > 1) UAVAC_HIP(ctx, hipEventSynchronize(ctx->pin_ev[1]));
    return UAVAC_OK;
}

int uavac_d2h(uavac_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!bytes) return UAVAC_OK;
    if (bytes > kStageLimit) {
        UAVAC_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return UAVAC_OK;
    }
    if (int rc = uavac_pin_reserve(ctx, 2 * kStageChunk)) return rc;
    char *dst = static_cast<char *>(dst_host);
    const char *src = static_cast<const char *>(src_dev);
    const size_t n_chunks = (bytes + kStageChunk - 1) / kStageChunk;
    auto len = [&](size_t i) { return (i + 1 < n_chunks) ? kStageChunk : bytes - i * kStageChunk; };
    auto issue = [&](size_t i) -> int {
        UAVAC_HIP(ctx, hipMemcpyAsync(ctx->h_pin + (i & 1) * kStageChunk, src + i * kStageChunk, len(i),
                                      hipMemcpyDeviceToHost, ctx->stream));
        UAVAC_HIP(ctx, hipEventRecord(ctx->pin_ev[i & 1], ctx->stream));
        return UAVAC_OK;
    };
    if (int rc = issue(0)) return rc;
    for (size_t i = 0; i < n_chunks; ++i) {
        if (i + 1 < n_chunks)
            if (int rc = issue(i + 1)) return rc;                        // its half was drained one iteration ago
        UAVAC_HIP(ctx, hipEventSynchronize(ctx->pin_ev[i & 1]));
        std::memcpy(dst + i * kStageChunk, ctx->h_pin + (i & 1) * kStageChunk, len(i));
    }
    return UAVAC_OK;
}

namespace {

template <class T> T *take(uavac_ctx *ctx, size_t count) { return static_cast<T *>(uavac_arena_take(ctx, count * sizeof(T))); }

bool finite_all(const double *p, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(p[i])) return false;
    return true;
}

int check_plan_args(uavac_ctx *ctx, const void *wp, int B, int m) {
    if (!ctx) return UAVAC_EINVAL;
    if (!wp) return uavac_fail(ctx, UAVAC_EINVAL, "null waypoint pointer");
    if (B < 1) return uavac_fail(ctx, UAVAC_EINVAL, "B must be >= 1");
    if (m < 1 || m > UAVAC_MAX_SEGMENTS) return uavac_fail(ctx, UAVAC_EINVAL, "m must be in [1, UAVAC_MAX_SEGMENTS]");
    return UAVAC_OK;
}

// The device-side flags are sticky until read: the host twins clear them on entry (so that what an earlier _dev call
// left behind is not reported against valid input) and read-and-clear them on exit.
int clear_flags(uavac_ctx *ctx) {
    UAVAC_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int32_t), ctx->stream));
    return UAVAC_OK;
}
int read_flags(uavac_ctx *ctx, int32_t out[4]) {
    UAVAC_HIP(ctx, hipMemcpyAsync(out, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int32_t), ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

}  // namespace

int uavac_arena_reserve(uavac_ctx *ctx, size_t bytes) {
    if (bytes > ctx->arena_cap) {
        UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));          // nothing enqueued may still use the old block
        if (ctx->d_arena) UAVAC_HIP(ctx, hipFree(ctx->d_arena));
        ctx->d_arena = nullptr;
        ctx->arena_cap = 0;
        size_t want = bytes + bytes / 4;
        if (want < ((size_t)1 << 20)) want = (size_t)1 << 20;
        void *p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            want = bytes;
            UAVAC_HIP(ctx, hipMalloc(&p, want));
        }
        ctx->d_arena = static_cast<char *>(p);
        ctx->arena_cap = want;
    }
    ctx->arena_top = 0;
    return UAVAC_OK;
}

void *uavac_arena_take(uavac_ctx *ctx, size_t bytes) {
    const size_t n = uavac_arena_size(bytes ? bytes : 8);
    if (ctx->arena_top + n > ctx->arena_cap) return nullptr;         // reserve() was given too small a total: a bug
    void *p = ctx->d_arena + ctx->arena_top;
    ctx->arena_top += n;
    return p;
}

int uavac_scratch(uavac_ctx *ctx, size_t bytes, void **out) {
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(bytes))) return rc;
    *out = uavac_arena_take(ctx, bytes);
    return UAVAC_OK;
}

int uavac_pin_reserve(uavac_ctx *ctx, size_t bytes) {
    if (!ctx->pin_ev[0]) {
        UAVAC_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_ev[0], hipEventDisableTiming));
        UAVAC_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_ev[1], hipEventDisableTiming));
    }
    if (bytes <= ctx->pin_cap) return UAVAC_OK;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_pin) UAVAC_HIP(ctx, hipHostFree(ctx->h_pin));
    ctx->h_pin = nullptr;
    ctx->pin_cap = 0;
    void *p = nullptr;
    UAVAC_HIP(ctx, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    ctx->h_pin = static_cast<char *>(p);
    ctx->pin_cap = bytes;
    return UAVAC_OK;
}

VehK uavac_make_vehk(const uavac_vehicle &V) {
    VehK k{};
    k.g = V.g; k.dt = V.dt; k.dt_outer = V.dt_outer; k.mass = V.mass; k.inv_mass = 1.0 / V.mass;
    for (int i = 0; i < 3; ++i) { k.I[i] = V.inertia[i]; k.inv_I[i] = 1.0 / V.inertia[i]; }
    k.arm = V.arm; k.inv_arm = 1.0 / V.arm; k.kappa = V.kappa; k.inv_kappa = 1.0 / V.kappa;
    k.kf = V.kf; k.inv_kf = 1.0 / V.kf;
    k.min_thrust = V.min_thrust; k.max_thrust = V.max_thrust;
    k.c_min = 4.0 * V.min_thrust; k.c_max = 4.0 * V.max_thrust;
    k.resp_rise = 1.0 - std::exp(-V.dt / V.tau_rise);
    k.resp_fall = 1.0 - std::exp(-V.dt / V.tau_fall);
    k.max_ascent = V.max_ascent; k.max_descent = V.max_descent; k.max_speed_xy = V.max_speed_xy;
    k.max_horiz_accel = V.max_horiz_accel; k.max_tilt = V.max_tilt;
    k.kp_xy = V.kp_xy; k.kd_xy = V.kd_xy; k.kp_z = V.kp_z; k.kd_z = V.kd_z; k.ki_z = V.ki_z;
    k.kp_roll = V.kp_roll; k.kp_pitch = V.kp_pitch; k.kp_yaw = V.kp_yaw;
    k.ikp[0] = V.inertia[0] * V.kp_p; k.ikp[1] = V.inertia[1] * V.kp_q; k.ikp[2] = V.inertia[2] * V.kp_r;
    k.hover_omega = std::sqrt(V.mass * V.g / (4.0 * V.kf));
    k.lit_tiny = 2.2250738585072014e-308; k.lit_h2_small = 1.0e-3;
    k.lit_c8 = 1.0 / 40320; k.lit_c6 = -1.0 / 720; k.lit_c4 = 1.0 / 24;
    k.lit_s9 = 1.0 / 362880; k.lit_s7 = -1.0 / 5040; k.lit_s5 = 1.0 / 120; k.lit_s3 = -1.0 / 6;
    k.lit_375 = 0.375; k.lit_e_small = 1.0e-6;
    k.F = V.inner_per_outer;
    k.ground = V.ground ? 1 : 0;
    k.ground_z = V.ground_z;
    k.ground_zc = V.ground_z - V.ground_clearance;
    k.ground_k = k.ground ? 1.0 / (V.ground_timeconst * V.ground_timeconst) : 0.0;
    k.ground_b = k.ground ? 2.0 / V.ground_timeconst : 0.0;
    return k;
}

int uavac_check_vehicle(uavac_ctx *ctx, const uavac_vehicle *V) {
    if (!V) return uavac_fail(ctx, UAVAC_EINVAL, "null vehicle");
    const double pos[] = {V->g, V->dt, V->dt_outer, V->mass, V->inertia[0], V->inertia[1], V->inertia[2], V->arm,
                          V->kf, V->kappa, V->max_thrust, V->tau_rise, V->tau_fall};
    for (double v : pos)
        if (!(v > 0.0) || !std::isfinite(v)) return uavac_fail(ctx, UAVAC_EINVAL, "vehicle constant must be finite and > 0");
    if (!(V->min_thrust >= 0.0) || !(V->max_thrust > V->min_thrust))
        return uavac_fail(ctx, UAVAC_EINVAL, "thrust limits must satisfy 0 <= min < max");
    if (V->inner_per_outer < 1) return uavac_fail(ctx, UAVAC_EINVAL, "inner_per_outer must be >= 1");
    if (V->ground) {
        if (!std::isfinite(V->ground_z) || !(V->ground_clearance >= 0.0) || !std::isfinite(V->ground_clearance) ||
            !(V->ground_timeconst > 0.0) || !std::isfinite(V->ground_timeconst))
            return uavac_fail(ctx, UAVAC_EINVAL, "ground plane needs finite z, clearance >= 0 and time constant > 0");
        if (!(V->ground_timeconst >= 2.0 * V->dt))      // the explicit step of the contact law is stable for dt <= tc / 2
            return uavac_fail(ctx, UAVAC_EINVAL, "ground_timeconst must be at least 2 dt");
    }
    return UAVAC_OK;
}

extern "C" {

int uavac_version(void) { return UAVAC_VERSION; }

int uavac_create(uavac_ctx **out, int device_id) {
    if (!out) return UAVAC_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return UAVAC_EHIP;   // no CPU fallback, ever
    uavac_ctx *ctx = new (std::nothrow) uavac_ctx();
    if (!ctx) return UAVAC_ENOMEM;
    if (device_id >= 0) {
        if (device_id >= ndev) { delete ctx; return UAVAC_EHIP; }
        ctx->device = device_id;
    } else if (hipGetDevice(&ctx->device) != hipSuccess) { delete ctx; return UAVAC_EHIP; }
    uavac_device_guard guard(ctx);                // stream and buffers are created on the ctx's device
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) { delete ctx; return UAVAC_EHIP; }
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return UAVAC_EHIP; }
    ctx->stream = ctx->own_stream;
    if (hipMalloc(&ctx->d_flags, 4 * sizeof(int32_t)) != hipSuccess ||
        hipMemset(ctx->d_flags, 0, 4 * sizeof(int32_t)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return UAVAC_EHIP;
    }
    if (const char *e = getenv("UAVAC_YAW_GROUP")) { const int v = atoi(e); if (v == 1 || v == 4 || v == 16) ctx->yaw_group = v; }
    if (const char *e = getenv("UAVAC_ROLLOUT_ALIGN")) ctx->rollout_align = (e[0] == '0') ? 0 : 1;
    if (const char *e = getenv("UAVAC_SAMPLER_WAVES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ctx->sampler_waves = v; }
    if (const char *e = getenv("UAVAC_SAMPLER_GROUP")) { const int v = atoi(e); if (v >= 1 && v <= 64) ctx->sampler_group = v; }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0)
            ctx->n_simds = 4 * cus;
    }
    // The rollout's yaw scan (device library atan2) and the sampler's (uavac_yaw::heading) must agree bit for bit, or plan-fed and
    // row-fed flights part at np.unwrap's forks.  Both are compiled INTO this library (the device library is linked as bitcode at
    // build time), so the answer is a property of the build: checked on the device by the first context of the process (~0.1 ms)
    // and remembered -- later contexts skip it (round-5 advice).  A failure is not remembered: every uavac_create refuses.
    static std::atomic<int> heading_checked{0};
    if (!getenv("UAVAC_SKIP_SELFCHECK") && heading_checked.load(std::memory_order_acquire) == 0) {
        int bad = -1;
        const int rc = uavac_heading_selfcheck(ctx, &bad);
        if (rc == UAVAC_OK && bad == 0) heading_checked.store(1, std::memory_order_release);
        if (rc != UAVAC_OK || bad != 0) {
            if (rc == UAVAC_OK)
                fprintf(stderr, "libuavac: self-check failed: heading() and the atan2 of the device library this build linked differ on %d of "
                                "65680 operand pairs (built with: %s).  The device library's atan2 has changed: heading() in "
                                "csrc/minsnap_yaw.h (polynomial, operation order) must be re-derived from it -- rebuilding alone "
                                "reproduces this failure\n", bad, uavac_build_info());
            (void)hipFree(ctx->d_flags);
            (void)hipStreamDestroy(ctx->own_stream);
            delete ctx;
            return rc != UAVAC_OK ? rc : UAVAC_ETOOLCHAIN;
        }
    }
    *out = ctx;
    return UAVAC_OK;
}

void uavac_destroy(uavac_ctx *ctx) {
    if (!ctx) return;
    uavac_device_guard guard(ctx);
    (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->d_totals) (void)hipFree(ctx->d_totals);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_plan) (void)hipFree(ctx->d_plan);
    if (ctx->d_arena) (void)hipFree(ctx->d_arena);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (hipEvent_t e : ctx->pin_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *uavac_last_error(const uavac_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int uavac_set_stream(uavac_ctx *ctx, void *hip_stream) {
    if (!ctx) return UAVAC_EINVAL;
    ctx->stream = static_cast<hipStream_t>(hip_stream);      // NULL = HIP's legacy default stream
    return UAVAC_OK;
}

int uavac_reset_stream(uavac_ctx *ctx) {
    if (!ctx) return UAVAC_EINVAL;
    ctx->stream = ctx->own_stream;
    return UAVAC_OK;
}

int uavac_synchronize(uavac_ctx *ctx) {
    UAVAC_ENTER(ctx);
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_device(const uavac_ctx *ctx) { return ctx ? ctx->device : UAVAC_EINVAL; }

int uavac_set_option(uavac_ctx *ctx, const char *name, int value) {
    if (!ctx || !name) return UAVAC_EINVAL;
    const std::string n(name);
    if (n == "yaw_group") {
        if (value != 1 && value != 4 && value != 8 && value != 16) return uavac_fail(ctx, UAVAC_EINVAL, "yaw_group is 1, 4, 8 or 16");
        ctx->yaw_group = value;
    } else if (n == "sampler_waves") {
        if (value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return uavac_fail(ctx, UAVAC_EINVAL, "sampler_waves is 1, 2, 4, 8 or 16");
        ctx->sampler_waves = value;
    } else if (n == "sampler_group") {
        if (value < 1 || value > 64) return uavac_fail(ctx, UAVAC_EINVAL, "sampler_group is 1 .. 64");
        ctx->sampler_group = value;
    } else if (n == "rollout_align") {
        ctx->rollout_align = value ? 1 : 0;
    } else if (n == "log_pitch") {
        if (value < 0) return uavac_fail(ctx, UAVAC_EINVAL, "log_pitch is 0 (= B) or a number of doubles >= B");
        ctx->log_pitch = value;
    } else if (n == "late_handover") {
        ctx->late_handover = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "solve_order") {
        if (value != 0 && value != 1 && value != -1) return uavac_fail(ctx, UAVAC_EINVAL, "solve_order is 0 (one-ended), 1 (two-ended) or -1 (by the launch)");
        ctx->solve_order = value;
    } else if (n == "solve_park") {
        ctx->solve_park = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "solve_keep") {
        ctx->solve_keep = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "solve_lanes") {
        if (value != -1 && value != 64 && value != 32 && value != 16) return uavac_fail(ctx, UAVAC_EINVAL, "solve_lanes is -1, 64, 32 or 16");
        ctx->solve_lanes = value;
    } else if (n == "coeff_dma") {
        ctx->coeff_dma = value < 0 ? -1 : (value > 2 ? 2 : value);
    } else if (n == "idle_waves") {
        ctx->idle_waves = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "cu_balance") {
        ctx->cu_balance = value ? 1 : 0;
    } else if (n == "lds_pad") {
        if (value < 0 || value > 120 * 1024) return uavac_fail(ctx, UAVAC_EINVAL, "lds_pad is 0 .. 122880 bytes");
        ctx->lds_pad = value & ~7;
    } else {
        return uavac_fail(ctx, UAVAC_EINVAL, "unknown option");
    }
    return UAVAC_OK;
}

const char *uavac_last_rollout_kernel(const uavac_ctx *ctx) { return ctx ? ctx->last_rollout.c_str() : ""; }

int uavac_last_rollout_vgprs(const uavac_ctx *ctx) { return ctx ? ctx->last_rollout_vgprs : UAVAC_EINVAL; }

#define UAVAC_STR2(x) #x
#define UAVAC_STR(x) UAVAC_STR2(x)
const char *uavac_build_info(void) {
    return "libuavac " UAVAC_STR(UAVAC_VERSION) "; gfx950; HIP " UAVAC_STR(HIP_VERSION_MAJOR) "." UAVAC_STR(HIP_VERSION_MINOR) "."
           UAVAC_STR(HIP_VERSION_PATCH) "; " __VERSION__;
}

int uavac_device_identity(uavac_ctx *ctx, char *buf, int n) {
    UAVAC_ENTER(ctx);
    if (!buf || n < 1) return uavac_fail(ctx, UAVAC_EINVAL, "null buffer");
    hipDeviceProp_t prop;
    UAVAC_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    char uuid[33] = {0}, pci[32] = {0};
    for (int i = 0; i < 16; ++i) snprintf(uuid + 2 * i, 3, "%02x", (unsigned)(unsigned char)prop.uuid.bytes[i]);
    if (hipDeviceGetPCIBusId(pci, (int)sizeof pci, ctx->device) != hipSuccess) pci[0] = 0;
    snprintf(buf, (size_t)n, "uuid=%s;pci=%s;name=%s", uuid, pci, prop.gcnArchName);
    return UAVAC_OK;
}

int uavac_take_flags(uavac_ctx *ctx, int32_t flags[4]) {
    UAVAC_ENTER(ctx);
    if (!flags) return uavac_fail(ctx, UAVAC_EINVAL, "null flags");
    return read_flags(ctx, flags);
}

void uavac_vehicle_default(uavac_vehicle *V) {
    if (!V) return;
    std::memset(V, 0, sizeof(*V));
    V->g = 9.81; V->dt = 0.001; V->inner_per_outer = 10; V->dt_outer = V->dt * V->inner_per_outer;
    V->mass = 0.5; V->inertia[0] = 0.0023; V->inertia[1] = 0.0023; V->inertia[2] = 0.0046;
    V->arm = 0.120208; V->kf = 1.0; V->kappa = 0.016;
    V->min_thrust = 0.1; V->max_thrust = 4.5; V->tau_rise = 0.0125; V->tau_fall = 0.025;
    V->max_ascent = 3.0; V->max_descent = 2.0; V->max_speed_xy = 3.0; V->max_horiz_accel = 12.0; V->max_tilt = 0.7;
    // second_order_gains(tau, zeta) = (1/tau^2, 2 zeta/tau): quad.py:124-127 with the constants of :42-51
    V->kp_xy = 1.0 / (0.25 * 0.25); V->kd_xy = 2.0 * 0.875 / 0.25;
    V->kp_z = 1.0 / (0.2 * 0.2);    V->kd_z = 2.0 * 0.8 / 0.2;
    V->ki_z = 0.1;
    V->kp_roll = 1.0 / 0.07; V->kp_pitch = 1.0 / 0.07; V->kp_yaw = 1.0 / 0.25;
    V->kp_p = 1.0 / 0.008; V->kp_q = 1.0 / 0.008; V->kp_r = 1.0 / 0.09;
    // free flight by default; the plane of lab_course.xml:34 and the body box of :101 when switched on
    V->ground = 0; V->ground_z = 0.0; V->ground_clearance = 0.02; V->ground_timeconst = 0.02;
}

// ------------------------------------------------------------------------------- planning, device
int uavac_minsnap_row_counts_dev(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt,
                                 double *times, int32_t *seg_rows, int64_t *row_offsets) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!times || !seg_rows || !row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null output pointer");
    if (!std::isfinite(velocity) || !std::isfinite(dt)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite velocity or dt");
    if (!(velocity > 0.0) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity and dt must be > 0");
    return uavac_launch_row_counts(ctx, wp, B, m, velocity, dt, times, seg_rows, row_offsets);
}

int uavac_minsnap_solve_dev(uavac_ctx *ctx, const double *wp, const double *times, int B, int m, double *coeffs,
                            int32_t *status) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!times || !coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    return uavac_launch_solve_bt(ctx, wp, times, B, m, coeffs, status);
}

int uavac_minsnap_solve_banded_dev(uavac_ctx *ctx, const double *wp, const double *times, int B, int m,
                                   double *coeffs, int32_t *status) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!times || !coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    return uavac_launch_solve(ctx, wp, times, B, m, coeffs, status);
}

static int check_sample_args(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *row_offsets,
                             int B, int m, double dt, const double *traj) {
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (!seg_rows || !row_offsets || !traj) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    return UAVAC_OK;
}

int uavac_minsnap_sample_dev(uavac_ctx *ctx, const double *coeffs, const double *times, const int32_t *seg_rows,
                             const int64_t *row_offsets, int B, int m, double dt, double *traj) {
    (void)times;
    UAVAC_ENTER(ctx);
    if (int rc = check_sample_args(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj)) return rc;
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, SampleExtras{});
}

int uavac_minsnap_sample_yaw_dev(uavac_ctx *ctx, const double *coeffs, const double *times, const int32_t *seg_rows,
                                 const int64_t *row_offsets, int B, int m, double dt, double *traj, double *yaw) {
    (void)times;
    UAVAC_ENTER(ctx);
    if (int rc = check_sample_args(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj)) return rc;
    if (!yaw) return uavac_fail(ctx, UAVAC_EINVAL, "null yaw");
    SampleExtras x;
    x.yaw_dense = yaw;
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x);
}

int uavac_minsnap_sample_hits_dev(uavac_ctx *ctx, const double *coeffs, const double *times, const int32_t *seg_rows,
                                  const int64_t *row_offsets, int B, int m, double dt, double *traj,
                                  const double *aabb, int32_t *hit) {
    (void)times;
    UAVAC_ENTER(ctx);
    if (int rc = check_sample_args(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj)) return rc;
    if (!aabb || !hit) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    SampleExtras x;
    x.aabb = aabb;
    x.hit = hit;
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x);
}

int uavac_minsnap_sample_derivs_dev(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows,
                                    const int64_t *row_offsets, int B, int m, double dt, double *traj, double *yaw,
                                    double *first_yaw, double *jerk, double *snap) {
    UAVAC_ENTER(ctx);
    if (int rc = check_sample_args(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj)) return rc;
    SampleExtras x;
    x.yaw_dense = yaw;
    x.first_yaw = first_yaw;
    x.jerk = jerk;
    x.snap = snap;
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, m, dt, traj, x);
}

// ------------------------------------------------------------------------------- planning, device, ragged batches
int uavac_minsnap_row_counts_ragged_dev(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B, int max_m,
                                        double velocity, double dt, double *times, int32_t *seg_rows, int64_t *row_offsets) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, max_m)) return rc;
    if (!seg_offsets || !times || !seg_rows || !row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(velocity) || !std::isfinite(dt)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite velocity or dt");
    if (!(velocity > 0.0) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity and dt must be > 0");
    return uavac_launch_row_counts(ctx, wp, B, max_m, velocity, dt, times, seg_rows, row_offsets, seg_offsets);
}

int uavac_minsnap_solve_ragged_dev(uavac_ctx *ctx, const double *wp, const double *times, const int64_t *seg_offsets, int B,
                                   int max_m, double *coeffs, int32_t *status) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, max_m)) return rc;
    if (!seg_offsets || !times || !coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    return uavac_launch_solve_bt(ctx, wp, times, B, max_m, coeffs, status, seg_offsets);
}

int uavac_minsnap_sample_ragged_dev(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *seg_offsets,
                                    const int64_t *row_offsets, int B, int max_m, int64_t total_segments, double dt,
                                    double *traj, int64_t traj_capacity_rows, const double *aabb, int32_t *hit,
                                    double *first_yaw) {
    UAVAC_ENTER(ctx);
    if (int rc = check_sample_args(ctx, coeffs, seg_rows, row_offsets, B, max_m, dt, traj)) return rc;
    if (!seg_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null seg_offsets");
    if (total_segments < B || total_segments > (int64_t)B * max_m)
        return uavac_fail(ctx, UAVAC_EINVAL, "total_segments must lie in [B, B * max_m]");
    if ((aabb == nullptr) != (hit == nullptr)) return uavac_fail(ctx, UAVAC_EINVAL, "aabb and hit go together");
    SampleExtras x;
    x.aabb = aabb;
    x.hit = hit;
    x.first_yaw = first_yaw;
    x.capacity_rows = traj_capacity_rows;
    x.seg_offsets = seg_offsets;
    x.total_segments = total_segments;
    return uavac_launch_sample(ctx, coeffs, seg_rows, row_offsets, B, max_m, dt, traj, x);
}

int uavac_minsnap_plan_dev(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt, double *times,
                           int32_t *seg_rows, int64_t *row_offsets, double *coeffs, int32_t *status, double *traj,
                           int64_t traj_capacity_rows, double *yaw, double *first_yaw) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!times || !seg_rows || !row_offsets || !coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(velocity) || !std::isfinite(dt)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite velocity or dt");
    if (!(velocity > 0.0) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity and dt must be > 0");
    if (!traj) {
        // Rows-free chain: no row buffer, hence nothing to refuse -- times, row counts and offsets go straight into the caller's
        // arrays; the one value of the sampler a plan-fed rollout needs comes from the first-heading kernel.
        if (yaw) return uavac_fail(ctx, UAVAC_EINVAL, "a dense yaw column needs the rows: traj is NULL");
        if (int rc = uavac_launch_row_counts(ctx, wp, B, m, velocity, dt, times, seg_rows, row_offsets)) return rc;
        if (int rc = uavac_launch_solve_bt(ctx, wp, times, B, m, coeffs, status)) return rc;
        return first_yaw ? uavac_launch_first_yaw(ctx, coeffs, seg_rows, nullptr, B, m, dt, first_yaw) : UAVAC_OK;
    }
    if (traj_capacity_rows < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative capacity");
    // The whole chain enqueued from here: nothing returns to the caller (or to an interpreter) between the launches.
    // Times, row counts and offsets go to ctx scratch first; whether the plan fits the caller's row buffer is only known on
    // the device (row_offsets_s[B]), so every later stage reads that one word: the commit kernel copies the three arrays
    // into the caller's only when it fits, the solver and the sampler do nothing when it does not (the sampler raises
    // flag 2).  A refused plan leaves times, seg_rows, row_offsets, coeffs, rows and first_yaw exactly as they were.
    const size_t nseg = (size_t)B * m;
    const size_t o_rows = uavac_arena_size(nseg * 8), o_offs = o_rows + uavac_arena_size(nseg * 4);
    const size_t need = o_offs + uavac_arena_size(((size_t)B + 1) * 8);
    if (need > ctx->plan_cap) {
        UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));          // nothing enqueued may still use the old block
        if (ctx->d_plan) UAVAC_HIP(ctx, hipFree(ctx->d_plan));
        ctx->d_plan = nullptr;
        ctx->plan_cap = 0;
        void *p = nullptr;
        UAVAC_HIP(ctx, hipMalloc(&p, need));
        ctx->d_plan = static_cast<char *>(p);
        ctx->plan_cap = need;
    }
    double *times_s = reinterpret_cast<double *>(ctx->d_plan);
    int32_t *seg_rows_s = reinterpret_cast<int32_t *>(ctx->d_plan + o_rows);
    int64_t *row_offsets_s = reinterpret_cast<int64_t *>(ctx->d_plan + o_offs);
    if (int rc = uavac_launch_row_counts(ctx, wp, B, m, velocity, dt, times_s, seg_rows_s, row_offsets_s)) return rc;
    if (int rc = uavac_launch_plan_commit(ctx, times_s, seg_rows_s, row_offsets_s, B, m, traj_capacity_rows, times, seg_rows,
                                          row_offsets)) return rc;
    // (Round 6 tried to hide this solve behind the sampler: the batch cut into 2 / 4 / 8 mission blocks, block i sampled on an
    // auxiliary stream while block i + 1 was being solved.  Bit-identical and SLOWER -- +1.3 % / +13 % / +23 % at 65 536 x 12 --
    // because a solve wave cannot get onto a SIMD the sampler's grid keeps full: commit 53921dc, profiles/r06_plan_blocks_ab*.jsonl.)
    if (int rc = uavac_launch_solve_bt(ctx, wp, times_s, B, m, coeffs, status, nullptr, row_offsets_s + B, traj_capacity_rows))
        return rc;
    SampleExtras x;
    x.yaw_dense = yaw;
    x.first_yaw = first_yaw;
    x.capacity_rows = traj_capacity_rows;         // the sampler refuses (flag 2) instead of overrunning the buffer
    return uavac_launch_sample(ctx, coeffs, seg_rows_s, row_offsets_s, B, m, dt, traj, x);
}

int uavac_minsnap_first_yaw_dev(uavac_ctx *ctx, const double *coeffs, const int32_t *seg_rows, const int64_t *seg_offsets, int B,
                                int m, double dt, double *first_yaw) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (!seg_rows || !first_yaw) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    return uavac_launch_first_yaw(ctx, coeffs, seg_rows, seg_offsets, B, m, dt, first_yaw);
}

int uavac_minsnap_row_offsets_dev(uavac_ctx *ctx, const int32_t *seg_rows, int B, int m, int64_t *row_offsets) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, seg_rows, B, m)) return rc;
    if (!row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null row_offsets");
    return uavac_launch_row_offsets(ctx, seg_rows, B, m, row_offsets);
}

int uavac_minsnap_row_offsets_ragged_dev(uavac_ctx *ctx, const int32_t *seg_rows, const int64_t *seg_offsets, int B, int max_m,
                                         int64_t *row_offsets) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, seg_rows, B, max_m)) return rc;
    if (!seg_offsets || !row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    return uavac_launch_row_offsets(ctx, seg_rows, B, max_m, row_offsets, seg_offsets);
}

int uavac_minsnap_obstacle_round_dev(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B, int max_m,
                                     double velocity, double dt, const double *aabb, int32_t *active, int32_t *overflow,
                                     int32_t *touched, double *wp_out, int64_t *seg_offsets_out, int32_t *counters,
                                     double *times, int32_t *seg_rows, int64_t *row_offsets, double *coeffs, int32_t *hit) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, max_m)) return rc;
    if (!seg_offsets || !aabb || !active || !overflow || !touched || !wp_out || !seg_offsets_out || !counters || !times ||
        !seg_rows || !row_offsets || !coeffs || !hit)
        return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(velocity) || !std::isfinite(dt)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite velocity or dt");
    if (!(velocity > 0.0) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity and dt must be > 0");
    UAVAC_HIP(ctx, hipMemsetAsync(counters, 0, 4 * sizeof(int32_t), ctx->stream));
    if (int rc = uavac_launch_row_counts(ctx, wp, B, max_m, velocity, dt, times, seg_rows, row_offsets, seg_offsets)) return rc;
    if (int rc = uavac_launch_solve_bt(ctx, wp, times, B, max_m, coeffs, nullptr, seg_offsets, nullptr, 0, active)) return rc;
    return uavac_launch_obstacle_scan_and_insert(ctx, wp, seg_offsets, coeffs, seg_rows, B, max_m, dt, aabb, active, overflow,
                                                 touched, hit, wp_out, seg_offsets_out, counters);
}

int uavac_yaw_scan_dev(uavac_ctx *ctx, const double *velocities, const int64_t *offsets, int B, double *yaws) {
    UAVAC_ENTER(ctx);
    if (!velocities || !offsets || !yaws || B < 1) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    return uavac_launch_yaw_scan(ctx, velocities, offsets, B, yaws);
}

// --------------------------------------------------------------------------------- planning, host
int uavac_minsnap_row_counts(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double dt,
                             double *times, int32_t *seg_rows, int64_t *row_offsets) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "null row_offsets");
    const size_t nwp = (size_t)B * (m + 1) * 3, nseg = (size_t)B * m;
    if (!finite_all(wp, nwp) || !std::isfinite(velocity) || !std::isfinite(dt))
        return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite waypoint, velocity or dt");
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(nwp * 8) + uavac_arena_size(nseg * 8) +
                                              uavac_arena_size(nseg * 4) + uavac_arena_size(((size_t)B + 1) * 8))) return rc;
    double *dwp = take<double>(ctx, nwp), *dt_ = take<double>(ctx, nseg);
    int32_t *dsr = take<int32_t>(ctx, nseg);
    int64_t *dro = take<int64_t>(ctx, (size_t)B + 1);
    if (int rc = clear_flags(ctx)) return rc;
    if (int rc = h2d_staged(ctx, dwp, wp, nwp * 8)) return rc;
    if (int rc = uavac_minsnap_row_counts_dev(ctx, dwp, B, m, velocity, dt, dt_, dsr, dro)) return rc;
    if (times) if (int rc = d2h_staged(ctx, times, dt_, nseg * 8)) return rc;
    if (seg_rows) if (int rc = d2h_staged(ctx, seg_rows, dsr, nseg * 4)) return rc;
    if (int rc = d2h_staged(ctx, row_offsets, dro, ((size_t)B + 1) * 8)) return rc;
    int32_t fl[4];
    if (int rc = read_flags(ctx, fl)) return rc;
    if (fl[0]) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite segment duration");
    if (fl[3]) return uavac_fail(ctx, UAVAC_EINVAL, "a mission has more than 2^31-1 rows");
    return UAVAC_OK;
}

int uavac_minsnap_solve(uavac_ctx *ctx, const double *wp, int B, int m, double velocity, double *coeffs,
                        double *times) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, wp, B, m)) return rc;
    if (!coeffs) return uavac_fail(ctx, UAVAC_EINVAL, "null coeffs");
    const size_t nwp = (size_t)B * (m + 1) * 3, nseg = (size_t)B * m, nco = (size_t)B * 24 * m;
    if (!finite_all(wp, nwp) || !std::isfinite(velocity))
        return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite waypoint or velocity");
    if (!(velocity > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity must be > 0");
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(nwp * 8) + uavac_arena_size(nseg * 8) +
                                              uavac_arena_size(nseg * 4) + uavac_arena_size(((size_t)B + 1) * 8) +
                                              uavac_arena_size(nco * 8))) return rc;
    double *dwp = take<double>(ctx, nwp), *dt_ = take<double>(ctx, nseg);
    int32_t *dsr = take<int32_t>(ctx, nseg);
    int64_t *dro = take<int64_t>(ctx, (size_t)B + 1);
    double *dco = take<double>(ctx, nco);
    if (int rc = clear_flags(ctx)) return rc;
    if (int rc = h2d_staged(ctx, dwp, wp, nwp * 8)) return rc;
    // dt only shapes the row counts, which this entry point does not return
    if (int rc = uavac_launch_row_counts(ctx, dwp, B, m, velocity, 1.0, dt_, dsr, dro)) return rc;
    if (int rc = uavac_launch_solve_bt(ctx, dwp, dt_, B, m, dco, nullptr)) return rc;
    if (int rc = d2h_staged(ctx, coeffs, dco, nco * 8)) return rc;
    if (times) if (int rc = d2h_staged(ctx, times, dt_, nseg * 8)) return rc;
    int32_t fl[4];
    if (int rc = read_flags(ctx, fl)) return rc;
    if (fl[0]) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite segment duration");
    if (fl[1]) return uavac_fail(ctx, UAVAC_ESINGULAR, "singular knot system (repeated waypoint?)");
    return UAVAC_OK;
}

int uavac_minsnap_sample(uavac_ctx *ctx, const double *coeffs, const double *times, int B, int m, double dt,
                         const int64_t *row_offsets, double *traj) {
    UAVAC_ENTER(ctx);
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (!times || !row_offsets || !traj) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    const size_t nseg = (size_t)B * m, nco = (size_t)B * 24 * m;
    // rows per segment, as uavac_minsnap_row_counts computes them; must agree with the caller's offsets
    std::vector<int32_t> sr(nseg);
    for (int b = 0; b < B; ++b) {
        int64_t tot = 0;
        for (int s = 0; s < m; ++s) {
            double q = std::ceil(times[(size_t)b * m + s] / dt);
            int32_t r = (std::isfinite(q) && q > 0.0 && q < 2.0e9) ? (int32_t)q : 0;
            sr[(size_t)b * m + s] = r;
            tot += r;
        }
        if (row_offsets[b + 1] - row_offsets[b] != tot)
            return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets inconsistent with ceil(times/dt)");
    }
    const int64_t total = row_offsets[B] - row_offsets[0];
    if (row_offsets[0] != 0 || total < 0) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must start at 0");
    const size_t ntr = (size_t)total * UAVAC_TRAJ_COLS;
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(nco * 8) + uavac_arena_size(nseg * 4) +
                                              uavac_arena_size(((size_t)B + 1) * 8) + uavac_arena_size(ntr * 8))) return rc;
    double *dco = take<double>(ctx, nco);
    int32_t *dsr = take<int32_t>(ctx, nseg);
    int64_t *dro = take<int64_t>(ctx, (size_t)B + 1);
    double *dtr = take<double>(ctx, ntr);
    if (int rc = h2d_staged(ctx, dco, coeffs, nco * 8)) return rc;
    if (int rc = h2d_staged(ctx, dsr, sr.data(), nseg * 4)) return rc;
    if (int rc = h2d_staged(ctx, dro, row_offsets, ((size_t)B + 1) * 8)) return rc;
    if (int rc = uavac_launch_sample(ctx, dco, dsr, dro, B, m, dt, dtr, SampleExtras{})) return rc;
    if (int rc = d2h_staged(ctx, traj, dtr, ntr * 8)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

// One call for a ragged batch on host buffers: durations, row offsets, coefficients and rows.  `traj` may be NULL (or too small:
// traj_capacity_rows): then everything but the rows is produced, row_offsets[B] says how many rows a second call needs.
int uavac_minsnap_plan_ragged(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B, double velocity, double dt,
                              double *times, int64_t *row_offsets, double *coeffs, double *traj, int64_t traj_capacity_rows) {
    UAVAC_ENTER(ctx);
    if (B < 1 || !wp || !seg_offsets || !row_offsets) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    if (seg_offsets[0] != 0) return uavac_fail(ctx, UAVAC_EINVAL, "seg_offsets must start at 0");
    int max_m = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t mb = seg_offsets[b + 1] - seg_offsets[b];
        if (mb < 1 || mb > UAVAC_MAX_SEGMENTS)
            return uavac_fail(ctx, UAVAC_EINVAL, "every mission needs 1 .. UAVAC_MAX_SEGMENTS segments");
        if (mb > max_m) max_m = (int)mb;
    }
    const size_t S = (size_t)seg_offsets[B], nwp = (S + (size_t)B) * 3, nco = S * 24;
    if (!finite_all(wp, nwp) || !std::isfinite(velocity) || !std::isfinite(dt))
        return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite waypoint, velocity or dt");
    if (!(velocity > 0.0) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity and dt must be > 0");
    const size_t fixed = uavac_arena_size(nwp * 8) + 2 * uavac_arena_size(((size_t)B + 1) * 8) + uavac_arena_size(S * 8) +
                         uavac_arena_size(S * 4) + uavac_arena_size(nco * 8);
    auto stage_inputs = [&](size_t row_bytes, double *&dwp, int64_t *&dso, double *&dtm, int32_t *&dsr, int64_t *&dro,
                            double *&dco, double *&dtr) -> int {
        if (int rc = uavac_arena_reserve(ctx, fixed + uavac_arena_size(row_bytes))) return rc;
        dwp = take<double>(ctx, nwp); dso = take<int64_t>(ctx, (size_t)B + 1); dtm = take<double>(ctx, S);
        dsr = take<int32_t>(ctx, S); dro = take<int64_t>(ctx, (size_t)B + 1); dco = take<double>(ctx, nco);
        dtr = row_bytes ? take<double>(ctx, row_bytes / 8) : nullptr;
        if (int rc = h2d_staged(ctx, dwp, wp, nwp * 8)) return rc;
        return h2d_staged(ctx, dso, seg_offsets, ((size_t)B + 1) * 8);
    };
    double *dwp, *dtm, *dco, *dtr;
    int64_t *dso, *dro;
    int32_t *dsr;
    if (int rc = clear_flags(ctx)) return rc;
    if (int rc = stage_inputs(0, dwp, dso, dtm, dsr, dro, dco, dtr)) return rc;
    if (int rc = uavac_launch_row_counts(ctx, dwp, B, max_m, velocity, dt, dtm, dsr, dro, dso)) return rc;
    if (int rc = d2h_staged(ctx, row_offsets, dro, ((size_t)B + 1) * 8)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int32_t fl[4];
    if (int rc = read_flags(ctx, fl)) return rc;
    if (fl[0]) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite segment duration");
    if (fl[3]) return uavac_fail(ctx, UAVAC_EINVAL, "a mission has more than 2^31-1 rows");
    const int64_t total = row_offsets[B];
    const bool rows = traj && traj_capacity_rows >= total && total > 0;
    if (rows) {             // the arena grows for the rows: stage again (cheap next to the rows) and redo the counts
        if (int rc = stage_inputs((size_t)total * UAVAC_TRAJ_COLS * 8, dwp, dso, dtm, dsr, dro, dco, dtr)) return rc;
        if (int rc = uavac_launch_row_counts(ctx, dwp, B, max_m, velocity, dt, dtm, dsr, dro, dso)) return rc;
    }
    if (int rc = uavac_launch_solve_bt(ctx, dwp, dtm, B, max_m, dco, nullptr, dso)) return rc;
    if (rows) {
        SampleExtras x;
        x.seg_offsets = dso;
        x.total_segments = (int64_t)S;
        x.capacity_rows = total;
        if (int rc = uavac_launch_sample(ctx, dco, dsr, dro, B, max_m, dt, dtr, x)) return rc;
        if (int rc = d2h_staged(ctx, traj, dtr, (size_t)total * UAVAC_TRAJ_COLS * 8)) return rc;
    }
    if (times) if (int rc = d2h_staged(ctx, times, dtm, S * 8)) return rc;
    if (coeffs) if (int rc = d2h_staged(ctx, coeffs, dco, nco * 8)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (int rc = read_flags(ctx, fl)) return rc;
    if (fl[1]) return uavac_fail(ctx, UAVAC_ESINGULAR, "singular knot system (repeated waypoint?)");
    if (traj && !rows && total > 0) return uavac_fail(ctx, UAVAC_EINVAL, "traj_capacity_rows < row_offsets[B]: nothing sampled");
    return UAVAC_OK;
}

// The obstacle loop of MinimumSnap._generate_collision_free_trajectory (minimum_snap.py:63-95) for B ragged missions on HOST
// buffers: every round is uavac_minsnap_obstacle_round_dev on device scratch of this ctx, the host keeps the bookkeeping the
// reference keeps (obstacles in order, earlier ones not re-checked; a bounded number of rounds per obstacle; optional re-check
// sweeps over the missions that received midpoints).  Returns the final waypoint lists; sample them with
// uavac_minsnap_plan_ragged.
int uavac_minsnap_obstacle_waypoints(uavac_ctx *ctx, const double *wp, const int64_t *seg_offsets, int B, double velocity,
                                     double dt, const double *cuboids, int n_cuboids, int max_iterations, int recheck_passes,
                                     double *wp_out, int64_t wp_capacity, int64_t *seg_offsets_out, int32_t *converged) {
    UAVAC_ENTER(ctx);
    if (B < 1 || !wp || !seg_offsets || !wp_out || !seg_offsets_out || n_cuboids < 0 || (n_cuboids > 0 && !cuboids))
        return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null pointer");
    if (seg_offsets[0] != 0) return uavac_fail(ctx, UAVAC_EINVAL, "seg_offsets must start at 0");
    if (max_iterations < 0 || recheck_passes < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative iteration count");
    int max_m = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t mb = seg_offsets[b + 1] - seg_offsets[b];
        if (mb < 1 || mb > UAVAC_MAX_SEGMENTS)
            return uavac_fail(ctx, UAVAC_EINVAL, "every mission needs 1 .. UAVAC_MAX_SEGMENTS segments");
        if (mb > max_m) max_m = (int)mb;
    }
    const size_t S0 = (size_t)seg_offsets[B], nB = (size_t)B;
    if (!finite_all(wp, (S0 + nB) * 3) || (n_cuboids && !finite_all(cuboids, (size_t)n_cuboids * 6)) || !std::isfinite(velocity) ||
        !std::isfinite(dt))
        return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite waypoint, cuboid, velocity or dt");
    if (!(velocity > 0.0) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "velocity and dt must be > 0");
    const size_t S_cap = nB * UAVAC_MAX_SEGMENTS, W_cap = (S_cap + nB) * 3;         // no mission ever has more segments
    if (int rc = uavac_arena_reserve(ctx, 2 * uavac_arena_size(W_cap * 8) + 2 * uavac_arena_size((nB + 1) * 8) +
                                              4 * uavac_arena_size(nB * 4) + uavac_arena_size(16) + uavac_arena_size(S_cap * 8) +
                                              2 * uavac_arena_size(S_cap * 4) + uavac_arena_size((nB + 1) * 8) +
                                              uavac_arena_size(S_cap * 192) + uavac_arena_size((size_t)(n_cuboids ? n_cuboids : 1) * 48)))
        return rc;
    double *wp_a = take<double>(ctx, W_cap), *wp_b = take<double>(ctx, W_cap);
    int64_t *so_a = take<int64_t>(ctx, nB + 1), *so_b = take<int64_t>(ctx, nB + 1);
    int32_t *d_active = take<int32_t>(ctx, nB), *d_overflow = take<int32_t>(ctx, nB), *d_touched = take<int32_t>(ctx, nB);
    int32_t *d_spare = take<int32_t>(ctx, nB);
    (void)d_spare;
    int32_t *d_counters = take<int32_t>(ctx, 4);
    double *d_times = take<double>(ctx, S_cap);
    int32_t *d_seg_rows = take<int32_t>(ctx, S_cap), *d_hit = take<int32_t>(ctx, S_cap);
    int64_t *d_row_offsets = take<int64_t>(ctx, nB + 1);
    double *d_coeffs = take<double>(ctx, S_cap * 24);
    double *d_cub = take<double>(ctx, (size_t)(n_cuboids ? n_cuboids : 1) * 6);
    if (int rc = clear_flags(ctx)) return rc;
    if (int rc = h2d_staged(ctx, wp_a, wp, (S0 + nB) * 24)) return rc;
    if (int rc = h2d_staged(ctx, so_a, seg_offsets, (nB + 1) * 8)) return rc;
    if (n_cuboids) if (int rc = h2d_staged(ctx, d_cub, cuboids, (size_t)n_cuboids * 48)) return rc;
    UAVAC_HIP(ctx, hipMemsetAsync(d_overflow, 0, nB * 4, ctx->stream));
    std::vector<int32_t> todo(nB, 1), failed(nB, 0), active(nB), touched(nB), overflow(nB);
    for (int sweep = 0; sweep <= recheck_passes && n_cuboids > 0; ++sweep) {
        UAVAC_HIP(ctx, hipMemsetAsync(d_touched, 0, nB * 4, ctx->stream));
        for (int c = 0; c < n_cuboids; ++c) {
            int n_active = 0;
            for (size_t b = 0; b < nB; ++b) { active[b] = todo[b] && !failed[b]; n_active += active[b]; }
            if (int rc = h2d_staged(ctx, d_active, active.data(), nB * 4)) return rc;
            int it = 0;
            for (; it <= max_iterations && n_active > 0; ++it) {
                if (int rc = uavac_minsnap_obstacle_round_dev(ctx, wp_a, so_a, B, max_m, velocity, dt, d_cub + 6 * c, d_active,
                                                              d_overflow, d_touched, wp_b, so_b, d_counters, d_times, d_seg_rows,
                                                              d_row_offsets, d_coeffs, d_hit)) return rc;
                std::swap(wp_a, wp_b);
                std::swap(so_a, so_b);
                int32_t cnt[4];
                if (int rc = d2h_staged(ctx, cnt, d_counters, 16)) return rc;           // the round's one read-back (synchronises)
                n_active = cnt[0];
                if (cnt[2] > max_m) max_m = cnt[2];
                if (cnt[1]) {                                                           // outgrew UAVAC_MAX_SEGMENTS: stays as it is
                    if (int rc = d2h_staged(ctx, overflow.data(), d_overflow, nB * 4)) return rc;
                    for (size_t b = 0; b < nB; ++b) failed[b] |= overflow[b];
                }
            }
            if (n_active > 0) {                                                         // the bounded loop ran out
                if (int rc = d2h_staged(ctx, active.data(), d_active, nB * 4)) return rc;
                for (size_t b = 0; b < nB; ++b) failed[b] |= (active[b] != 0);
            }
        }
        if (int rc = d2h_staged(ctx, touched.data(), d_touched, nB * 4)) return rc;
        int again = 0;
        for (size_t b = 0; b < nB; ++b) { todo[b] = touched[b] && !failed[b]; again += todo[b]; }
        if (!again) break;                                                              // only missions that changed can have new conflicts
    }
    if (int rc = d2h_staged(ctx, seg_offsets_out, so_a, (nB + 1) * 8)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n_wp = seg_offsets_out[B] + B;
    int32_t fl[4];
    if (int rc = read_flags(ctx, fl)) return rc;
    if (fl[0]) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite segment duration");
    if (n_wp > wp_capacity) return uavac_fail(ctx, UAVAC_EINVAL, "wp_capacity too small: seg_offsets_out[B] + B waypoints are needed");
    if (int rc = d2h_staged(ctx, wp_out, wp_a, (size_t)n_wp * 24)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (converged) for (size_t b = 0; b < nB; ++b) converged[b] = failed[b] ? 0 : 1;
    // a singular knot system (a repeated waypoint) gives NaN coefficients, and NaN positions are inside no cuboid: such a
    // mission would pass for collision-free.  The outputs are complete, the call says so (like uavac_minsnap_solve).
    if (fl[1]) return uavac_fail(ctx, UAVAC_ESINGULAR, "a mission's knot system is singular (repeated waypoint?): its collision scan is void");
    return UAVAC_OK;
}

int uavac_yaw_scan(uavac_ctx *ctx, const double *velocities, int64_t n, double *yaws) {
    UAVAC_ENTER(ctx);
    if (n < 0 || (n > 0 && (!velocities || !yaws))) return uavac_fail(ctx, UAVAC_EINVAL, "bad n or null pointer");
    if (n == 0) return UAVAC_OK;
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size((size_t)n * 24) + uavac_arena_size((size_t)n * 8) +
                                              uavac_arena_size(16))) return rc;
    double *dv = take<double>(ctx, (size_t)n * 3), *dy = take<double>(ctx, (size_t)n);
    int64_t *doff = take<int64_t>(ctx, 2);
    const int64_t off[2] = {0, n};
    if (int rc = h2d_staged(ctx, dv, velocities, (size_t)n * 24)) return rc;
    if (int rc = h2d_staged(ctx, doff, off, 16)) return rc;
    if (int rc = uavac_launch_yaw_scan(ctx, dv, doff, 1, dy)) return rc;
    if (int rc = d2h_staged(ctx, yaws, dy, (size_t)n * 8)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

// -------------------------------------------------------------------------------- control, device
int uavac_state_init_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *positions, int B, int hover,
                         double *state, int32_t *istate) {
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || !state || !istate) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null state");
    return uavac_launch_state_init(ctx, uavac_make_vehk(*V), positions, B, hover, state, istate);
}

int uavac_control_rollout_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                              double *state, int32_t *istate, int B, int K, double *state_log, double *cmd_log,
                              const double *aabbs, int n_obs) {
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (B < 1 || K < 0 || !traj || !row_offsets || !state || !istate || n_obs < 0)
        return uavac_fail(ctx, UAVAC_EINVAL, "bad size or null pointer");
    if (K == 0) return UAVAC_OK;
    if ((state_log || cmd_log) && ctx->log_pitch > 0 && ctx->log_pitch < B)
        return uavac_fail(ctx, UAVAC_EINVAL, "option log_pitch is smaller than B");
    return uavac_launch_rollout(ctx, uavac_make_vehk(*V), traj, row_offsets, state, istate, B, K, state_log, cmd_log,
                                aabbs, n_obs);
}

int uavac_control_rollout_plan_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *coeffs, const int32_t *seg_rows,
                                   const int64_t *row_offsets, const double *yaw, const double *first_yaw, int m, double dt,
                                   double *state, int32_t *istate, int B, int K, double *state_log, double *cmd_log,
                                   const double *aabbs, int n_obs) {
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (int rc = check_plan_args(ctx, coeffs, B, m)) return rc;
    if (K < 0 || !seg_rows || !row_offsets || !state || !istate || n_obs < 0)
        return uavac_fail(ctx, UAVAC_EINVAL, "bad size or null pointer");
    if (!yaw && !first_yaw) return uavac_fail(ctx, UAVAC_EINVAL, "need the dense yaw column or the missions' first headings");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    if (K == 0) return UAVAC_OK;
    if ((state_log || cmd_log) && ctx->log_pitch > 0 && ctx->log_pitch < B)
        return uavac_fail(ctx, UAVAC_EINVAL, "option log_pitch is smaller than B");
    PlanRef plan;
    plan.coeffs = coeffs; plan.seg_rows = seg_rows; plan.yaw = yaw; plan.first_yaw = first_yaw; plan.dt = dt; plan.m = m;
    return uavac_launch_rollout(ctx, uavac_make_vehk(*V), nullptr, row_offsets, state, istate, B, K, state_log, cmd_log,
                                aabbs, n_obs, &plan);
}

int uavac_control_rollout_plan_ragged_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *coeffs, const int32_t *seg_rows,
                                          const int64_t *seg_offsets, const int64_t *row_offsets, const double *first_yaw,
                                          int max_m, double dt, double *state, int32_t *istate, int B, int K,
                                          double *state_log, double *cmd_log, const double *aabbs, int n_obs) {
    UAVAC_ENTER(ctx);
    if (int rc = uavac_check_vehicle(ctx, V)) return rc;
    if (int rc = check_plan_args(ctx, coeffs, B, max_m)) return rc;
    if (K < 0 || !seg_rows || !seg_offsets || !row_offsets || !first_yaw || !state || !istate || n_obs < 0)
        return uavac_fail(ctx, UAVAC_EINVAL, "bad size or null pointer");
    if (!std::isfinite(dt) || !(dt > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "dt must be finite and > 0");
    if (K == 0) return UAVAC_OK;
    if ((state_log || cmd_log) && ctx->log_pitch > 0 && ctx->log_pitch < B)
        return uavac_fail(ctx, UAVAC_EINVAL, "option log_pitch is smaller than B");
    PlanRef plan;
    plan.coeffs = coeffs; plan.seg_rows = seg_rows; plan.first_yaw = first_yaw; plan.dt = dt; plan.m = max_m;
    plan.seg_offsets = seg_offsets;
    return uavac_launch_rollout(ctx, uavac_make_vehk(*V), nullptr, row_offsets, state, istate, B, K, state_log, cmd_log,
                                aabbs, n_obs, &plan);
}

int uavac_control_step_dev(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                           double *state, int32_t *istate, int B) {
    return uavac_control_rollout_dev(ctx, V, traj, row_offsets, state, istate, B, 1, nullptr, nullptr, nullptr, 0);
}

// ---------------------------------------------------------------------------------- control, host
int uavac_state_init(uavac_ctx *ctx, const uavac_vehicle *V, const double *positions, int B, int hover,
                     double *state, int32_t *istate) {
    UAVAC_ENTER(ctx);
    if (B < 1 || !state || !istate) return uavac_fail(ctx, UAVAC_EINVAL, "bad B or null state");
    if (positions && !finite_all(positions, (size_t)B * 3)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite position");
    const size_t ns = (size_t)B * UAVAC_STATE_ROWS, ni = (size_t)B * UAVAC_ISTATE_ROWS;
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(ns * 8) + uavac_arena_size(ni * 4) +
                                              uavac_arena_size((size_t)B * 24))) return rc;
    double *ds = take<double>(ctx, ns);
    int32_t *di = take<int32_t>(ctx, ni);
    double *dp = positions ? take<double>(ctx, (size_t)B * 3) : nullptr;
    if (positions) if (int rc = h2d_staged(ctx, dp, positions, (size_t)B * 24)) return rc;
    if (int rc = uavac_state_init_dev(ctx, V, dp, B, hover, ds, di)) return rc;
    if (int rc = d2h_staged(ctx, state, ds, ns * 8)) return rc;
    if (int rc = d2h_staged(ctx, istate, di, ni * 4)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_control_rollout(uavac_ctx *ctx, const uavac_vehicle *V, const double *traj, const int64_t *row_offsets,
                          double *state, int32_t *istate, int B, int K, double *state_log, double *cmd_log,
                          const double *aabbs, int n_obs) {
    UAVAC_ENTER(ctx);
    if (B < 1 || K < 0 || !traj || !row_offsets || !state || !istate || n_obs < 0)
        return uavac_fail(ctx, UAVAC_EINVAL, "bad size or null pointer");
    if (row_offsets[0] != 0) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must start at 0");
    for (int b = 0; b < B; ++b)
        if (row_offsets[b + 1] < row_offsets[b]) return uavac_fail(ctx, UAVAC_EINVAL, "row_offsets must be non-decreasing");
    const size_t total = (size_t)row_offsets[B];
    if (!finite_all(state, (size_t)B * UAVAC_STATE_ROWS)) return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite state");
    const size_t ntr = total * UAVAC_TRAJ_COLS, ns = (size_t)B * UAVAC_STATE_ROWS, ni = (size_t)B * UAVAC_ISTATE_ROWS;
    const size_t nsl = state_log ? (size_t)K * 13 * B : 0, ncl = cmd_log ? (size_t)K * UAVAC_CMD_COLS * B : 0;
    const bool obs = aabbs && n_obs > 0;
    if (int rc = uavac_arena_reserve(ctx, uavac_arena_size(ntr * 8) + uavac_arena_size(((size_t)B + 1) * 8) +
                                              uavac_arena_size(ns * 8) + uavac_arena_size(ni * 4) +
                                              uavac_arena_size(nsl * 8) + uavac_arena_size(ncl * 8) +
                                              uavac_arena_size((size_t)n_obs * 48))) return rc;
    double *dtr = take<double>(ctx, ntr);
    int64_t *dro = take<int64_t>(ctx, (size_t)B + 1);
    double *ds = take<double>(ctx, ns);
    int32_t *di = take<int32_t>(ctx, ni);
    double *dsl = state_log ? take<double>(ctx, nsl) : nullptr;
    double *dcl = cmd_log ? take<double>(ctx, ncl) : nullptr;
    double *dab = obs ? take<double>(ctx, (size_t)n_obs * 6) : nullptr;
    if (int rc = h2d_staged(ctx, dtr, traj, ntr * 8)) return rc;
    if (int rc = h2d_staged(ctx, dro, row_offsets, ((size_t)B + 1) * 8)) return rc;
    if (int rc = h2d_staged(ctx, ds, state, ns * 8)) return rc;
    if (int rc = h2d_staged(ctx, di, istate, ni * 4)) return rc;
    if (obs) if (int rc = h2d_staged(ctx, dab, aabbs, (size_t)n_obs * 48)) return rc;
    {
        const int64_t pitch = ctx->log_pitch;          // the host logs are dense [K][13 | 12][B]
        ctx->log_pitch = 0;
        const int rc = uavac_control_rollout_dev(ctx, V, dtr, dro, ds, di, B, K, dsl, dcl, dab, obs ? n_obs : 0);
        ctx->log_pitch = pitch;
        if (rc) return rc;
    }
    if (int rc = d2h_staged(ctx, state, ds, ns * 8)) return rc;
    if (int rc = d2h_staged(ctx, istate, di, ni * 4)) return rc;
    if (state_log) if (int rc = d2h_staged(ctx, state_log, dsl, nsl * 8)) return rc;
    if (cmd_log) if (int rc = d2h_staged(ctx, cmd_log, dcl, ncl * 8)) return rc;
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

}  // extern "C"
