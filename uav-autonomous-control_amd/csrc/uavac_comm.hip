// Multi-GPU side of the C ABI: the ONE exchange of the path -- the final gather of the ragged
// trajectory row blocks to a root rank -- over RCCL point-to-point operations on xGMI.
//
// One process per GPU, one uavac_ctx each; missions are sharded by contiguous index blocks and every kernel works on
// its own shard, so nothing is exchanged while planning or flying.  At the end every peer owns one direct xGMI link
// to the root: ncclGroupStart + one ncclRecv per peer on the root / one ncclSend on each peer + ncclGroupEnd
// (rccl.h:700,722,923) lets the 7 transfers of an 8-GPU node run concurrently, each on its own link.  A ring
// all-gather would push 7/8 of the total through every link and deliver to ranks that do not want the data.
// (The reference has no counterpart: it plans and flies one mission per process, uav_ac/main.py:87-120.)

#include "uavac_internal.h"

#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

int nccl_fail(uavac_ctx *ctx, const char *what, ncclResult_t r) {
    ctx->err = std::string(what) + ": " + ncclGetErrorString(r);
    return UAVAC_ECOMM;
}

#define UAVAC_NCCL(ctx, call)                                                                    \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) return nccl_fail((ctx), #call, r_);                               \
    } while (0)

// After the operations of a group have been enqueued and the stream has drained: did the communicator see an
// asynchronous failure (a peer that died, a transport error)?
int check_async(uavac_ctx *ctx, ncclComm_t comm) {
    ncclResult_t async = ncclSuccess;
    UAVAC_NCCL(ctx, ncclCommGetAsyncError(comm, &async));
    if (async != ncclSuccess) return nccl_fail(ctx, "asynchronous RCCL error", async);
    return UAVAC_OK;
}

int comm_shape(uavac_ctx *ctx, ncclComm_t comm, int *world, int *rank) {
    UAVAC_NCCL(ctx, ncclCommCount(comm, world));
    UAVAC_NCCL(ctx, ncclCommUserRank(comm, rank));
    return UAVAC_OK;
}

}  // namespace

extern "C" {

int uavac_comm_unique_id(uavac_ctx *ctx, char id[UAVAC_COMM_ID_BYTES]) {
    UAVAC_ENTER(ctx);
    if (!id) return uavac_fail(ctx, UAVAC_EINVAL, "null id");
    static_assert(UAVAC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "uavac.h and rccl.h disagree on the id size");
    ncclUniqueId u;
    UAVAC_NCCL(ctx, ncclGetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return UAVAC_OK;
}

int uavac_comm_init_rank(uavac_ctx *ctx, const char id[UAVAC_COMM_ID_BYTES], int world, int rank, void **nccl_comm) {
    UAVAC_ENTER(ctx);
    if (!id || !nccl_comm) return uavac_fail(ctx, UAVAC_EINVAL, "null id or output pointer");
    if (world < 1 || rank < 0 || rank >= world) return uavac_fail(ctx, UAVAC_EINVAL, "need 0 <= rank < world");
    *nccl_comm = nullptr;
    // This library is compiled against /opt/rocm's rccl.h and binds, at run time, whichever librccl.so.1 the process
    // mapped first -- in a Python process that is the RCCL bundled with torch (2.26 against a 2.27 header on this image).
    // Only entry points whose signatures have been stable since NCCL 2.10 are called (group, send / recv, all-gather,
    // communicator set-up and queries) and the only structure that crosses is the 128-byte ncclUniqueId, so a different
    // MINOR version is accepted; a different major version, or a runtime older than 2.10, is refused here instead of
    // failing somewhere inside a collective.
    {
        int rt = 0;
        UAVAC_NCCL(ctx, ncclGetVersion(&rt));
        const int rt_major = rt >= 10000 ? rt / 10000 : rt / 1000;
        const int rt_minor = rt >= 10000 ? (rt / 100) % 100 : (rt / 100) % 10;
        if (rt_major != NCCL_MAJOR || rt_minor < 10) {
            char msg[160];
            snprintf(msg, sizeof msg, "RCCL runtime version %d does not match the rccl.h this library was built with (%d)",
                     rt, NCCL_VERSION_CODE);
            return uavac_fail(ctx, UAVAC_ECOMM, msg);
        }
    }
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    UAVAC_NCCL(ctx, ncclCommInitRank(&comm, world, u, rank));     // binds to the current device = ctx->device
    *nccl_comm = comm;
    return UAVAC_OK;
}

int uavac_comm_destroy(uavac_ctx *ctx, void *nccl_comm) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm) return UAVAC_OK;
    UAVAC_NCCL(ctx, ncclCommDestroy(static_cast<ncclComm_t>(nccl_comm)));
    return UAVAC_OK;
}

int uavac_comm_abort(uavac_ctx *ctx, void *nccl_comm) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm) return UAVAC_OK;
    UAVAC_NCCL(ctx, ncclCommAbort(static_cast<ncclComm_t>(nccl_comm)));
    return UAVAC_OK;
}

int uavac_comm_shape(uavac_ctx *ctx, void *nccl_comm, int *world, int *rank) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm || !world || !rank) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    return comm_shape(ctx, static_cast<ncclComm_t>(nccl_comm), world, rank);
}

int uavac_gather_counts(uavac_ctx *ctx, void *nccl_comm, int64_t n_rows, int64_t *counts) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm || !counts) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (n_rows < 0) return uavac_fail(ctx, UAVAC_EINVAL, "n_rows must be >= 0");
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int world = 0, rank = 0;
    if (int rc = comm_shape(ctx, comm, &world, &rank)) return rc;
    int64_t *d = nullptr;                                          // [1 + world] i64 of device scratch
    if (int rc = uavac_scratch(ctx, (size_t)(1 + world) * 8, reinterpret_cast<void **>(&d))) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(d, &n_rows, 8, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_NCCL(ctx, ncclAllGather(d, d + 1, 1, ncclInt64, comm, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(counts, d + 1, (size_t)world * 8, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_async(ctx, comm);
}

int uavac_gather_rows_dev(uavac_ctx *ctx, void *nccl_comm, const double *rows, int64_t n_rows, int row_elems,
                          const int64_t *counts, int root, double *out) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm || !counts) return uavac_fail(ctx, UAVAC_EINVAL, "null communicator or counts");
    if (row_elems < 1 || n_rows < 0) return uavac_fail(ctx, UAVAC_EINVAL, "bad row shape");
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int world = 0, rank = 0;
    if (int rc = comm_shape(ctx, comm, &world, &rank)) return rc;
    if (root < 0 || root >= world) return uavac_fail(ctx, UAVAC_EINVAL, "root out of range");
    for (int r = 0; r < world; ++r)
        if (counts[r] < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (counts[rank] != n_rows) return uavac_fail(ctx, UAVAC_EINVAL, "counts[rank] != n_rows");
    if (n_rows > 0 && !rows) return uavac_fail(ctx, UAVAC_EINVAL, "null rows");
    const size_t re = (size_t)row_elems;
    if (rank == root) {
        std::vector<size_t> off((size_t)world + 1, 0);
        for (int r = 0; r < world; ++r) off[r + 1] = off[r] + (size_t)counts[r];
        if (!out && off[world] > 0) return uavac_fail(ctx, UAVAC_EINVAL, "null out on the root");
        if (n_rows > 0 && out + off[rank] * re != rows)            // the root's own block: a device-to-device copy
            UAVAC_HIP(ctx, hipMemcpyAsync(out + off[rank] * re, rows, (size_t)n_rows * re * 8, hipMemcpyDeviceToDevice,
                                          ctx->stream));
        // one grouped launch: the receives run concurrently, one per direct xGMI link into the root
        UAVAC_NCCL(ctx, ncclGroupStart());
        for (int r = 0; r < world; ++r) {
            if (r == root || counts[r] == 0) continue;
            ncclResult_t res = ncclRecv(out + off[r] * re, (size_t)counts[r] * re, ncclDouble, r, comm, ctx->stream);
            if (res != ncclSuccess) { (void)ncclGroupEnd(); return nccl_fail(ctx, "ncclRecv", res); }
        }
        UAVAC_NCCL(ctx, ncclGroupEnd());
    } else if (n_rows > 0) {
        UAVAC_NCCL(ctx, ncclGroupStart());
        ncclResult_t res = ncclSend(rows, (size_t)n_rows * re, ncclDouble, root, comm, ctx->stream);
        if (res != ncclSuccess) { (void)ncclGroupEnd(); return nccl_fail(ctx, "ncclSend", res); }
        UAVAC_NCCL(ctx, ncclGroupEnd());
    }
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;                    // enqueued on the ctx stream; uavac_comm_finish() synchronises and checks
}

// Segments [first[r], first[r] + count[r]) of every rank's block (whose sizes are seg_counts) travel to the root and land where
// the gather of the whole blocks puts them.  An array travels when it is non-NULL (the caller keeps that consistent over ranks).
static int gather_plan_ranges(uavac_ctx *ctx, ncclComm_t comm, int world, int rank, const double *coeffs, const double *times,
                              const int32_t *seg_rows, const int64_t *seg_counts, const int64_t *first, const int64_t *count, int root,
                              double *coeffs_out, double *times_out, int32_t *seg_rows_out) {
    const size_t f = (size_t)first[rank], n = (size_t)count[rank];
    if (rank == root) {
        std::vector<size_t> off((size_t)world + 1, 0);
        for (int r = 0; r < world; ++r) off[r + 1] = off[r] + (size_t)seg_counts[r];
        const size_t o = off[rank] + f;
        if (n > 0) {                                                // the root's own block: device-to-device copies
            if (coeffs && coeffs_out + o * 24 != coeffs + f * 24)
                UAVAC_HIP(ctx, hipMemcpyAsync(coeffs_out + o * 24, coeffs + f * 24, n * 24 * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (times_out && times && times_out + o != times + f)
                UAVAC_HIP(ctx, hipMemcpyAsync(times_out + o, times + f, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (seg_rows && seg_rows_out + o != seg_rows + f)
                UAVAC_HIP(ctx, hipMemcpyAsync(seg_rows_out + o, seg_rows + f, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        // one grouped launch for all peers and all arrays: the receives of different peers run concurrently, one per
        // direct xGMI link into the root
        UAVAC_NCCL(ctx, ncclGroupStart());
        ncclResult_t res = ncclSuccess;
        for (int r = 0; r < world && res == ncclSuccess; ++r) {
            if (r == root || count[r] == 0) continue;
            const size_t c = (size_t)count[r], at = off[r] + (size_t)first[r];
            if (coeffs_out) res = ncclRecv(coeffs_out + at * 24, c * 24, ncclDouble, r, comm, ctx->stream);
            if (res == ncclSuccess && times_out) res = ncclRecv(times_out + at, c, ncclDouble, r, comm, ctx->stream);
            if (res == ncclSuccess && seg_rows_out) res = ncclRecv(seg_rows_out + at, c, ncclInt32, r, comm, ctx->stream);
        }
        if (res != ncclSuccess) { (void)ncclGroupEnd(); return nccl_fail(ctx, "ncclRecv", res); }
        UAVAC_NCCL(ctx, ncclGroupEnd());
    } else if (n > 0) {
        UAVAC_NCCL(ctx, ncclGroupStart());
        ncclResult_t res = ncclSuccess;
        if (coeffs) res = ncclSend(coeffs + f * 24, n * 24, ncclDouble, root, comm, ctx->stream);
        if (res == ncclSuccess && times) res = ncclSend(times + f, n, ncclDouble, root, comm, ctx->stream);
        if (res == ncclSuccess && seg_rows) res = ncclSend(seg_rows + f, n, ncclInt32, root, comm, ctx->stream);
        if (res != ncclSuccess) { (void)ncclGroupEnd(); return nccl_fail(ctx, "ncclSend", res); }
        UAVAC_NCCL(ctx, ncclGroupEnd());
    }
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;                    // enqueued on the ctx stream; uavac_comm_finish() synchronises and checks
}

int uavac_gather_plan_dev(uavac_ctx *ctx, void *nccl_comm, const double *coeffs, const double *times, const int32_t *seg_rows,
                          int64_t n_segments, const int64_t *seg_counts, int root, double *coeffs_out, double *times_out,
                          int32_t *seg_rows_out) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm || !seg_counts) return uavac_fail(ctx, UAVAC_EINVAL, "null communicator or counts");
    if (n_segments < 0) return uavac_fail(ctx, UAVAC_EINVAL, "n_segments must be >= 0");
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int world = 0, rank = 0;
    if (int rc = comm_shape(ctx, comm, &world, &rank)) return rc;
    if (root < 0 || root >= world) return uavac_fail(ctx, UAVAC_EINVAL, "root out of range");
    for (int r = 0; r < world; ++r)
        if (seg_counts[r] < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (seg_counts[rank] != n_segments) return uavac_fail(ctx, UAVAC_EINVAL, "seg_counts[rank] != n_segments");
    if (n_segments > 0 && (!coeffs || !seg_rows)) return uavac_fail(ctx, UAVAC_EINVAL, "null coeffs or seg_rows");
    if (rank == root) {
        int64_t total = 0;
        for (int r = 0; r < world; ++r) total += seg_counts[r];
        if (total > 0 && (!coeffs_out || !seg_rows_out)) return uavac_fail(ctx, UAVAC_EINVAL, "null output on the root");
        if (n_segments > 0 && (times_out != nullptr) != (times != nullptr))
            return uavac_fail(ctx, UAVAC_EINVAL, "times and times_out go together (on every rank, or on none)");
    }
    std::vector<int64_t> zero((size_t)world, 0);
    return gather_plan_ranges(ctx, comm, world, rank, coeffs, times, seg_rows, seg_counts, zero.data(), seg_counts, root, coeffs_out,
                              times_out, seg_rows_out);
}

int uavac_gather_plan_part_dev(uavac_ctx *ctx, void *nccl_comm, const double *coeffs, const double *times, const int32_t *seg_rows,
                               const int64_t *seg_counts, const int64_t *part_first, const int64_t *part_counts, int root,
                               double *coeffs_out, double *times_out, int32_t *seg_rows_out) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm || !seg_counts || !part_first || !part_counts) return uavac_fail(ctx, UAVAC_EINVAL, "null communicator or counts");
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int world = 0, rank = 0;
    if (int rc = comm_shape(ctx, comm, &world, &rank)) return rc;
    if (root < 0 || root >= world) return uavac_fail(ctx, UAVAC_EINVAL, "root out of range");
    for (int r = 0; r < world; ++r)
        if (seg_counts[r] < 0 || part_first[r] < 0 || part_counts[r] < 0 || part_first[r] + part_counts[r] > seg_counts[r])
            return uavac_fail(ctx, UAVAC_EINVAL, "a part must lie inside its rank's block");
    if (!coeffs && !times && !seg_rows && part_counts[rank] > 0) return uavac_fail(ctx, UAVAC_EINVAL, "nothing to send");
    if (rank == root) {
        // what the root receives is decided by its *_out pointers; its own block must offer the same arrays
        if (!coeffs_out && !times_out && !seg_rows_out) return uavac_fail(ctx, UAVAC_EINVAL, "null output on the root");
        if (part_counts[rank] > 0 && ((coeffs_out != nullptr) != (coeffs != nullptr) || (times_out != nullptr) != (times != nullptr) ||
                                      (seg_rows_out != nullptr) != (seg_rows != nullptr)))
            return uavac_fail(ctx, UAVAC_EINVAL, "the root's own arrays and its outputs must name the same arrays");
    }
    return gather_plan_ranges(ctx, comm, world, rank, coeffs, times, seg_rows, seg_counts, part_first, part_counts, root, coeffs_out,
                              times_out, seg_rows_out);
}

int uavac_comm_versions(int *built_with, int *runtime) {
    if (built_with) *built_with = NCCL_VERSION_CODE;
    int rt = 0;
    if (ncclGetVersion(&rt) != ncclSuccess) return UAVAC_ECOMM;
    if (runtime) *runtime = rt;
    return UAVAC_OK;
}

int uavac_comm_finish(uavac_ctx *ctx, void *nccl_comm) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm) return uavac_fail(ctx, UAVAC_EINVAL, "null communicator");
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_async(ctx, static_cast<ncclComm_t>(nccl_comm));
}

int uavac_comm_loopback_dev(uavac_ctx *ctx, void *nccl_comm, const double *src, double *dst, int64_t n) {
    UAVAC_ENTER(ctx);
    if (!nccl_comm || !src || !dst || n < 1) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer or n < 1");
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int world = 0, rank = 0;
    if (int rc = comm_shape(ctx, comm, &world, &rank)) return rc;
    // the very pair of calls the gather uses, with this rank as its own peer: exercises RCCL's send/recv path on
    // a single GPU (a self send/recv must sit in one group)
    UAVAC_NCCL(ctx, ncclGroupStart());
    ncclResult_t a = ncclSend(src, (size_t)n, ncclDouble, rank, comm, ctx->stream);
    ncclResult_t b = ncclRecv(dst, (size_t)n, ncclDouble, rank, comm, ctx->stream);
    ncclResult_t e = ncclGroupEnd();
    if (a != ncclSuccess) return nccl_fail(ctx, "ncclSend (loopback)", a);
    if (b != ncclSuccess) return nccl_fail(ctx, "ncclRecv (loopback)", b);
    if (e != ncclSuccess) return nccl_fail(ctx, "ncclGroupEnd (loopback)", e);
    return UAVAC_OK;
}

}  // extern "C"
