// Probe: can the sampler's fast mode be PRODUCED instead of waited for?  The row buffer as a virtual range backed by
// separately created physical chunks (hipMemCreate / hipMemMap), mapped in creation order or in a shuffled order, for
// several chunk sizes -- next to a plain hipMalloc buffer.  The whole planning chain runs through the C ABI
// (uavac_minsnap_plan_dev); only the sampler (the last of its four launches) depends on the row buffer.
// Build: hipcc --offload-arch=gfx950 -O2 tools/vmm_placement_probe.cpp -Iinclude -Luav-autonomous-control_amd/lib -luavac \
//        -Wl,-rpath,$PWD/uav-autonomous-control_amd/lib -o tools/vmm_placement_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "uavac.h"
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define UA(x) do { int r_ = (x); if (r_) { printf("%s failed %d: %s\n", #x, r_, uavac_last_error(ctx)); exit(1); } } while (0)

struct Mapped { void *va = nullptr; size_t size = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
static Mapped map_chunks(size_t bytes, size_t chunk, bool shuffle) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    chunk = (chunk + gran - 1) / gran * gran;
    Mapped m;
    const size_t n = (bytes + chunk - 1) / chunk;
    m.size = n * chunk;
    HIP(hipMemAddressReserve(&m.va, m.size, 0, nullptr, 0));
    m.h.resize(n);
    for (size_t i = 0; i < n; ++i) HIP(hipMemCreate(&m.h[i], chunk, &prop, 0));
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    if (shuffle) { std::mt19937_64 g(12345); std::shuffle(order.begin(), order.end(), g); }
    for (size_t i = 0; i < n; ++i) HIP(hipMemMap((char *)m.va + i * chunk, chunk, 0, m.h[order[i]], 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    HIP(hipMemSetAccess(m.va, m.size, &acc, 1));
    return m;
}
static void unmap(Mapped &m) {
    HIP(hipMemUnmap(m.va, m.size));
    for (auto h : m.h) HIP(hipMemRelease(h));
    HIP(hipMemAddressFree(m.va, m.size));
}

int main() {
    const int B = 65536, m = 12;
    uavac_ctx *ctx = nullptr;
    UA(uavac_create(&ctx, 0));
    std::vector<double> wp((size_t)B * (m + 1) * 3);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> N(0, 1);
    for (int b = 0; b < B; ++b) {
        double p[3] = {24 * U(g), 14 * U(g), -3.0};
        for (int k = 0; k <= m; ++k) {
            for (int a = 0; a < 3; ++a) wp[((size_t)b * (m + 1) + k) * 3 + a] = p[a];
            double d[3] = {N(g), N(g), 0.25 * N(g)}, n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), L = 2.5 + U(g);
            for (int a = 0; a < 3; ++a) p[a] += L * d[a] / n;
        }
    }
    double *dwp, *times, *coeffs, *first_yaw;
    int32_t *seg_rows, *status;
    int64_t *row_offsets;
    HIP(hipMalloc(&dwp, wp.size() * 8)); HIP(hipMalloc(&times, (size_t)B * m * 8)); HIP(hipMalloc(&coeffs, (size_t)B * m * 192));
    HIP(hipMalloc(&first_yaw, (size_t)B * 8)); HIP(hipMalloc(&seg_rows, (size_t)B * m * 4)); HIP(hipMalloc(&status, (size_t)B * 4));
    HIP(hipMalloc(&row_offsets, ((size_t)B + 1) * 8));
    HIP(hipMemcpy(dwp, wp.data(), wp.size() * 8, hipMemcpyHostToDevice));
    UA(uavac_minsnap_row_counts_dev(ctx, dwp, B, m, 3.0, 0.01, times, seg_rows, row_offsets));
    HIP(hipDeviceSynchronize());
    int64_t total = 0;
    HIP(hipMemcpy(&total, row_offsets + B, 8, hipMemcpyDeviceToHost));
    const size_t bytes = (size_t)total * UAVAC_TRAJ_COLS * 8;
    printf("%lld rows = %.2f GB\n", (long long)total, bytes / 1e9);
    hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    auto timed = [&](double *traj) {
        for (int i = 0; i < 6; ++i) UA(uavac_minsnap_plan_dev(ctx, dwp, B, m, 3.0, 0.01, times, seg_rows, row_offsets, coeffs, status, traj, total, nullptr, first_yaw));
        HIP(hipDeviceSynchronize());
        HIP(hipEventRecord(e0, 0));      // the ctx's stream is the default stream of this thread unless set otherwise
        const int n = 20;
        for (int i = 0; i < n; ++i) UA(uavac_minsnap_plan_dev(ctx, dwp, B, m, 3.0, 0.01, times, seg_rows, row_offsets, coeffs, status, traj, total, nullptr, first_yaw));
        HIP(hipDeviceSynchronize());
        HIP(hipEventRecord(e1, 0)); HIP(hipEventSynchronize(e1));
        float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
        return ms / n;
    };
    for (int rep = 0; rep < 2; ++rep) {
        double *plain; HIP(hipMalloc(&plain, bytes));
        printf("plain hipMalloc:                      planning chain %.3f ms\n", timed(plain));
        for (size_t chunk : {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30})
            for (int shuffle = 0; shuffle < 2; ++shuffle) {
                Mapped mp = map_chunks(bytes, chunk, shuffle);
                printf("chunks of %5zu MB mapped %s: planning chain %.3f ms\n", chunk >> 20, shuffle ? "shuffled" : "in order", timed((double *)mp.va));
                unmap(mp);
            }
        HIP(hipFree(plain));
    }
    return 0;
}
