// Probe (round 4): is a row buffer's "fast" or "slow" kind (DESIGN K2) a property of the PHYSICAL memory behind it, chunk by
// chunk?  Physical chunks of 1 GiB are created one by one (hipMemCreate), each mapped on its own and written with two
// store-only patterns -- a plain fill, and the one-wave-per-mission shape of the sampler (sparse write heads 114 KB apart) --
// then 8-chunk row buffers are assembled from the fastest and from the slowest chunks and from chunks in creation order, and
// the same patterns run on those.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_speed_map.hip -o tools/vmm_speed_map.bin       Run: ./tools/vmm_speed_map.bin [chunks]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int R = 1306, PAIRS = 64 * 11 / 2;
__device__ __forceinline__ size_t xcd_block(size_t block, size_t n) { return (block % 8) * (n / 8) + block / 8; }
// one wave per mission of R rows, 64-row chunks one after the other; missions of an XCD contiguous
__global__ void __launch_bounds__(64) heads(double *traj) {
    double *base = traj + xcd_block(blockIdx.x, gridDim.x) * R * 11;
    const int npairs = R * 11 / 2;
    for (int c = 0; c * 64 < R; ++c)
        for (int p = threadIdx.x; p < PAIRS && c * PAIRS + p < npairs; p += 64) {
            d2 v = {1.0 + c, 2.0 + p};
            *(d2 *)(base + 2 * (size_t)(c * PAIRS + p)) = v;
        }
}
__global__ void __launch_bounds__(256) fill(double *traj, size_t pairs) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) {
        d2 v = {1.0, 2.0};
        *(d2 *)(traj + 2 * p) = v;
    }
}
int main(int argc, char **argv) {
    const size_t GB = (size_t)1 << 30;
    int want = argc > 1 ? atoi(argv[1]) : 200;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h;
    for (int i = 0; i < want; ++i) {
        hipMemGenericAllocationHandle_t x;
        if (hipMemCreate(&x, GB, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        h.push_back(x);
    }
    const int n = (int)h.size();
    printf("%d chunks of 1 GiB created\n", n);
    void *va; HIP(hipMemAddressReserve(&va, (size_t)n * GB, 0, nullptr, 0));
    for (int i = 0; i < n; ++i) HIP(hipMemMap((char *)va + i * GB, GB, 0, h[i], 0));
    HIP(hipMemSetAccess(va, (size_t)n * GB, &acc, 1));
    hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    auto timed = [&](int shape, double *t, size_t bytes) {
        const int missions = (int)(bytes / ((size_t)R * 88)) / 8 * 8;
        auto go = [&] { if (shape == 0) fill<<<2048, 256>>>(t, bytes / 16); else heads<<<missions, 64>>>(t); };
        float best = 1e9f;
        go();
        for (int r = 0; r < 3; ++r) {
            HIP(hipEventRecord(e0));
            for (int i = 0; i < 4; ++i) go();
            HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
            float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 4);
        }
        return best;
    };
    std::vector<float> tf(n), th(n);
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < n; ++i) { tf[i] = timed(0, (double *)((char *)va + i * GB), GB); th[i] = timed(1, (double *)((char *)va + i * GB), GB); }
        printf("pass %d, us per chunk (fill / heads), in creation order:\n", rep);
        for (int i = 0; i < n; ++i) printf("%3d %6.1f %6.1f%s", i, tf[i] * 1e3, th[i] * 1e3, i % 4 == 3 ? "\n" : "   |  ");
        printf("\n");
    }
    // row buffers of 8 chunks: creation order (three of them), the 8 fastest, the 8 slowest (by the heads pattern)
    HIP(hipMemUnmap(va, (size_t)n * GB));
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return th[a] < th[b]; });
    auto run8 = [&](const char *name, const int *idx) {
        for (int i = 0; i < 8; ++i) HIP(hipMemMap((char *)va + i * GB, GB, 0, h[idx[i]], 0));
        HIP(hipMemSetAccess(va, 8 * GB, &acc, 1));
        const size_t bytes = (size_t)65536 * R * 88;
        printf("%-28s chunks", name);
        for (int i = 0; i < 8; ++i) printf(" %3d", idx[i]);
        printf(":  fill %.3f ms   heads %.3f ms   (7.53 GB)\n", timed(0, (double *)va, bytes), timed(1, (double *)va, bytes));
        HIP(hipMemUnmap(va, 8 * GB));
    };
    if (n >= 24) {
        int idx[8];
        for (int k = 0; k < 3; ++k) { for (int i = 0; i < 8; ++i) idx[i] = k * 8 + i; run8("creation order", idx); }
        run8("8 fastest", order.data());
        run8("next 8 fastest", order.data() + 8);
        run8("8 slowest", order.data() + n - 8);
        for (int i = 0; i < 8; ++i) idx[i] = i % 2 ? order[i / 2] : order[n - 1 - i / 2];
        run8("4 slowest + 4 fastest", idx);
    }
    return 0;
}
