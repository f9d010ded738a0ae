// Probe (round 4): what do the fast 8-chunk row buffers have in common?  N physical chunks of 1 GiB; the full pairwise matrix
// (us for two chunks written at the same time, tools/vmm_pair_probe.hip) and then several hundred random ordered 8-subsets
// mapped as a 7.53 GB row buffer with the sampler's store pattern timed on each.  Output for offline analysis.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_search_probe.hip -o tools/vmm_search_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int R = 1306, PAIRS = 64 * 11 / 2, HALF = 4608;
__device__ __forceinline__ void mission_rows(double *base) {
    const int npairs = R * 11 / 2;
    for (int c = 0; c * 64 < R; ++c)
        for (int p = threadIdx.x; p < PAIRS && c * PAIRS + p < npairs; p += 64) {
            d2 v = {1.0 + c, 2.0 + p};
            *(d2 *)(base + 2 * (size_t)(c * PAIRS + p)) = v;
        }
}
__global__ void __launch_bounds__(64) heads2(double *a, double *b) {
    const size_t blk = blockIdx.x >> 1, n = gridDim.x >> 1;
    mission_rows(((blockIdx.x & 1) ? b : a) + ((blk % 8) * (n / 8) + blk / 8) * R * 11);
}
__global__ void __launch_bounds__(64) heads(double *traj) {
    mission_rows(traj + ((blockIdx.x % 8) * (size_t)(gridDim.x / 8) + blockIdx.x / 8) * R * 11);
}
int main(int argc, char **argv) {
    const size_t GB = (size_t)1 << 30;
    const int want = argc > 1 ? atoi(argv[1]) : 64, trials = argc > 2 ? atoi(argv[2]) : 300;
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
    void *va, *vb;
    HIP(hipMemAddressReserve(&va, (size_t)n * GB, 0, nullptr, 0));
    HIP(hipMemAddressReserve(&vb, 8 * GB, 0, nullptr, 0));
    for (int i = 0; i < n; ++i) HIP(hipMemMap((char *)va + i * GB, GB, 0, h[i], 0));
    HIP(hipMemSetAccess(va, (size_t)n * GB, &acc, 1));
    hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    auto chunk = [&](int i) { return (double *)((char *)va + (size_t)i * GB); };
    for (int w = 0; w < 60; ++w) heads2<<<2 * HALF, 64>>>(chunk(0), chunk(1));
    printf("PAIRS %d\n", n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            float best = 1e9f;
            if (j > i) {
                for (int r = 0; r < 2; ++r) {
                    HIP(hipEventRecord(e0));
                    for (int q = 0; q < 2; ++q) heads2<<<2 * HALF, 64>>>(chunk(i), chunk(j));
                    HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
                    float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms / 2);
                }
            } else best = 0;
            printf("%.0f ", best * 1e3f);
        }
        printf("\n");
    }
    // (a physical chunk can be mapped at two addresses at once: the composites go to a second range)
    std::mt19937 g(17);
    printf("COMPOSITES %d\n", trials);
    for (int t = 0; t < trials; ++t) {
        std::vector<int> idx(n);
        for (int i = 0; i < n; ++i) idx[i] = i;
        std::shuffle(idx.begin(), idx.end(), g);
        if (t == 0) for (int i = 0; i < 8; ++i) idx[i] = i;                       // creation order first
        for (int i = 0; i < 8; ++i) HIP(hipMemMap((char *)vb + i * GB, GB, 0, h[idx[i]], 0));
        HIP(hipMemSetAccess(vb, 8 * GB, &acc, 1));
        float best = 1e9f;
        heads<<<65536, 64>>>((double *)vb);
        for (int r = 0; r < 2; ++r) {
            HIP(hipEventRecord(e0));
            for (int q = 0; q < 2; ++q) heads<<<65536, 64>>>((double *)vb);
            HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
            float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 2);
        }
        for (int i = 0; i < 8; ++i) printf("%d ", idx[i]);
        printf("%.3f\n", best);
        HIP(hipMemUnmap(vb, 8 * GB));
    }
    return 0;
}
