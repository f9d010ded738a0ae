// Probe (round 4): does a row buffer that is ONE physical allocation mapped through the VMM API (hipMemCreate of 7.53 GB +
// hipMemMap) come in the same kinds as a hipMalloc buffer?  (Buffers assembled from 1 GiB chunks were always of the slow
// kind: the chunking, or the VMM mapping itself?)  Fourteen draws each, one alive at a time; the one-wave-per-mission store pattern.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_whole_probe.hip -o tools/vmm_whole_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int R = 1306, PAIRS = 64 * 11 / 2;
__device__ __forceinline__ size_t xcd_block(size_t block, size_t n) { return (block % 8) * (n / 8) + block / 8; }
__global__ void __launch_bounds__(64) heads(double *traj) {
    double *base = traj + xcd_block(blockIdx.x, gridDim.x) * R * 11;
    const int npairs = R * 11 / 2;
    for (int c = 0; c * 64 < R; ++c)
        for (int p = threadIdx.x; p < PAIRS && c * PAIRS + p < npairs; p += 64) {
            d2 v = {1.0 + c, 2.0 + p};
            *(d2 *)(base + 2 * (size_t)(c * PAIRS + p)) = v;
        }
}
static hipEvent_t e0, e1;
static float timed(double *t) {
    float best = 1e9f;
    for (int i = 0; i < 12; ++i) heads<<<65536, 64>>>(t);          // (first touch, and the clock back up after the allocation)
    for (int r = 0; r < 3; ++r) {
        HIP(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) heads<<<65536, 64>>>(t);
        HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
        float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
        best = best < ms / 4 ? best : ms / 4;
    }
    return best;
}
int main() {
    HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    const size_t bytes = (size_t)65536 * R * 88;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t size = (bytes + gran - 1) / gran * gran;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int round = 0; round < 1; ++round) {
        printf("hipMalloc, one alive:");
        for (int i = 0; i < 14; ++i) {
            double *p; HIP(hipMalloc(&p, bytes));
            printf(" %.3f", timed(p));
            HIP(hipFree(p));
        }
        printf("\nhipMemCreate(whole) + hipMemMap, one alive (granularity %zu):", gran);
        for (int i = 0; i < 14; ++i) {
            hipMemGenericAllocationHandle_t h;
            HIP(hipMemCreate(&h, size, &prop, 0));
            void *va; HIP(hipMemAddressReserve(&va, size, 0, nullptr, 0));
            HIP(hipMemMap(va, size, 0, h, 0));
            HIP(hipMemSetAccess(va, size, &acc, 1));
            printf(" %.3f", timed((double *)va));
            HIP(hipMemUnmap(va, size)); HIP(hipMemRelease(h)); HIP(hipMemAddressFree(va, size));
        }
        printf("\n");
    }
    // all alive side by side: 30 whole-buffer VMM allocations, then (after releasing them) 30 hipMalloc ones
    {
        printf("hipMemCreate(whole) + hipMemMap, ALL ALIVE:");
        std::vector<hipMemGenericAllocationHandle_t> hs; std::vector<void *> vas;
        for (int i = 0; i < 30; ++i) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, size, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            void *va; HIP(hipMemAddressReserve(&va, size, 0, nullptr, 0));
            HIP(hipMemMap(va, size, 0, h, 0));
            HIP(hipMemSetAccess(va, size, &acc, 1));
            hs.push_back(h); vas.push_back(va);
            printf(" %.3f", timed((double *)va));
        }
        printf("\n");
        for (size_t i = 0; i < hs.size(); ++i) { HIP(hipMemUnmap(vas[i], size)); HIP(hipMemRelease(hs[i])); HIP(hipMemAddressFree(vas[i], size)); }
        printf("hipMalloc, ALL ALIVE:");
        std::vector<double *> ps;
        for (int i = 0; i < 30; ++i) {
            double *p;
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            ps.push_back(p);
            printf(" %.3f", timed(p));
        }
        printf("\n");
        for (auto p : ps) HIP(hipFree(p));
    }
    return 0;
}
