// Probe (round 4): is a row buffer's kind a matter of its VIRTUAL address?  (All 8-chunk row buffers mapped at one virtual range
// were of one kind whatever physical chunks they were made of, and of another kind in the next process: tools/vmm_search_probe.hip.)
// The SAME eight physical 1 GiB chunks mapped at virtual ranges of different alignment and offset, the sampler's store pattern on each.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_va_probe.hip -o tools/vmm_va_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int R = 1306, PAIRS = 64 * 11 / 2;
__global__ void __launch_bounds__(64) heads(double *traj) {
    double *base = traj + ((blockIdx.x % 8) * (size_t)(gridDim.x / 8) + blockIdx.x / 8) * R * 11;
    const int npairs = R * 11 / 2;
    for (int c = 0; c * 64 < R; ++c)
        for (int p = threadIdx.x; p < PAIRS && c * PAIRS + p < npairs; p += 64) {
            d2 v = {1.0 + c, 2.0 + p};
            *(d2 *)(base + 2 * (size_t)(c * PAIRS + p)) = v;
        }
}
int main() {
    const size_t GB = (size_t)1 << 30, MB = (size_t)1 << 20;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h[2][8];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 8; ++i) HIP(hipMemCreate(&h[s][i], GB, &prop, 0));
    hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    auto timed = [&](double *t) {
        float best = 1e9f;
        for (int i = 0; i < 6; ++i) heads<<<65536, 64>>>(t);
        for (int r = 0; r < 3; ++r) {
            HIP(hipEventRecord(e0));
            for (int q = 0; q < 3; ++q) heads<<<65536, 64>>>(t);
            HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
            float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 3);
        }
        return best;
    };
    // one big reservation, 64 GiB aligned; the 8 GiB window is placed at different offsets inside it
    void *big; HIP(hipMemAddressReserve(&big, 80 * GB, 64 * GB, nullptr, 0));
    printf("reservation at %p\n", big);
    const size_t offs[] = {0, 1 * GB, 2 * GB, 4 * GB, 8 * GB, 16 * GB, 32 * GB, 2 * MB, 4 * MB, 64 * MB, 512 * MB, GB + 2 * MB, 3 * GB + 130 * MB, 5 * GB + 666 * MB, 7 * GB + 1022 * MB, 40 * GB + 2 * MB};
    for (int rep = 0; rep < 2; ++rep)
        for (int set = 0; set < 2; ++set) {
            printf("physical set %d:", set);
            for (size_t o : offs) {
                char *va = (char *)big + o;
                for (int i = 0; i < 8; ++i) HIP(hipMemMap(va + i * GB, GB, 0, h[set][i], 0));
                HIP(hipMemSetAccess(va, 8 * GB, &acc, 1));
                printf(" %.3f", timed((double *)va));
                HIP(hipMemUnmap(va, 8 * GB));
            }
            printf("\n");
        }
    // and separate small reservations with the runtime's default alignment
    printf("separate reservations (default alignment):");
    for (int k = 0; k < 8; ++k) {
        void *va; HIP(hipMemAddressReserve(&va, 8 * GB + k * 2 * MB, 0, nullptr, 0));
        for (int i = 0; i < 8; ++i) HIP(hipMemMap((char *)va + i * GB, GB, 0, h[0][i], 0));
        HIP(hipMemSetAccess(va, 8 * GB, &acc, 1));
        printf(" %p %.3f", va, timed((double *)va));
        HIP(hipMemUnmap(va, 8 * GB));
    }
    printf("\n");
    return 0;
}
