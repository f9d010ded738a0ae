// Probe (round 4): which pieces of physical memory get in each other's way?  240 physical chunks of 1 GiB (hipMemCreate, each
// mapped on its own).  The one-wave-per-mission store pattern runs on chunk 0 and on chunk k AT THE SAME TIME (even workgroups
// write one, odd ones the other; 2 x 4 608 missions), for every k: the time against k shows what period the device's address
// map has for concurrent write streams, if creation order is address order.  Then the same for a few other anchors.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_pair_probe.hip -o tools/vmm_pair_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int R = 1306, PAIRS = 64 * 11 / 2, HALF = 4608;
// workgroup b: half (b & 1), mission (b >> 1) of that half; the missions of an XCD contiguous inside each half
__global__ void __launch_bounds__(64) heads2(double *a, double *b, int shift, size_t byte_shift) {
    const size_t blk = blockIdx.x >> 1, n = gridDim.x >> 1;
    size_t mission = (blk % 8) * (n / 8) + blk / 8;
    if (blockIdx.x & 1) mission = (mission + shift) % n;                     // the second stream `shift` missions ahead of the first
    double *base = ((blockIdx.x & 1) ? b + byte_shift / 8 : a) + mission * R * 11;
    const int npairs = R * 11 / 2;
    for (int c = 0; c * 64 < R; ++c)
        for (int p = threadIdx.x; p < PAIRS && c * PAIRS + p < npairs; p += 64) {
            d2 v = {1.0 + c, 2.0 + p};
            *(d2 *)(base + 2 * (size_t)(c * PAIRS + p)) = v;
        }
}
int main(int argc, char **argv) {
    const size_t GB = (size_t)1 << 30;
    const int want = argc > 1 ? atoi(argv[1]) : 240;
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
    void *va; HIP(hipMemAddressReserve(&va, (size_t)n * GB, 0, nullptr, 0));
    for (int i = 0; i < n; ++i) HIP(hipMemMap((char *)va + i * GB, GB, 0, h[i], 0));
    HIP(hipMemSetAccess(va, (size_t)n * GB, &acc, 1));
    printf("%d chunks of 1 GiB\n", n);
    hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    auto chunk = [&](int i) { return (double *)((char *)va + (size_t)i * GB); };
    int shift = 0; size_t byte_shift = 0;
    auto timed = [&](int i, int j) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            HIP(hipEventRecord(e0));
            for (int q = 0; q < 4; ++q) heads2<<<2 * HALF, 64>>>(chunk(i), chunk(j), shift, byte_shift);
            HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
            float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
            best = best < ms / 4 ? best : ms / 4;
        }
        return best * 1e3f;
    };
    for (int w = 0; w < 40; ++w) heads2<<<2 * HALF, 64>>>(chunk(0), chunk(1), 0, 0);      // clocks up
    for (int anchor : {0, 1}) {
        if (anchor >= n) continue;
        for (int variant = 0; variant < 5; ++variant) {
            shift = variant == 1 ? 1 : variant == 2 ? 2311 : 0;
            byte_shift = variant == 3 ? 4096 : variant == 4 ? 65536 + 256 : 0;
            printf("us for chunk %d together with chunk k (second stream %d missions ahead, %zu bytes up), k = 0 ...:\n", anchor, shift, byte_shift);
            for (int k = 0; k < n; ++k) printf("%4.0f%s", timed(anchor, k), k % 24 == 23 ? "\n" : "");
            printf("\n");
        }
    }
    return 0;
}
