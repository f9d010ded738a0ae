// Probe (round 4): physical memory comes in CLASSES -- two write streams into different 1 GiB chunks of the same class get in each
// other's way (201 us against 163 us for 2 x 0.53 GB, tools/vmm_pair_probe.hip).  Here: every chunk is classified (streamed
// together with one anchor per class found so far), then 8-chunk row buffers are assembled by class -- all eight from one class,
// four + four from two, round-robin over all classes -- and the sampler's store pattern and a fill run on them.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_class_probe.hip -o tools/vmm_class_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
__global__ void __launch_bounds__(256) fill(double *traj, size_t pairs) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) {
        d2 v = {1.0, 2.0};
        *(d2 *)(traj + 2 * p) = v;
    }
}
int main(int argc, char **argv) {
    const size_t GB = (size_t)1 << 30;
    const int want = argc > 1 ? atoi(argv[1]) : 96;
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
    hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    auto chunk = [&](int i) { return (double *)((char *)va + (size_t)i * GB); };
    auto pair_us = [&](int i, int j) {
        float best = 1e9f;
        for (int r = 0; r < 2; ++r) {
            HIP(hipEventRecord(e0));
            for (int q = 0; q < 3; ++q) heads2<<<2 * HALF, 64>>>(chunk(i), chunk(j));
            HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
            float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 3);
        }
        return best * 1e3f;
    };
    for (int w = 0; w < 60; ++w) heads2<<<2 * HALF, 64>>>(chunk(0), chunk(1));
    // classes: a chunk belongs to the class of the first anchor it conflicts with (> 12 % above the fastest pairing seen)
    std::vector<int> cls(n, -1), anchors;
    float fastest = 1e9f;
    for (int k = 1; k < std::min(n, 12); ++k) fastest = std::min(fastest, pair_us(0, k));
    for (int i = 0; i < n; ++i) {
        for (size_t a = 0; a < anchors.size() && cls[i] < 0; ++a)
            if (anchors[a] != i && pair_us(anchors[a], i) > 1.12f * fastest) cls[i] = (int)a;
        if (cls[i] < 0) { cls[i] = (int)anchors.size(); anchors.push_back(i); }
    }
    printf("%d chunks of 1 GiB, fastest pairing %.0f us, %zu classes; class of every chunk in creation order:\n", n, fastest, anchors.size());
    for (int i = 0; i < n; ++i) printf("%d%s", cls[i], i % 48 == 47 ? "\n" : "");
    printf("\n");
    std::vector<std::vector<int>> members(anchors.size());
    for (int i = 0; i < n; ++i) members[cls[i]].push_back(i);
    for (size_t c = 0; c < members.size(); ++c) printf("class %zu: %zu chunks\n", c, members[c].size());
    // row buffers of 8 chunks
    HIP(hipMemUnmap(va, (size_t)n * GB));
    auto run8 = [&](const char *name, const std::vector<int> &idx) {
        if (idx.size() < 8) { printf("%-44s (not enough chunks)\n", name); return; }
        for (int i = 0; i < 8; ++i) HIP(hipMemMap((char *)va + i * GB, GB, 0, h[idx[i]], 0));
        HIP(hipMemSetAccess(va, 8 * GB, &acc, 1));
        const size_t bytes = (size_t)65536 * R * 88;
        float t[2];
        for (int shape = 0; shape < 2; ++shape) {
            float best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                HIP(hipEventRecord(e0));
                for (int q = 0; q < 3; ++q) { if (shape) fill<<<2048, 256>>>((double *)va, bytes / 16); else heads<<<65536, 64>>>((double *)va); }
                HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
                float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / 3);
            }
            t[shape] = best;
        }
        printf("%-44s classes", name);
        for (int i = 0; i < 8; ++i) printf(" %d", cls[idx[i]]);
        printf(":  heads %.3f ms   fill %.3f ms\n", t[0], t[1]);
        HIP(hipMemUnmap(va, 8 * GB));
    };
    std::vector<size_t> order(members.size());
    for (size_t c = 0; c < order.size(); ++c) order[c] = c;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return members[a].size() > members[b].size(); });
    for (size_t oc = 0; oc < std::min<size_t>(3, order.size()); ++oc) {
        const auto &m = members[order[oc]];
        run8("eight chunks of ONE class", std::vector<int>(m.begin(), m.begin() + std::min<size_t>(8, m.size())));
    }
    if (order.size() >= 2) {
        std::vector<int> v;
        for (int i = 0; i < 4; ++i) { if ((int)members[order[0]].size() > i) v.push_back(members[order[0]][i]); if ((int)members[order[1]].size() > i) v.push_back(members[order[1]][i]); }
        run8("two classes, alternating", v);
        std::vector<int> w;
        for (int i = 0; i < 4 && i < (int)members[order[0]].size(); ++i) w.push_back(members[order[0]][i]);
        for (int i = 0; i < 4 && i < (int)members[order[1]].size(); ++i) w.push_back(members[order[1]][i]);
        run8("two classes, four then four", w);
    }
    {
        std::vector<int> v;
        for (int i = 0; (int)v.size() < 8 && i < n; ++i)
            for (size_t c = 0; c < order.size() && (int)v.size() < 8; ++c)
                if ((int)members[order[c]].size() > i) v.push_back(members[order[c]][i]);
        run8("round-robin over all classes", v);
    }
    { std::vector<int> v; for (int i = 0; i < 8; ++i) v.push_back(i); run8("creation order 0..7", v); }
    return 0;
}
