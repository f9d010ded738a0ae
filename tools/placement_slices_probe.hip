// Probe (round 4): at what granularity is a row buffer "fast" or "slow" (DESIGN K2)?  Twelve 7.53 GB buffers from hipMalloc side
// by side; the one-wave-per-mission store pattern (sparse heads 114 KB apart) and a plain fill over each whole buffer, the fill over each of
// its 512 MiB slices on its own (sixteen passes: the heads pattern on a slice is a single batch of waves and says nothing); then all buffers are freed and allocated again (does the pattern repeat?), then buffers of other sizes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_slices_probe.hip -o tools/placement_slices_probe.bin
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
// a compact front: grid-stride fill of the range, `passes` times over (long enough to time a 512 MiB slice on its own)
__global__ void __launch_bounds__(256) fill(double *traj, size_t pairs, int passes) {
    for (int k = 0; k < passes; ++k)
        for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) {
            d2 v = {1.0 + k, 2.0};
            *(d2 *)(traj + 2 * p) = v;
        }
}
static hipEvent_t e0, e1;
static float timed_fill(double *t, size_t bytes, int passes) {
    float best = 1e9f;
    fill<<<2048, 256>>>(t, bytes / 16, 1);
    for (int r = 0; r < 3; ++r) {
        HIP(hipEventRecord(e0));
        fill<<<2048, 256>>>(t, bytes / 16, passes);
        HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
        float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
        best = best < ms / passes ? best : ms / passes;
    }
    return best;
}
static float timed(double *t, size_t bytes) {
    const int missions = (int)(bytes / ((size_t)R * 88)) / 8 * 8;
    float best = 1e9f;
    heads<<<missions, 64>>>(t);
    for (int r = 0; r < 3; ++r) {
        HIP(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) heads<<<missions, 64>>>(t);
        HIP(hipEventRecord(e1)); HIP(hipDeviceSynchronize());
        float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
        best = best < ms / 4 ? best : ms / 4;
    }
    return best;
}
int main() {
    HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    const size_t bytes = (size_t)65536 * R * 88, S = (size_t)512 << 20;
    for (int round = 0; round < 2; ++round) {
        std::vector<double *> bufs(12);
        for (auto &p : bufs) HIP(hipMalloc(&p, bytes));
        printf("round %d: whole buffer ms (heads, fill), then FILL us per 512 MiB slice (16 passes each)\n", round);
        for (auto p : bufs) {
            printf("%p  %.3f %.3f |", (void *)p, timed(p, bytes), timed_fill(p, bytes, 2));
            for (size_t o = 0; o + S <= bytes; o += S) printf(" %3.0f", timed_fill((double *)((char *)p + o), S, 16) * 1e3f);
            printf("\n");
        }
        for (auto p : bufs) HIP(hipFree(p));
    }
    for (size_t gb : {1, 2, 4, 8, 16, 32}) {
        std::vector<double *> bufs(6);
        const size_t n = gb << 30;
        for (auto &p : bufs) HIP(hipMalloc(&p, n));
        printf("buffers of %2zu GiB, us per GiB:", gb);
        for (auto p : bufs) printf(" %.1f", timed(p, n) * 1e3 / gb);
        printf("\n");
        for (auto p : bufs) HIP(hipFree(p));
    }
    return 0;
}
