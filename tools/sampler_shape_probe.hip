// Probe: is the sampler's placement sensitivity (DESIGN K2) a property of its SHAPE -- one wave streaming one mission's
// 114 KB, thousands of sparse streams -- and would fatter, fewer streams be indifferent to where the buffer lies?
// Store-only, 65 536 missions x 1 306 rows x 88 B, twelve 7.5 GB buffers allocated side by side, three shapes on each:
//   A  one wave per mission, 64-row chunks one after the other           (the shipped shape)
//   B  four waves per mission, chunk c by wave c % 4, a barrier per round (22.5 KB per round and mission)
//   C  one wave per chunk, chunks in memory order                         (a compact front, like a fill)
//   D  eight waves per mission (45 KB per round), E  sixteen (90 KB per round: a mission in two rounds)   (round 3)
//   F  like A, but the missions of an XCD are visited with a stride of 37 (neighbouring waves far apart)     (round 3)
// Build: hipcc --offload-arch=gfx950 -O3 tools/sampler_shape_probe.hip -o tools/sampler_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int R = 1306, PAIRS = 64 * 11 / 2;       // rows per mission; 16-byte pairs per 64-row chunk
__device__ __forceinline__ void chunk(double *base, int c, int lane) {
    const int npairs_total = R * 11 / 2;
    const int p0 = c * PAIRS;
    for (int p = lane; p < PAIRS && p0 + p < npairs_total; p += 64) {
        d2 v = {1.0 + c, 2.0 + p};
        *(d2 *)(base + 2 * (size_t)(p0 + p)) = v;
    }
}
__device__ __forceinline__ size_t xcd_mission(size_t block, size_t n) { return (block % 8) * (n / 8) + block / 8; }
__global__ void __launch_bounds__(64) shapeA(double *traj) {
    double *base = traj + xcd_mission(blockIdx.x, gridDim.x) * R * 11;
    for (int c = 0; c * 64 < R; ++c) chunk(base, c, threadIdx.x);
}
__global__ void __launch_bounds__(256) shapeB(double *traj) {
    double *base = traj + xcd_mission(blockIdx.x, gridDim.x) * R * 11;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c0 = 0; c0 * 64 < R; c0 += 4) {
        if ((c0 + w) * 64 < R) chunk(base, c0 + w, lane);
        __builtin_amdgcn_s_barrier();
    }
}
template <int W>
__global__ void __launch_bounds__(64 * W) shapeW(double *traj) {
    double *base = traj + xcd_mission(blockIdx.x, gridDim.x) * R * 11;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c0 = 0; c0 * 64 < R; c0 += W) {
        if ((c0 + w) * 64 < R) chunk(base, c0 + w, lane);
        __builtin_amdgcn_s_barrier();
    }
}
__global__ void __launch_bounds__(64) shapeF(double *traj) {
    const size_t n = gridDim.x, per = n / 8, x = blockIdx.x % 8, i = blockIdx.x / 8;
    double *base = traj + (x * per + (i * 37) % per) * R * 11;          // 37 and 8192 are coprime: a permutation
    for (int c = 0; c * 64 < R; ++c) chunk(base, c, threadIdx.x);
}
__global__ void __launch_bounds__(64) shapeC(double *traj, int chunks_per_mission) {
    const size_t g = xcd_mission(blockIdx.x, gridDim.x);
    const size_t mission = g / chunks_per_mission;
    chunk(traj + mission * R * 11, (int)(g % chunks_per_mission), threadIdx.x);
}
int main() {
    const int B = 65536, NB = 12, cpm = (R + 63) / 64;
    const size_t bytes = (size_t)B * R * 88;
    std::vector<double *> bufs(NB);
    for (auto &p : bufs) if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timed = [&](int shape, double *t) {
        auto go = [&] {
            if (shape == 0) shapeA<<<B, 64>>>(t);
            else if (shape == 1) shapeB<<<B, 256>>>(t);
            else if (shape == 2) shapeC<<<B * cpm, 64>>>(t, cpm);
            else if (shape == 3) shapeW<8><<<B, 512>>>(t);
            else if (shape == 4) shapeW<16><<<B, 1024>>>(t);
            else shapeF<<<B, 64>>>(t);
        };
        go(); go();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) go();
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 5;
    };
    for (int rep = 0; rep < 2; ++rep)
        for (int shape = 0; shape < 6; ++shape) {
            printf("shape %c:", 'A' + shape);
            for (auto p : bufs) printf(" %.3f", timed(shape, p));
            printf("  ms per buffer\n");
        }
    return 0;
}
