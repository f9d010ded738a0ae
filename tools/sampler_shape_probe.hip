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
// round 4: G = C without the XCD blocking (chunk g by workgroup g: one dense front, neighbours on different XCDs);
// H = a plain fill (grid-stride, 16 bytes per lane); I = persistent waves (20 per CU), wave k takes chunks k, k + N, k + 2N, ...
__global__ void __launch_bounds__(64) shapeG(double *traj, int chunks_per_mission) {
    const size_t g = blockIdx.x;
    chunk(traj + (g / chunks_per_mission) * R * 11, (int)(g % chunks_per_mission), threadIdx.x);
}
__global__ void __launch_bounds__(256) shapeH(double *traj, size_t pairs) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) {
        d2 v = {1.0, 2.0};
        *(d2 *)(traj + 2 * p) = v;
    }
}
__global__ void __launch_bounds__(64) shapeI(double *traj, int chunks_per_mission, size_t chunks) {
    for (size_t g = blockIdx.x; g < chunks; g += gridDim.x)
        chunk(traj + (g / chunks_per_mission) * R * 11, (int)(g % chunks_per_mission), threadIdx.x);
}
// round 4: A with the missions of an XCD contiguous only inside windows of nw missions: mission = ((j / nw) * 8 + x) * nw + j % nw
// for the j-th workgroup of XCD x (nw = missions / 8: shape A; nw = 1: no XCD blocking at all)
__global__ void __launch_bounds__(64) shapeAw(double *traj, int nw) {
    const size_t x = blockIdx.x % 8, j = blockIdx.x / 8;
    double *base = traj + ((j / nw * 8 + x) * nw + j % nw) * R * 11;
    for (int c = 0; c * 64 < R; ++c) chunk(base, c, threadIdx.x);
}
// round 4: A with the j-th workgroup of XCD x taking mission (j + phase[x]) mod n of the XCD's eighth: the eight write windows
// keep their width and move at the same speed, but sit at other distances from each other
struct Phases { int p[8]; };
__global__ void __launch_bounds__(64) shapeAp(double *traj, Phases ph) {
    const size_t x = blockIdx.x % 8, n = gridDim.x / 8, j = (blockIdx.x / 8 + ph.p[x]) % n;
    double *base = traj + (x * n + j) * R * 11;
    for (int c = 0; c * 64 < R; ++c) chunk(base, c, threadIdx.x);
}
int main() {
    const int B = 65536, NB = 12, cpm = (R + 63) / 64;
    const size_t bytes = (size_t)B * R * 88;
    std::vector<double *> bufs(NB);
    // the last three buffers physically contiguous (hipDeviceMallocContiguous), when the runtime grants that
    for (size_t i = 0; i < bufs.size(); ++i) {
        if (false && hipExtMallocWithFlags((void **)&bufs[i], bytes, hipDeviceMallocContiguous) == hipSuccess) { printf("buffer %zu contiguous\n", i); continue; }
        if (hipMalloc(&bufs[i], bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    Phases ph = {};
    auto timed = [&](int shape, double *t) {
        auto go = [&] {
            if (shape == 0) shapeA<<<B, 64>>>(t);
            else if (shape == 1) shapeB<<<B, 256>>>(t);
            else if (shape == 2) shapeC<<<B * cpm, 64>>>(t, cpm);
            else if (shape == 3) shapeW<8><<<B, 512>>>(t);
            else if (shape == 4) shapeW<16><<<B, 1024>>>(t);
            else if (shape == 5) shapeF<<<B, 64>>>(t);
            else if (shape == 6) shapeG<<<B * cpm, 64>>>(t, cpm);
            else if (shape == 7) shapeH<<<256 * 8, 256>>>(t, bytes / 16);
            else if (shape == 8) shapeI<<<256 * 20, 64>>>(t, cpm, (size_t)B * cpm);
            else if (shape < 100) shapeAw<<<B, 64>>>(t, 1 << (shape - 9));
            else shapeAp<<<B, 64>>>(t, ph);
        };
        go(); go();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) go();
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 5;
    };
    for (int rep = 0; rep < 1; ++rep)
        for (int shape = 0; shape < 9 + 14; ++shape) {
            if (shape >= 9 && shape < 9 + 12) continue;
            if (shape == 1 || shape == 3 || shape == 5 || shape == 6 || shape == 8) continue;
            if (shape < 9) printf("shape %c:", 'A' + shape); else printf("A, windows of %5d:", 1 << (shape - 9));
            for (auto p : bufs) printf(" %.3f", timed(shape, p));
            printf("  ms per buffer\n");
        }
    // phase sets: all equal (= shape A), steps of 1/16, 1/11, 1/3 of an eighth, random ones
    srand(5);
    for (int set = 0; set < 12; ++set) {
        for (int x = 0; x < 8; ++x)
            ph.p[x] = set == 0 ? 0 : set == 1 ? x * 512 : set == 2 ? x * 745 : set == 3 ? x * 2731 : set == 4 ? (x % 2) * 4096 : rand() % 8192;
        printf("phases");
        for (int x = 0; x < 8; ++x) printf(" %4d", ph.p[x]);
        printf(":");
        for (auto p : bufs) printf(" %.3f", timed(100, p));
        printf("\n");
    }
    return 0;
}
