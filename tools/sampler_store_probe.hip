// Probe: the sampler's write pattern without its arithmetic -- one wave per mission streams R rows of 88 B
// (5.6 KB chunks, 16-byte stores) into a contiguous ~7.5 GB buffer.  What is the store-only ceiling, and does
// it depend on the alignment of a mission's block to 128-byte lines (R = 1306: blocks start at any multiple of
// 16 B; R = 1312: every block and every chunk starts on a line) or on the size of the piece a wave writes
// before moving on (chunk rows)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int CHUNK_ROWS, int XCD = 0>
__global__ void __launch_bounds__(64) k(double *traj, int rows_per_mission) {
    constexpr int PAIRS = CHUNK_ROWS * 11 / 2;
    const int lane = threadIdx.x;
    // XCD: the workgroups of one XCD (blockIdx % 8) take a contiguous eighth of the missions
    const size_t mission = XCD ? (size_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    double *base = traj + mission * rows_per_mission * 11;
    const int npairs_total = rows_per_mission * 11 / 2;
    for (int c0 = 0; c0 < npairs_total; c0 += PAIRS) {
        for (int p = lane; p < PAIRS && c0 + p < npairs_total; p += 64) {
            d2 v = {1.0 + c0, 2.0 + p};
            *(d2 *)(base + 2 * (size_t)(c0 + p)) = v;
        }
    }
}
template <int CHUNK_ROWS, int XCD = 0> void run(double *traj, int B, int R, const char *what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<CHUNK_ROWS, XCD><<<B, 64>>>(traj, R);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<CHUNK_ROWS, XCD><<<B, 64>>>(traj, R);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s R=%d: %.3f ms => %.2f TB/s\n", what, R, ms, (double)B * R * 88 / ms / 1e9);
}
int main() {
    const int B = 65536;
    double *traj; if (hipMalloc(&traj, (size_t)B * 1312 * 88) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        run<64>(traj, B, 1306, "64-row chunks, blocks 16-B aligned");
        run<64>(traj, B, 1312, "64-row chunks, blocks line aligned");
        run<256>(traj, B, 1306, "256-row pieces, blocks 16-B aligned");
        run<256>(traj, B, 1312, "256-row pieces, blocks line aligned");
        run<1312>(traj, B, 1312, "whole mission in one sweep, line aligned");
        run<64, 1>(traj, B, 1306, "64-row chunks, XCD-contiguous missions");
        run<64, 1>(traj, B, 1312, "64-row chunks, XCD-contiguous, line aligned");
    }
    return 0;
}
