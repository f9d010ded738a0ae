// Probe: store-only ceilings of candidate layouts for the rollout's state log, with the rollout's own launch shape
// (B/64 workgroups, one storing wave each, K ticks, 13 doubles per UAV per tick).
//   0  [K][13][B]                 the shipped layout: 13 x 512-B wave stores per tick, rows 8 B pitch apart
//   1  [K][B/64][13][64]          tick-major tiles: the same 13 stores, 6.5 KB contiguous per workgroup and tick
//   2  [K/2][B/64][13][64][2]     two ticks per 16-B lane store: 13 x 1 KB per two ticks
//   3  [K/4][B/64][13][64][4]     four ticks per lane: 26 x 1 KB per four ticks
//   4  [B/64][K][13][64]          workgroup-major: every workgroup streams its own contiguous 6.5 KB x K region
//   5  [K][13][B], XCD-contiguous  layout 0, but the workgroups of one XCD (blockIdx % 8) own one contiguous eighth of
//                                  every row instead of every eighth 512-B piece
// Build: hipcc --offload-arch=gfx950 -O3 tools/log_layout_probe.hip -o tools/log_layout_probe.bin 2>/dev/null
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(64) k(double *log, int B, int K) {
    const int lane = threadIdx.x;
    const size_t g = blockIdx.x, G = gridDim.x, sB = (size_t)B;
    if (MODE == 0) {
        for (int t = 0; t < K; ++t)
            for (int r = 0; r < 13; ++r) log[((size_t)t * 13 + r) * sB + g * 64 + lane] = 1.0 + t;
    } else if (MODE == 1) {
        for (int t = 0; t < K; ++t)
            for (int r = 0; r < 13; ++r) log[(((size_t)t * G + g) * 13 + r) * 64 + lane] = 1.0 + t;
    } else if (MODE == 2) {
        for (int t = 0; t < K; t += 2)
            for (int r = 0; r < 13; ++r) {
                d2 v = {1.0 + t, 2.0 + t};
                *(d2 *)(log + ((((size_t)(t / 2) * G + g) * 13 + r) * 64 + lane) * 2) = v;
            }
    } else if (MODE == 3) {
        for (int t = 0; t < K; t += 4)
            for (int r = 0; r < 13; ++r) {
                d2 v = {1.0 + t, 2.0 + t};
                double *p = log + ((((size_t)(t / 4) * G + g) * 13 + r) * 64) * 4;
                *(d2 *)(p + lane * 2) = v;                 // 64 lanes x 16 B = first KB of the row's 2 KB
                *(d2 *)(p + 128 + lane * 2) = v;
            }
    } else if (MODE == 4) {
        for (int t = 0; t < K; ++t)
            for (int r = 0; r < 13; ++r) log[((g * K + t) * 13 + r) * 64 + lane] = 1.0 + t;
    } else {
        const size_t g2 = (g % 8) * (G / 8) + g / 8;
        for (int t = 0; t < K; ++t)
            for (int r = 0; r < 13; ++r) log[((size_t)t * 13 + r) * sB + g2 * 64 + lane] = 1.0 + t;
    }
}
template <int MODE> void run(double *log, int B, int K, const char *name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<B / 64, 64>>>(log, B, K);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) k<MODE><<<B / 64, 64>>>(log, B, K);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-34s %.3f ms per %d ticks => %.2f TB/s\n", name, ms, K, 104.0 * B * K / ms / 1e9);
}
int main() {
    const int B = 65536, K = 1000;
    double *log; if (hipMalloc(&log, (size_t)K * 13 * B * 8) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(log, B, K, "[K][13][B]");
        run<1>(log, B, K, "[K][B/64][13][64]");
        run<2>(log, B, K, "[K/2][B/64][13][64][2]");
        run<3>(log, B, K, "[K/4][B/64][13][64][4]");
        run<4>(log, B, K, "[B/64][K][13][64]");
        run<5>(log, B, K, "[K][13][B] XCD-contiguous");
    }
    return 0;
}
