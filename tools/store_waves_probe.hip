// Probe: is the rollout's log stream limited by having ONE storing wave per SIMD?  Store-only kernel with the rollout's
// log pattern ([K][13][B], 512-B wave stores, XCD-contiguous tiles, 1 024 workgroups) where W waves share the 13 rows of
// a 64-UAV tile (wave w stores rows w, w + W, ...), W = 1, 2, 4; and 16-byte lane stores of two ticks for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_waves_probe.hip -o tools/store_waves_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>
__global__ void __launch_bounds__(64 * W) k(double *log, int B, int K) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t G = gridDim.x, g = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8, sB = (size_t)B;
    for (int t = 0; t < K; ++t)
        for (int r = w; r < 13; r += W) log[((size_t)t * 13 + r) * sB + g * 64 + lane] = 1.0 + t;
}
template <int W> void run(double *log, int B, int K) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<W><<<B / 64, 64 * W>>>(log, B, K);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) k<W><<<B / 64, 64 * W>>>(log, B, K);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%d storing wave(s) per 64-UAV tile: %.3f ms per %d ticks => %.2f TB/s\n", W, ms, K, 104.0 * B * K / ms / 1e9);
}
int main() {
    const int B = 65536, K = 1000;
    double *log; if (hipMalloc(&log, (size_t)K * 13 * B * 8) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep) { run<1>(log, B, K); run<2>(log, B, K); run<4>(log, B, K); run<8>(log, B, K); }
    (void)hipMemset(log, 0, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemsetD32Async(log, 1, (size_t)K * 13 * B * 2, 0);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) (void)hipMemsetD32Async(log, 1, (size_t)K * 13 * B * 2, 0);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("runtime fill of the same buffer: %.3f ms => %.2f TB/s\n", ms, 104.0 * B * K / ms / 1e9);
    return 0;
}
