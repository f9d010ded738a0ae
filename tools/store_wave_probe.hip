// Probe: how fast can ONE dedicated store wave per CU stream the rollout's log pattern?
// Pattern: K ticks x 13 rows x B doubles, block j owns columns [256 j, 256 j + 256) of every row (2 KB).
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_wave_probe.hip -o tools/store_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int WIDE, int WAVES, int TILED = 0, int NT = 0>   // NT: nontemporal stores; TILED: tick-major tiles [K][B/256][13][256] instead of [K][13][B]; WIDE: 1 = 16 B per lane (2 stores per row), 0 = 8 B per lane (4 stores per row)
__global__ void __launch_bounds__(64 * WAVES) k(double *log, int B, int K, size_t pitch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t col0 = (size_t)blockIdx.x * 256;
    for (int t = 0; t < K; ++t) {
        for (int r = w; r < 13; r += WAVES) {
            double *row = TILED ? log + ((size_t)t * (B / 256) + blockIdx.x) * 13 * 256 + (size_t)r * 256
                                : log + ((size_t)t * 13 + r) * pitch + col0;
            if (WIDE) {
                d2 v = {1.0 + t, 2.0 + r};
                *(d2 *)(row + lane * 2) = v;
                *(d2 *)(row + 128 + lane * 2) = v;
            } else {
                for (int q = 0; q < 4; ++q) { if (NT) __builtin_nontemporal_store(1.0 + t, &row[q * 64 + lane]); else row[q * 64 + lane] = 1.0 + t; }
            }
        }
    }
}
template <int WIDE, int WAVES, int TILED = 0, int NT = 0> void run(double *log, int B, int K, size_t pitch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<WIDE, WAVES, TILED, NT><<<B / 256, 64 * WAVES>>>(log, B, K, pitch);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) k<WIDE, WAVES, TILED, NT><<<B / 256, 64 * WAVES>>>(log, B, K, pitch);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("nt=%d tiled=%d pitch=%zu wide=%d store-waves/CU=%d: %.3f ms for %d ticks => %.2f TB/s\n", NT, TILED, pitch, WIDE, WAVES, ms, K, 104.0 * B * K / ms / 1e9);
}
int main() {
    const int B = 65536, K = 1000;
    double *log; if (hipMalloc(&log, (size_t)K * 13 * (B + 4096) * 8) != hipSuccess) return 1;
    run<0, 4>(log, B, K, B); run<0, 8>(log, B, K, B); run<0, 13>(log, B, K, B); run<1, 13>(log, B, K, B); run<1, 13, 1>(log, B, K, B);
    return 0;
}
