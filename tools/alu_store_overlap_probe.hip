// Probe: do a store-streaming wave and an fp64-ALU wave that share a SIMD slow each other down?
// 1 024 workgroups x 2 waves (the rollout's shape: one of each wave on every SIMD).  Wave 0 runs a dependent chain
// of NF fp64 FMAs per "tick" (4 independent chains), wave 1 writes 13 x 512 B per tick into a [K][13][B] log.
// Modes: ALU only, stores only, both (no synchronisation between the two waves), both with a barrier per tick.
// Build: hipcc --offload-arch=gfx950 -O3 tools/alu_store_overlap_probe.hip -o tools/alu_store_overlap_probe.bin 2>/dev/null
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int TPB = 1>      // TPB: ticks per hand-over (mode 15/31 only); 1 ALU, 2 store, 3 both, 7 both + barrier per tick, 15 = 7 + the values go through LDS slabs; +16: XCD-contiguous columns
__global__ void __launch_bounds__(128) k(double *log, double *sink, int B, int K, int NF) {
    __shared__ double slab[2 * 13 * 64 * TPB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (MODE & 16) ? (size_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (size_t)blockIdx.x, sB = (size_t)B;
    if (wave == 0) {
        double a = 1.0 + lane, b = 2.0, c = 3.0, d = 4.0;
        for (int t = 0; t < K; ++t) {
            if (MODE & 1)
                for (int i = 0; i < NF; i += 4) {
                    a = __builtin_fma(a, 1.0000001, 0.5); b = __builtin_fma(b, 0.9999999, 0.25);
                    c = __builtin_fma(c, 1.0000002, 0.125); d = __builtin_fma(d, 0.9999998, 0.0625);
                }
            if (MODE & 8) {
                double *my = slab + (((t / TPB) & 1) * TPB + (t % TPB)) * 13 * 64 + lane;
                for (int r = 0; r < 13; ++r) my[r * 64] = a + r;
            }
            if ((MODE & 4) && (t % TPB == TPB - 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (a + b + c + d == 12345.678) sink[g * 64 + lane] = a;
    } else {
        for (int t = 0; t < K; ++t) {
            if (MODE & 8) {
                if (t % TPB == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const double *src = slab + (((t / TPB) & 1) * TPB + (t % TPB)) * 13 * 64 + lane;
                double v[13];
                for (int r = 0; r < 13; ++r) v[r] = src[r * 64];
                for (int r = 0; r < 13; ++r) log[((size_t)t * 13 + r) * sB + g * 64 + lane] = v[r];
                continue;
            }
            if (MODE & 2)
                for (int r = 0; r < 13; ++r) log[((size_t)t * 13 + r) * sB + g * 64 + lane] = 1.0 + t;
            if (MODE & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
}
template <int MODE, int TPB = 1> float run(double *log, double *sink, int B, int K, int NF) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, TPB><<<B / 64, 128>>>(log, sink, B, K, NF);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) k<MODE, TPB><<<B / 64, 128>>>(log, sink, B, K, NF);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
    const int B = 65536, K = 1000;
    double *log, *sink;
    if (hipMalloc(&log, (size_t)K * 13 * B * 8) != hipSuccess || hipMalloc(&sink, B * 8) != hipSuccess) return 1;
    for (int NF = 120; NF <= 240; NF += 40) {
        const float a = run<1>(log, sink, B, K, NF), s = run<2>(log, sink, B, K, NF), b = run<3>(log, sink, B, K, NF),
                    bb = run<7>(log, sink, B, K, NF), bl = run<15>(log, sink, B, K, NF), sx = run<18>(log, sink, B, K, NF),
                    blx = run<31>(log, sink, B, K, NF), bl2 = run<15, 2>(log, sink, B, K, NF), bl4 = run<15, 4>(log, sink, B, K, NF);
        printf("NF=%3d FMAs/tick: ALU only %.3f ms, stores only %.3f ms, both %.3f ms, both + barrier per tick %.3f ms, "
               "+ LDS hand-over %.3f ms (every 2 ticks %.3f, every 4 ticks %.3f) | XCD-contiguous columns: stores only %.3f ms, full hand-over %.3f ms\n", NF, a, s, b, bb, bl, bl2, bl4, sx, blx);
    }
    return 0;
}
