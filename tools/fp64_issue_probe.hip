// Micro-probe: cycles per v_fma_f64 for one wave per SIMD vs two, independent vs dependent chains (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_issue_probe.hip -o /tmp/fp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void k(double *out, long long *cyc, int iters, double a, double b) {
    double acc[ILP];
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) acc[i] = fma(acc[i], a, b);
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP> void run(int waves_per_simd) {
    int blocks = 256 * 4 * waves_per_simd, iters = 2000;
    double *out; long long *cyc;
    hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ILP><<<blocks, 64>>>(out, cyc, iters, 1.0000001, 1e-9);
    hipEventRecord(e0);
    k<ILP><<<blocks, 64>>>(out, cyc, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    double n = (double)iters * 16 * ILP;
    printf("ILP=%d waves/SIMD=%d: %.2f clock64-ticks per fma per wave, wall %.3f ms => %.2f ns per fma per wave\n", ILP,
           waves_per_simd, h[0] / n, ms, ms * 1e6 / n);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(1); run<1>(2); run<1>(4); run<4>(1); run<4>(2); run<4>(4); run<8>(1); run<8>(2);
    return 0;
}
