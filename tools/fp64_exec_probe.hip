// Micro-probe: does a wave whose EXEC mask has only 16 or 32 lanes set pay less per fp64 instruction (gfx950)?
// One wave per SIMD, four independent v_fma_f64 chains; active lanes = the first N of the wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_exec_probe.hip -o tools/fp64_exec_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, int iters, double sa, double sb, int active) {
    double a0 = threadIdx.x + 1.0, a1 = threadIdx.x + 2.0, a2 = threadIdx.x + 3.0, a3 = threadIdx.x + 4.0;
    double va = sa, vb = sb;
    asm volatile("" : "+v"(va), "+v"(vb));
    if ((int)threadIdx.x < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(va), "v"(vb));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
    const int blocks = 1024, iters = 20000;
    double *out; (void)hipMalloc(&out, (size_t)blocks * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int active : {64, 48, 32, 16, 8, 1}) {
            k<<<blocks, 64>>>(out, iters, 1.0000001, 1e-9, active);
            (void)hipEventRecord(e0);
            k<<<blocks, 64>>>(out, iters, 1.0000001, 1e-9, active);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("active lanes %2d: %.2f ns per v_fma_f64 per wave\n", active, ms * 1e6 / ((double)iters * 64));
        }
    return 0;
}
