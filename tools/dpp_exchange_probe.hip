// Would splitting one UAV's tick over the 4 lanes of a quad (rotor-parallel allocation / motors, axis-parallel rates,
// component-parallel quaternion update: SURVEY.md H5) shorten the tick at small batch sizes?  The answer hangs on what a
// single wave pays for (a) an fp64 FMA, dependent or not, and (b) handing one double to another lane of the quad
// (two v_mov_b32 with a DPP quad_perm).  One wave per SIMD, like the rollout's compute wave.  (gfx950)
// Build: hipcc --offload-arch=gfx950 -O3 tools/dpp_exchange_probe.hip -o tools/dpp_exchange_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double quad_rot(double v) {           // lane i of every quad takes the value of lane (i+1)&3
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x39, 0xf, 0xf, true);      // quad_perm:[1,2,3,0]
    hi = __builtin_amdgcn_mov_dpp(hi, 0x39, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// MODE 0: 16 dependent FMAs          MODE 1: 16 FMAs in 4 independent chains
// MODE 2: 16 x (exchange + dependent FMA)   -- a value that crosses lanes before every use
// MODE 3: 16 x (exchange) only, dependent chain of exchanges
template <int MODE>
__global__ void k(double *out, long long *cyc, int iters, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) x0 = fma(x0, a, b);
            if (MODE == 1) { if ((r & 3) == 0) x0 = fma(x0, a, b); if ((r & 3) == 1) x1 = fma(x1, a, b);
                             if ((r & 3) == 2) x2 = fma(x2, a, b); if ((r & 3) == 3) x3 = fma(x3, a, b); }
            if (MODE == 2) x0 = fma(quad_rot(x0), a, b);
            if (MODE == 3) x0 = quad_rot(x0);
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *what) {
    const int blocks = 256 * 4, iters = 4000;
    double *out; long long *cyc;
    (void)hipMalloc(&out, blocks * 64 * 8); (void)hipMalloc(&cyc, blocks * 8);
    k<MODE><<<blocks, 64>>>(out, cyc, iters, 1.0000001, 1e-9);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, cyc, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %6.2f ns per step of the chain (wall %.3f ms / %d steps)\n", what, ms * 1e6 / (iters * 16.0), ms, iters * 16);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0>("16 dependent v_fma_f64");
    run<1>("16 v_fma_f64 in 4 independent chains");
    run<2>("quad exchange of the operand + dependent v_fma_f64");
    run<3>("quad exchange alone (2 x v_mov_b32 dpp), dependent");
    return 0;
}
