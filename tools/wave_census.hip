// In-situ census of where the hardware puts the waves of rollout-shaped workgroups, callable on any HIP stream between
// real kernels (tools/placement_after_sampler.py).  Launch shape of control_rollout_kernel at B = 65 536: 1 024
// workgroups x 128 threads, 33 792 B of dynamic LDS => four workgroups per CU, all resident; other shapes (256 / 512
// threads per workgroup with 2x / 4x the LDS) for comparison.  Each wave records HW_REG_HW_ID and HW_REG_XCC_ID, then
// waits (bounded) until every wave has reported.  `noop` is an "aligner" candidate: an empty kernel of any shape.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/wave_census.hip -o tools/libwave_census.so
#include <hip/hip_runtime.h>
__global__ void census_kernel(unsigned *out, int *arrived, int total_waves) {
    extern __shared__ double slab[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpw = blockDim.x >> 6;
    if (lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * (blockIdx.x * wpw + wave)] = hw;
        out[2 * (blockIdx.x * wpw + wave) + 1] = xcc;
        __threadfence();
        atomicAdd(arrived, 1);
        for (int spin = 0; spin < 200000 && atomicAdd(arrived, 0) < total_waves; ++spin) __builtin_amdgcn_s_sleep(8);
    }
    slab[threadIdx.x] = 1.0;
    __syncthreads();
}
__global__ void noop_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 0; }
extern "C" int wave_census(void *stream, unsigned *out, int *arrived, int wgs, int threads, int lds_bytes) {
    (void)hipMemsetAsync(arrived, 0, 4, (hipStream_t)stream);
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(census_kernel, dim3(wgs), dim3(threads), lds_bytes, (hipStream_t)stream, out, arrived,
                       wgs * (threads / 64));
    return (int)hipGetLastError();
}
extern "C" int noop(void *stream, int wgs, int threads, int lds_bytes) {
    hipLaunchKernelGGL(noop_kernel, dim3(wgs), dim3(threads), lds_bytes, (hipStream_t)stream, (int *)nullptr);
    return (int)hipGetLastError();
}
