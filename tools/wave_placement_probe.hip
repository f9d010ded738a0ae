// Probe: where does the hardware put the two waves (compute, store) of the rollout's 128-thread workgroups?
// Each wave records HW_REG_HW_ID (wave slot, SIMD, CU, SE) and HW_REG_XCC_ID; the host counts, per CU and SIMD,
// how many "compute" (wave 0) and "store" (wave 1) waves landed there.  All workgroups stay resident (they spin
// on a flag until every wave has reported), like the real kernel's 1 024 workgroups at B = 65 536.
// Build: hipcc --offload-arch=gfx950 -O3 tools/wave_placement_probe.hip -o tools/wave_placement_probe.bin 2>/dev/null
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
__global__ void __launch_bounds__(128) k(unsigned *out, int *arrived, int total_waves) {
    extern __shared__ double slab[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * (blockIdx.x * 2 + wave)] = hw;
        out[2 * (blockIdx.x * 2 + wave) + 1] = xcc;
        __threadfence();
        atomicAdd(arrived, 1);
        while (atomicAdd(arrived, 0) < total_waves) __builtin_amdgcn_s_sleep(8);
    }
    slab[threadIdx.x] = 1.0;
    __syncthreads();
}
int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 1024, lds = 2 * 13 * 64 * 8;
    unsigned *out; int *arrived;
    (void)hipMalloc(&out, wgs * 2 * 2 * 4); (void)hipMalloc(&arrived, 4); (void)hipMemset(arrived, 0, 4);
    k<<<wgs, 128, lds>>>(out, arrived, wgs * 2);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * 4);
    (void)hipMemcpy(h.data(), out, wgs * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int[2]> per_simd;        // key: xcc | se | cu | simd
    for (int w = 0; w < wgs * 2; ++w) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = (xcc << 16) | (se << 12) | (sh << 11) | (cu << 4) | simd;
        per_simd[key][w & 1]++;
    }
    int hist[8][8]; memset(hist, 0, sizeof hist);
    for (auto &kv : per_simd) hist[kv.second[0] > 7 ? 7 : kv.second[0]][kv.second[1] > 7 ? 7 : kv.second[1]]++;
    printf("%d workgroups x (1 compute + 1 store wave): %zu SIMDs used\n", wgs, per_simd.size());
    for (int c = 0; c < 8; ++c)
        for (int s = 0; s < 8; ++s)
            if (hist[c][s]) printf("  SIMDs holding %d compute + %d store waves: %d\n", c, s, hist[c][s]);
    for (int w = 0; w < 8; ++w) {
        const unsigned hw = h[2 * w];
        printf("  wg %d wave %d: xcc %u se %u cu %u simd %u slot %u\n", w / 2, w & 1, h[2 * w + 1] & 0xf, (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3, hw & 0xf);
    }
    return 0;
}
