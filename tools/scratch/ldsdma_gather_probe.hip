// Probe: global_load_lds_dwordx4 with a per-lane source address, under a partial EXEC mask, with instruction offsets:
// lane l of instruction p must land at LDS base_p + 16 l; masked-off lanes must leave their 16 bytes alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int P>
__device__ __forceinline__ void dma_pairs(const double *mine, double *tile) {
    // the instruction offset is added to the GLOBAL address and to the LDS address alike: lane l lands at M0 + offset + 16 l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)mine,
                                     (__attribute__((address_space(3))) void *)((char *)tile + P * (1024 - 16)), 16, 16 * P, 0);
    if constexpr (P + 1 < 12) dma_pairs<P + 1>(mine, tile);
}
__global__ void k(const double *src, double *out, const int *segof) {
    __shared__ double tile[12 * 64 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 12 * 128; i += 64) tile[i] = -1.0;
    __syncthreads();
    const double *mine = src + 24 * segof[lane];
    if (lane % 3 != 0) dma_pairs<0>(mine, tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k2 = 0; k2 < 24; ++k2) out[lane * 24 + k2] = tile[(k2 >> 1) * 128 + lane * 2 + (k2 & 1)];
}
int main() {
    const int NS = 1000;
    std::vector<double> h(NS * 24);
    for (int i = 0; i < NS * 24; ++i) h[i] = i;
    std::vector<int> seg(64);
    for (int l = 0; l < 64; ++l) seg[l] = (l * 37 + 11) % NS;
    double *d, *o; int *s;
    (void)hipMalloc(&d, h.size() * 8); (void)hipMalloc(&o, 64 * 24 * 8); (void)hipMalloc(&s, 64 * 4);
    (void)hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(s, seg.data(), 64 * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, s);
    std::vector<double> r(64 * 24);
    (void)hipMemcpy(r.data(), o, r.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 24; ++j) {
            const double want = (l % 3 != 0) ? (double)(seg[l] * 24 + j) : -1.0;
            if (r[l * 24 + j] != want) { if (bad < 5) printf("lane %d coeff %d: got %g want %g\n", l, j, r[l * 24 + j], want); ++bad; }
        }
    printf("ldsdma gather probe: %d mismatches (status %d)\n", bad, (int)hipGetLastError());
    return bad != 0;
}
