// Probe (round-4): do the four SIMDs of a CU share their fp64 rate?  One workgroup of W = 1, 2, 4 wavefronts per CU
// (a CU deals the waves of a workgroup to SIMDs s, s+2, s+1, s+3: every wave has a SIMD of its own), every wave runs the same
// loop of fp64 FMAs -- CH independent dependency chains (1 = latency-bound like a control tick's critical path, 8 = issue-
// bound) -- and stamps s_memtime around it.  If cycles per FMA of a wave grow with W, waves on DIFFERENT SIMDs contend for
// something per CU.  The same with 32-bit integer multiply-adds as a control (a per-SIMD resource by all accounts).
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_cu_sharing_probe.hip -o tools/fp64_cu_sharing_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int ITER = 4096;
template <int CH, bool F64>
__global__ void probe(double *sink, long long *cycles, double seed) {
    double a[CH];
    int ia[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; ia[i] = (int)threadIdx.x + i; }
    const double m = 1.0000001, c = 1e-9;
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (F64) a[i] = __builtin_fma(a[i], m, c);
            else ia[i] = ia[i] * 3 + it;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += F64 ? a[i] : (double)ia[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int CH, bool F64>
void run(int W, int blocks, double *sink, long long *cyc) {
    std::vector<long long> h((size_t)blocks * W);
    for (int rep = 0; rep < 3; ++rep) probe<CH, F64><<<blocks, 64 * W>>>(sink, cyc, 1.0 + rep);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%s chains=%d waves_per_CU=%d blocks=%d : median %.2f cycles per op per wave (min %.2f max %.2f)\n", F64 ? "fp64_fma" : "int_mad ",
           CH, W, blocks, (double)h[h.size() / 2] / ITER / CH, (double)h[0] / ITER / CH, (double)h.back() / ITER / CH);
}
// STRAIGHT-LINE code: the same FMAs, but UNROLL iterations of CH chains written out (UNROLL * CH * 8 bytes of instructions per
// loop body: 16-64 KB, far more than a wave's instruction buffer holds) -- is instruction FETCH what waves on different SIMDs share?
template <int CH, int UNROLL>
__global__ void probe_long(double *sink, long long *cycles, double seed) {
    double a[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double m = 1.0000001, c = 1e-9;
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER / UNROLL; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = __builtin_fma(a[i], m, c);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += a[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int CH, int UNROLL>
void run_long(int W, int blocks, double *sink, long long *cyc) {
    std::vector<long long> h((size_t)blocks * W);
    for (int rep = 0; rep < 3; ++rep) probe_long<CH, UNROLL><<<blocks, 64 * W>>>(sink, cyc, 1.0 + rep);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("straight-line fp64_fma chains=%d body=%d KB waves_per_CU=%d blocks=%d : median %.3f cycles per op per wave (max %.3f)\n", CH,
           UNROLL * CH * 8 / 1024, W, blocks, (double)h[h.size() / 2] / ITER / CH, (double)h.back() / ITER / CH);
}

// the transcendental-class fp64 instructions of a control tick (v_rcp_f64, v_rsq_f64) and v_ldexp_f64 / v_mul_f64 / v_add_f64
template <int OP>
__global__ void probe_op(double *sink, long long *cycles, double seed) {
    double a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_rcp(a[i]) + 1.0;
            else if (OP == 1) a[i] = __builtin_amdgcn_rsq(a[i]) + 1.0;
            else if (OP == 2) a[i] = __builtin_amdgcn_ldexp(a[i], (it & 1) ? 1 : -1);
            else if (OP == 3) a[i] = a[i] * 1.0000001;
            else a[i] = a[i] + 1e-9;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run_op(const char *name, int W, double *sink, long long *cyc) {
    std::vector<long long> h((size_t)256 * W);
    for (int rep = 0; rep < 3; ++rep) probe_op<OP><<<256, 64 * W>>>(sink, cyc, 1.5 + rep);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-22s waves_per_CU=%d : median %.2f cycles per iteration-op per wave (max %.2f)\n", name, W, (double)h[h.size() / 2] / ITER / 4,
           (double)h.back() / ITER / 4);
}

int main() {
    double *sink; long long *cyc;
    (void)hipMalloc(&sink, 256 * 1024 * 8); (void)hipMalloc(&cyc, 256 * 16 * 8);
    for (int W : {1, 2, 4}) {
        run<1, true>(W, 256, sink, cyc); run<2, true>(W, 256, sink, cyc); run<8, true>(W, 256, sink, cyc);
        run<1, false>(W, 256, sink, cyc); run<8, false>(W, 256, sink, cyc);
    }
    for (int W : {1, 2, 4}) {
        run_op<0>("v_rcp_f64 + v_add_f64", W, sink, cyc); run_op<1>("v_rsq_f64 + v_add_f64", W, sink, cyc);
        run_op<2>("v_ldexp_f64", W, sink, cyc); run_op<3>("v_mul_f64", W, sink, cyc); run_op<4>("v_add_f64", W, sink, cyc);
    }
    for (int W : {1, 2, 4}) {
        run_long<1, 512>(W, 256, sink, cyc); run_long<2, 1024>(W, 256, sink, cyc); run_long<8, 512>(W, 256, sink, cyc);
        run_long<2, 1024>(W, 1, sink, cyc);
    }
    // the same with ONE CU busy (a single workgroup): is it the CU or the chip (clock / power) that is shared?
    for (int W : {1, 2, 4}) { run<1, true>(W, 1, sink, cyc); run<8, true>(W, 1, sink, cyc); }
    return 0;
}
