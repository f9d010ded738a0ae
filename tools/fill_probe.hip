// Probe: what shape of plain store stream reaches the 6.5 TB/s that torch's fill_ reaches on this box?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
// MODE 0: grid-stride fill, block of 256 threads writes 4 KB per iteration (16 B per lane), contiguous per iteration
// MODE 1: one block per 16 KB contiguous region (each lane 4 x 16 B at 4 KB stride inside the region)
// MODE 2: chunk-major: one 64-thread block per 5632-B chunk, chunks in address order
// MODE 3: like MODE 2 but each block handles 21 consecutive chunks (= the sampler's mission-major order)
template <int MODE>
__global__ void k(double *p, size_t n_doubles) {
    if (MODE == 0) {
        size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2, stride = (size_t)gridDim.x * blockDim.x * 2;
        for (; i + 1 < n_doubles; i += stride) { d2 v = {1.0, 2.0}; *(d2 *)(p + i) = v; }
    } else if (MODE == 1) {
        double *b = p + (size_t)blockIdx.x * 2048;
        for (int q = 0; q < 4; ++q) { d2 v = {1.0, 2.0}; *(d2 *)(b + q * 512 + threadIdx.x * 2) = v; }
    } else {
        const int per = MODE == 2 ? 1 : 21;
        for (int c = 0; c < per; ++c) {
            double *b = p + ((size_t)blockIdx.x * per + c) * 704;
            for (int q = threadIdx.x; q < 352; q += 64) { d2 v = {1.0, 2.0}; *(d2 *)(b + q * 2) = v; }
        }
    }
}
template <int MODE> void run(double *p, size_t n, int grid, int block, const char *name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(p, n);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<grid, block>>>(p, n);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-58s %.3f ms => %.2f TB/s\n", name, ms, n * 8.0 / ms / 1e9);
}
int main() {
    const size_t chunks = 65536ull * 21, n = chunks * 704;             // 7.75 GB
    double *p; if (hipMalloc(&p, n * 8) != hipSuccess) return 1;
    run<0>(p, n, 256 * 8, 256, "grid-stride, 2048 blocks x 256 thr, 16 B/lane");
    run<0>(p, n, 256 * 32, 256, "grid-stride, 8192 blocks x 256 thr, 16 B/lane");
    run<1>(p, n, (int)(n / 2048), 256, "one 256-thr block per contiguous 16 KB");
    run<2>(p, n, (int)chunks, 64, "chunk-major: one wave per 5.6 KB chunk, address order");
    run<3>(p, n, 65536, 64, "mission-major: one wave per 21 consecutive chunks");
    return 0;
}
