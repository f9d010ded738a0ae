// Probe: the sampler's write pattern without its arithmetic -- one wave per mission streams ~1306 rows of 88 B
// (5.6 KB chunks, 16-byte stores) into a contiguous 7.5 GB buffer.  What is the store-only ceiling?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64) k(double *traj, int rows_per_mission) {
    const int lane = threadIdx.x;
    double *base = traj + (size_t)blockIdx.x * rows_per_mission * 11;
    const int npairs_total = rows_per_mission * 11 / 2;
    for (int c0 = 0; c0 < npairs_total; c0 += 352) {               // 352 pairs = one 64-row chunk
        for (int p = lane; p < 352 && c0 + p < npairs_total; p += 64) {
            d2 v = {1.0 + c0, 2.0 + p};
            *(d2 *)(base + 2 * (size_t)(c0 + p)) = v;
        }
    }
}
int main() {
    const int B = 65536, R = 1306;
    double *traj; if (hipMalloc(&traj, (size_t)B * R * 88) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<B, 64>>>(traj, R);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<<<B, 64>>>(traj, R);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("sampler-pattern store only: %.3f ms => %.2f TB/s\n", ms, (double)B * R * 88 / ms / 1e9);
    return 0;
}
