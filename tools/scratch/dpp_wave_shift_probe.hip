// Probe: what v_mov_b32_dpp wave_shr:1 / wave_shl:1 do on gfx950 (lane l reads lane l-1 / l+1?), incl. the end lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
    int v = threadIdx.x + 100;
    int a = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
    int b = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
    o[threadIdx.x] = a;
    o[64 + threadIdx.x] = b;
    int c = v, d = v;                  // the same moves IN PLACE (destination register == source register)
    asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c));
    asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(d));
    o[128 + threadIdx.x] = c;
    o[192 + threadIdx.x] = d;
}
int main() {
    int *d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("hip status %d\n", (int)hipGetLastError());
    printf("wave_shr:1 :"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n");
    printf("wave_shl:1 :"); for (int i = 0; i < 64; ++i) printf(" %d", h[64 + i]); printf("\n");
    printf("in place shr:"); for (int i = 0; i < 64; ++i) printf(" %d", h[128 + i]); printf("\n");
    printf("in place shl:"); for (int i = 0; i < 64; ++i) printf(" %d", h[192 + i]); printf("\n");
    return 0;
}
