// Micro-probe: what one wave on a SIMD pays per fp64 instruction, by instruction and by where the operands come from
// (gfx950).  Four independent chains per wave, one wave per SIMD (the regime of small batches), and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_operand_probe.hip -o tools/fp64_operand_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(X) X(0) X(1) X(2) X(3)
template <int KIND>
__global__ void k(double *out, int iters, double sa, double sb) {
    double a0 = threadIdx.x + 1.0, a1 = threadIdx.x + 2.0, a2 = threadIdx.x + 3.0, a3 = threadIdx.x + 4.0;
    double va = sa, vb = sb;
    asm volatile("" : "+v"(va), "+v"(vb));          // the same constants in vector registers
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (KIND == 0) {          // v_fma_f64, three vector operands
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(va), "v"(vb));
            } else if (KIND == 1) {   // v_fma_f64, one scalar operand
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sa), "v"(vb));
            } else if (KIND == 2) {   // v_fma_f64, inline constant addend
                asm volatile("v_fma_f64 %0, %0, %4, 1.0\n v_fma_f64 %1, %1, %4, 1.0\n v_fma_f64 %2, %2, %4, 1.0\n v_fma_f64 %3, %3, %4, 1.0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(va));
            } else if (KIND == 3) {   // v_fmac_f64 (VOP2: dst is the addend)
                asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(va), "v"(vb));
            } else if (KIND == 4) {   // v_add_f64 vector + vector
                asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(vb));
            } else if (KIND == 5) {   // v_mul_f64 vector * vector
                asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(va));
            } else if (KIND == 6) {   // v_mul_f64 vector * scalar
                asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sa));
            } else if (KIND == 7) {   // v_mov_b64
                asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (KIND == 8) {   // v_add_u32 (a 32-bit VALU instruction for comparison)
                int i0 = (int)a0, i1 = (int)a1, i2 = (int)a2, i3 = (int)a3;
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(r));
                a0 = i0; a1 = i1; a2 = i2; a3 = i3;
            } else if (KIND == 9) {   // v_cmp_ge_f64 into vcc
                asm volatile("v_cmp_ge_f64 vcc, %0, %4\n v_cmp_ge_f64 vcc, %1, %4\n v_cmp_ge_f64 vcc, %2, %4\n v_cmp_ge_f64 vcc, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(vb) : "vcc");
            } else if (KIND == 10) {  // dependent v_fma_f64 chain (three vector operands)
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(va), "v"(vb));
            } else if (KIND == 11) {  // v_rsq_f64 (transcendental unit)
                asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int KIND> void run(const char *name) {
    for (int wps = 1; wps <= 2; ++wps) {
        int blocks = 256 * 4 * wps, iters = 4000;
        double *out; hipMalloc(&out, (size_t)blocks * 64 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, 64>>>(out, iters, 1.0000001, 1e-9);
        hipEventRecord(e0);
        k<KIND><<<blocks, 64>>>(out, iters, 1.0000001, 1e-9);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double n = (double)iters * 16 * 4;
        printf("%-44s %d wave(s) per SIMD: %.2f ns per instruction per wave\n", name, wps, ms * 1e6 / n);
        hipFree(out);
    }
}
int main() {
    run<0>("v_fma_f64 v, v, v, v");
    run<1>("v_fma_f64 v, v, s, v");
    run<2>("v_fma_f64 v, v, v, 1.0");
    run<3>("v_fmac_f64 v, v, v");
    run<4>("v_add_f64 v, v, v");
    run<5>("v_mul_f64 v, v, v");
    run<6>("v_mul_f64 v, v, s");
    run<7>("v_mov_b64 v, v");
    run<8>("v_add_u32 v, v, v (+ 2 cvt per instruction)");
    run<9>("v_cmp_ge_f64 vcc, v, v");
    run<10>("v_fma_f64 v, v, v, v (dependent chain)");
    run<11>("v_rsq_f64 v, v");
    return 0;
}
