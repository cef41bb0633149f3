// VALU throughput microbenchmark for gfx950: instructions per cycle per SIMD for a few op classes at 1..8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2_t p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) {  // v_fma_f32 x8
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (OP == 1) {  // v_mul_f32 / v_add_f32 alternating x8
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (OP == 2) {  // v_pk_fma_f32 x4 (8 lanes-worth of fma)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (OP == 3) {  // v_pk_mul_f32 / v_pk_add_f32 x4
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (OP == 4) {  // v_max_f32 / v_min_f32 / v_cndmask x8
                asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %9\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %9\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_max3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            } else if (OP == 5) {  // v_rcp_f32 / v_sqrt_f32 / v_sin x8 (transcendental)
                asm volatile("v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_rcp_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_rcp_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (OP == 6) {  // integer: v_mul_lo_u32 / v_add_u32 / v_xor / v_lshrrev
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %9\n v_xor_b32 %2, %2, %8\n v_lshrrev_b32 %3, 3, %3\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %9\n v_xor_b32 %6, %6, %8\n v_lshrrev_b32 %7, 3, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (OP == 7) {  // v_cmp + v_cndmask pairs
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP>
void run(const char* name, int instr_per_unroll, float flops_per_instr) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * sizeof(float));
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 8}) {       // waves per SIMD
        int blocks = cus * wps;             // 256 threads = 4 waves = one per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, 100, 1.0001f, 0.5f);
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instrs_per_simd = (double)iters * 8 * instr_per_unroll * wps;   // wave-instructions issued on one SIMD
        double rate = instrs_per_simd / (ms * 1e-3);                            // wave-instr / s / SIMD
        double tflops = rate * cus * 4 * 64 * flops_per_instr / 1e12;
        printf("%-28s waves/SIMD=%d  %.3f ms  %.3f Ginstr/s/SIMD (=%.2f cyc/instr @2.4GHz)  %.1f TFLOP/s\n", name, wps, ms, rate / 1e9, 2.4e9 / rate, tflops);
    }
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 8, 2);
    run<1>("v_mul_f32+v_add_f32", 8, 1);
    run<2>("v_pk_fma_f32", 4, 4);
    run<3>("v_pk_mul_f32+v_pk_add_f32", 4, 2);
    run<4>("max/min/cndmask/max3", 8, 1);
    run<5>("v_rcp/v_sqrt", 8, 1);
    run<6>("int mul_lo/add/xor/shift", 8, 1);
    run<7>("v_cmp+v_cndmask", 8, 1);
    return 0;
}
