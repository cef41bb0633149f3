// Issue rate of single VALU opcodes on gfx950 at 8 waves per SIMD (8 independent dependency chains per wave).
//   hipcc -O3 --offload-arch=gfx950 ops.hip -o ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define EIGHT(INS, TAIL) asm volatile(INS " %0, %0" TAIL "\n" INS " %1, %1" TAIL "\n" INS " %2, %2" TAIL "\n" INS " %3, %3" TAIL "\n" \
                                      INS " %4, %4" TAIL "\n" INS " %5, %5" TAIL "\n" INS " %6, %6" TAIL "\n" INS " %7, %7" TAIL "\n" \
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(sa) : "vcc")
#define PAIR4(INS, TAIL) asm volatile(INS " %0, %0" TAIL "\n" INS " %1, %1" TAIL "\n" INS " %2, %2" TAIL "\n" INS " %3, %3" TAIL "\n" \
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb))
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void ops(float* out, int iters, float a, float b, unsigned long long sa) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2_t p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(x0), "v"(32.0f) : "vcc");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) EIGHT("v_fma_f32", ", %8, %9");
            if (OP == 1) EIGHT("v_mul_f32", ", %8");
            if (OP == 2) EIGHT("v_add_f32", ", %8");
            if (OP == 3) EIGHT("v_sub_f32", ", %8");
            if (OP == 4) EIGHT("v_max_f32", ", %8");
            if (OP == 5) EIGHT("v_med3_f32", ", %8, %9");
            if (OP == 6) EIGHT("v_cndmask_b32", ", %8, vcc");
            if (OP == 7) EIGHT("v_cndmask_b32", ", %8, %10");
            if (OP == 8) EIGHT("v_and_b32", ", %8");
            if (OP == 9) EIGHT("v_xor_b32", ", %8");
            if (OP == 10) EIGHT("v_lshrrev_b32", ", 7");
            if (OP == 11) EIGHT("v_lshl_add_u32", ", 4, %8");
            if (OP == 12) EIGHT("v_add3_u32", ", %8, %9");
            if (OP == 13) EIGHT("v_mul_lo_u32", ", %8");
            if (OP == 14) EIGHT("v_mul_u32_u24", ", %8");
            if (OP == 15) EIGHT("v_mad_u32_u24", ", %8, %9");
            if (OP == 16) EIGHT("v_sub_u32", ", %8");
            if (OP == 17) EIGHT("v_mov_b32", "");
            if (OP == 18) EIGHT("v_fmac_f32", ", %8");
            if (OP == 19) EIGHT("v_cvt_f32_u32", "");
            if (OP == 20) EIGHT("v_rcp_f32", "");
            if (OP == 21) EIGHT("v_sqrt_f32", "");
            if (OP == 22) EIGHT("v_bfe_u32", ", 3, 9");
            if (OP == 23) EIGHT("v_mad_u64_u32 %0, vcc, %0, %8, 0 ; ", "");
            if (OP == 24) PAIR4("v_pk_fma_f32", ", %4, %5");
            if (OP == 25) PAIR4("v_pk_mul_f32", ", %4");
            if (OP == 26) PAIR4("v_pk_add_f32", ", %4");
            if (OP == 27) EIGHT("v_mul_hi_u32", ", %8");
            if (OP == 28) EIGHT("v_alignbit_b32", ", %8, 13");
            if (OP == 29) EIGHT("v_xad_u32", ", %8, %9");
            if (OP == 30) EIGHT("v_cvt_f32_ubyte0", "");
            if (OP == 31) EIGHT("v_fract_f32", "");
            if (OP == 32) EIGHT("v_floor_f32", "");
            if (OP == 33) EIGHT("v_sin_f32", "");
            if (OP == 34) EIGHT("v_or3_b32", ", %8, %9");
            if (OP == 35) EIGHT("v_and_or_b32", ", %8, %9");
            if (OP == 36) EIGHT("v_fma_f32", ", |%8|, -%9");
            if (OP == 37) EIGHT("v_min3_f32", ", %8, %9");
            if (OP == 38) EIGHT("v_mul_legacy_f32", ", %8");
            if (OP == 39) EIGHT("v_ldexp_f32", ", %8");
            if (OP == 40) EIGHT("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0] ; ", "");
            if (OP == 41) EIGHT("v_fma_mix_f32 %0, %8, %9, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0] ; ", "");
            if (OP == 42) EIGHT("v_perm_b32", ", %8, %9");
            if (OP == 43) EIGHT("v_cvt_f32_f16", "");
            if (OP == 44) EIGHT("v_cvt_f32_ubyte2", "");
            if (OP == 45) EIGHT("v_or_b32", ", %8");
            if (OP == 46) EIGHT("v_max3_f32", ", %8, %9");
            if (OP == 47) EIGHT("v_cmp_lt_f32 vcc, %0, %8 ; ", "");
            if (OP == 48) EIGHT("v_min_f32", ", %8");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y;
}

template <int OP>
void run(const char* name, float* out, int cus) {
    const int wps = 8, blocks = cus * wps, iters = 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(ops<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f, 0x5555555555555555ull);
    hipEventRecord(e0);
    hipLaunchKernelGGL(ops<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, 0x5555555555555555ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int per = (OP >= 24 && OP <= 26) ? 4 : 8;
    const double rate = (double)iters * 8 * per * wps / (ms * 1e-3);
    printf("%-28s %.3f Ginstr/s/SIMD\n", name, rate / 1e9);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
#define R(n, s) run<n>(s, out, cus)
    R(0, "v_fma_f32"); R(1, "v_mul_f32"); R(2, "v_add_f32"); R(3, "v_sub_f32"); R(4, "v_max_f32"); R(5, "v_med3_f32"); R(6, "v_cndmask_b32 vcc");
    R(7, "v_cndmask_b32 sgpr"); R(8, "v_and_b32"); R(9, "v_xor_b32"); R(10, "v_lshrrev_b32"); R(11, "v_lshl_add_u32"); R(12, "v_add3_u32");
    R(13, "v_mul_lo_u32"); R(14, "v_mul_u32_u24"); R(15, "v_mad_u32_u24"); R(16, "v_sub_u32"); R(17, "v_mov_b32"); R(18, "v_fmac_f32");
    R(19, "v_cvt_f32_u32"); R(20, "v_rcp_f32"); R(21, "v_sqrt_f32"); R(22, "v_bfe_u32"); R(24, "v_pk_fma_f32"); R(25, "v_pk_mul_f32");
    R(26, "v_pk_add_f32"); R(27, "v_mul_hi_u32"); R(28, "v_alignbit_b32"); R(29, "v_xad_u32"); R(30, "v_cvt_f32_ubyte0"); R(31, "v_fract_f32");
    R(32, "v_floor_f32"); R(33, "v_sin_f32"); R(34, "v_or3_b32"); R(35, "v_and_or_b32"); R(36, "v_fma_f32 |b| -c"); R(37, "v_min3_f32");
    R(38, "v_mul_legacy_f32"); R(39, "v_ldexp_f32");
    R(40, "v_fma_mix_f32 lo"); R(41, "v_fma_mix_f32 hi"); R(42, "v_perm_b32"); R(43, "v_cvt_f32_f16"); R(44, "v_cvt_f32_ubyte2");
    R(45, "v_or_b32"); R(46, "v_max3_f32"); R(47, "v_cmp_lt_f32"); R(48, "v_min_f32");
    return 0;
}
