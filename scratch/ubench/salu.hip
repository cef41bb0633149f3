// Scalar-instruction issue on gfx950 at 8 waves per SIMD: alone, and interleaved with vector multiply-adds at several
// scalar : vector ratios (the render kernel issues 0.58 scalar instructions per vector instruction).
//   hipcc -O3 --offload-arch=gfx950 salu.hip -o salu
#include <hip/hip_runtime.h>
#include <cstdio>

#define V8 asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n" \
                        "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n" \
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b))
#define S1(r) asm volatile("s_add_u32 %0, %0, %1" : "+s"(r) : "s"(k) : "scc")

// NS scalar adds per eight vector multiply-adds; NV = 0: scalar adds only
template <int NS, int NV>
__global__ void mix(float* out, int iters, float a, float b, unsigned k) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned s0 = k, s1 = k + 1, s2 = k + 2, s3 = k + 3, s4 = k + 4, s5 = k + 5, s6 = k + 6, s7 = k + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (NV) V8;
            if (NS > 0) S1(s0);
            if (NS > 1) S1(s1);
            if (NS > 2) S1(s2);
            if (NS > 3) S1(s3);
            if (NS > 4) S1(s4);
            if (NS > 5) S1(s5);
            if (NS > 6) S1(s6);
            if (NS > 7) S1(s7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
}

template <int NS, int NV>
void run(float* out, int cus) {
    const int wps = 8, blocks = cus * wps, iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<NS, NV>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f, 3u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mix<NS, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves x iters x 8 x (8 vector | NS scalar) instructions; a CU has four SIMDs
    const double v = NV ? (double)iters * 8 * 8 * wps / (ms * 1e-3) / 1e9 : 0.0, s = (double)iters * 8 * NS * wps / (ms * 1e-3) / 1e9;
    printf("%d scalar per 8 vector: vector %.3f G/s/SIMD, scalar %.3f G/s/SIMD = %.3f G/s/CU\n", NS, v, s, 4 * s);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    float* out;
    (void)hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * sizeof(float));
    printf("%s, %d CUs, %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<8, 0>(out, p.multiProcessorCount);
    run<0, 1>(out, p.multiProcessorCount);
    run<2, 1>(out, p.multiProcessorCount);
    run<4, 1>(out, p.multiProcessorCount);
    run<5, 1>(out, p.multiProcessorCount);
    run<6, 1>(out, p.multiProcessorCount);
    run<8, 1>(out, p.multiProcessorCount);
    return 0;
}
