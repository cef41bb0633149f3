// Microbenchmarks behind the box stage of render_kernel (gfx950):
//  1. issue rate of the instruction classes a box step is made of, at 1..8 waves per SIMD;
//  2. the box step itself -- entry read from LDS, box test, next-entry select -- as a bare loop with all 64 lanes
//     active and no scheduling code around it: the ceiling of the box stage, for the 32-byte float entries and for the
//     16-byte packed slots.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../../include -I../../ray-tracer_amd/csrc box.hip -o box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "rtx_device.hpp"
using namespace rtxd;

template <int OP>
__global__ void ops(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define EIGHT(INS, TAIL) asm volatile(INS " %0, %0" TAIL "\n" INS " %1, %1" TAIL "\n" INS " %2, %2" TAIL "\n" INS " %3, %3" TAIL "\n" \
                                      INS " %4, %4" TAIL "\n" INS " %5, %5" TAIL "\n" INS " %6, %6" TAIL "\n" INS " %7, %7" TAIL "\n" \
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc")
            if (OP == 0) EIGHT("v_fma_f32", ", %8, %9");
            if (OP == 1) EIGHT("v_fma_mix_f32", ", %8, %9 op_sel_hi:[1,0,0]");
            if (OP == 2) EIGHT("v_fma_mix_f32", ", |%8|, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]");
            if (OP == 3) EIGHT("v_min_f32", ", %8");
            if (OP == 4) EIGHT("v_max3_f32", ", %8, %9");
            if (OP == 5) EIGHT("v_cndmask_b32", ", %8, vcc");
            if (OP == 6) EIGHT("v_add_u32", ", %8");
            if (OP == 7) EIGHT("v_lshlrev_b32", ", 4");
            if (OP == 8) {
                asm volatile("v_cmp_nle_f32 vcc, %0, %8\n v_cmp_nle_f32 vcc, %1, %8\n v_cmp_nle_f32 vcc, %2, %8\n v_cmp_nle_f32 vcc, %3, %8\n"
                             "v_cmp_nle_f32 vcc, %4, %8\n v_cmp_nle_f32 vcc, %5, %8\n v_cmp_nle_f32 vcc, %6, %8\n v_cmp_nle_f32 vcc, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

extern __shared__ uint4 smem[];
// A bare walk over a synthetic threaded array in LDS: every lane starts at its own entry and takes `steps` box steps.
template <bool PK>
__global__ __launch_bounds__(1024, 8) void walk(const uint4* __restrict__ ents, uint32_t n_u4, int steps, uint32_t* out) {
    for (uint32_t i = threadIdx.x; i < n_u4; i += blockDim.x) smem[i] = ents[i];
    __syncthreads();
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    Ray r;
    r.ox = 13.0f + 0.01f * threadIdx.x; r.oy = 2.0f; r.oz = 3.0f;
    r.dx = -1.0f + 0.003f * (threadIdx.x & 63); r.dy = -0.2f; r.dz = -0.3f + 0.001f * threadIdx.x;
    r.time = 0.5f;
    const RayConst rc = ray_const<SLAB_FMA>(r);
    const uint32_t n_ent = PK ? n_u4 : n_u4 / 2;
    uint32_t off = ((threadIdx.x * 2654435761u) % n_ent) * (PK ? 16u : 32u);
    float tbest = 1e30f;
    uint32_t acc = 0;
    if (PK) {
        u32x4 e = *(lds_u4*)(uintptr_t)off;
        for (int s = 0; s < steps; s++) {
            const bool hit = pk_slab_hit(make_uint4(e.x, e.y, e.z, e.w), rc, 0.001f, tbest);
            off = hit ? off + 16u : e.w * 16u;
            acc += off;
            e = *(lds_u4*)(uintptr_t)off;
        }
    } else {
        u32x4 a = *(lds_u4*)(uintptr_t)off, b = *(lds_u4*)(uintptr_t)(off + 16u);
        for (int s = 0; s < steps; s++) {
            const bool hit = slab_hit<SLAB_FMA>(make_uint4(a.x, a.y, a.z, a.w), make_uint4(b.x, b.y, b.z, b.w), r, rc, 0.001f, tbest);
            off = hit ? off + 32u : b.z;
            acc += off;
            a = *(lds_u4*)(uintptr_t)off;
            b = *(lds_u4*)(uintptr_t)(off + 16u);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static uint16_t half_bits(float f) {  // round to nearest, normal range only (enough for the synthetic boxes)
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7FFFFFFFu;
    if (x < 0x38800000u) return (uint16_t)sign;
    return (uint16_t)(sign | ((x - 0x38000000u + 0x1000u) >> 13));
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * sizeof(float));
    const char* names[] = {"v_fma_f32", "v_fma_mix_f32 (f16 lo src0)", "v_fma_mix_f32 (f16 hi src0, |src1|)", "v_min_f32", "v_max3_f32", "v_cndmask_b32 (vcc)",
                           "v_add_u32", "v_lshlrev_b32", "v_cmp_nle_f32 (vcc)"};
    void (*kern[])(float*, int, float, float) = {ops<0>, ops<1>, ops<2>, ops<3>, ops<4>, ops<5>, ops<6>, ops<7>, ops<8>};
    for (int op = 0; op < 9; op++) {
        for (int wps : {1, 4, 8}) {
            const int blocks = cus * wps, iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kern[op], dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern[op], dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double rate = (double)iters * 64 * wps / (ms * 1e-3);
            printf("%-38s waves/SIMD=%d  %.3f Ginstr/s/SIMD\n", names[op], wps, rate / 1e9);
        }
    }
    // synthetic threaded arrays: 1400 boxes around the rays' region, random forward miss links, wrap at the end
    const uint32_t n = 1400;
    std::mt19937 gen(5);
    std::uniform_real_distribution<float> U(-11.0f, 11.0f), W(0.1f, 6.0f);
    std::vector<uint32_t> f32(8 * n), pk(4 * n);
    for (uint32_t i = 0; i < n; i++) {
        float c[3] = {U(gen), 0.2f, U(gen)}, h[3] = {W(gen), 0.3f, W(gen)};
        const uint32_t skip = (i + 2 + gen() % 40) % n, next_ok = (i + 1) % n;
        (void)next_ok;
        float b[6] = {c[0] - h[0], c[0] + h[0], c[1] - h[1], c[1] + h[1], c[2] - h[2], c[2] + h[2]};
        std::memcpy(&f32[8 * i], b, 24);
        f32[8 * i + 6] = skip * 32; f32[8 * i + 7] = 0;
        pk[4 * i + 0] = half_bits(c[0]) | ((uint32_t)half_bits(c[1]) << 16);
        pk[4 * i + 1] = half_bits(c[2]) | ((uint32_t)half_bits(h[0]) << 16);
        pk[4 * i + 2] = half_bits(h[1]) | ((uint32_t)half_bits(h[2]) << 16);
        pk[4 * i + 3] = skip;
    }
    // the last entries' "hit" successor would run off the end: make the final entry's box unhittable
    float far_box[6] = {1e9f, 2e9f, 1e9f, 2e9f, 1e9f, 2e9f};
    std::memcpy(&f32[8 * (n - 1)], far_box, 24); f32[8 * (n - 1) + 6] = 0;
    pk[4 * (n - 1) + 0] = 0x7BFFu | (0x7BFFu << 16); pk[4 * (n - 1) + 1] = 0x7BFFu; pk[4 * (n - 1) + 2] = 0; pk[4 * (n - 1) + 3] = 0;
    uint4 *d32, *dpk; uint32_t* dout;
    hipMalloc(&d32, f32.size() * 4); hipMalloc(&dpk, pk.size() * 4); hipMalloc(&dout, (size_t)cus * 2 * 1024 * 4);
    hipMemcpy(d32, f32.data(), f32.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; variant++) {
        const int steps = 20000;
        const uint32_t n_u4 = variant ? n : 2 * n;
        const size_t lds = (size_t)n_u4 * 16;
        hipFuncSetAttribute((const void*)(variant ? walk<true> : walk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (variant) hipLaunchKernelGGL(walk<true>, dim3(cus * 2), dim3(1024), lds, 0, dpk, n_u4, steps, dout);
            else hipLaunchKernelGGL(walk<false>, dim3(cus * 2), dim3(1024), lds, 0, d32, n_u4, steps, dout);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)steps * 8 / (ms * 1e-3);  // 32 waves per CU = 8 per SIMD
        printf("bare box walk, %s: %.3f ms, %.3f G wave-steps/s/SIMD = %.1f G lane-steps/s on the chip\n", variant ? "packed 16-byte slots" : "32-byte float entries",
               ms, per_simd / 1e9, per_simd * cus * 4 * 64 / 1e9);
    }
    return 0;
}
