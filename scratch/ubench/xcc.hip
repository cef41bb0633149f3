// prints which XCD (HW_REG_XCC_ID) the first 32 blocks of a launch ran on, and how many blocks each XCD got
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    if (threadIdx.x == 0) o[blockIdx.x] = x;
}
int main() {
    unsigned* d; unsigned h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(512), dim3(768), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int i = 0; i < 512; i++) cnt[h[i] & 15]++;
    for (int i = 0; i < 32; i++) printf("%u ", h[i]);
    printf("\n");
    for (int i = 0; i < 16; i++) printf("xcc %d: %d blocks\n", i, cnt[i]);
    return 0;
}
