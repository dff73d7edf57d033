#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * blockIdx.y] = x & 15;
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(16, 8), dim3(256), 0, 0, d);
    int h[128]; hipMemcpy(h, d, 128 * 4, hipMemcpyDeviceToHost);
    for (int y = 0; y < 8; ++y) { for (int x = 0; x < 16; ++x) printf("%d ", h[x + 16 * y]); printf("\n"); }
    return 0;
}
