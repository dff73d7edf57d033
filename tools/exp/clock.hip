// Experiment: shader clock seen by a small (1 block) vs a chip-filling launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* t, int iters) {
    long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a = 1.0 + threadIdx.x * 1e-9, x = 0.5;
    for (int i = 0; i < iters; ++i) x = fma(x, a, 1e-9);
    long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
int main() {
    double* out; long long* t; hipMalloc(&out, 1 << 24); hipMalloc(&t, 64);
    long long h[2];
    for (int rep = 0; rep < 3; ++rep)
    for (int blocks : {1, 16, 512, 4096}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, t, 200000); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("blocks %5d: %.3f ms, memtime %lld realtime(100MHz) %lld -> shader clock %.0f MHz, %.1f cycles per dependent v_fma_f64\n",
               blocks, ms, h[0], h[1], double(h[0]) / double(h[1]) * 100.0, double(h[0]) / 200000.0);
    }
    return 0;
}
