// Experiment: per-primitive costs inside one 256-thread workgroup (cycles from s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define T0 long long t0 = __builtin_amdgcn_s_memtime();
#define T1(i) { long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) res[i] = t1 - t0; }
__global__ __launch_bounds__(256) void k(double* out, long long* res, int n) {
    __shared__ double sh[4096];
    const int tid = threadIdx.x;
    double x = 1.0 + tid * 1e-9, acc = 0;
    for (int i = tid; i < 4096; i += 256) sh[i] = i * 1e-3;
    __syncthreads();
    { T0 for (int i = 0; i < n; ++i) __syncthreads(); T1(0) }
    { T0 for (int i = 0; i < n; ++i) { sh[tid] = x; __syncthreads(); x += sh[(tid + 1) & 255]; } T1(1) }       // store-barrier-load-add chain
    { double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
      T0 for (int i = 0; i < n; ++i) { a0 = fma(a0, 1.0000001, 1e-9); a1 = fma(a1, 1.0000001, 1e-9); a2 = fma(a2, 1.0000001, 1e-9); a3 = fma(a3, 1.0000001, 1e-9);
                                       a4 = fma(a4, 1.0000001, 1e-9); a5 = fma(a5, 1.0000001, 1e-9); a6 = fma(a6, 1.0000001, 1e-9); a7 = fma(a7, 1.0000001, 1e-9); } T1(2)
      acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }                                                         // 8 independent fma per iter
    { T0 for (int i = 0; i < n; ++i) { double r = __builtin_amdgcn_rcp(x); r = r * fma(-x, r, 2.0); x = x + r * 1e-9; } T1(3) }     // rcp + newton chain
    { T0 for (int i = 0; i < n; ++i) { x = x + sh[(tid * 17 + i) & 4095]; } T1(4) }                           // dependent LDS load + add
    { double v[16]; for (int i = 0; i < 16; ++i) v[i] = x + i;
      T0 for (int it = 0; it < n; ++it) { int jg = it & 15;
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] -= ((i > jg) ? x : 0.0) * sh[tid * 16 + i]; } T1(5)
      for (int i = 0; i < 16; ++i) acc += v[i]; }                                                            // 16 select+fma with LDS operand
    { T0 for (int i = 0; i < n; ++i) { x = sqrt(x + 1.0); } T1(6) }
    { T0 for (int i = 0; i < n; ++i) { x = 1.0 / (x + 1.0); } T1(7) }
    out[blockIdx.x * 256 + tid] = x + acc;
}
int main() {
    double* out; long long* res; hipMalloc(&out, 1 << 20); hipMalloc(&res, 256);
    const int n = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, res, n);
    long long h[8]; hipMemcpy(h, res, 64, hipMemcpyDeviceToHost);
    const char* names[8] = {"__syncthreads", "LDS store + barrier + load + add", "8 independent v_fma_f64", "rcp + newton + add chain",
                            "dependent LDS load + add", "16 x (select, fma, LDS operand)", "sqrt(double)", "1.0/double"};
    for (int i = 0; i < 8; ++i) printf("%-36s %8.1f cycles / iteration\n", names[i], double(h[i]) / n);
    return 0;
}
