// Experiment: latency of DEPENDENT fp64 operations (one wave per SIMD, 32 dependent ops per loop trip).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, int n) {
    __shared__ double sh[512];
    const int tid = threadIdx.x;
    sh[tid] = 1.0 + tid * 1e-9; sh[256 + tid] = 0.5;
    __syncthreads();
    double x = 1.0 + 1e-9 * tid, y = 1.0000001;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (KIND == 0) x = fma(x, y, 1e-9);
            else if (KIND == 1) x = x * y;
            else if (KIND == 2) x = x + y;
            else if (KIND == 3) x = __builtin_amdgcn_rcp(x) + 0.0 * y;     // rcp (+ fold-proof)
            else if (KIND == 4) x = (x > y) ? x : y + 1e-9;                // cmp + select
            else if (KIND == 5) x = __builtin_amdgcn_update_dpp(0.0, x, 0x150 + 3, 0xF, 0xF, true);
            else if (KIND == 6) { sh[tid] = x; __syncthreads(); x = sh[(tid + 1) & 255] + 1e-9; }    // store, barrier, load, add
            else if (KIND == 7) { float f = (float)x; f = fmaf(f, 1.0000001f, 1e-9f); x = f; }          // cvt + f32 fma + cvt
            else if (KIND == 8) x = __builtin_amdgcn_rsq(x);
        }
    }
    out[blockIdx.x * 256 + tid] = x;
}
template <int KIND> void run(const char* name, double* out) {
    const int n = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256), 0, 0, out, n); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-40s %7.2f ns per dependent op\n", name, ms * 1e6 / n / 32);
}
int main() {
    double* out; hipMalloc(&out, 1 << 20);
    run<0>("v_fma_f64", out); run<1>("v_mul_f64", out); run<2>("v_add_f64", out); run<3>("v_rcp_f64 (+add)", out);
    run<4>("v_cmp_f64 + v_cndmask x2", out); run<5>("v_mov_b64_dpp row_newbcast", out);
    run<6>("ds_write + s_barrier + ds_read + add", out); run<7>("cvt f64->f32, fma f32, cvt back", out); run<8>("v_rsq_f64", out);
    return 0;
}
