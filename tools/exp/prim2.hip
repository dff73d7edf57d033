// Experiment: issue cost of the instruction kinds the Cholesky inner loops are made of, for one
// workgroup of 256 (1 wave / SIMD) and 512 / 1024 threads (2 / 4 waves / SIMD).  ns per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v16d __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(1024) void k(double* out, int n) {
    __shared__ __attribute__((aligned(16))) double sh[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) sh[i] = 1e-3 * i;
    __syncthreads();
    v16d v; for (int i = 0; i < 16; ++i) v[i] = 1.0 + tid * 1e-9 + i;
    double x = 1.0 + 1e-9 * tid;
    const double* row = sh + (tid & 3) * 16;
    for (int it = 0; it < n; ++it) {
        if (KIND == 0) {            // 16 independent fma, register operands
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fma(v[i], x, 1e-9);
        } else if (KIND == 1) {     // 16 fma with LDS operands (8 ds_read_b128)
            const double* r = row + (it & 31) * 64;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fma(r[i], x, v[i]);
        } else if (KIND == 2) {     // 16 x (uniform select + fma), LDS operands
            const double* r = row + (it & 31) * 64; const int jg = it & 15;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] -= ((i > jg) ? x : 0.0) * r[i];
        } else if (KIND == 3) {     // dependent chain: mul -> dpp bcast -> fma
            int lo = __double2loint(x * 1.0000001), hi = __double2hiint(x * 1.0000001);
            lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xF, 0xF, true);
            x = fma(__hiloint2double(hi, lo), 1e-9, x);
        } else if (KIND == 4) {     // 16 independent f32 fma
            float* f = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = fmaf(f[i], 1.0000001f, 1e-9f);
        } else if (KIND == 5) {     // 8 MFMA f64 16x16x4 independent
            typedef double v4d __attribute__((ext_vector_type(4)));
            v4d* a = reinterpret_cast<v4d*>(&v);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a[i], 0, 0, 0);
        }
    }
    double s = x; for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + tid] = s;
}
template <int KIND> void run(const char* name, int per_iter, double* out) {
    const int n = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 1024}) {
        float ms = 0;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, n); hipEventRecord(e1);
            hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-44s threads %4d: %7.2f ns / iteration, %6.2f ns per wave-instruction (per SIMD: %6.2f)\n", name, threads, ms * 1e6 / n,
               ms * 1e6 / n / per_iter, ms * 1e6 / n / per_iter / (threads / 256));
    }
}
int main() {
    double* out; hipMalloc(&out, 1 << 20);
    run<0>("16 indep v_fma_f64 (regs)", 16, out);
    run<1>("16 v_fma_f64 + 8 ds_read_b128", 16, out);
    run<2>("16 (select + fma) + 8 ds_read_b128", 16, out);
    run<3>("chain mul -> dpp -> fma", 1, out);
    run<4>("16 indep v_fma_f32", 16, out);
    run<5>("8 indep mfma_f64_16x16x4", 8, out);
    return 0;
}
