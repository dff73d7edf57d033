// Experiment: single-wave right-looking elimination of a 64 x 16 slab (lane = row, 16 columns in
// registers, pivot-row values fetched with v_readlane) -- time per slab.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__device__ __forceinline__ double rdlane(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int J>
__device__ __forceinline__ void slab_steps(double (&t)[16], double (&rp)[16], double (&pv)[16], const double* dsh, int base, double pivtol, int& nbad) {
    if constexpr (J < 16) {
        double p = rdlane(t[J], base + J);
        const double dj = dsh[base + J];
        const bool bad = !(p > pivtol * dj);
        p = bad ? fmax(dj, 1e-300) : p;
        nbad += bad;
        double rcp = __builtin_amdgcn_rcp(p);
        rcp = rcp * fma(-p, rcp, 2.0);
        const double l = t[J] * rcp;
#pragma unroll
        for (int c = J + 1; c < 16; ++c) t[c] -= l * rdlane(t[J], base + c);
        rp[J] = rcp; pv[J] = p;
        slab_steps<J + 1>(t, rp, pv, dsh, base, pivtol, nbad);
    }
}

__global__ __launch_bounds__(256) void k(const double* A, double* out, int reps, int b) {
    __shared__ double dsh[64];
    const int tid = threadIdx.x;
    if (tid < 64) dsh[tid] = A[tid * 64 + tid];
    __syncthreads();
    if (tid >= 64) return;
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        double t[16], rp[16], pv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = A[tid * 64 + 16 * b + c] + 1e-12 * r;
        int nbad = 0;
        slab_steps<0>(t, rp, pv, dsh, 16 * b, 1e-13, nbad);
#pragma unroll
        for (int c = 0; c < 16; ++c) acc += t[c] * rp[c] + pv[c];
        acc += nbad;
    }
    out[tid] = acc;
}
int main() {
    std::vector<double> A(4096);
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) A[i * 64 + j] = 1.0 / (1.0 + abs(i - j)) + (i == j ? 2.0 : 0.0);
    double *dA, *dO; hipMalloc(&dA, 4096 * 8); hipMalloc(&dO, 4096); hipMemcpy(dA, A.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 2000; float ms = 0;
    for (int w = 0; w < 2; ++w) { hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dA, dO, reps, 0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    printf("slab elimination: %.2f us per slab (%.1f ns per pivot)\n", ms * 1e3 / reps, ms * 1e6 / reps / 16);
    return 0;
}
