// Experiment: right-looking register substitution vs the left-looking one (one 256-thread WG).
#include "../../multiband-rf-pulse-design_amd/csrc/chol.hip"
#include <vector>
#include <cmath>
using namespace mbfir;

template <int JJ>
__device__ __forceinline__ double quad_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, JJ * 0x55, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, JJ * 0x55, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

#define RL_STEP(JJ)                                                                   \
    {                                                                                 \
        const double* lcol = &Lt[4 * jg + JJ][q * 16];                                \
        const double x = quad_bcast<JJ>(cur * dm);                                    \
        const double lc = lcol[jg];                                                   \
        cur = (q == JJ) ? x : ((q > JJ) ? cur - lc * x : cur);                        \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) v[i] -= ((i > jg) ? x : 0.0) * lcol[i]; \
    }

// Lt[j][cperm(t)] = L[t][j]
__device__ __forceinline__ void subst64_rl(const double (*Lt)[SLD], const double* dinv, v16d& v) {
    const int q = threadIdx.x & 3;
#pragma unroll 1
    for (int jg = 0; jg < 16; ++jg) {
        double cur = sel16(v, jg);
        const double dm = dinv[4 * jg + q];
        RL_STEP(0) RL_STEP(1) RL_STEP(2) RL_STEP(3)
        put16(v, jg, cur);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_exp(const double* L, const double* B, double* X, int reps) {
    __shared__ __attribute__((aligned(16))) double smem[PANEL_LDS];
    double(*Sp)[SLD] = reinterpret_cast<double(*)[SLD]>(smem);
    double* dinv = smem + CB * SLD;
    const int tid = threadIdx.x;
    for (int e = tid; e < CB * CB; e += 256) {
        if (MODE == 0) Sp[e >> 6][cperm(e & 63)] = L[e];
        else Sp[e & 63][cperm(e >> 6)] = L[e];
    }
    if (tid < CB) dinv[tid] = 1.0 / L[tid * 65];
    __syncthreads();
    const int c = tid >> 2, q = tid & 3;
    const double* col = B + (long)blockIdx.x * 4096 + c;
    double* xo = X + (long)blockIdx.x * 4096 + c;
    for (int r = 0; r < reps; ++r) {
        v16d v;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = col[(4 * i + q) * 64];
        if (MODE == 0) subst64(Sp, dinv, v); else subst64_rl(Sp, dinv, v);
#pragma unroll
        for (int i = 0; i < 16; ++i) xo[(4 * i + q) * 64] = v[i];
    }
}

int main() {
    std::vector<double> L(4096, 0.0), B(4096 * 16), X(4096 * 16), Xr(4096);
    srand(1);
    for (int i = 0; i < 64; ++i) for (int j = 0; j <= i; ++j) L[i * 64 + j] = (i == j) ? 1.0 + 0.5 * (rand() / (double)RAND_MAX) : 0.3 * (rand() / (double)RAND_MAX - 0.5);
    for (auto& b : B) b = rand() / (double)RAND_MAX - 0.5;
    for (int c = 0; c < 64; ++c) for (int t = 0; t < 64; ++t) {
        double s = B[t * 64 + c];
        for (int j = 0; j < t; ++j) s -= L[t * 64 + j] * Xr[j * 64 + c];
        Xr[t * 64 + c] = s / L[t * 65];
    }
    double *dL, *dB, *dX; hipMalloc(&dL, 4096 * 8); hipMalloc(&dB, 4096 * 16 * 8); hipMalloc(&dX, 4096 * 16 * 8);
    hipMemcpy(dL, L.data(), 4096 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 4096 * 16 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) for (int nb : {1, 16}) {
        const int reps = 200; float ms = 0;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_exp<0>, dim3(nb), dim3(256), 0, 0, dL, dB, dX, reps);
            else hipLaunchKernelGGL(k_exp<1>, dim3(nb), dim3(256), 0, 0, dL, dB, dX, reps);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(X.data(), dX, 4096 * 8, hipMemcpyDeviceToHost);
        double err = 0; for (int i = 0; i < 4096; ++i) err = fmax(err, fabs(X[i] - Xr[i]));
        printf("mode %d blocks %2d: %.2f us per substitution, max err %.2e\n", mode, nb, ms * 1e3 / reps, err);
    }
    return 0;
}
