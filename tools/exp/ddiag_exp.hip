// Where the 32 x 32 double-double diagonal block (k_ddchol_diag) spends its 1.3 us per pivot: timing variants of the kernel
// (results are only checked for variant 0 / 1, the others are timing experiments).
//   hipcc -O3 --offload-arch=gfx950 ddiag_exp.hip -o ddiag_exp ; ./ddiag_exp
#include "../../multiband-rf-pulse-design_amd/csrc/ddlin.hip"
#include <vector>
#include <cmath>
using namespace mbfir;

template <int VAR, int NT>
__global__ __launch_bounds__(NT) void k_diag_var(double* __restrict__ Hh, double* __restrict__ Hl, int np, int k0,
                                                 double* __restrict__ outh, double* __restrict__ outl, long long* ticks) {
    __shared__ double Dh[DNB][DNB + 1], Dl[DNB][DNB + 1];
    __shared__ double Fh[DNB][DNB + 1], Fl[DNB][DNB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < DNB * DNB; e += NT) {
        const int i = e / DNB, c = e - i * DNB;
        const bool lo = c <= i;
        Dh[i][c] = lo ? Hh[(long)(k0 + i) * np + k0 + c] : 0.0;
        Dl[i][c] = lo ? Hl[(long)(k0 + i) * np + k0 + c] : 0.0;
    }
    long long t0 = 0;
    for (int j = 0; j < DNB; ++j) {
        if (VAR != 3) __syncthreads();
        if (j == 1 && tid == 0) t0 = __builtin_amdgcn_s_memtime();
        dd p = dd_make(Dh[j][j], Dl[j][j]);
        dd ri;
        if (VAR == 0) {
            double x = 1.0 / sqrt(p.h);
            x = x * (1.5 - 0.5 * p.h * x * x);
            const dd e1 = dd_sub(dd_make(1.0), dd_mul_d(dd_mul_d(p, x), x));
            ri = dd_add_d(dd_mul_d(e1, 0.5 * x), x);
        } else {
            double x = __builtin_amdgcn_rsq(p.h);
            x = x * (1.5 - 0.5 * p.h * x * x);
            x = x * (1.5 - 0.5 * p.h * x * x);
            const dd t = two_prod(x, x);
            double e1 = __builtin_fma(-p.h, t.h, 1.0);
            e1 = __builtin_fma(-p.h, t.l, e1);
            e1 = __builtin_fma(-p.l, t.h, e1);
            ri = quick_two_sum(x, 0.5 * x * e1);
        }
        for (int e = tid; e < DNB * DNB; e += NT) {
            const int i = e / DNB, c = e - i * DNB;
            if (i == j && c == j) {
                const dd r = dd_mul(p, ri);
                Fh[j][j] = r.h; Fl[j][j] = r.l;
            } else if (c == j && i > j) {
                const dd v = dd_mul(dd_make(Dh[i][j], Dl[i][j]), ri);
                Fh[i][j] = v.h; Fl[i][j] = v.l;
            } else if (c > j && i >= c) {
                const dd li = dd_mul(dd_make(Dh[i][j], Dl[i][j]), ri), lc = dd_mul(dd_make(Dh[c][j], Dl[c][j]), ri);
                const dd v = VAR == 2 ? dd_make(Dh[i][c] - li.h * lc.h, 0.0) : dd_fnma(dd_make(Dh[i][c], Dl[i][c]), li, lc);
                Dh[i][c] = v.h; Dl[i][c] = v.l;
            }
        }
    }
    if (tid == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
    __syncthreads();
    for (int e = tid; e < DNB * DNB; e += NT) {
        const int i = e / DNB, c = e - i * DNB;
        if (c <= i) { outh[i * DNB + c] = Fh[i][c]; outl[i * DNB + c] = Fl[i][c]; }
    }
}

template <int VAR, int NT>
static void run(const char* what, double* dH, double* dHl, int np, double* oh, double* ol, long long* dt, std::vector<double>* ref) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    hipLaunchKernelGGL((k_diag_var<VAR, NT>), dim3(1), dim3(NT), 0, 0, dH, dHl, np, 0, oh, ol, dt);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_diag_var<VAR, NT>), dim3(1), dim3(NT), 0, 0, dH, dHl, np, 0, oh, ol, dt);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    std::vector<double> h(DNB * DNB), l(DNB * DNB);
    hipMemcpy(h.data(), oh, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(l.data(), ol, l.size() * 8, hipMemcpyDeviceToHost);
    double dmax = 0;
    if (ref && !ref->empty()) for (size_t q = 0; q < h.size(); ++q) dmax = std::max(dmax, std::fabs((h[q] - (*ref)[q]) + (l[q] - (*ref)[h.size() + q])));
    if (ref && ref->empty()) { ref->insert(ref->end(), h.begin(), h.end()); ref->insert(ref->end(), l.begin(), l.end()); }
    printf("%-64s %3d threads: %6.2f us per launch back to back, pivots 1..31 %.2f us (100 MHz ticks), max |diff to variant 0| %.2e  %s\n", what, NT, 1e3 * ms / reps,
           0.01 * double(t), dmax, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int np = 64;
    std::vector<double> H(np * np), Hl(np * np, 0.0);
    for (int i = 0; i < np; ++i) for (int j = 0; j < np; ++j) H[i * np + j] = 1.0 / (1.0 + abs(i - j)) + (i == j ? 2.0 : 0.0);
    double *dH, *dHl, *oh, *ol; long long* dt;
    hipMalloc(&dH, np * np * 8); hipMalloc(&dHl, np * np * 8); hipMalloc(&oh, DNB * DNB * 8); hipMalloc(&ol, DNB * DNB * 8); hipMalloc(&dt, 64);
    hipMemcpy(dH, H.data(), np * np * 8, hipMemcpyHostToDevice); hipMemcpy(dHl, Hl.data(), np * np * 8, hipMemcpyHostToDevice);
    std::vector<double> ref;
    run<0, 1024>("as in ddlin.hip (1 / sqrt, full dd Newton step)", dH, dHl, np, oh, ol, dt, &ref);
    run<0, 256>("the same, four entries per thread", dH, dHl, np, oh, ol, dt, &ref);
    run<1, 1024>("rsq + 2 double Newton steps, residual by three fma", dH, dHl, np, oh, ol, dt, &ref);
    run<1, 256>("the same, four entries per thread", dH, dHl, np, oh, ol, dt, &ref);
    run<2, 1024>("short 1/sqrt and a DOUBLE update (timing only)", dH, dHl, np, oh, ol, dt, &ref);
    run<3, 1024>("short 1/sqrt, no barrier (timing only)", dH, dHl, np, oh, ol, dt, &ref);
    return 0;
}
