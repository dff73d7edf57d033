// K7: tap extraction of the arbitrary-phase designer -- autocorrelation -> minimum-phase taps by
// FFT spectral factorisation, the device restatement of fir_ap_cvx.m:185-186 (reshape),
// :264-284 (fmp2) and :294-304 (mag2mp).  Three FFTs of length lp = 8*2^ceil(log2(2n-1))
// (<= 65536): one workgroup, in-place radix-2 on global scratch, twiddles by sincospi.
#include "dev_common.h"

namespace mbfir {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-place FFT of length n = 2^logn; sign = -1 forward, +1 inverse (unscaled).
__device__ void fft_inplace(double2* a, int n, int logn, int sign) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n; i += nt) {
        int j = int(__brev((unsigned)i) >> (32 - logn));
        if (i < j) { double2 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    __syncthreads();
    for (int s = 1; s <= logn; ++s) {
        const int m = 1 << s, half = m >> 1;
        for (int idx = tid; idx < n / 2; idx += nt) {
            int j = idx & (half - 1);
            int k = (idx >> (s - 1)) << s;
            double sn, cs;
            sincospi(2.0 * double(j) / double(m), &sn, &cs);
            double2 w = make_double2(cs, sign * sn);
            double2 u = a[k + j], t = cmul(w, a[k + j + half]);
            a[k + j] = make_double2(u.x + t.x, u.y + t.y);
            a[k + j + half] = make_double2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
}

// blockIdx.x = design of a lock-step unit: every pointer moves by blockIdx.x * lane_bytes
__global__ __launch_bounds__(1024) void k_specfact(const double* __restrict__ x, int n, int lp, int loglp,
                                                   double2* __restrict__ B0, double* __restrict__ hout, size_t lane_bytes) {
    {
        const size_t off = (size_t)blockIdx.x * lane_bytes;
        x = reinterpret_cast<const double*>(reinterpret_cast<const char*>(x) + off);
        B0 = reinterpret_cast<double2*>(reinterpret_cast<char*>(B0) + off);
        hout = reinterpret_cast<double*>(reinterpret_cast<char*>(hout) + off);
    }
    double2* __restrict__ B1 = B0 + lp;
    double2* __restrict__ B2 = B0 + 2 * lp;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int l = 2 * n - 1;
    const int pad_lo = (lp - l + 1) / 2;                 // ceil((lp-l)/2)            :273
    const int hlp = lp / 2;
    // B0 = fftshift(hp), hp = [zeros r zeros], r two-sided Hermitian              :185-186,273
    for (int i = tid; i < lp; i += nt) {
        int q = (i + hlp) & (lp - 1);                    // index into hp
        int t = q - pad_lo;
        double2 v = make_double2(0, 0);
        if (t >= 0 && t < l) {
            int lag = t - (n - 1);
            int al = lag < 0 ? -lag : lag;
            double re = x[al];
            double im = al == 0 ? 0.0 : x[n - 1 + al];
            v = make_double2(re, lag < 0 ? -im : im);
        }
        B0[i] = v;
    }
    __syncthreads();
    fft_inplace(B0, lp, loglp, -1);                      // hpf = fftshift(B0)               :274
    // xl = log(sqrt(abs(hpf)))                                                   :281,296
    for (int i = tid; i < lp; i += nt) {
        double2 v = B0[(i + hlp) & (lp - 1)];
        B1[i] = make_double2(log(sqrt(hypot(v.x, v.y))), 0.0);
    }
    __syncthreads();
    fft_inplace(B1, lp, loglp, -1);                      // xlf                             :297
    for (int i = tid; i < lp; i += nt) {                 // :298-301
        double2 v = B1[i];
        if (i >= 1 && i < hlp) v = make_double2(2 * v.x, 2 * v.y);
        else if (i > hlp) v = make_double2(0, 0);
        B1[i] = v;
    }
    __syncthreads();
    fft_inplace(B1, lp, loglp, +1);                      // xlaf * lp                        :302
    const double inv = 1.0 / double(lp);
    for (int i = tid; i < lp; i += nt) {                 // a = exp(xlaf)                    :303
        double2 v = B1[i];
        double e = exp(v.x * inv), sn, cs;
        sincos(v.y * inv, &sn, &cs);
        B1[i] = make_double2(e * cs, e * sn);
    }
    __syncthreads();
    for (int i = tid; i < lp; i += nt) {                 // fftshift(conj(hpfmp))            :282
        double2 v = B1[(i + hlp) & (lp - 1)];
        B2[i] = make_double2(v.x, -v.y);
    }
    __syncthreads();
    fft_inplace(B2, lp, loglp, +1);
    for (int i = tid; i < n; i += nt) {                  // hmp = hpmp(1:(l+1)/2)            :283
        hout[2 * i] = B2[i].x * inv;
        hout[2 * i + 1] = B2[i].y * inv;
    }
}

int specfact_lp(int n) {
    int l = 2 * n - 1, p = 1;
    while (p < l) p <<= 1;                               // 2^ceil(log2(l))                  :272
    return 8 * p;
}

void specfact_launch(const double* x, int n, double* work, double* hout, hipStream_t st, int nlanes, size_t lane_bytes) {
    int lp = specfact_lp(n), loglp = 0;
    while ((1 << loglp) < lp) ++loglp;
    hipLaunchKernelGGL(k_specfact, dim3(nlanes), dim3(1024), 0, st, x, n, lp, loglp, reinterpret_cast<double2*>(work), hout, lane_bytes);
}

}  // namespace mbfir
