// Inverse SLR step on the device: beta polynomial -> minimum-phase alpha -> RF pulse,
// the restatement of b2a.m:15-32 (with mag2mp.m:21-31) and ab2rf.m:14-29 as dzrf_mb.m:239-240 calls them.
//   b2a   : DFTs of length blp = 8 n (any n: direct O(blp^2) DFT, twiddles by rotation recurrence with an
//           exact sincospi seed every 256 terms), elementwise steps in between
//   ab2rf : the n-step inverse SLR recursion in one workgroup, the two polynomials in LDS (ping-pong)
#include "dev_common.h"

namespace mbfir {

__device__ __forceinline__ double2 cmul2(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// out[k] = scale * sum_j in[j] exp(sign * 2 pi i j k / N); one thread per k, input staged through LDS
constexpr int DFT_TILE = 256;
__global__ __launch_bounds__(256) void k_dft_any(const double2* __restrict__ in, double2* __restrict__ out, int N, int sign,
                                                 double scale) {
    __shared__ double2 tile[DFT_TILE];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int kk = k < N ? k : 0;
    double sw, cw;
    sincospi(2.0 * double(kk) / double(N), &sw, &cw);
    sw *= sign;
    double2 acc = make_double2(0, 0);
    for (int j0 = 0; j0 < N; j0 += DFT_TILE) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        tile[threadIdx.x] = j < N ? in[j] : make_double2(0, 0);
        __syncthreads();
        const long ph = ((long)kk * j0) % N;              // exact phase of the tile's first term
        double s, c;
        sincospi(2.0 * double(ph) / double(N), &s, &c);
        s *= sign;
        const int cnt = min(DFT_TILE, N - j0);
        for (int q = 0; q < cnt; ++q) {
            const double2 v = tile[q];
            acc.x += v.x * c - v.y * s;
            acc.y += v.x * s + v.y * c;
            const double cn = c * cw - s * sw;
            s = s * cw + c * sw;
            c = cn;
        }
    }
    if (k < N) out[k] = make_double2(acc.x * scale, acc.y * scale);
}

// B0 = [b ; zeros] (length N)
__global__ void k_slr_pad(const double* __restrict__ b_re, const double* __restrict__ b_im, int n, int N, double2* __restrict__ B0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) B0[i] = i < n ? make_double2(b_re[i], b_im[i]) : make_double2(0, 0);
}
// xl = log(sqrt(1 - |bf|^2)) with bf scaled by 1 / (1e-8 + max|bf|) when max|bf| >= 1     (b2a.m:24-29, mag2mp.m:24)
__global__ __launch_bounds__(1024) void k_slr_logmag(const double2* __restrict__ bf, int N, double2* __restrict__ xl) {
    __shared__ double sh[17];
    double m = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) m = fmax(m, hypot(bf[i].x, bf[i].y));
    m = block_max(m, sh);
    const double sc = m >= 1.0 ? 1.0 / (1e-8 + m) : 1.0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double re = bf[i].x * sc, im = bf[i].y * sc;
        xl[i] = make_double2(log(sqrt(1.0 - (re * re + im * im))), 0.0);
    }
}
// keep DC and N/2, double 1 .. N/2-1, zero the rest                                        (mag2mp.m:26-29)
__global__ void k_slr_window(double2* __restrict__ x, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double2 v = x[i];
    if (i >= 1 && i < N / 2) v = make_double2(2 * v.x, 2 * v.y);
    else if (i > N / 2) v = make_double2(0, 0);
    x[i] = v;
}
// a = exp(xlaf)                                                                            (mag2mp.m:31)
__global__ void k_slr_exp(double2* __restrict__ x, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double2 v = x[i];
    double s, c;
    sincos(v.y, &s, &c);
    const double e = exp(v.x);
    x[i] = make_double2(e * c, e * s);
}
// aca(n:-1:1)                                                                              (b2a.m:31-32)
__global__ void k_slr_out(const double2* __restrict__ aca, int n, double* __restrict__ a_il) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a_il[2 * i] = aca[n - 1 - i].x; a_il[2 * i + 1] = aca[n - 1 - i].y; }
}

// inverse SLR recursion; a_il / b_il / rf_il interleaved (re, im); n <= SLR_MAXN
constexpr int SLR_MAXN = 2048;
__global__ __launch_bounds__(1024) void k_ab2rf(const double* __restrict__ a_il, const double* __restrict__ b_il, int n,
                                                double* __restrict__ rf_il) {
    __shared__ double2 A[2][SLR_MAXN], B[2][SLR_MAXN];
    __shared__ double2 cs[2];                             // (c, 0), s
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        A[0][i] = make_double2(a_il[2 * i], a_il[2 * i + 1]);
        B[0][i] = make_double2(b_il[2 * i], b_il[2 * i + 1]);
    }
    __syncthreads();
    int cur = 0;
    for (int i = n; i >= 1; --i) {
        if (threadIdx.x == 0) {
            const double2 ai = A[cur][i - 1], bi = B[cur][i - 1];
            const double den = ai.x * ai.x + ai.y * ai.y;
            const double2 q = make_double2((bi.x * ai.x + bi.y * ai.y) / den, (bi.y * ai.x - bi.x * ai.y) / den);   // b / a
            const double c = sqrt(1.0 / (1.0 + (q.x * q.x + q.y * q.y)));
            const double2 s = make_double2(c * q.x, -c * q.y);                                                      // conj(c b / a)
            const double theta = atan2(hypot(s.x, s.y), c), psi = atan2(s.y, s.x);
            rf_il[2 * (i - 1)] = 2 * theta * cos(psi);
            rf_il[2 * (i - 1) + 1] = 2 * theta * sin(psi);
            cs[0] = make_double2(c, 0);
            cs[1] = s;
        }
        __syncthreads();
        const double c = cs[0].x;
        const double2 s = cs[1], ms = make_double2(-s.x, s.y);                                                      // -conj(s)
        const int nxt = cur ^ 1;
        for (int k = threadIdx.x; k < i; k += blockDim.x) {
            const double2 ak = A[cur][k], bk = B[cur][k];
            const double2 sb = cmul2(s, bk), msa = cmul2(ms, ak);
            const double2 acn = make_double2(c * ak.x + sb.x, c * ak.y + sb.y);
            const double2 bcn = make_double2(msa.x + c * bk.x, msa.y + c * bk.y);
            if (k >= 1) A[nxt][k - 1] = acn;              // ac = acn(2:i)
            if (k < i - 1) B[nxt][k] = bcn;               // bc = bcn(1:i-1)
        }
        __syncthreads();
        cur = nxt;
    }
}

// Forward simulation of an RF pulse over off-resonance: Cayley-Klein parameters per position.
//   mode 0: rf_tools/abrm.m:40-57 -- one rotation about (Re rf, Im rf, x g_m) per sample
//   mode 1: the hard-pulse model the inverse SLR transform inverts exactly -- free precession by x g_m on beta, then
//           the hard pulse of the sample
// one thread per position, the pulse staged through LDS; g may be null (2 pi / n per sample).
__global__ __launch_bounds__(256) void k_abr(const double* __restrict__ rf_il, const double* __restrict__ g, int n,
                                             const double* __restrict__ x, int nx, int mode, double* __restrict__ a_il,
                                             double* __restrict__ b_il) {
    __shared__ double2 srf[256];
    __shared__ double sg[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double xv = i < nx ? x[i] : 0.0;
    double2 a = make_double2(1, 0), b = make_double2(0, 0);
    const double g0 = 2.0 * M_PI / n;
    for (int m0 = 0; m0 < n; m0 += 256) {
        __syncthreads();
        const int mm = m0 + threadIdx.x;
        if (mm < n) { srf[threadIdx.x] = make_double2(rf_il[2 * mm], rf_il[2 * mm + 1]); sg[threadIdx.x] = g ? g[mm] : g0; }
        __syncthreads();
        const int cnt = min(256, n - m0);
        for (int q = 0; q < cnt; ++q) {
            const double2 r = srf[q];
            const double om = xv * sg[q];
            double2 av, bv;                              // step: a' = av a - conj(bv) b ; b' = bv a + conj(av) b
            if (mode == 0) {
                const double phi = sqrt(r.x * r.x + r.y * r.y + om * om);
                double sn, cs;
                sincos(0.5 * phi, &sn, &cs);
                const double inv = phi > 0 ? sn / phi : 0.0;
                av = make_double2(cs, -om * inv);
                bv = make_double2(r.y * inv, -r.x * inv);                  // -i (n1 + i n2) sin
                const double2 an = make_double2(av.x * a.x - av.y * a.y - (bv.x * b.x + bv.y * b.y),
                                                av.x * a.y + av.y * a.x - (bv.x * b.y - bv.y * b.x));
                const double2 bn = make_double2(bv.x * a.x - bv.y * a.y + (av.x * b.x + av.y * b.y),
                                                bv.x * a.y + bv.y * a.x + (av.x * b.y - av.y * b.x));
                a = an; b = bn;
            } else {
                const double th = hypot(r.x, r.y);
                double sn, cs, sz, cz;
                sincos(0.5 * th, &sn, &cs);
                sincos(-om, &sz, &cz);                                   // z^-1
                const double2 zb = make_double2(cz * b.x - sz * b.y, cz * b.y + sz * b.x);
                const double inv = th > 0 ? sn / th : 0.0;
                const double2 S = make_double2(-r.y * inv, r.x * inv);    // i e^{i arg rf} sin(th/2)
                const double2 an = make_double2(cs * a.x - (S.x * zb.x + S.y * zb.y), cs * a.y - (S.x * zb.y - S.y * zb.x));
                const double2 bn = make_double2(S.x * a.x - S.y * a.y + cs * zb.x, S.x * a.y + S.y * a.x + cs * zb.y);
                a = an; b = bn;
            }
        }
    }
    if (i < nx) { a_il[2 * i] = a.x; a_il[2 * i + 1] = a.y; b_il[2 * i] = b.x; b_il[2 * i + 1] = b.y; }
}
void slr_abr_launch(const double* rf_il, const double* g, int n, const double* x, int nx, int mode, double* a_il, double* b_il,
                    hipStream_t st) {
    hipLaunchKernelGGL(k_abr, dim3(cdiv(nx, 256)), dim3(256), 0, st, rf_il, g, n, x, nx, mode, a_il, b_il);
}

// ------------------------------------------------------------------------------------------------
// Bloch-equation simulation with relaxation: bloch_simulation/blochC.c calcrotmat (:171-236),
// blochsim (:283-418), blochsimfz (:422-512).  One thread per (off-resonance, position) pair -- the reference's
// two outer loops -- and the time loop inside; the per-sample quantities (rotation components of the pulse,
// gradient, interval, E1, E2) are staged through LDS 256 samples at a time.
//   mode bit 0: steady state (propagate A, B with M' = A M + B, then M = (I - A)^-1 B), bit 1: record every sample.
// step[t] = (rotx, roty, gx, gy, gz (each * gamma * dt), dt * TWOPI, e1, e2); pos3 = (x, y, z) per position.
struct Rot3 {
    double m[9];           // column-major like the reference: m[i + 3 j]
};
__device__ __forceinline__ void bloch_rotmat(double nx, double ny, double nz, Rot3& R) {
    const double phi = sqrt(nx * nx + ny * ny + nz * nz);
    if (phi == 0.0) {
        R.m[0] = 1; R.m[1] = 0; R.m[2] = 0; R.m[3] = 0; R.m[4] = 1; R.m[5] = 0; R.m[6] = 0; R.m[7] = 0; R.m[8] = 1;
        return;
    }
    double sn, cp;
    sincos(0.5 * phi, &sn, &cp);
    const double sp = sn / phi;
    const double ar = cp, ai = -nz * sp, br = ny * sp, bi = -nx * sp;
    R.m[0] = ar * ar - ai * ai - br * br + bi * bi;
    R.m[1] = -2 * ar * ai - 2 * br * bi;
    R.m[2] = -2 * ar * br + 2 * ai * bi;
    R.m[3] = 2 * ar * ai - 2 * br * bi;
    R.m[4] = ar * ar - ai * ai + br * br - bi * bi;
    R.m[5] = -2 * ai * br - 2 * ar * bi;
    R.m[6] = 2 * ar * br + 2 * ai * bi;
    R.m[7] = 2 * ar * bi - 2 * ai * br;
    R.m[8] = ar * ar + ai * ai - br * br - bi * bi;
}
__device__ __forceinline__ void rot_vec(const Rot3& R, const double v[3], double o[3]) {
    o[0] = R.m[0] * v[0] + R.m[3] * v[1] + R.m[6] * v[2];
    o[1] = R.m[1] * v[0] + R.m[4] * v[1] + R.m[7] * v[2];
    o[2] = R.m[2] * v[0] + R.m[5] * v[1] + R.m[8] * v[2];
}
constexpr int BLOCH_CH = 256;
__global__ __launch_bounds__(256) void k_bloch(const double* __restrict__ step, int ntime, const double* __restrict__ df, int nf,
                                               const double* __restrict__ pos3, int npos, int mode, double* __restrict__ mx,
                                               double* __restrict__ my, double* __restrict__ mz) {
    __shared__ double sst[BLOCH_CH][8];
    const long pair = (long)blockIdx.x * 256 + threadIdx.x, npair = (long)nf * npos;
    const bool live = pair < npair;
    const int fi = live ? int(pair / npos) : 0, pi = live ? int(pair - (long)fi * npos) : 0;
    const double dfv = df[fi], px = pos3[3 * pi], py = pos3[3 * pi + 1], pz = pos3[3 * pi + 2];
    const int ntout = (mode & 2) ? ntime : 1;
    const long o0 = pair * ntout;
    double m[3] = {0, 0, 1};
    if (live) { m[0] = mx[o0]; m[1] = my[o0]; m[2] = mz[o0]; }          // initial magnetisation sits in the output (:826-841)
    for (int pass = (mode & 1) ? 0 : 1; pass < 2; ++pass) {
        if (pass == 1 && mode == 1) break;                               // steady state only
        double A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, B[3] = {0, 0, 0};
        for (int t0 = 0; t0 < ntime; t0 += BLOCH_CH) {
            __syncthreads();
            const int cnt = min(BLOCH_CH, ntime - t0);
            for (int e = threadIdx.x; e < cnt * 8; e += 256) sst[e >> 3][e & 7] = step[(long)t0 * 8 + e];
            __syncthreads();
            if (!live) continue;
            for (int q = 0; q < cnt; ++q) {
                const double* s = sst[q];
                const double rotz = -((s[2] * px + s[3] * py + s[4] * pz) + dfv * s[5]);
                Rot3 R;
                bloch_rotmat(s[0], s[1], rotz, R);
                const double e1 = s[6], e2 = s[7];
                if (pass == 0) {
                    double c[3], o[3];
                    for (int j = 0; j < 3; ++j) {                        // A <- D R A, column by column
                        c[0] = A[3 * j]; c[1] = A[3 * j + 1]; c[2] = A[3 * j + 2];
                        rot_vec(R, c, o);
                        A[3 * j] = e2 * o[0]; A[3 * j + 1] = e2 * o[1]; A[3 * j + 2] = e1 * o[2];
                    }
                    rot_vec(R, B, o);
                    B[0] = e2 * o[0]; B[1] = e2 * o[1]; B[2] = e1 * o[2] + (1 - e1);
                } else {
                    double o[3];
                    rot_vec(R, m, o);
                    m[0] = e2 * o[0]; m[1] = e2 * o[1]; m[2] = e1 * o[2] + (1 - e1);
                    if (mode & 2) { mx[o0 + t0 + q] = m[0]; my[o0 + t0 + q] = m[1]; mz[o0 + t0 + q] = m[2]; }
                }
            }
        }
        if (pass == 0 && live) {
            // M = (I - A)^-1 B by the adjugate (the reference's invmat)
            double K[9];
            for (int e = 0; e < 9; ++e) K[e] = ((e == 0 || e == 4 || e == 8) ? 1.0 : 0.0) - A[e];
            const double c00 = K[4] * K[8] - K[7] * K[5], c01 = K[7] * K[2] - K[1] * K[8], c02 = K[1] * K[5] - K[4] * K[2];
            const double det = K[0] * c00 + K[3] * c01 + K[6] * c02;
            const double inv[9] = {c00 / det, c01 / det, c02 / det,
                                   (K[6] * K[5] - K[3] * K[8]) / det, (K[0] * K[8] - K[6] * K[2]) / det, (K[3] * K[2] - K[0] * K[5]) / det,
                                   (K[3] * K[7] - K[6] * K[4]) / det, (K[6] * K[1] - K[0] * K[7]) / det, (K[0] * K[4] - K[3] * K[1]) / det};
            for (int i = 0; i < 3; ++i) m[i] = inv[i] * B[0] + inv[3 + i] * B[1] + inv[6 + i] * B[2];
        }
    }
    if (live && !(mode & 2)) { mx[o0] = m[0]; my[o0] = m[1]; mz[o0] = m[2]; }
}
void bloch_launch(const double* step, int ntime, const double* df, int nf, const double* pos3, int npos, int mode, double* mx,
                  double* my, double* mz, hipStream_t st) {
    hipLaunchKernelGGL(k_bloch, dim3(cdiv((long)nf * npos, 256)), dim3(256), 0, st, step, ntime, df, nf, pos3, npos, mode, mx, my, mz);
}

// work: 3 * 8n double2.  a_il / rf_il: device arrays of 2n doubles (interleaved).
void slr_b2a_launch(const double* b_re, const double* b_im, int n, double* work, double* a_il, hipStream_t st) {
    const int N = 8 * n;
    double2* B0 = reinterpret_cast<double2*>(work);
    double2* B1 = B0 + N;
    double2* B2 = B1 + N;
    const dim3 g(cdiv(N, 256)), b(256);
    hipLaunchKernelGGL(k_slr_pad, g, b, 0, st, b_re, b_im, n, N, B0);
    hipLaunchKernelGGL(k_dft_any, g, b, 0, st, B0, B1, N, -1, 1.0);              // bf = fft(bcp)
    hipLaunchKernelGGL(k_slr_logmag, dim3(1), dim3(1024), 0, st, B1, N, B2);     // xl
    hipLaunchKernelGGL(k_dft_any, g, b, 0, st, B2, B0, N, -1, 1.0);              // xlf = fft(xl)
    hipLaunchKernelGGL(k_slr_window, g, b, 0, st, B0, N);
    hipLaunchKernelGGL(k_dft_any, g, b, 0, st, B0, B2, N, +1, 1.0 / N);          // xlaf = ifft(xlfp)
    hipLaunchKernelGGL(k_slr_exp, g, b, 0, st, B2, N);                           // afa
    hipLaunchKernelGGL(k_dft_any, g, b, 0, st, B2, B1, N, -1, 1.0 / N);          // aca = fft(afa) / blp
    hipLaunchKernelGGL(k_slr_out, dim3(cdiv(n, 256)), b, 0, st, B1, n, a_il);
}
void slr_ab2rf_launch(const double* a_il, const double* b_il, int n, double* rf_il, hipStream_t st) {
    if (n > SLR_MAXN) throw HipError("ab2rf: more than 2048 taps");
    hipLaunchKernelGGL(k_ab2rf, dim3(1), dim3(1024), 0, st, a_il, b_il, n, rf_il);
}

}  // namespace mbfir
