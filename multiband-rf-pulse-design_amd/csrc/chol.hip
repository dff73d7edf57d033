// K4: dense KKT factorisation.  H = L L' (blocked right-looking Cholesky, 64-wide panels) and
// M = L^-1 (diagonal 64 x 64 inverses from the panel kernel, off-diagonal blocks by recursive
// doubling  X21 = -inv22 * L21 * inv11 -- log2(N/64) levels of batched MFMA GEMMs), so that every
// later solve is two triangular GEMVs (x = M'(M b)) instead of two latency-bound substitutions.
// All fp64; products on v_mfma_f64_16x16x4_f64.
#include "dev_common.h"

namespace mbfir {

constexpr int CB = 64;       // panel width
constexpr int CLD = 66;      // padded LDS leading dimension

// ---- 64x64 MFMA helper: each of the 4 waves owns a 32x32 quadrant (2x2 MFMA blocks) -----------
// acc[a][b] += sum_k  Aop[i][k] * Bop[k][j]   with  Aop[i][k] = As[i][k]  (As row-major [64][CLD])
// and Bop[k][j] = transB ? Bs[j][k] : Bs[k][j].
template <bool TRANSB>
__device__ __forceinline__ void mma64(const double (*As)[CLD], const double (*Bs)[CLD], int kbeg, int kend,
                                      v4d acc[2][2]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    for (int k0 = kbeg; k0 < kend; k0 += 4) {
        const int k = k0 + (lane >> 4);
        double af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = As[wi * 32 + a * 16 + (lane & 15)][k];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = wj * 32 + b * 16 + (lane & 15);
            bf[b] = TRANSB ? Bs[j][k] : Bs[k][j];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
}

// Visit the (i, j, value) triples of a wave's accumulator quadrant.
template <class F>
__device__ __forceinline__ void acc_foreach(v4d acc[2][2], F f) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wi = wv >> 1, wj = wv & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                f(wi * 32 + a * 16 + (lane >> 4) + 4 * r, wj * 32 + b * 16 + (lane & 15), acc[a][b][r]);
}

__device__ __forceinline__ void load_block(double (*S)[CLD], const double* __restrict__ src, int ld) {
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) S[e >> 6][e & 63] = src[(long)(e >> 6) * ld + (e & 63)];
}

// Panel step k: every block factorises the 64x64 diagonal block in LDS (redundantly -- 87 kflop);
// block 0 publishes L_kk and its inverse (a diagonal block of M); block b>0 forms
// L_ik = A_ik L_kk^-T by forward substitution (multiplying by the explicit inverse is not
// backward stable and breaks the factorisation on the near-singular late IPM iterates).
// Pivot rule: a pivot that is not above pivtol * H_jj is rounding noise; it is replaced by 1e128,
// which removes that column from the factorisation (flag counts the replacements).
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ H, double* __restrict__ M, int np,
                                                    int k, const double* __restrict__ d0, double pivtol,
                                                    double* __restrict__ Dfac, int* __restrict__ flag) {
    __shared__ double S[CB][CLD];
    __shared__ double X[CB][CLD];
    const int tid = threadIdx.x;
    const long kk = (long)k * CB;
    for (int e = tid; e < CB * CB; e += 256) {
        int i = e >> 6, j = e & 63;
        S[i][j] = j <= i ? H[(kk + i) * np + kk + j] : 0.0;
    }
    // unblocked right-looking Cholesky of S
    for (int j = 0; j < CB; ++j) {
        __syncthreads();
        double p = S[j][j];
        if (!(p > pivtol * d0[kk + j])) {
            if (tid == 0 && blockIdx.x == 0) atomicAdd(flag, 1);
            p = 1e128;
        }
        const double r = sqrt(p), rinv = 1.0 / r;
        __syncthreads();
        if (tid == 0) S[j][j] = r;
        for (int i = j + 1 + tid; i < CB; i += 256) S[i][j] *= rinv;
        __syncthreads();
        const int nrem = CB - 1 - j;
        for (int e = tid; e < nrem * nrem; e += 256) {
            int ii = e / nrem, cc = e - ii * nrem;
            if (cc <= ii) S[j + 1 + ii][j + 1 + cc] -= S[j + 1 + ii][j] * S[j + 1 + cc][j];
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        // X = S^-1 (lower): thread c solves S x = e_c by forward substitution
        for (int e = tid; e < CB * CB; e += 256) X[e >> 6][e & 63] = 0.0;
        __syncthreads();
        if (tid < CB) {
            const int c = tid;
            for (int i = c; i < CB; ++i) {
                double sum = (i == c) ? 1.0 : 0.0;
                for (int j = c; j < i; ++j) sum -= S[i][j] * X[j][c];
                X[i][c] = sum / S[i][i];
            }
        }
        __syncthreads();
        for (int e = tid; e < CB * CB; e += 256) {
            int i = e >> 6, j = e & 63;
            // L_kk goes to a side buffer: the other blocks of this launch may still be reading A_kk
            // from H; k_finish_L copies it into H after the last panel
            Dfac[(kk + i) * CB + j] = S[i][j];
            M[(kk + i) * np + kk + j] = X[i][j];
        }
        return;
    }
    // rows of A_ik: 4 threads per row (same wave), x_j = (a_j - sum_{t<j} x_t S[j][t]) / S[j][j]
    const long ii = (long)(k + blockIdx.x) * CB;
    load_block(X, H + ii * np + kk, np);
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;
    for (int j = 0; j < CB; ++j) {
        double part = 0;
        for (int t = q; t < j; t += 4) part += X[r][t] * S[j][t];
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        if (q == 0) X[r][j] = (X[r][j] - part) / S[j][j];
        __syncthreads();                      // the row's 4 lanes (and the compiler) see the new x_j
    }
    double* dst = H + ii * np + kk;
    for (int e = tid; e < CB * CB; e += 256) dst[(long)(e >> 6) * np + (e & 63)] = X[e >> 6][e & 63];
}

// Trailing update after panel k:  A_ij -= L_ik L_jk'   for k < j <= i.
__global__ __launch_bounds__(256) void k_chol_trail(double* __restrict__ H, int np, int k) {
    __shared__ double P[CB][CLD];
    __shared__ double Q[CB][CLD];
    // decode lower-triangular tile index
    int t = blockIdx.x;
    int ti = int((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while ((long)ti * (ti + 1) / 2 > t) --ti;
    int tj = t - ti * (ti + 1) / 2;
    const long kk = (long)k * CB, i0 = (long)(k + 1 + ti) * CB, j0 = (long)(k + 1 + tj) * CB;
    load_block(P, H + i0 * np + kk, np);
    load_block(Q, H + j0 * np + kk, np);
    __syncthreads();
    v4d acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    mma64<true>(P, Q, 0, CB, acc);
    double* dst = H + i0 * np + j0;
    acc_foreach(acc, [&](int i, int j, double v) { dst[(long)i * np + j] -= v; });
}

// Zero the strict upper triangle of H and drop in the diagonal-block factors (H then holds a clean L).
__global__ void k_finish_L(double* __restrict__ H, int np, const double* __restrict__ Dfac) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)np * np) return;
    long i = e / np, j = e - i * np;
    if (j > i) H[e] = 0.0;
    else if ((i / CB) == (j / CB)) H[e] = Dfac[i * CB + (j % CB)];
}

// Batched GEMM used by the inverse assembly.  For pair z at level `u` (unit size in elements):
//   o = z*2u, b1 = u, b2 = min(u, np-o-u)   (skipped when o+u >= np)
//   stage 0:  W[o+u.., o..]  =  L21 * inv11          (L21 = H[o+u.., o..], inv11 = M[o.., o..])
//   stage 1:  M[o+u.., o..]  = -inv22 * W[o+u.., o..] (inv22 = M[o+u.., o+u..])
// inv11 / inv22 are lower triangular, so the k range is clipped to the non-zero tiles.
__global__ __launch_bounds__(256) void k_inv_gemm(const double* __restrict__ H, double* __restrict__ M,
                                                  double* __restrict__ W, int np, int u, int stage) {
    __shared__ double As[CB][CLD];
    __shared__ double Bs[CB][CLD];
    const long o = (long)blockIdx.z * 2 * u;
    if (o + u >= np) return;
    const int b1 = u, b2 = int(np - o - u < u ? np - o - u : u);
    const int tj = blockIdx.x, ti = blockIdx.y;      // tile (ti,tj) of the b2 x b1 result
    if (ti * CB >= b2 || tj * CB >= b1) return;
    const double *A, *B;
    double* C;
    int kbeg, kend;                                   // in tiles
    if (stage == 0) {
        A = H + (o + u) * np + o;                     // L21 (b2 x b1)
        B = M + o * np + o;                           // inv11 (b1 x b1), lower: B[k][j]!=0 for k>=j
        C = W + (o + u) * np + o;
        kbeg = tj; kend = b1 / CB;
    } else {
        A = M + (o + u) * np + (o + u);               // inv22 (b2 x b2), lower: A[i][k]!=0 for k<=i
        B = W + (o + u) * np + o;                     // T (b2 x b1)
        C = M + (o + u) * np + o;
        kbeg = 0; kend = ti + 1;
    }
    v4d acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    for (int kt = kbeg; kt < kend; ++kt) {
        __syncthreads();
        load_block(As, A + (long)ti * CB * np + (long)kt * CB, np);
        load_block(Bs, B + (long)kt * CB * np + (long)tj * CB, np);
        __syncthreads();
        mma64<false>(As, Bs, 0, CB, acc);
    }
    double* dst = C + (long)ti * CB * np + (long)tj * CB;
    const double sgn = stage == 0 ? 1.0 : -1.0;
    acc_foreach(acc, [&](int i, int j, double v) { dst[(long)i * np + j] = sgn * v; });
}

// Lower-triangular tile products for the Newton correction of M = L^-1:
//   mode 0:  C = I - A B     (E = I - L M)
//   mode 1:  C = A + A B     (M_new = M + M E)
// A, B, C lower triangular np x np; tile (ti,tj), ti >= tj, sums k = tj..ti.
__global__ __launch_bounds__(256) void k_tri_gemm(const double* __restrict__ A, const double* __restrict__ B,
                                                  double* __restrict__ C, int np, int mode) {
    __shared__ double As[CB][CLD];
    __shared__ double Bs[CB][CLD];
    const int tj = blockIdx.x, ti = blockIdx.y;
    if (tj > ti) return;
    v4d acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    for (int kt = tj; kt <= ti; ++kt) {
        __syncthreads();
        load_block(As, A + (long)ti * CB * np + (long)kt * CB, np);
        load_block(Bs, B + (long)kt * CB * np + (long)tj * CB, np);
        __syncthreads();
        mma64<false>(As, Bs, 0, CB, acc);
    }
    const long base = (long)ti * CB * np + (long)tj * CB;
    acc_foreach(acc, [&](int i, int j, double v) {
        const long o = base + (long)i * np + j;
        if (mode == 0) C[o] = ((ti == tj && i == j) ? 1.0 : 0.0) - v;
        else C[o] = A[o] + v;
    });
}

__global__ void k_transpose(const double* __restrict__ M, double* __restrict__ Mt, int np) {
    __shared__ double tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) tile[r][threadIdx.x] = M[(long)(by + r) * np + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) Mt[(long)(bx + r) * np + by + threadIdx.x] = tile[threadIdx.x][r];
}

__global__ void k_diag_copy(const double* __restrict__ H, int np, double* __restrict__ d0) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < np) d0[j] = H[(long)j * np + j];
}

void chol_inv_launch(double* H, double* M, double* Mt, double* W1, int np, int* flag, hipStream_t st,
                     double* Lcopy) {
    const int nblk = np / CB;
    const double pivtol = 1e-13;                 // oracle/conic_ipm.py PIVTOL
    // W1 layout: [0, np^2) GEMM workspace | np doubles: original diagonal | 64 np doubles: L_kk blocks
    double* d0 = W1 + (size_t)np * np;
    double* Dfac = d0 + np;
    hipMemsetAsync(M, 0, sizeof(double) * np * np, st);
    hipMemsetAsync(flag, 0, sizeof(int), st);
    hipLaunchKernelGGL(k_diag_copy, dim3(cdiv(np, 256)), dim3(256), 0, st, H, np, d0);
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(k_chol_panel, dim3(nblk - k), dim3(256), 0, st, H, M, np, k, d0, pivtol, Dfac, flag);
        int nrem = nblk - k - 1;
        if (nrem > 0)
            hipLaunchKernelGGL(k_chol_trail, dim3(nrem * (nrem + 1) / 2), dim3(256), 0, st, H, np, k);
    }
    hipLaunchKernelGGL(k_finish_L, dim3(cdiv((long)np * np, 256)), dim3(256), 0, st, H, np, Dfac);
    for (int u = CB; u < np; u *= 2) {
        int pairs = cdiv(np, 2 * u);
        dim3 grid(u / CB, u / CB, pairs);
        hipLaunchKernelGGL(k_inv_gemm, grid, dim3(256), 0, st, H, M, W1, np, u, 0);
        hipLaunchKernelGGL(k_inv_gemm, grid, dim3(256), 0, st, H, M, W1, np, u, 1);
    }
    if (Lcopy) hipMemcpyAsync(Lcopy, H, sizeof(double) * np * np, hipMemcpyDeviceToDevice, st);
    // One Newton step  M <- M + M (I - L M): the recursive-doubling products lose ~30x accuracy
    // against a substitution-based inverse on the ill-conditioned late IPM factors; the correction
    // restores it (measured: solve residual 6.5e-4 -> 2.5e-5 at cond(H) = 3e9, same as LAPACK trtri).
    hipMemsetAsync(W1, 0, sizeof(double) * np * np, st);
    hipLaunchKernelGGL(k_tri_gemm, dim3(nblk, nblk), dim3(256), 0, st, H, M, W1, np, 0);     // E = I - L M
    hipMemsetAsync(H, 0, sizeof(double) * np * np, st);
    hipLaunchKernelGGL(k_tri_gemm, dim3(nblk, nblk), dim3(256), 0, st, M, W1, H, np, 1);     // H <- M + M E
    hipMemcpyAsync(M, H, sizeof(double) * np * np, hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(k_transpose, dim3(np / 32, np / 32), dim3(32, 8), 0, st, M, Mt, np);
}

// y[v][i] = sum_j T[i][j] b[v][j] over the stored triangle; one wave per row, 16-byte loads.
template <int NV>
__global__ __launch_bounds__(256) void k_trigemv(const double* __restrict__ T, int np, int upper,
                                                 const double* __restrict__ b, double* __restrict__ y, int ldv) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= np) return;
    int jlo = upper ? (i & ~1) : 0;
    int jhi = upper ? np : ((i + 2) & ~1);            // exclusive, even
    const double* row = T + (long)i * np;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0;
    for (int j = jlo + 2 * lane; j < jhi; j += 128) {
        double2 t = *reinterpret_cast<const double2*>(row + j);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double2 bb = *reinterpret_cast<const double2*>(b + (long)v * ldv + j);
            acc[v] += t.x * bb.x + t.y * bb.y;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = wave_sum(acc[v]);
        if (lane == 0) y[(long)v * ldv + i] = s;
    }
}

void trigemv_launch(const double* T, int np, int upper, const double* b, double* y, int nv, int ldv,
                    hipStream_t st) {
    dim3 grid(cdiv(np, 4));
    if (nv == 1) hipLaunchKernelGGL(k_trigemv<1>, grid, dim3(256), 0, st, T, np, upper, b, y, ldv);
    else if (nv == 2) hipLaunchKernelGGL(k_trigemv<2>, grid, dim3(256), 0, st, T, np, upper, b, y, ldv);
    else throw HipError("trigemv_launch: nv must be 1 or 2");
}

}  // namespace mbfir
