// K4: dense KKT factorisation.  H = L L' (blocked right-looking Cholesky, 64-wide panels) with
// M = L^-1 carried along by forward substitution on the identity, so that every later solve is
// two triangular GEMVs (x = M'(M b)) instead of two latency-bound substitutions.
// All fp64; tile products on v_mfma_f64_16x16x4_f64.
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

// =================================================================================================
// Right-looking blocked Cholesky that carries M = L^-1 along (two launches per 64-wide panel):
//
//   step A_k : (a) panel blocks -- every block factorises A_kk in LDS (redundantly, 87 kflop);
//                  block 0 publishes L_kk (side buffer Dfac), block b>0 solves  L_ik L_kk' = A_ik
//                  by forward substitution (NOT by multiplying with inv(L_kk): that is not backward
//                  stable and breaks the factorisation on the near-singular late IPM iterates);
//              (b) R-update tiles of the previous panel:  M_ij -= L_i,k-1 M_k-1,j   (i >= k, j < k)
//   step B_k : (a) trailing tiles  A_ij -= L_ik L_jk'   (k < j <= i);
//              (b) row block k of the inverse:  M_kj = L_kk^-1 R_kj  (j <= k) by forward substitution
//                  over the 64 columns of each tile (R lives in the M buffer, initialised to I).
// M computed this way has the accuracy of a substitution-based inverse (measured: solve residual
// 3e-5 at cond(H)=3e9, same as LAPACK trtri; multiplying explicit 64x64 inverses gives 2e-3).
// Pivot rule: a pivot not above pivtol * H_jj is rounding noise and is replaced by H_jj itself; by
// Cauchy-Schwarz the rest of that Schur-complement column is at noise level too, so the column is
// effectively decoupled and M'M stays a non-singular preconditioner (flag counts the replacements).
// =================================================================================================
typedef double v16d __attribute__((ext_vector_type(16)));   // register-resident 16-vector (an array would go to scratch)

// sum over the 4 lanes of a quad (DPP quad_perm, no LDS traffic)
__device__ __forceinline__ double quad_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo1 = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true), hi1 = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    v += __hiloint2double(hi1, lo1);                      // + lane ^ 1
    lo = __double2loint(v); hi = __double2hiint(v);
    lo1 = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi1 = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
    return v + __hiloint2double(hi1, lo1);                // + lane ^ 2
}

// Column permutation used for the LDS images below: column t is stored at position
// perm(t) = (t & 3) * 16 + (t >> 2), so the 16 columns {q, q+4, ...} that one lane needs are
// contiguous (wide LDS reads, no per-element address math).
__device__ __forceinline__ int cperm(int t) { return (t & 3) * 16 + (t >> 2); }
constexpr int SLD = 66;      // row stride of the permuted images (16-byte aligned rows)
constexpr int PANEL_LDS = 2 * CB * CLD;              // Sp | dsh | dinv | colbuf (panel) or two MFMA tiles

// NOTE on code shape (all measured on MI355X with tools/exp/chol_exp.hip):
//  * a dependent fp64 VALU op costs ~40 cycles, an LDS round trip ~130, so the per-pivot chain is
//    what matters; a read-modify-write loop over LDS serialises on that latency because the
//    compiler cannot prove the arrays disjoint;
//  * fully unrolling the 64 steps (straight-line code executed once, ~10k instructions) is
//    instruction-fetch bound and 2x slower still;
//  * so: operands live in registers with static indices, the outer loop over groups of 4 pivots
//    is rolled, the 4 steps inside are unrolled, and the one register that must be picked by the
//    loop counter is selected / put back once per group with wave-uniform compares.
__device__ __forceinline__ double sel16(const v16d& v, int idx) {
    double r = v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) r = (i == idx) ? v[i] : r;
    return r;
}
__device__ __forceinline__ void put16(v16d& v, int idx, double x) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (i == idx) ? x : v[i];
}

// Forward substitution  S x = a  for 64 right-hand sides, one per group of 4 adjacent lanes.
// Lane q (= lane & 3) holds a[t] on entry / x[t] on exit for t = 4 i + q in v[i].
// Sp is the column-permuted lower-triangular factor in LDS, dinv[j] = 1 / S[j][j].
// Per group of 4 unknowns: the contribution of all earlier groups ("hist") is formed for the 4 rows
// at once (64 independent FMAs), then the 4 in-group steps run the short chain
// term -> quad_sum -> (a - part) * dinv.
__device__ __forceinline__ void subst64(const double (*Sp)[SLD], const double* dinv, v16d& v) {
    const int q = threadIdx.x & 3;
#pragma unroll 1
    for (int jg = 0; jg < 16; ++jg) {
        const double acur = sel16(v, jg);                 // a_{4 jg + q}
        v16d vm;
#pragma unroll
        for (int i = 0; i < 16; ++i) vm[i] = (i < jg) ? v[i] : 0.0;
        double hist0 = 0, hist1 = 0, hist2 = 0, hist3 = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const double* srow = &Sp[4 * jg + jj][q * 16];
            double h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                h0 += vm[i] * srow[i]; h1 += vm[i + 1] * srow[i + 1];
                h2 += vm[i + 2] * srow[i + 2]; h3 += vm[i + 3] * srow[i + 3];
            }
            const double hh = (h0 + h1) + (h2 + h3);
            if (jj == 0) hist0 = hh; else if (jj == 1) hist1 = hh; else if (jj == 2) hist2 = hh; else hist3 = hh;
        }
        double vn = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * jg + jj;
            const double term = (q < jj) ? vn * Sp[j][q * 16 + jg] : 0.0;
            const double hj = jj == 0 ? hist0 : jj == 1 ? hist1 : jj == 2 ? hist2 : hist3;
            const double part = quad_sum(hj + term);
            if (q == jj) vn = (acur - part) * dinv[j];
        }
        put16(v, jg, vn);
    }
}

// Unblocked right-looking Cholesky of a 64x64 block held in registers: thread (tx = tid & 63,
// ty = tid >> 6) owns row tx, columns ty + 4 i in rv[i].  Column j is broadcast through the
// double-buffered, column-permuted LDS vector cb (one barrier per pivot).  Square-root free inner
// loop (S_ic -= S_ij S_cj / p_j, 1/p from v_rcp_f64 + one Newton step); the pivots go to piv[] and
// the caller applies L_ij = S_ij / sqrt(p_j) in one parallel pass.  Only the lower triangle of the
// result is meaningful.
__device__ __forceinline__ void potf2_regs(v16d& rv, double* colbuf, const double* dsh, double* piv,
                                           double pivtol, int* flag, bool count) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ptx = cperm(tx);
#pragma unroll 1
    for (int jg = 0; jg < 16; ++jg) {
        double cur = sel16(rv, jg);                       // element (tx, 4 jg + ty)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * jg + jj;
            double* cb = colbuf + (jj & 1) * CB;
            if (ty == jj) cb[ptx] = cur;                  // publish column j of the Schur complement
            __syncthreads();
            double p = cb[jj * 16 + jg];                  // cperm(j)
            const double dj = dsh[j];
            if (!(p > pivtol * dj)) {
                if (count && threadIdx.x == 0) atomicAdd(flag, 1);
                p = fmax(dj, 1e-300);
            }
            double rcp = __builtin_amdgcn_rcp(p);         // ~26 bits
            rcp = rcp * fma(-p, rcp, 2.0);                // 1/p to rounding
            const double a = cb[ptx] * rcp;
            const double* mine = cb + ty * 16;            // elements (c, j), c = ty + 4 i
            if (ty > jj) cur -= a * mine[jg];             // this thread's element of column group jg
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] -= ((i > jg) ? a : 0.0) * mine[i];
            if (ty == jj && tx == j) { cur = p; piv[j] = p; }
        }
        put16(rv, jg, cur);
    }
}

__device__ __forceinline__ void tile_decode(int t, int& ti, int& tj) {
    ti = int((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while ((long)ti * (ti + 1) / 2 > t) --ti;
    tj = t - ti * (ti + 1) / 2;
}

// C_tile (64x64 at dst) -= A_tile * B_tile (TRANSB: B_tile') through the MFMA helper
template <bool TRANSB>
__device__ __forceinline__ void tile_update(double* smem, const double* __restrict__ Ag, const double* __restrict__ Bg,
                                            double* __restrict__ dst, int np) {
    double(*P)[CLD] = reinterpret_cast<double(*)[CLD]>(smem);
    double(*Q)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + CB * CLD);
    load_block(P, Ag, np);
    load_block(Q, Bg, np);
    __syncthreads();
    v4d acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    mma64<TRANSB>(P, Q, 0, CB, acc);
    acc_foreach(acc, [&](int i, int j, double v) { dst[(long)i * np + j] -= v; });
}

__global__ __launch_bounds__(256) void k_chol_stepA(double* __restrict__ H, double* __restrict__ M, int np, int nblk,
                                                    int k, const double* __restrict__ d0, double pivtol,
                                                    double* __restrict__ Dfac, int* __restrict__ flag) {
    __shared__ __attribute__((aligned(16))) double smem[PANEL_LDS];
    const int tid = threadIdx.x;
    const int npanel = nblk - k;
    if ((int)blockIdx.x >= npanel) {
        // (b) R-update of panel k-1:  M_ij -= L_i,k-1 * M_k-1,j   for i >= k, j < k
        const int t = blockIdx.x - npanel;
        const int i = k + t / k, j = t % k;
        const long km = (long)(k - 1) * CB;
        tile_update<false>(smem, H + (long)i * CB * np + km, M + km * np + (long)j * CB,
                           M + (long)i * CB * np + (long)j * CB, np);
        return;
    }
    // (a) panel
    double(*Sp)[SLD] = reinterpret_cast<double(*)[SLD]>(smem);
    double* dsh = smem + CB * SLD;                        // original diagonal of this block (64)
    double* dinv = dsh + CB;                              // pivots, then 1 / L_jj (64)
    double* colbuf = dinv + CB;                           // 2 x 64
    const long kk = (long)k * CB;
    const int tx = tid & 63, ty = tid >> 6;
    v16d rv;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = ty + 4 * i;
        rv[i] = c <= tx ? H[(kk + tx) * np + kk + c] : 0.0;
    }
    if (tid < CB) dsh[tid] = d0[kk + tid];
    __syncthreads();
    potf2_regs(rv, colbuf, dsh, dinv, pivtol, flag, blockIdx.x == 0);
    __syncthreads();
    if (tid < CB) {                                       // 1 / sqrt(pivot): v_rsq_f64 + two Newton steps
        const double p = dinv[tid];
        double y = __builtin_amdgcn_rsq(p);
        y = y * (1.5 - 0.5 * p * y * y);
        y = y * (1.5 - 0.5 * p * y * y);
        dinv[tid] = y;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {                        // L_ij = S_ij / sqrt(p_j), image position cperm(c)
        const int c = ty + 4 * i;
        Sp[tx][ty * 16 + i] = c <= tx ? rv[i] * dinv[c] : 0.0;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        // L_kk goes to a side buffer: the other blocks of this launch may still be reading A_kk from H
        for (int e = tid; e < CB * CB; e += 256) Dfac[kk * CB + e] = Sp[e >> 6][cperm(e & 63)];
        return;
    }
    // rows of A_ik:  X L_kk' = A_ik  <=>  L_kk x_r' = a_r'   (rhs r = tid / 4, lane q holds t = 4 i + q)
    const long ii = (long)(k + blockIdx.x) * CB;
    const int r = tid >> 2, q = tid & 3;
    double* row = H + (ii + r) * np + kk;
    v16d v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = row[4 * i + q];
    subst64(Sp, dinv, v);
#pragma unroll
    for (int i = 0; i < 16; ++i) row[4 * i + q] = v[i];
}

__global__ __launch_bounds__(256) void k_chol_stepB(double* __restrict__ H, double* __restrict__ M, int np, int nblk,
                                                    int k, const double* __restrict__ Dfac) {
    __shared__ __attribute__((aligned(16))) double smem[PANEL_LDS];
    const int tid = threadIdx.x;
    const int nrem = nblk - k - 1;
    const int ntrail = nrem * (nrem + 1) / 2;
    const long kk = (long)k * CB;
    if ((int)blockIdx.x < ntrail) {
        // (a) trailing update  A_ij -= L_ik L_jk'
        int ti, tj;
        tile_decode(blockIdx.x, ti, tj);
        const long i0 = (long)(k + 1 + ti) * CB, j0 = (long)(k + 1 + tj) * CB;
        tile_update<true>(smem, H + i0 * np + kk, H + j0 * np + kk, H + i0 * np + j0, np);
        return;
    }
    // (b) M_kj = L_kk^-1 R_kj : column c = tid / 4 of the tile is one right-hand side
    const int j = blockIdx.x - ntrail;
    double(*Sp)[SLD] = reinterpret_cast<double(*)[SLD]>(smem);
    double* dinv = smem + CB * SLD;
    for (int e = tid; e < CB * CB; e += 256) Sp[e >> 6][cperm(e & 63)] = Dfac[kk * CB + e];
    __syncthreads();
    if (tid < CB) dinv[tid] = 1.0 / Sp[tid][cperm(tid)];
    __syncthreads();
    const int c = tid >> 2, q = tid & 3;
    double* col = M + kk * np + (long)j * CB + c;
    v16d v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = col[(long)(4 * i + q) * np];
    subst64(Sp, dinv, v);
#pragma unroll
    for (int i = 0; i < 16; ++i) col[(long)(4 * i + q) * np] = v[i];
}

__global__ void k_diag_copy(const double* __restrict__ H, int np, double* __restrict__ d0, double* __restrict__ M) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < np) { d0[j] = H[(long)j * np + j]; M[(long)j * np + j] = 1.0; }      // R starts as the identity
}

// L (np x np, clean lower triangle) from the factored H and the diagonal-block side buffer
__global__ void k_extract_L(const double* __restrict__ H, int np, const double* __restrict__ Dfac, double* __restrict__ Lout) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)np * np) return;
    long i = e / np, j = e - i * np;
    double v = 0.0;
    if (j <= i) v = ((i / CB) == (j / CB)) ? Dfac[(i / CB) * CB * CB + (i % CB) * CB + (j % CB)] : H[e];
    Lout[e] = v;
}

__global__ void k_transpose(const double* __restrict__ M, double* __restrict__ Mt, int np) {
    __shared__ double tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) tile[r][threadIdx.x] = M[(long)(by + r) * np + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) Mt[(long)(bx + r) * np + by + threadIdx.x] = tile[threadIdx.x][r];
}

void chol_inv_launch(double* H, double* M, double* Mt, double* W1, int np, int* flag, hipStream_t st,
                     double* Lcopy) {
    const int nblk = np / CB;
    const double pivtol = 1e-13;                 // oracle/conic_ipm.py PIVTOL
    // W1 layout: np doubles: original diagonal | 64 np doubles: L_kk blocks
    double* d0 = W1;
    double* Dfac = W1 + np;
    hipMemsetAsync(M, 0, sizeof(double) * np * np, st);
    hipMemsetAsync(flag, 0, sizeof(int), st);
    hipLaunchKernelGGL(k_diag_copy, dim3(cdiv(np, 256)), dim3(256), 0, st, H, np, d0, M);
    for (int k = 0; k < nblk; ++k) {
        const int npanel = nblk - k, nrem = nblk - k - 1;
        hipLaunchKernelGGL(k_chol_stepA, dim3(npanel + npanel * k), dim3(256), 0, st, H, M, np, nblk, k, d0, pivtol, Dfac, flag);
        hipLaunchKernelGGL(k_chol_stepB, dim3(nrem * (nrem + 1) / 2 + k + 1), dim3(256), 0, st, H, M, np, nblk, k, Dfac);
    }
    if (Lcopy) hipLaunchKernelGGL(k_extract_L, dim3(cdiv((long)np * np, 256)), dim3(256), 0, st, H, np, Dfac, Lcopy);
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
