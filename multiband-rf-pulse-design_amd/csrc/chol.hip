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
constexpr int CLDP = 65;     // odd LDS stride: lanes = rows is conflict free

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
constexpr int PANEL_LDS = 2 * CB * SLD + 4 * CB;     // Sp | Xs | dsh | dinv | colbuf(2 x 64)

// NOTE on code shape: these loops are deliberately ROLLED with LDS-resident data.  A fully
// unrolled register-resident version (64 steps of straight-line code, ~10k instructions executed
// once) ran 4x slower: it is instruction-fetch bound.

// Forward substitution  S x = a  for 64 right-hand sides, one per group of 4 adjacent lanes
// (rhs index r = tid >> 2).  Xs[r][q*16 + i] holds a[t] on entry and x[t] on exit for t = 4 i + q;
// lane q only ever touches its own 16-element segment, so the loop needs no barrier.
__device__ __forceinline__ void subst64(const double (*Sp)[SLD], const double* dinv, double (*Xs)[SLD]) {
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    double* xrow = &Xs[r][q * 16];
    for (int j = 0; j < CB; ++j) {
        const double* srow = &Sp[j][q * 16];
        const int ni = (j - q + 3) >> 2;                  // number of t = 4 i + q below j
        double p0 = 0, p1 = 0, p2 = 0, p3 = 0;            // four chains hide the FMA latency
        int i = 0;
        for (; i + 4 <= ni; i += 4) {
            p0 += xrow[i] * srow[i]; p1 += xrow[i + 1] * srow[i + 1];
            p2 += xrow[i + 2] * srow[i + 2]; p3 += xrow[i + 3] * srow[i + 3];
        }
        for (; i < ni; ++i) p0 += xrow[i] * srow[i];
        double part = quad_sum((p0 + p1) + (p2 + p3));
        if (q == (j & 3)) xrow[j >> 2] = (xrow[j >> 2] - part) * dinv[j];
    }
}

// Unblocked right-looking Cholesky of the 64x64 block in Sp (column-permuted, LDS).  Thread
// (tx = tid & 63, ty = tid >> 6) owns the segment Sp[tx][ty*16 .. +16) = columns ty + 4 i of row tx.
// Column j is broadcast through the double-buffered vector cb, one barrier per step.
// A dependent fp64 VALU op costs ~40 cycles on gfx950 (measured), so the pivot chain is kept as
// short as possible: the loop is square-root free (S_ic -= S_ij S_cj / p_j, with 1/p from v_rcp_f64
// + one Newton step) and the column scaling L_ij = S_ij / sqrt(p_j) is applied in one parallel pass
// afterwards.  Only the lower triangle of the result is meaningful.
__device__ __forceinline__ void potf2_lds(double (*Sp)[SLD], double* colbuf, const double* dsh, double* dinv,
                                          double pivtol, int* flag, bool count) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ptx = cperm(tx);
    double* seg = &Sp[tx][ty * 16];
    double* piv = dinv;                                   // pivots first, 1/sqrt(pivot) after the loop
    for (int j = 0; j < CB; ++j) {
        double* cb = colbuf + (j & 1) * CB;
        const int jg = j >> 2, jq = j & 3;
        if (ty == jq) cb[ptx] = seg[jg];                  // element (tx, j) of the Schur complement
        __syncthreads();
        double p = cb[cperm(j)];
        const double dj = dsh[j];
        if (!(p > pivtol * dj)) {
            if (count && threadIdx.x == 0) atomicAdd(flag, 1);
            p = fmax(dj, 1e-300);
        }
        double rcp = __builtin_amdgcn_rcp(p);             // ~26 bits
        rcp = rcp * fma(-p, rcp, 2.0);                    // 1/p to rounding
        const double a = cb[ptx] * rcp;
        const double* mine = cb + ty * 16;                // elements (c, j), c = ty + 4 i
        const int i0 = ty > jq ? jg : jg + 1;             // columns c = ty + 4 i > j
        // Three phases (all loads, all FMAs, all stores): interleaving them serialises on LDS latency
        // because the compiler cannot prove seg and mine disjoint (measured 1290 vs ~300 cycles).
        // Finished column groups are skipped (wave-uniform tests).
        double rv[16], cv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (4 * g + 3 >= i0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { rv[4 * g + u] = seg[4 * g + u]; cv[4 * g + u] = mine[4 * g + u]; }
            }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (4 * g + 3 >= i0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) rv[4 * g + u] -= (4 * g + u >= i0 ? a : 0.0) * cv[4 * g + u];
            }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (4 * g + 3 >= i0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) seg[4 * g + u] = rv[4 * g + u];
            }
        if (ty == jq && tx == j) { seg[jg] = p; piv[j] = p; }
    }
    __syncthreads();
    if (threadIdx.x < CB) {
        const double p = piv[threadIdx.x];
        double y = __builtin_amdgcn_rsq(p);
        y = y * (1.5 - 0.5 * p * y * y);
        y = y * (1.5 - 0.5 * p * y * y);
        dinv[threadIdx.x] = y;                            // 1 / L_jj
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) seg[i] *= dinv[ty + 4 * i];    // L_ij = S_ij / sqrt(p_j)  (L_jj = p_j / sqrt(p_j))
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
    double(*Xs)[SLD] = reinterpret_cast<double(*)[SLD]>(smem + CB * SLD);
    double* dsh = smem + 2 * CB * SLD;                    // original diagonal of this block (64)
    double* dinv = dsh + CB;                              // 1 / L_jj (64)
    double* colbuf = dinv + CB;                           // 2 x 64
    const long kk = (long)k * CB;
    for (int e = tid; e < CB * CB; e += 256) {
        int i = e >> 6, j = e & 63;
        Sp[i][cperm(j)] = j <= i ? H[(kk + i) * np + kk + j] : 0.0;
    }
    if (tid < CB) dsh[tid] = d0[kk + tid];
    __syncthreads();
    potf2_lds(Sp, colbuf, dsh, dinv, pivtol, flag, blockIdx.x == 0);
    __syncthreads();
    if (blockIdx.x == 0) {
        // L_kk goes to a side buffer: the other blocks of this launch may still be reading A_kk from H
        for (int e = tid; e < CB * CB; e += 256) {
            int i = e >> 6, j = e & 63;
            Dfac[kk * CB + e] = j <= i ? Sp[i][cperm(j)] : 0.0;
        }
        return;
    }
    // rows of A_ik:  X L_kk' = A_ik  <=>  L_kk x_r' = a_r'   (rhs r = row r of the tile)
    const long ii = (long)(k + blockIdx.x) * CB;
    double* tile = H + ii * np + kk;
    for (int e = tid; e < CB * CB; e += 256) Xs[e >> 6][cperm(e & 63)] = tile[(long)(e >> 6) * np + (e & 63)];
    __syncthreads();
    subst64(Sp, dinv, Xs);
    __syncthreads();
    for (int e = tid; e < CB * CB; e += 256) tile[(long)(e >> 6) * np + (e & 63)] = Xs[e >> 6][cperm(e & 63)];
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
    // (b) M_kj = L_kk^-1 R_kj : column c of the tile is one right-hand side (Xs row c)
    const int j = blockIdx.x - ntrail;
    double(*Sp)[SLD] = reinterpret_cast<double(*)[SLD]>(smem);
    double(*Xs)[SLD] = reinterpret_cast<double(*)[SLD]>(smem + CB * SLD);
    double* dinv = smem + 2 * CB * SLD;
    double* tile = M + kk * np + (long)j * CB;
    for (int e = tid; e < CB * CB; e += 256) {
        Sp[e >> 6][cperm(e & 63)] = Dfac[kk * CB + e];
        Xs[e & 63][cperm(e >> 6)] = tile[(long)(e >> 6) * np + (e & 63)];      // transposed: rhs = column
    }
    __syncthreads();
    if (tid < CB) dinv[tid] = 1.0 / Sp[tid][cperm(tid)];
    __syncthreads();
    subst64(Sp, dinv, Xs);
    __syncthreads();
    for (int e = tid; e < CB * CB; e += 256) tile[(long)(e >> 6) * np + (e & 63)] = Xs[e & 63][cperm(e >> 6)];
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
